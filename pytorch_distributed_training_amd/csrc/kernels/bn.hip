// BatchNorm training kernels for NHWC bf16 activations.
//
// Reference stack per BN layer: batch_norm_stats, gather_stats_with_counts, batch_norm_elemt,
// relu, (add), and in backward relu_backward, batch_norm_backward_reduce, backward_elemt.
// Here:  stats (only when the producing conv did not emit them in its epilogue)
//        finalize            mean / invstd / running stats, one tiny launch, no host sync
//        apply               normalise + affine (+ residual) (+ ReLU), one pass
//        bwd_reduce          ReLU mask fused into the two BN reductions
//        bwd_elemt           grad_input (+ the masked gradient for the residual branch)
// Access pattern: NHWC makes the channel the fastest axis, so a thread owns 8 consecutive
// channels (one 16 B load) and walks rows; its per-channel coefficients stay in registers.
// The cross-row reductions are deterministic (LDS tree + partial slab, no float atomics).
#include <cstdlib>
#include "common.h"
#include "kernels.h"
#include "../comm/ipc_device.h"

namespace pdt {
bool stream_nt(long bytes);
namespace {

constexpr int kThreads = 256;
constexpr long kElemtTableBlocks = 4096;  // bwd_elemt: above this many chunks the coefficient table pays
constexpr long kSyncPrologueBlocks = 32;  // SyncBN: largest grid whose blocks exchange the sums in their prologue
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
constexpr int kUnrollReduce = 2;  // the backward reduction (two loads, no store per row): 8.0 -> 7.8 ms per step with 2 rows in flight (non-temporal loads)
constexpr int kUnroll2 = 4;  // k_bn_bwd_elemt2
constexpr int kUnroll = 4;  // rows per thread and trip in the streaming kernels (bn_apply+res: 1: 5.3, 2: 5.9, 4: 5.8, 8: 5.6 TB/s; 2 vs 4 is a wash end to end)

struct Map {
    int G;       // channel groups of 8 per row (C/8), power of two <= 256
    int R;       // rows covered per pass by one block
    int cg, rl;  // this thread's channel group / row lane
};
__device__ __forceinline__ Map make_map(int C) {
    Map m;
    m.G = C >> 3;
    m.R = kThreads / m.G;
    m.cg = threadIdx.x % m.G;
    m.rl = threadIdx.x / m.G;
    return m;
}
__device__ __forceinline__ void block_rows(long M, long& r0, long& r1) {
    const long per = (M + gridDim.x - 1) / gridDim.x;
    r0 = (long)blockIdx.x * per;
    r1 = r0 + per < M ? r0 + per : M;
}
// The activation streams of these kernels are read once and written once, gigabytes at a time: non-temporal loads and
// stores (no allocation in the caches on the way) measured 6.25 TB/s against 5.68 TB/s with plain ones on c = a + b
// (bench/micro/hbm_nt.hip).  Tensors that fit the caches (the CIFAR-shaped step: 16 MiB and less) are better off with
// plain accesses (that step: 4.08 vs 4.17 ms), so every kernel exists in both forms (NT) and the launchers pick by
// size (stream_nt).  -DPDT_BN_PLAIN_LDST builds the plain version only.
template <bool NT>
__device__ __forceinline__ bf16x8 ld8(const bf16* p) {
#ifdef PDT_BN_PLAIN_LDST
    return *reinterpret_cast<const bf16x8*>(p);
#else
    if constexpr (NT) return __builtin_bit_cast(bf16x8, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
    else return *reinterpret_cast<const bf16x8*>(p);
#endif
}
template <bool NT>
__device__ __forceinline__ void st8(bf16* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)v[k];
#ifdef PDT_BN_PLAIN_LDST
    *reinterpret_cast<bf16x8*>(p) = o;
#else
    if constexpr (NT) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<bf16x8*>(p) = o;
#endif
}
__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
}

// Two 8-channel accumulators per thread -> slab[part][2][C]
__device__ __forceinline__ void block_reduce_store(const Map& m, int C, const float (&s0)[8], const float (&s1)[8],
                                                   float* slab) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);  // [R][G][16]
    float* mine = red + ((long)m.rl * m.G + m.cg) * 16;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mine[k] = s0[k];
        mine[8 + k] = s1[k];
    }
    __syncthreads();
    // thread t < 2C sums column t over the R row lanes
    for (int t = threadIdx.x; t < 2 * C; t += kThreads) {
        const int which = t / C, c = t % C;
        const int cg = c >> 3, k = c & 7;
        float a = 0.f;
        for (int r = 0; r < m.R; ++r) a += red[((long)r * m.G + cg) * 16 + which * 8 + k];
        slab[(long)blockIdx.x * 2 * C + t] = a;
    }
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void k_bn_stats(const bf16* __restrict__ x, long M, int C,
                                                       float* __restrict__ slab) {
    const Map m = make_map(C);
    long r0, r1;
    block_rows(M, r0, r1);
    float s[8] = {0}, q[8] = {0};
    for (long r = r0 + m.rl; r < r1; r += m.R) {
        const bf16x8 v = ld8<NT>(x + r * C + m.cg * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float f = (float)v[k];
            s[k] += f;
            q[k] += f * f;
        }
    }
    block_reduce_store(m, C, s, q, slab);
}

// stats: partial rows [nparts][2][C] (conv epilogue / stats kernel).  256 threads = 8 row lanes x 32
// channels: the partial rows are summed here (fixed order -> deterministic), so no separate
// reduction launch sits between the conv and the normalisation.  Thread 0 also bumps
// num_batches_tracked: one launch does everything the reference needs 3 kernels + a host add for.
//
// SyncBatchNorm (mb.world > 1): the kernel also does the exchange between the ranks (comm/ipc_device.h), in place
// of a packed tensor, an all-reduce launch and the kernels around them.  Each rank turns its sums into
// (count, mean, M2 = sum (x - mean)^2), pushes them into every peer's mailbox, polls its own and merges the W
// triples in rank order with the pairwise update of Chan et al. (what torch's batch_norm_gather_stats_with_counts
// does with Welford's): no E[x^2] - E[x]^2 over the global batch, identical bits on every rank.  The mean travels
// as two floats (hi + lo), the count as an integer.  Message layout: [mean hi (C)][mean lo (C)][M2 (C)][count].
__global__ __launch_bounds__(kThreads) void k_bn_finalize(const float* __restrict__ stats, int nparts, int C,
                                                          double count, const float* __restrict__ count_dev, float eps,
                                                          float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var,
                                                          long* __restrict__ num_batches_tracked,
                                                          float* __restrict__ mean, float* __restrict__ invstd,
                                                          float* __restrict__ sums_out, const IpcMailboxes mb,
                                                          float* __restrict__ count_out) {
    const bool exchange = mb.world > 1 && mean != nullptr;
    const uint32_t seq = exchange ? ipc_begin(mb) : 0u;
    bool late = exchange && ipc_dead(mb);
    // The partial rows are summed in DOUBLE and the variance is formed in double: E[x^2] - E[x]^2 cancels
    // catastrophically in fp32 once |mean| >> std (a million fp32 additions lose ~1e-6 of sum(x^2), which is
    // (mean/std)^2 * 1e-6 of the variance).  With double here the only fp32 rounding left is inside one tile's
    // partial row (<= a few thousand additions); this kernel is a few microseconds either way.  A NaN / Inf in the
    // sums stays visible: the clamp below only catches a (tiny) negative result, it never turns a NaN into 0.
    __shared__ double red[8][32][2];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0, q = 0.0;
    if (c < C) {
        // (four rows per trip with their loads issued first; same summation order as row by row)
        int p = rl;
        for (; p + 24 < nparts; p += 32) {
            float a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = stats[(long)(p + 8 * u) * 2 * C + c];
                b[u] = stats[(long)(p + 8 * u) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s += (double)a[u];
                q += (double)b[u];
            }
        }
        for (; p < nparts; p += 8) {
            s += (double)stats[(long)p * 2 * C + c];
            q += (double)stats[(long)p * 2 * C + C + c];
        }
    }
    red[rl][cl][0] = s;
    red[rl][cl][1] = q;
    __syncthreads();
    if (rl == 0 && c < C) {
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            s += red[r][cl][0];
            q += red[r][cl][1];
        }
        if (sums_out) {
            sums_out[c] = (float)s;
            sums_out[C + c] = (float)q;
        }
        if (mean) {
            double n = count_dev ? (double)count_dev[0] : count;
            double mu = s / n;
            double var = q / n - mu * mu;
            if (var < 0.0) var = 0.0;  // (NaN compares false: it is kept)
            if (exchange) {
                const float hi = (float)mu;
                ipc_push(mb, seq, c, hi);
                ipc_push(mb, seq, C + c, (float)(mu - (double)hi));
                ipc_push(mb, seq, 2 * C + c, (float)(var * n));
                if (c == 0) ipc_push_bits(mb, seq, 3 * C, (uint32_t)(n + 0.5));
                const long long t0 = wall_clock64();
                double nt = 0.0, wsum = 0.0;
                for (int r = 0; r < mb.world; ++r) {
                    const double nr = (double)ipc_wait_bits(mb, seq, r, 3 * C, &late, t0);
                    const double mr = (double)ipc_wait(mb, seq, r, c, &late, t0) + (double)ipc_wait(mb, seq, r, C + c, &late, t0);
                    nt += nr;
                    wsum += nr * mr;
                }
                const double gm = wsum / nt;
                double m2 = 0.0;
                for (int r = 0; r < mb.world; ++r) {  // (the words have arrived: these reads return at once)
                    const double nr = (double)ipc_wait_bits(mb, seq, r, 3 * C, &late, t0);
                    const double mr = (double)ipc_wait(mb, seq, r, c, &late, t0) + (double)ipc_wait(mb, seq, r, C + c, &late, t0);
                    m2 += (double)ipc_wait(mb, seq, r, 2 * C + c, &late, t0) + nr * (mr - gm) * (mr - gm);
                }
                n = nt;
                mu = gm;
                var = m2 / nt;
                if (c == 0 && count_out) count_out[0] = (float)nt;
            }
            mean[c] = (float)mu;
            invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (running_mean) {
                const double unbiased = var * (n / (n > 2.0 ? n - 1.0 : 1.0));
                running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mu);
                running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
            }
        }
    }
    if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) num_batches_tracked[0] += 1;
    if (exchange) ipc_end(mb, seq, late);
}

// With RELU the kernel also writes one bit per element (uint8 per 8 channels): backward reads this
// 16x smaller mask instead of re-reading the bf16 output to know where the ReLU was active.
template <bool NT, bool RELU, bool RES>
__global__ __launch_bounds__(kThreads) void k_bn_apply(const bf16* __restrict__ x, const bf16* __restrict__ res,
                                                       bf16* __restrict__ z, uint8_t* __restrict__ mask, long M, int C,
                                                       const float* __restrict__ mean,
                                                       const float* __restrict__ invstd,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, BnAffine ra) {
    const Map m = make_map(C);
    float a[8], b[8];
    {
        float mu[8], is[8], ga[8], be[8];
        ld8f(mean + m.cg * 8, mu);
        ld8f(invstd + m.cg * 8, is);
        ld8f(gamma + m.cg * 8, ga);
        ld8f(beta + m.cg * 8, be);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            a[k] = is[k] * ga[k];
            b[k] = be[k] - mu[k] * a[k];
        }
    }
    // residual that is itself the INPUT of a (ReLU-less) BatchNorm, e.g. the downsample branch: its affine is
    // applied here, so that BN never writes its output tensor (one write + one read of a block output saved)
    float ra_s[8];
    const bool res_affine = RES && ra.mean != nullptr;
    if (res_affine) {
        float mu[8], is[8], ga[8], be[8];
        ld8f(ra.mean + m.cg * 8, mu);
        ld8f(ra.invstd + m.cg * 8, is);
        ld8f(ra.gamma + m.cg * 8, ga);
        ld8f(ra.beta + m.cg * 8, be);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ra_s[k] = is[k] * ga[k];
            b[k] += be[k] - mu[k] * ra_s[k];  // both shifts folded into one
        }
    }
    // grid-stride rows (the whole grid walks memory front to back), kUnroll rows per trip with all their loads
    // issued before the first use: enough bytes in flight per CU to cover HBM latency.  The rows of a trip are
    // adjacent, so at any moment the grid touches ONE contiguous window per tensor: with S-strided rows the same
    // kernels ran at 4.3 TB/s where a plain 2R+1W add reaches 6 (bench/micro/hbm_ceiling.py)
    const long S = m.R;                       // a trip of a block covers kUnroll * R ADJACENT rows ...
    const long T = (long)gridDim.x * kUnroll * m.R;  // ... and the grid one contiguous window (few DRAM streams)
    for (long rb = (long)blockIdx.x * kUnroll * m.R + m.rl; rb < M; rb += T) {
        bf16x8 v[kUnroll], rv[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const long r = rb + u * S;
            if (r < M) {
                v[u] = ld8<NT>(x + r * C + m.cg * 8);
                if (RES) rv[u] = ld8<NT>(res + r * C + m.cg * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const long r = rb + u * S;
            if (r >= M) break;
            const long off = r * C + m.cg * 8;
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = (float)v[u][k] * a[k] + b[k];
            if (RES) {
                if (res_affine) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] += (float)rv[u][k] * ra_s[k];
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] += (float)rv[u][k];
                }
            }
            if (RELU) {
                unsigned bits = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    // NaN-preserving ReLU: "o <= 0" is false for a NaN, so it passes through (and its bit is set)
                    // instead of being clamped to 0 as fmaxf / "o > 0 ? o : 0" would do
                    const bool off = o[k] <= 0.f;
                    bits |= (off ? 0u : 1u) << k;
                    o[k] = off ? 0.f : o[k];
                }
                mask[r * m.G + m.cg] = (uint8_t)bits;
            }
            st8<NT>(z + off, o);
        }
    }
}

template <bool NT, bool RELU>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_reduce(const bf16* __restrict__ dz, const uint8_t* __restrict__ mask,
                                                            const bf16* __restrict__ x, long M, int C,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ invstd,
                                                            float* __restrict__ slab) {
    const Map m = make_map(C);
    float mu[8], is[8];
    ld8f(mean + m.cg * 8, mu);
    ld8f(invstd + m.cg * 8, is);
    float s0[8] = {0}, s1[8] = {0};
    const long S = m.R;                       // a trip of a block covers kUnrollReduce * R ADJACENT rows ...
    const long T = (long)gridDim.x * kUnrollReduce * m.R;  // ... and the grid one contiguous window (few DRAM streams)
    for (long rb = (long)blockIdx.x * kUnrollReduce * m.R + m.rl; rb < M; rb += T) {
        bf16x8 g[kUnrollReduce], xv[kUnrollReduce];
        unsigned bits[kUnrollReduce];
#pragma unroll
        for (int u = 0; u < kUnrollReduce; ++u) {
            const long r = rb + u * S;
            bits[u] = 0xffu;
            if (r < M) {
                g[u] = ld8<NT>(dz + r * C + m.cg * 8);
                xv[u] = ld8<NT>(x + r * C + m.cg * 8);
                if (RELU) bits[u] = mask[r * m.G + m.cg];
            }
        }
#pragma unroll
        for (int u = 0; u < kUnrollReduce; ++u) {
            if (rb + u * S >= M) break;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float gg = (float)g[u][k];
                if (RELU) gg = ((bits[u] >> k) & 1u) ? gg : 0.f;
                s0[k] += gg;
                s1[k] += gg * ((float)xv[u][k] - mu[k]) * is[k];
            }
        }
    }
    block_reduce_store(m, C, s0, s1, slab);
}

// Per-channel coefficient table [3][C] of k_bn_bwd_elemt plus the summed [2][C] (second reduction level):
// launched ahead of the streaming kernel when that one runs one chunk per block and a per-block
// prologue over nps partial rows would cost as much as the chunk.
__global__ __launch_bounds__(kThreads) void k_bn_bwd_coef(const float* __restrict__ psums, int nps, int C,
                                                          const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, float count,
                                                          const float* __restrict__ count_dev,
                                                          float* __restrict__ sums_out, float* __restrict__ coef,
                                                          const IpcMailboxes mb) {
    // 32 channels per block, 8 lanes per channel over the partial rows (fixed order -> deterministic)
    // SyncBatchNorm (mb.world > 1): sums_out keeps the LOCAL sums (the affine gradients, averaged by DDP), the
    // coefficients use the sums over all ranks, exchanged here (message: [sum dz (C)][sum dz*xhat (C)])
    __shared__ float red[8][32][2];
    const int cl = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const bool exchange = mb.world > 1;
    const uint32_t seq = exchange ? ipc_begin(mb) : 0u;
    bool late = exchange && ipc_dead(mb);
    float sa = 0.f, sb = 0.f;
    if (c < C)
        for (int p = pl; p < nps; p += 8) {
            sa += psums[(long)p * 2 * C + c];
            sb += psums[(long)p * 2 * C + C + c];
        }
    red[pl][cl][0] = sa;
    red[pl][cl][1] = sb;
    __syncthreads();
    if (pl == 0 && c < C) {
#pragma unroll
        for (int r = 1; r < 8; ++r) {
            sa += red[r][cl][0];
            sb += red[r][cl][1];
        }
        sums_out[c] = sa;
        sums_out[C + c] = sb;
        if (exchange) {
            ipc_push(mb, seq, c, sa);
            ipc_push(mb, seq, C + c, sb);
            const long long t0 = wall_clock64();
            sa = 0.f;
            sb = 0.f;
            for (int r = 0; r < mb.world; ++r) {
                sa += ipc_wait(mb, seq, r, c, &late, t0);
                sb += ipc_wait(mb, seq, r, C + c, &late, t0);
            }
        }
        const float invn = 1.f / (count_dev ? count_dev[0] : count);
        const float is = invstd[c], g_is = gamma[c] * is;
        const float c1 = sa * invn, c2 = sb * invn;
        coef[c] = g_is;
        coef[C + c] = -g_is * c2 * is;
        coef[2 * C + c] = g_is * (c2 * is * mean[c] - c1);
    }
    if (exchange) ipc_end(mb, seq, late);
}

// dx = k1*dzm + k2*x + k3 with per-channel k's (see host comment); dres = dzm
template <bool NT, bool RELU, bool DRES>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_elemt(const bf16* __restrict__ dz, const uint8_t* __restrict__ mask,
                                                           const bf16* __restrict__ x, bf16* __restrict__ dx,
                                                           bf16* __restrict__ dres, long M, int C,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ psums, int nps,
                                                           float* __restrict__ sums_out, float count,
                                                           const float* __restrict__ count_dev,
                                                           const float* __restrict__ coef, const IpcMailboxes mb) {
    const Map m = make_map(C);
    float k1[8], k2[8], k3[8];
    // SyncBatchNorm on a small tensor (no table kernel in front): block 0 pushes this rank's sums, the first row
    // lane of EVERY block polls the mailbox for its 8 channels and hands the totals to the block through LDS
    const bool exchange = mb.world > 1 && coef == nullptr;
    const uint32_t seq = exchange ? ipc_begin(mb) : 0u;
    bool late = exchange && ipc_dead(mb);
    if (coef != nullptr) {  // big tensors: k_bn_bwd_coef made the table, a block only streams its one chunk
        ld8f(coef + m.cg * 8, k1);
        ld8f(coef + C + m.cg * 8, k2);
        ld8f(coef + 2 * C + m.cg * 8, k3);
    } else {
        const float invn = 1.f / (count_dev ? count_dev[0] : count);
        float mu[8], is[8], ga[8], sa[8] = {0}, sb[8] = {0};
        ld8f(mean + m.cg * 8, mu);
        ld8f(invstd + m.cg * 8, is);
        ld8f(gamma + m.cg * 8, ga);
        // second reduction level fused here: psums is [nps <= 32][2][C] (fixed order -> deterministic)
        for (int p = 0; p < nps; ++p) {
            float ta[8], tb[8];
            ld8f(psums + (long)p * 2 * C + m.cg * 8, ta);
            ld8f(psums + (long)p * 2 * C + C + m.cg * 8, tb);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sa[k] += ta[k];
                sb[k] += tb[k];
            }
        }
        if (sums_out != nullptr && blockIdx.x == 0 && m.rl == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sums_out[m.cg * 8 + k] = sa[k];
                sums_out[C + m.cg * 8 + k] = sb[k];
            }
        }
        if (exchange) {
            __shared__ float tot[2][2048];  // C <= 2048 (bn_channels_ok)
            if (m.rl == 0) {
                if (blockIdx.x == 0) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        ipc_push(mb, seq, m.cg * 8 + k, sa[k]);
                        ipc_push(mb, seq, C + m.cg * 8 + k, sb[k]);
                    }
                }
                const long long t0 = wall_clock64();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float ta = 0.f, tb = 0.f;
                    for (int r = 0; r < mb.world; ++r) {
                        ta += ipc_wait(mb, seq, r, m.cg * 8 + k, &late, t0);
                        tb += ipc_wait(mb, seq, r, C + m.cg * 8 + k, &late, t0);
                    }
                    tot[0][m.cg * 8 + k] = ta;
                    tot[1][m.cg * 8 + k] = tb;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                sa[k] = tot[0][m.cg * 8 + k];
                sb[k] = tot[1][m.cg * 8 + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float g_is = ga[k] * is[k];
            const float c1 = sa[k] * invn, c2 = sb[k] * invn;
            k1[k] = g_is;
            k2[k] = -g_is * c2 * is[k];
            k3[k] = g_is * (c2 * is[k] * mu[k] - c1);
        }
    }
    const long S = m.R;                       // a trip of a block covers kUnroll * R ADJACENT rows ...
    const long T = (long)gridDim.x * kUnroll * m.R;  // ... and the grid one contiguous window (few DRAM streams)
    for (long rb = (long)blockIdx.x * kUnroll * m.R + m.rl; rb < M; rb += T) {
        bf16x8 g[kUnroll], xv[kUnroll];
        unsigned bits[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const long r = rb + u * S;
            bits[u] = 0xffu;
            if (r < M) {
                g[u] = ld8<NT>(dz + r * C + m.cg * 8);
                xv[u] = ld8<NT>(x + r * C + m.cg * 8);
                if (RELU) bits[u] = mask[r * m.G + m.cg];
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const long r = rb + u * S;
            if (r >= M) break;
            const long off = r * C + m.cg * 8;
            float o[8], gm[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float gg = (float)g[u][k];
                if (RELU) gg = ((bits[u] >> k) & 1u) ? gg : 0.f;
                gm[k] = gg;
                o[k] = k1[k] * gg + k2[k] * (float)xv[u][k] + k3[k];
            }
            st8<NT>(dx + off, o);
            if (DRES) st8<NT>(dres + off, gm);
        }
    }
    if (exchange) ipc_end(mb, seq, late);
}

// Two BatchNorms that receive the SAME gradient (the last BN of a bottleneck and the BN of its downsample branch:
// z = relu(bn3(x) + bn_ds(x_ds))) in one streaming pass: dz and the ReLU mask are read once instead of twice.
// Both coefficient tables come from k_bn_bwd_coef.
template <bool NT>
__global__ __launch_bounds__(kThreads) void k_bn_bwd_elemt2(const bf16* __restrict__ dz, const uint8_t* __restrict__ mask,
                                                            const bf16* __restrict__ xa, const bf16* __restrict__ xb,
                                                            bf16* __restrict__ dxa, bf16* __restrict__ dxb, long M,
                                                            int C, const float* __restrict__ coef_a,
                                                            const float* __restrict__ coef_b) {
    const Map m = make_map(C);
    float a1[8], a2[8], a3[8], b1[8], b2[8], b3[8];
    ld8f(coef_a + m.cg * 8, a1);
    ld8f(coef_a + C + m.cg * 8, a2);
    ld8f(coef_a + 2 * C + m.cg * 8, a3);
    ld8f(coef_b + m.cg * 8, b1);
    ld8f(coef_b + C + m.cg * 8, b2);
    ld8f(coef_b + 2 * C + m.cg * 8, b3);
    constexpr int U = kUnroll2;  // rows in flight per trip (three input streams)
    const long S = m.R;
    const long T = (long)gridDim.x * U * m.R;
    for (long rb = (long)blockIdx.x * U * m.R + m.rl; rb < M; rb += T) {
        bf16x8 g[U], va[U], vb[U];
        unsigned bits[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = rb + u * S;
            bits[u] = 0xffu;
            if (r < M) {
                g[u] = ld8<NT>(dz + r * C + m.cg * 8);
                va[u] = ld8<NT>(xa + r * C + m.cg * 8);
                vb[u] = ld8<NT>(xb + r * C + m.cg * 8);
                if (mask != nullptr) bits[u] = mask[r * m.G + m.cg];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = rb + u * S;
            if (r >= M) break;
            const long off = r * C + m.cg * 8;
            float oa[8], ob[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gg = ((bits[u] >> k) & 1u) ? (float)g[u][k] : 0.f;
                oa[k] = a1[k] * gg + a2[k] * (float)va[u][k] + a3[k];
                ob[k] = b1[k] * gg + b2[k] * (float)vb[u][k] + b3[k];
            }
            st8<NT>(dxa + off, oa);
            st8<NT>(dxb + off, ob);
        }
    }
}

// Grid of the streaming kernels: ONE chunk (kUnroll * R adjacent rows, 16 KiB per tensor) per block, blocks
// dispatched in order -> the chip walks each tensor front to back like a plain vectorised elementwise
// kernel does (5.8 TB/s; a 2048-block grid-stride loop over the same rows reached 5.1, hbm_ceiling.py).
long stream_blocks(long M, int R, long cap) {
    static const long env_cap = [] {
        const char* e = getenv("PDT_BN_GRID_CAP");
        return e ? atol(e) : 0L;
    }();
    if (env_cap > 0) cap = env_cap;
    long blocks = (M + (long)R * kUnroll - 1) / ((long)R * kUnroll);
    if (cap > 0 && blocks > cap) blocks = cap;
    return blocks < 1 ? 1 : blocks;
}

// ---------------------------------------------------------------------------------------------------------
// Stem tail fused: y = maxpool3x3/2/1(relu(bn(u))).  The full-resolution activation relu(bn(u)) is never
// written: the forward reads the conv output u once and writes the pooled tensor, the winning tap per element
// and the 1-bit ReLU mask of the full-resolution pixels (each window owns the 2x2 pixels at its taps (1..2, 1..2),
// which tile the image exactly once).  The backward is the unfused one (pool backward, then the BN kernels with
// the mask): a variant that gathered the pooled gradient inside the BN reduce / grad-input kernels measured
// 1.7x SLOWER than those three streaming passes (the gather ran twice), so it was dropped.
struct PoolGeom {
    int B, H, W, C, Ho, Wo;
    FastDiv dw, dh;  // forward: divide by Wo, Ho; backward: by W, H
};

__device__ __forceinline__ void bn_coeffs(const Map& m, const float* mean, const float* invstd, const float* gamma,
                                          const float* beta, float (&a)[8], float (&b)[8]) {
    float mu[8], is[8], ga[8], be[8];
    ld8f(mean + m.cg * 8, mu);
    ld8f(invstd + m.cg * 8, is);
    ld8f(gamma + m.cg * 8, ga);
    ld8f(beta + m.cg * 8, be);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = is[k] * ga[k];
        b[k] = be[k] - mu[k] * a[k];
    }
}

template <bool NT>
__global__ __launch_bounds__(kThreads) void k_bn_relu_maxpool_fwd(const bf16* __restrict__ u, bf16* __restrict__ y,
                                                                 uint8_t* __restrict__ argmax,
                                                                 uint8_t* __restrict__ mask, PoolGeom g,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta) {
    const Map m = make_map(g.C);
    float a[8], b[8];
    bn_coeffs(m, mean, invstd, gamma, beta, a, b);
    const uint32_t P = (uint32_t)g.B * g.Ho * g.Wo;
    // pooled pixel: one chunk of R pixels per block; runs of 64 consecutive chunks (about 37 image rows) go to one
    // XCD so that the input rows two neighbouring output rows share are fetched into ONE L2
    uint32_t blk = blockIdx.x;
    {
        constexpr uint32_t GRP = 64;
        const uint32_t full = gridDim.x / (8 * GRP) * (8 * GRP);
        if (blk < full) {
            const uint32_t xcd = blk & 7, slot = blk >> 3;
            blk = ((slot / GRP) * 8 + xcd) * GRP + slot % GRP;
        }
    }
    const uint32_t p = blk * m.R + m.rl;
    if (p >= P) return;
    uint32_t t = fdiv(p, g.dw);
    const int wo = (int)(p - t * g.dw.d);
    const uint32_t bb = fdiv(t, g.dh);
    const int ho = (int)(t - bb * g.dh.d);
    // all nine taps are loaded up front from CLAMPED coordinates (the out-of-image ones are ignored below)
    bf16x8 v9[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int hi = min(max(ho * 2 - 1 + r, 0), g.H - 1);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int wi = min(max(wo * 2 - 1 + s, 0), g.W - 1);
            v9[r * 3 + s] = ld8<NT>(u + (uint32_t)((((int)bb * g.H + hi) * g.W + wi) * g.C + m.cg * 8));
        }
    }
    // The kernel is VALU-bound (a first version: 1500 vector instructions per thread = the whole run time), so value
    // and winning tap travel in ONE integer key per channel: post-ReLU values are >= +0, so their bf16 bit patterns
    // order like the numbers; key = bf16 << 16 | (8 - tap), and one v_max_u32 per tap keeps the largest value and,
    // among equals, the FIRST tap in scan order (what ATen's pool records).
    uint32_t key[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) key[k] = 0u;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int hi = ho * 2 - 1 + r;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int wi = wo * 2 - 1 + s;
            if (hi < 0 || hi >= g.H || wi < 0 || wi >= g.W) continue;
            const bf16x8 v = v9[r * 3 + s];
            const uint32_t code = (uint32_t)(8 - (r * 3 + s));
            unsigned bits = 0;
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const float p0 = (float)v[k] * a[k] + b[k], p1 = (float)v[k + 1] * a[k + 1] + b[k + 1];
                if (r >= 1 && s >= 1) bits |= ((p0 <= 0.f ? 0u : 1u) << k) | ((p1 <= 0.f ? 0u : 1u) << (k + 1));
                // bf16(relu(a*u + b)), the value the unfused path would have stored (+0 where inactive; a NaN stays)
                const bf16x2 pk = {(bf16)(p0 <= 0.f ? 0.f : p0), (bf16)(p1 <= 0.f ? 0.f : p1)};
                const uint32_t w2 = __builtin_bit_cast(uint32_t, pk);
                key[k] = max(key[k], (w2 << 16) | code);
                key[k + 1] = max(key[k + 1], (w2 & 0xffff0000u) | code);
            }
            if (r >= 1 && s >= 1)  // this window owns pixel (2ho + r - 1, 2wo + s - 1)
                mask[(uint32_t)((((int)bb * g.H + hi) * g.W + wi)) * (uint32_t)m.G + m.cg] = (uint8_t)bits;
        }
    }
    const uint32_t off = p * (uint32_t)g.C + m.cg * 8;
    uint32_t ow[4], plo = 0, phi = 0;
#pragma unroll
    for (int k = 0; k < 8; k += 2) ow[k >> 1] = (key[k] >> 16) | (key[k + 1] & 0xffff0000u);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        plo |= (8u - (key[k] & 0xfu)) << (8 * k);
        phi |= (8u - (key[4 + k] & 0xfu)) << (8 * k);
    }
    if constexpr (NT) {
        st_stream(y + off, u32x4{ow[0], ow[1], ow[2], ow[3]});
        __builtin_nontemporal_store(u32x2_t{plo, phi}, reinterpret_cast<u32x2_t*>(argmax + off));
    } else {
        *reinterpret_cast<uint4*>(y + off) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        *reinterpret_cast<uint2*>(argmax + off) = make_uint2(plo, phi);
    }
}

// Backward of the fused stem tail: the gradient w.r.t. relu(bn(u)) is the max-pool scatter of the pooled gradient, and
// it is only ever consumed by the two BN-backward passes.  Both passes GATHER it themselves (2x2 blocks of
// full-resolution pixels, four window loads and nine tap compares per channel, see k_maxpool_bwd_2x2 in pool.hip), so
// the full-resolution gradient (2 GB at B = 1280) is never written or read: 2.2 ms of pool-backward + reduce + elemt
// become 1.3.  (An earlier fused version with the per-pixel gather was 1.7x SLOWER than the three passes.)
// MODE 0: partial rows [gridDim][2][C] of [sum g, sum g * xhat] (grid-stride, block reduction);  MODE 1: du = k1 g + k2 u + k3.
// The gathered value is rounded to bf16 first, as the unfused path stored it.
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_bn_relu_maxpool_bwd(const bf16* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                                 const bf16* __restrict__ u, const uint8_t* __restrict__ mask,
                                                                 bf16* __restrict__ du, PoolGeom g /* dw: W/2, dh: H/2 */,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 const float* __restrict__ coef, float* __restrict__ slab) {
    const Map m = make_map(g.C);
    const uint32_t Hb = (uint32_t)g.H >> 1, Wb = (uint32_t)g.W >> 1;
    const uint32_t P = (uint32_t)g.B * Hb * Wb;
    float c1[8], c2[8], c3[8];  // MODE 0: mean, invstd, -;  MODE 1: k1, k2, k3
    if (MODE == 0) {
        ld8f(mean + m.cg * 8, c1);
        ld8f(invstd + m.cg * 8, c2);
    } else {
        ld8f(coef + m.cg * 8, c1);
        ld8f(coef + g.C + m.cg * 8, c2);
        ld8f(coef + 2 * g.C + m.cg * 8, c3);
    }
    float s0[8] = {0}, s1[8] = {0};
    for (uint32_t p = blockIdx.x * (uint32_t)m.R + m.rl; p < P; p += gridDim.x * (uint32_t)m.R) {
        const uint32_t t = fdiv(p, g.dw);
        const int j = (int)(p - t * Wb);
        const uint32_t bb = fdiv(t, g.dh);
        const int i = (int)(t - bb * Hb);
        const int b = (int)bb;
        bf16x8 gy[4];
        uint2 am[4];
#pragma unroll
        for (int da = 0; da < 2; ++da)
#pragma unroll
            for (int dc = 0; dc < 2; ++dc) {
                const int ho = i + da, wo = j + dc;
                const bool ok = ho < g.Ho && wo < g.Wo;
                const uint32_t off = (uint32_t)(((b * g.Ho + min(ho, g.Ho - 1)) * g.Wo + min(wo, g.Wo - 1)) * g.C) + m.cg * 8;
                gy[da * 2 + dc] = *reinterpret_cast<const bf16x8*>(dy + off);
                const uint2 a = *reinterpret_cast<const uint2*>(argmax + off);
                am[da * 2 + dc] = ok ? a : make_uint2(0xffffffffu, 0xffffffffu);
            }
        bf16x8 uv[4];
        unsigned bits[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t pix = (uint32_t)((b * g.H + 2 * i + (q >> 1)) * g.W + 2 * j + (q & 1));
            uv[q] = ld8<true>(u + pix * (uint32_t)g.C + m.cg * 8);
            bits[q] = mask[pix * (uint32_t)m.G + m.cg];
        }
        float gz[4][8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int sh = 8 * (k & 3);
            uint32_t c[4];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c[q] = ((k < 4 ? am[q].x : am[q].y) >> sh) & 0xffu;
                v[q] = (float)gy[q][k];
            }
            gz[0][k] = (float)(bf16)(c[0] == 4u ? v[0] : 0.f);
            gz[1][k] = (float)(bf16)((c[0] == 5u ? v[0] : 0.f) + (c[1] == 3u ? v[1] : 0.f));
            gz[2][k] = (float)(bf16)((c[0] == 7u ? v[0] : 0.f) + (c[2] == 1u ? v[2] : 0.f));
            gz[3][k] = (float)(bf16)((((c[0] == 8u ? v[0] : 0.f) + (c[1] == 6u ? v[1] : 0.f)) + (c[2] == 2u ? v[2] : 0.f)) +
                                     (c[3] == 0u ? v[3] : 0.f));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gg = ((bits[q] >> k) & 1u) ? gz[q][k] : 0.f;
                if (MODE == 0) {
                    s0[k] += gg;
                    s1[k] += gg * ((float)uv[q][k] - c1[k]) * c2[k];
                } else {
                    o[k] = c1[k] * gg + c2[k] * (float)uv[q][k] + c3[k];
                }
            }
            if (MODE == 1) {
                const uint32_t pix = (uint32_t)((b * g.H + 2 * i + (q >> 1)) * g.W + 2 * j + (q & 1));
                st8<true>(du + pix * (uint32_t)g.C + m.cg * 8, o);
            }
        }
    }
    if (MODE == 0) block_reduce_store(m, g.C, s0, s1, slab);
}

int grid_for(long M, int C) {
    const int R = kThreads / (C >> 3);
    long parts = (M + (long)R * 16 - 1) / ((long)R * 16);
    if (parts < 1) parts = 1;
    if (parts > 512) parts = 512;
    return (int)parts;
}

}  // namespace

// Non-temporal accesses from this tensor size on (bytes of ONE activation tensor of the launch); PDT_STREAM_NT_MIN_MB
// overrides (0: always, a huge value: never).  Same-box A/B (bench/micro/nt_ab.sh), ms per step at B = 1280 / B = 256 /
// CIFAR shape under hipGraph: never 84.48 / 20.98 / 4.131, from 512 MiB 83.30, from 128 MiB 82.79 / 20.80, from 32 MiB
// 82.90 / 20.90 / 4.131, always - / - / 4.178
bool stream_nt(long bytes) {
    static const long min_bytes = [] {
        const char* e = getenv("PDT_STREAM_NT_MIN_MB");
        return (e ? atol(e) : 128L) << 20;
    }();
    return bytes >= min_bytes;
}


int bn_parts(long M, int C) { return grid_for(M, C); }

void launch_bn_stats(const void* x, long M, int C, float* slab, int parts, hipStream_t s) {
    const size_t lds = (size_t)kThreads * 16 * sizeof(float);
    if (stream_nt(M * C * 2)) hipLaunchKernelGGL(k_bn_stats<true>, dim3(parts), dim3(kThreads), lds, s, (const bf16*)x, M, C, slab);
    else hipLaunchKernelGGL(k_bn_stats<false>, dim3(parts), dim3(kThreads), lds, s, (const bf16*)x, M, C, slab);
}

static IpcMailboxes no_exchange() {
    IpcMailboxes mb{};
    mb.world = 1;
    return mb;
}

void launch_bn_finalize(const float* stats, int nparts, int C, double count, const float* count_dev, float eps,
                        float momentum, float* running_mean, float* running_var, long* num_batches_tracked,
                        float* mean, float* invstd, float* sums_out, hipStream_t s, const IpcMailboxes* mb,
                        float* count_out) {
    hipLaunchKernelGGL(k_bn_finalize, dim3((C + 31) / 32), dim3(kThreads), 0, s, stats, nparts, C, count, count_dev,
                       eps, momentum, running_mean, running_var, num_batches_tracked, mean, invstd, sums_out,
                       mb ? *mb : no_exchange(), count_out);
}

void launch_bn_apply(const void* x, const void* residual, void* z, uint8_t* mask, long M, int C, const float* mean,
                     const float* invstd, const float* gamma, const float* beta, int relu, const BnAffine& res_affine,
                     hipStream_t s) {
    const int R = kThreads / (C >> 3);
    const long blocks = stream_blocks(M, R, 0);
    dim3 grid((int)blocks), block(kThreads);
    const bool nt = stream_nt(M * C * 2);
#define PDT_APPLY1(NT, RELU, RES)                                                                                    \
    hipLaunchKernelGGL((k_bn_apply<NT, RELU, RES>), grid, block, 0, s, (const bf16*)x, (const bf16*)residual, (bf16*)z, \
                       mask, M, C, mean, invstd, gamma, beta, res_affine)
#define PDT_APPLY(RELU, RES)              \
    do {                                  \
        if (nt) PDT_APPLY1(true, RELU, RES);  \
        else PDT_APPLY1(false, RELU, RES);    \
    } while (0)
    if (relu && residual) PDT_APPLY(true, true);
    else if (relu) PDT_APPLY(true, false);
    else if (residual) PDT_APPLY(false, true);
    else PDT_APPLY(false, false);
#undef PDT_APPLY
#undef PDT_APPLY1
}

void launch_bn_bwd_reduce(const void* dz, const void* z, const void* x, long M, int C, const float* mean,
                          const float* invstd, int relu, float* slab, int parts, hipStream_t s) {
    const size_t lds = (size_t)kThreads * 16 * sizeof(float);
    const bool nt = stream_nt(M * C * 2);
#define PDT_REDUCE(NT, RELU)                                                                                       \
    hipLaunchKernelGGL((k_bn_bwd_reduce<NT, RELU>), dim3(parts), dim3(kThreads), lds, s, (const bf16*)dz,           \
                       (const uint8_t*)z, (const bf16*)x, M, C, mean, invstd, slab)
    if (relu && nt) PDT_REDUCE(true, true);
    else if (relu) PDT_REDUCE(false, true);
    else if (nt) PDT_REDUCE(true, false);
    else PDT_REDUCE(false, false);
#undef PDT_REDUCE
}

void launch_bn_bwd_elemt(const void* dz, const void* z, const void* x, void* dx, void* dres, long M, int C,
                         const float* mean, const float* invstd, const float* gamma, const float* psums, int nps,
                         float* sums_out, float count, const float* count_dev, int relu, float* coef,
                         hipStream_t s, const IpcMailboxes* mbp) {
    const IpcMailboxes mb = mbp ? *mbp : no_exchange();
    const int R = kThreads / (C >> 3);
    // small tensors: one launch, <= 2048 blocks each folding the nps partial rows in its prologue;
    // big ones (coef workspace given): table kernel first, then one chunk per block
    // SyncBatchNorm: the blocks that exchange with the peers WAIT for them.  In the prologue of this kernel that is
    // every block of the grid, which is fine for a handful of blocks (tiny tensors, where a launch saved matters
    // most) but a grid that fills the chip with waiting waves would starve a peer that shares the GPU (the one-GPU
    // rehearsal of W > 1) and burn the chip for nothing elsewhere: larger grids take the table kernel (<= 64 blocks).
    const bool sync = mb.world > 1;
    const bool table = coef != nullptr && stream_blocks(M, R, 0) > (sync ? kSyncPrologueBlocks : kElemtTableBlocks);
    if (table)
        hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 31) / 32), dim3(kThreads), 0, s, psums, nps, C,
                           mean, invstd, gamma, count, count_dev, sums_out, coef, mb);
    else
        coef = nullptr;
    const long blocks = stream_blocks(M, R, table ? 0 : 2048);
    dim3 grid((int)blocks), block(kThreads);
    const bool nt = stream_nt(M * C * 2);
#define PDT_ELEMT1(NT, RELU, DRES)                                                                                  \
    hipLaunchKernelGGL((k_bn_bwd_elemt<NT, RELU, DRES>), grid, block, 0, s, (const bf16*)dz, (const uint8_t*)z,       \
                       (const bf16*)x, (bf16*)dx, (bf16*)dres, M, C, mean, invstd, gamma, psums, nps, sums_out, count,   \
                       count_dev, coef, mb)
#define PDT_ELEMT(RELU, DRES)                  \
    do {                                       \
        if (nt) PDT_ELEMT1(true, RELU, DRES);  \
        else PDT_ELEMT1(false, RELU, DRES);    \
    } while (0)
    if (relu && dres) PDT_ELEMT(true, true);
    else if (relu) PDT_ELEMT(true, false);
    else if (dres) PDT_ELEMT(false, true);
    else PDT_ELEMT(false, false);
#undef PDT_ELEMT
#undef PDT_ELEMT1
}

void launch_bn_bwd_elemt2(const void* dz, const void* mask, long M, int C, float count, const float* count_dev,
                          const BnBwdSide& a, const BnBwdSide& b, hipStream_t s, const IpcMailboxes* mbp) {
    const IpcMailboxes mb = mbp ? *mbp : no_exchange();
    const int R = kThreads / (C >> 3);
    const BnBwdSide* sides[2] = {&a, &b};
    for (const BnBwdSide* p : sides)  // SyncBatchNorm: one exchange per table, in the same order on every rank
        hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 31) / 32), dim3(kThreads), 0, s, p->psums, p->nps, C, p->mean,
                           p->invstd, p->gamma, count, count_dev, p->sums_out, p->coef, mb);
    const int U = kUnroll2;
    const long blocks = (M + (long)R * U - 1) / ((long)R * U);
#define PDT_ELEMT2(NT)                                                                                            \
    hipLaunchKernelGGL(k_bn_bwd_elemt2<NT>, dim3((unsigned)blocks), dim3(kThreads), 0, s, (const bf16*)dz,            \
                       (const uint8_t*)mask, (const bf16*)a.x, (const bf16*)b.x, (bf16*)a.dx, (bf16*)b.dx, M, C,     \
                       (const float*)a.coef, (const float*)b.coef)
    if (stream_nt(M * C * 2)) PDT_ELEMT2(true);
    else PDT_ELEMT2(false);
#undef PDT_ELEMT2
}

namespace {
PoolGeom pool_geom(int B, int H, int W, int C, int Ho, int Wo, bool backward) {
    PoolGeom g;
    g.B = B; g.H = H; g.W = W; g.C = C; g.Ho = Ho; g.Wo = Wo;
    g.dw = make_fastdiv((uint32_t)(backward ? W : Wo));
    g.dh = make_fastdiv((uint32_t)(backward ? H : Ho));
    return g;
}
void check_pool_index_space(long pixels, int C) {
    if (pixels * C >= (1L << 31)) {
        fprintf(stderr, "bn_relu_maxpool: %ld elements do not fit the 31-bit index space of the kernels\n", pixels * C);
        abort();
    }
}
}  // namespace

void launch_bn_relu_maxpool_fwd(const void* u, void* y, uint8_t* argmax, uint8_t* mask, int B, int H, int W, int C,
                                int Ho, int Wo,
                                const float* mean, const float* invstd, const float* gamma, const float* beta,
                                hipStream_t s) {
    check_pool_index_space((long)B * H * W, C);
    const int R = kThreads / (C >> 3);
    const long P = (long)B * Ho * Wo;
#define PDT_POOLF(NT)                                                                                                 \
    hipLaunchKernelGGL(k_bn_relu_maxpool_fwd<NT>, dim3((unsigned)((P + R - 1) / R)), dim3(kThreads), 0, s, (const bf16*)u, \
                       (bf16*)y, argmax, mask, pool_geom(B, H, W, C, Ho, Wo, false), mean, invstd, gamma, beta)
    if (stream_nt((long)B * H * W * C * 2)) PDT_POOLF(true);
    else PDT_POOLF(false);
#undef PDT_POOLF
}

// -> du; slab: [parts][2][C] workspace, sums_out [2][C], coef [3][C] workspace (see launch_bn_bwd_elemt for count / mb)
void launch_bn_relu_maxpool_bwd(const void* dy, const uint8_t* argmax, const void* u, const uint8_t* mask, void* du, int B,
                                int H, int W, int C, int Ho, int Wo, const float* mean, const float* invstd,
                                const float* gamma, float count, const float* count_dev, float* slab, int parts,
                                float* sums_out, float* coef, hipStream_t s, const IpcMailboxes* mbp) {
    // (plain pointers with unsigned 32-bit element indices: no 2 GiB window, the whole batch in one launch)
    if ((long)B * H * W * C >= (1L << 32) || H % 2 != 0 || W % 2 != 0 || Ho != H / 2 || Wo != W / 2 || parts < 1) {
        fprintf(stderr, "bn_relu_maxpool_bwd: unsupported geometry %d x %d -> %d x %d\n", H, W, Ho, Wo);
        abort();
    }
    const IpcMailboxes mb = mbp ? *mbp : no_exchange();
    const int R = kThreads / (C >> 3);
    PoolGeom g = pool_geom(B, H, W, C, Ho, Wo, false);
    g.dw = make_fastdiv((uint32_t)(W / 2));
    g.dh = make_fastdiv((uint32_t)(H / 2));
    const size_t lds = (size_t)kThreads * 16 * sizeof(float);
    hipLaunchKernelGGL(k_bn_relu_maxpool_bwd<0>, dim3(parts), dim3(kThreads), lds, s, (const bf16*)dy, argmax, (const bf16*)u,
                       mask, (bf16*)nullptr, g, mean, invstd, (const float*)nullptr, slab);
    hipLaunchKernelGGL(k_bn_bwd_coef, dim3((C + 31) / 32), dim3(kThreads), 0, s, (const float*)slab, parts, C, mean, invstd,
                       gamma, count, count_dev, sums_out, coef, mb);
    const long P = (long)B * (H / 2) * (W / 2);
    hipLaunchKernelGGL(k_bn_relu_maxpool_bwd<1>, dim3((unsigned)((P + R - 1) / R)), dim3(kThreads), 0, s, (const bf16*)dy, argmax,
                       (const bf16*)u, mask, (bf16*)du, g, mean, invstd, (const float*)coef, (float*)nullptr);
}

}  // namespace pdt
