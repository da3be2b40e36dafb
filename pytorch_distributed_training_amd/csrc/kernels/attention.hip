// Fused multi-head self-attention (forward + backward) on the gfx950 matrix cores, no mask, no dropout.
// Three families: whole-head kernels for short sequences at head_dim 64, one head per workgroup (up to 128 tokens) or,
// from 129 tokens on, persistent and software-pipelined (ViT-B/16: 197 tokens; the flagship path, middle of the file);
// and key-/query-block streaming kernels for any length at head_dim 32 / 64 / 128 (last part of the file).
//
//   qkv : [B, S, 3, H, 64] bf16  (the packed in-projection output, read in place: no permute copies)
//   out : [B, S, H, 64]    bf16  (= [B, S, D], what the out-projection GEMM wants)
//   lse : [B, H, S]        fp32  (log-sum-exp of the scaled scores, for the backward pass)
//
// One workgroup (4 waves) per (batch, head).  A whole head fits on chip: K, V (and Q, dO in the backward)
// live in LDS as [SP][64] bf16 images (SP = S rounded up to 32), the S x S score matrix only ever exists
// as MFMA accumulators: no [B,H,S,S] tensor touches HBM (the unfused path moves ~10 of them per layer).
//
//  * MFMA 32x32x16 bf16.  Scores are produced TRANSPOSED where that puts the softmax row in a lane
//    (forward, dQ pass: S^T = K Q^T, lane = query) so row max / sum are in-register reductions plus one
//    cross-half shuffle, and NON-transposed where the next GEMM contracts over queries (dK / dV pass).
//  * An accumulator tile feeds the next MFMA straight from registers as its A operand: the C layout gives
//    a lane the contraction indices (j&3) + 8*(j>>2) + 4*half of a 16-block, a permutation of the
//    8*half + j the A operand nominally wants; the B operand is fetched with the same permutation
//    (two ds_read_b64_tr_b16 at rows 4*half and 8 + 4*half), so no shuffles and no LDS round trip.
//  * one LDS swizzle serves both fragment kinds (16-B ds_read_b128 rows and transposing 8-B reads):
//    chunk ^= ((row>>1)&1)<<2 | ((row>>2)&3).
//  * softmax scale 1/8 is folded into Q when it is staged (exact in bf16).
//  * deterministic: no atomics; the backward recomputes the scores once per pass (dK/dV pass with a wave
//    per key tile, dQ pass with a wave per query tile) instead of reducing dQ across waves.
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

constexpr int HD = 64;        // head dim
constexpr int ROWB = HD * 2;  // LDS row bytes
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kNegBig = -1e30f;  // finite stand-in for -inf (the kernels are built with -ffast-math)

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

__device__ __forceinline__ int swz(int row, int chunk) {
    const int x = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
    return row * ROWB + ((chunk ^ x) << 4);
}
// MFMA A/B fragment whose contraction index runs along the 64-wide dim of a row-major [row][64] image:
// lane = row (l&31), elements d = 16*step + 8*(l>>5) + 0..7
__device__ __forceinline__ bf16x8 frag_rows(const char* img, int row, int step, int half) {
    return *reinterpret_cast<const bf16x8*>(img + swz(row, 2 * step + half));
}
// MFMA B fragment whose contraction index runs along the ROWS of a row-major [row][64] image, in the
// permuted order of an accumulator tile used as A operand: lane col = colbase + (l&31), contraction
// j<4 -> row r0 + 4*half + j, j>=4 -> row r0 + 8 + 4*half + (j-4)
__device__ __forceinline__ bf16x8 frag_cols(const char* img, int r0, int colbase, int lane) {
    const int q4 = (lane & 15) >> 2, p = lane & 3, h16 = (lane >> 4) & 1, half = lane >> 5;
    const int col = colbase + h16 * 16 + p * 4;
    const int ra = r0 + 4 * half + q4, rb = ra + 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + swz(ra, col >> 3) + (col & 7) * 2));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + swz(rb, col >> 3) + (col & 7) * 2));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// accumulator tile regs [8*blk, 8*blk+8) -> bf16 A operand
__device__ __forceinline__ bf16x8 acc_to_a(const f32x16& t, int blk) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (bf16)t[8 * blk + j];
    return a;
}
__device__ __forceinline__ int acc_row(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

// stage rows [0, SP) of one of q/k/v (which = 0/1/2) or of a [B,S,H,64] tensor (which < 0) into an LDS image
template <int NT, int NW>
__device__ __forceinline__ void stage(char* img, const bf16* src, long row_stride, int S, float scale, int tid) {
    constexpr int SP = NT * 32;
    for (int i = tid; i < SP * 8; i += 64 * NW) {
        const int row = i >> 3, ch = i & 7;
        bf16x8 v;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (bf16)0.f;
        if (row < S) v = *reinterpret_cast<const bf16x8*>(src + (long)row * row_stride + ch * 8);
        if (scale != 1.f) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] * scale);
        }
        *reinterpret_cast<bf16x8*>(img + swz(row, ch)) = v;
    }
}

// ------------------------------------------------------------------------------------------ forward
// NW waves per workgroup: 4, or 8 once a head has more than 4 tiles (ViT: 7), so that every tile has its own
// wave (one round instead of two uneven ones) and the CU holds twice the waves behind the same LDS images
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k_attn_fwd(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                  float* __restrict__ lse, int S, int H) {
    constexpr int SP = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + SP * ROWB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long D = (long)H * HD, rs = 3 * D;
    const bf16* base = qkv + (long)b * S * rs + h * HD;
    stage<NT, NW>(Ks, base + D, rs, S, 1.f, tid);
    stage<NT, NW>(Vs, base + 2 * D, rs, S, 1.f, tid);
    __syncthreads();

    for (int qt = wave; qt < NT; qt += NW) {
        // Q fragments straight from global (each query row is used by this wave only), scaled by 1/8
        const int q = qt * 32 + l31;
        bf16x8 bq[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int k = 0; k < 8; ++k) bq[s][k] = (bf16)0.f;
            if (q < S) {
                bq[s] = *reinterpret_cast<const bf16x8*>(base + (long)q * rs + s * 16 + half * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) bq[s][k] = (bf16)((float)bq[s][k] * 0.125f);
            }
        }
        // S^T tiles: rows = keys, lane = query
        f32x16 st[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, kt * 32 + l31, s, half), bq[s], st[kt], 0, 0, 0);
        }
        float m = kNegBig;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kt * 32 + acc_row(r, half) >= S) st[kt][r] = kNegBig;
                m = fmaxf(m, st[kt][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __expf(st[kt][r] - m);
                st[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 32, 64);
        // O = P V
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 pa = acc_to_a(st[kt], blk);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_cols(Vs, kt * 32 + blk * 16, dt * 32, lane),
                                                                   o[dt], 0, 0, 0);
            }
        const float inv = 1.f / sum;
        if (q < S && half == 0) lse[((long)b * H + h) * S + q] = m + __logf(sum);
        // o rows are queries (acc_row), this lane's 1/sum belongs to query l31: fetch the row's value
        bf16* orow = out + ((long)b * S * H + h) * HD;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qr = acc_row(r, half);
            const float invr = __shfl(inv, qr, 64);
            const int qq = qt * 32 + qr;
            if (qq < S) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) orow[(long)qq * D + dt * 32 + l31] = (bf16)(o[dt][r] * invr);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------- backward
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k_attn_bwd(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                  const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                  bf16* __restrict__ dqkv, int S, int H) {
    constexpr int SP = NT * 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;                       // scaled by 1/8
    char* Ks = Qs + SP * ROWB;
    char* Vs = Ks + SP * ROWB;
    char* Os = Vs + SP * ROWB;             // dO
    float* Ls = reinterpret_cast<float*>(Os + SP * ROWB);  // lse[q]   (huge on padding rows: P = 0 there)
    float* Ds = Ls + SP;                                   // D[q] = sum_d dO * O
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const long D = (long)H * HD, rs = 3 * D;
    const bf16* base = qkv + (long)b * S * rs + h * HD;
    const bf16* obase = out + ((long)b * S * H + h) * HD;
    const bf16* dobase = dout + ((long)b * S * H + h) * HD;
    stage<NT, NW>(Qs, base, rs, S, 0.125f, tid);
    stage<NT, NW>(Ks, base + D, rs, S, 1.f, tid);
    stage<NT, NW>(Vs, base + 2 * D, rs, S, 1.f, tid);
    stage<NT, NW>(Os, dobase, D, S, 1.f, tid);
    if (tid < SP) {
        float d = 0.f, l = -kNegBig;
        if (tid < S) {
            l = lse[((long)b * H + h) * S + tid];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(obase + (long)tid * D + c * 8);
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(dobase + (long)tid * D + c * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) d += (float)a[k] * (float)g[k];
            }
        }
        Ls[tid] = l;
        Ds[tid] = d;
    }
    __syncthreads();
    bf16* dq_base = dqkv + (long)b * S * rs + h * HD;

    // ---- pass A: a wave owns a KEY tile; scores non-transposed (lane = key, regs = queries) -> dK, dV
    for (int kt = wave; kt < NT; kt += NW) {
        const int key = kt * 32 + l31;
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = frag_rows(Ks, key, s, half);
            vf[s] = frag_rows(Vs, key, s, half);
        }
        f32x16 dk[2], dv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                dk[dt][r] = 0.f;
                dv[dt][r] = 0.f;
            }
        for (int qt = 0; qt < NT; ++qt) {
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = 0.f;
                dp[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Qs, qt * 32 + l31, s, half), kf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Os, qt * 32 + l31, s, half), vf[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + acc_row(r, half);
                const float p = key < S ? __expf(sc[r] - Ls[q]) : 0.f;
                sc[r] = p;                    // P
                dp[r] = p * (dp[r] - Ds[q]);  // dS
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 pa = acc_to_a(sc, blk), dsa = acc_to_a(dp, blk);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, frag_cols(Os, qt * 32 + blk * 16, dt * 32, lane),
                                                                    dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, frag_cols(Qs, qt * 32 + blk * 16, dt * 32, lane),
                                                                    dk[dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kk = kt * 32 + acc_row(r, half);
            if (kk < S) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    dq_base[(long)kk * rs + D + dt * 32 + l31] = (bf16)dk[dt][r];       // Qs carries the 1/8
                    dq_base[(long)kk * rs + 2 * D + dt * 32 + l31] = (bf16)dv[dt][r];
                }
            }
        }
    }

    // ---- pass B: a wave owns a QUERY tile; scores transposed (lane = query, regs = keys) -> dQ
    for (int qt = wave; qt < NT; qt += NW) {
        const int q = qt * 32 + l31;
        bf16x8 qf[4], gf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[s] = frag_rows(Qs, q, s, half);
            gf[s] = frag_rows(Os, q, s, half);
        }
        const float lq = Ls[q], dq_row = Ds[q];
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
        for (int kt = 0; kt < NT; ++kt) {
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = 0.f;
                dp[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, kt * 32 + l31, s, half), qf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Vs, kt * 32 + l31, s, half), gf[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kt * 32 + acc_row(r, half);
                const float p = kk < S ? __expf(sc[r] - lq) : 0.f;
                dp[r] = p * (dp[r] - dq_row);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 dsa = acc_to_a(dp, blk);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, frag_cols(Ks, kt * 32 + blk * 16, dt * 32, lane),
                                                                    dq[dt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = qt * 32 + acc_row(r, half);
            if (qq < S) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) dq_base[(long)qq * rs + dt * 32 + l31] = (bf16)(dq[dt][r] * 0.125f);
            }
        }
    }
}

// ===================================================================== whole head, persistent and software-pipelined
// The kernels above run one head per workgroup and fetch it with global loads that are written to LDS one at a time
// (load -> wait -> ds_write per 16 B, 11 dependent round trips before the first MFMA) with one workgroup per CU
// (201 / 180 registers): at ViT-B/16's 197 tokens they reach ~200 TFLOP/s, a third of what their own HBM traffic
// allows.  The kernels below keep the arithmetic and change the data movement:
//  * one workgroup per CU walks over heads (grid = CUs, head = blockIdx + k * grid);
//  * operands arrive by LDS-DMA (buffer_load ... lds, 1 KiB = 8 rows per wave instruction) with the swizzle applied to
//    the SOURCE chunk and rows past S sent out of range (zeros): no register round trip, everything in flight at once;
//  * the next head's operands are fetched while the current head is computed.  Forward: two (K, V) buffers.  Backward:
//    the key-owner pass reads Q / dO / O images and the wave's own K / V rows (prefetched into registers), the
//    query-owner pass reads K / V images and the wave's own Q / dO rows (taken from LDS at the end of the first pass),
//    so K / V of this head land during the first pass and Q / dO / O of the NEXT head during the second: five images
//    (140 KiB at 7 tiles) instead of eight;
//  * D[q] = sum_d dO * O comes from the two LDS images (the O image exists for this alone: 28 KiB of LDS against a
//    second read of O and dO by a pre-pass), the softmax scale moves from the staged Q to the fp32 scores;
//  * waits are counted: the stores of a head (issued after the next head's DMA) are not waited for.
template <int NT, int NW>
__device__ __forceinline__ void stage_dma(uint32_t img_lds, const bf16* src, long row_stride, int S, int wave, int lane) {
    i32x4 rsrc = make_rsrc_words(src, (uint32_t)((long)(S - 1) * row_stride * 2 + ROWB));
#pragma unroll
    for (int k = 0; k < 3; ++k) rsrc[k] = __builtin_amdgcn_readfirstlane(rsrc[k]);  // (head index: uniform, but the compiler cannot see it)
#pragma unroll
    for (int i = 0; i < (NT * 4 + NW - 1) / NW; ++i) {
        const int p = wave + i * NW;  // piece of 8 rows; wave-uniform
        if (p < NT * 4) {
            const int row = p * 8 + (lane >> 3);
            const int x = (((row >> 1) & 1) << 2) | ((row >> 2) & 3);
            const uint32_t off = row < S ? (uint32_t)row * (uint32_t)(row_stride * 2) + (uint32_t)(((lane & 7) ^ x) << 4) : kOOB;
            lds_dma16(rsrc, off, (uint32_t)__builtin_amdgcn_readfirstlane((int)(img_lds + (uint32_t)p * 1024u)));
        }
    }
}
// a wave's own 32 rows of a [*, row_stride] tensor as MFMA fragments from global memory; rows past S read row S - 1
// (their results are masked or never stored)
__device__ __forceinline__ void own_rows64(bf16x8* f, const bf16* src, long row_stride, int row, int S, int half) {
    const int rc = row < S ? row : S - 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const bf16x8*>(src + (long)rc * row_stride + s * 16 + half * 8);
}
__device__ __forceinline__ void pin(bf16x8* f) {  // the fragments exist in registers HERE (nothing of them is loaded later)
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" : "+v"(f[s])::"memory");
}

// one bf16 through a buffer descriptor: per-lane byte offset + uniform byte offset; lanes past the descriptor's range are
// dropped by the hardware (rows >= S of a head), so every lane issues every store and no address is kept per row
// descriptor over [base, base + bytes) for a wave-uniform base: the two address words go through readfirstlane, because a
// 64-bit product computed on the vector ALU (there is no scalar one) otherwise reaches the store as a VGPR pair and
// every store gets a waterfall loop around it
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_uniform(const void* base, uint32_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a);
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
    return make_rsrc(reinterpret_cast<const void*>(((unsigned long long)hi << 32) | lo),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)bytes));
}
__device__ __forceinline__ void store_bf16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float v) {
    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (bf16)v), r, (int)voff, (int)soff, 0);
}

template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k_attn_fwd_p(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                        float* __restrict__ lse, int S, int H, int BH) {
    constexpr int SP = NT * 32, IMG = SP * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][K image, V image]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long D = (long)H * HD, rs = 3 * D;
    const uint32_t lds0 = lds_address(smem);
    const int q = wave * 32 + l31;  // this wave's query tile is tile `wave` of every head
    auto head_base = [&](int bh) {
        const int hb = __builtin_amdgcn_readfirstlane(bh / H);  // (uniform, and now the compiler knows: scalar addresses, no waterfall)
        return qkv + (long)hb * S * rs + (bh - hb * H) * HD;
    };
    auto issue_kv = [&](int bh, int buf) {
        const bf16* base = head_base(bh);
        stage_dma<NT, NW>(lds0 + (uint32_t)(buf * 2 * IMG), base + D, rs, S, wave, lane);
        stage_dma<NT, NW>(lds0 + (uint32_t)(buf * 2 * IMG + IMG), base + 2 * D, rs, S, wave, lane);
    };
    int bh = blockIdx.x, buf = 0;
    if (bh >= BH) return;
    bf16x8 bqn[4];
    issue_kv(bh, 0);
    if (wave < NT) own_rows64(bqn, head_base(bh), rs, q, S, half);
    bool stored = false;  // (first head: nothing was issued after its DMA)
    for (; bh < BH; bh += gridDim.x, buf ^= 1) {
        bf16x8 bq[4];
        if (wave < NT) {
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < 8; ++k) bq[s][k] = (bf16)((float)bqn[s][k] * 0.125f);
            pin(bq);
        }
        // the previous head's 33 stores were issued after this head's DMA and query rows: a wave with a full tile
        // may leave them in flight
        if (stored && wave < NT - 1) asm volatile("s_waitcnt vmcnt(33)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stored = true;
        __builtin_amdgcn_s_barrier();  // this head's images are complete; nobody reads the other buffer any more
        const int nbh = bh + (int)gridDim.x;
        if (nbh < BH) issue_kv(nbh, buf ^ 1);
        if (wave >= NT) continue;
        const char* Ks = smem + buf * 2 * IMG;
        const char* Vs = Ks + IMG;
        const int b = __builtin_amdgcn_readfirstlane(bh / H), h = bh - b * H;
        // S^T tiles: rows = keys, lane = query
        f32x16 st[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s)
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows(Ks, kt * 32 + l31, s, half), bq[s], st[kt], 0, 0, 0);
        }
        // the next head's query rows, into the registers the scores no longer need (in flight during softmax and P V)
        __builtin_amdgcn_sched_barrier(0);
        if (nbh < BH) own_rows64(bqn, head_base(nbh), rs, q, S, half);
        __builtin_amdgcn_sched_barrier(0);
        float m = kNegBig;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (kt == NT - 1 && kt * 32 + acc_row(r, half) >= S) st[kt][r] = kNegBig;  // (S > 32 * (NT - 1): only the last tile has padding)
                m = fmaxf(m, st[kt][r]);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float mneg = -m * kLog2e;
        float sum = 0.f;
        bf16x8 pa[NT][2];  // P as the bf16 A operands of the second product (half the registers of the fp32 scores)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], kLog2e, mneg));  // one FMA + v_exp_f32
                st[kt][r] = p;
                sum += p;
            }
            pa[kt][0] = acc_to_a(st[kt], 0);
            pa[kt][1] = acc_to_a(st[kt], 1);
        }
        sum += __shfl_xor(sum, 32, 64);
        f32x16 o[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[kt][blk], frag_cols(Vs, kt * 32 + blk * 16, dt * 32, lane),
                                                                   o[dt], 0, 0, 0);
            }
        const float inv = 1.f / sum;
        // a wave with a full tile issues 33 stores here, after the next head's DMA: "at most 33 outstanding" at the top of
        // the loop therefore means that DMA has landed (the count may only be exceeded, never undercut: the buffer
        // stores below are issued for every row, rows past S are dropped by the descriptor)
        if (q < S && half == 0) lse[((long)b * H + h) * S + q] = m + __logf(sum);
        const __amdgpu_buffer_rsrc_t ro = make_rsrc_uniform(out + ((long)b * S * H + h) * HD, (uint32_t)((long)(S - 1) * D * 2 + ROWB));
        const uint32_t ov = (uint32_t)(wave * 32 + 4 * half) * (uint32_t)(D * 2) + (uint32_t)(l31 * 2);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float invr = __shfl(inv, acc_row(r, half), 64);
            const uint32_t so = (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)(D * 2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) store_bf16(ro, ov, so + dt * 64, o[dt][r] * invr);
        }
    }
}

template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void k_attn_bwd_p(const bf16* __restrict__ qkv, const bf16* __restrict__ out,
                                                        const bf16* __restrict__ dout, const float* __restrict__ lse,
                                                        bf16* __restrict__ dqkv, int S, int H, int BH) {
    constexpr int SP = NT * 32, IMG = SP * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;                       // raw (the 1/8 goes onto the fp32 scores)
    char* Os = Qs + IMG;                   // dO
    char* Zs = Os + IMG;                   // O (for D only)
    char* Ks = Zs + IMG;
    char* Vs = Ks + IMG;
    float* Ls = reinterpret_cast<float*>(Vs + IMG);  // -8 lse[q]   (hugely negative on padding rows: P = 0 there)
    float* Ds = Ls + SP;                             // -D[q], D[q] = sum_d dO * O
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long D = (long)H * HD, rs = 3 * D;
    const uint32_t lds0 = lds_address(smem);
    const int own = wave * 32 + l31;  // this wave's key (first pass) / query (second pass) row
    auto head_base = [&](int bh) {
        const int hb = __builtin_amdgcn_readfirstlane(bh / H);  // (uniform, and now the compiler knows: scalar addresses, no waterfall)
        return qkv + (long)hb * S * rs + (bh - hb * H) * HD;
    };
    auto issue_x = [&](int bh) {
        const int hb = __builtin_amdgcn_readfirstlane(bh / H);
        const long o = ((long)hb * S * H + (bh - hb * H)) * HD;
        stage_dma<NT, NW>(lds0, head_base(bh), rs, S, wave, lane);
        stage_dma<NT, NW>(lds0 + (uint32_t)IMG, dout + o, D, S, wave, lane);
        stage_dma<NT, NW>(lds0 + (uint32_t)(2 * IMG), out + o, D, S, wave, lane);
    };
    auto issue_y = [&](int bh) {
        stage_dma<NT, NW>(lds0 + (uint32_t)(3 * IMG), head_base(bh) + D, rs, S, wave, lane);
        stage_dma<NT, NW>(lds0 + (uint32_t)(4 * IMG), head_base(bh) + 2 * D, rs, S, wave, lane);
    };
    int bh = blockIdx.x;
    if (bh >= BH) return;
    bf16x8 kfn[4], vfn[4];
    float ln = 0.f;
    auto prefetch = [&](int hd) {
        if (wave < NT) {
            own_rows64(kfn, head_base(hd) + D, rs, own, S, half);
            own_rows64(vfn, head_base(hd) + 2 * D, rs, own, S, half);
        }
        if (tid < SP) ln = tid < S ? lse[(long)hd * S + tid] : -kNegBig;
    };
    issue_x(bh);
    prefetch(bh);
    bool stored = false;  // (first head: nothing was issued after its DMA)
    for (; bh < BH; bh += gridDim.x) {
        bf16x8 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            kf[s] = kfn[s];
            vf[s] = vfn[s];
        }
        float lcur = ln;
        if (wave < NT) {
            pin(kf);
            pin(vf);
        }
        asm volatile("" : "+v"(lcur)::"memory");
        // second-pass stores of the previous head (32 per wave with a full tile) came after this head's DMA and rows
        if (stored && wave < NT - 1) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stored = true;
        __builtin_amdgcn_s_barrier();  // Q / dO / O images complete; the K / V images are free
        issue_y(bh);
        if (tid < SP) {
            float d = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {  // same row, same physical chunk of both images = same logical chunk
                const int cc = ((c + tid) & 7) * 16;  // (lanes are rows 128 B apart: rotate the chunk so that 8 lanes cover the 32 banks)
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(Zs + tid * ROWB + cc);
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(Os + tid * ROWB + cc);
#pragma unroll
                for (int k = 0; k < 8; ++k) d += (float)a[k] * (float)g[k];
            }
            Ls[tid] = -8.f * lcur;  // the score accumulators START at -8 lse[q] and -D[q]: P = exp2(c * acc) with
            Ds[tid] = -d;           // c = log2(e) / 8, dS = P * acc', no subtraction and no L / D registers in the loops
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int b = __builtin_amdgcn_readfirstlane(bh / H), h = bh - b * H;
        // this head's rows of dqkv: row S (the next image, or the end of the tensor) starts past the range
        const __amdgpu_buffer_rsrc_t rd = make_rsrc_uniform(dqkv + (long)b * S * rs + h * HD, (uint32_t)((long)(S - 1) * rs * 2 + D * 4 + ROWB));
        const uint32_t dvo = (uint32_t)(wave * 32 + 4 * half) * (uint32_t)(rs * 2) + (uint32_t)(l31 * 2);
        bf16x8 qf[4], gf[4];

        // ---- first pass: this wave's KEY tile; scores non-transposed (lane = key, regs = queries) -> dK, dV
        if (wave < NT) {
            f32x16 dk[2], dv[2];
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    dk[dt][r] = 0.f;
                    dv[dt][r] = 0.f;
                }
#pragma unroll 1
            for (int qt = 0; qt < NT; ++qt) {
                // every LDS read of the first two products, and this tile's L / D, in flight before the first MFMA
                // (left alone, the compiler reads one fragment, waits, issues one MFMA: the loop then runs at LDS latency)
                bf16x8 qa[4], ga[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    qa[s] = frag_rows(Qs, qt * 32 + l31, s, half);
                    ga[s] = frag_rows(Os, qt * 32 + l31, s, half);
                }
                f32x16 sc, dp;  // registers 4j .. 4j+3 of a tile are queries 8j + 4*half + 0..3
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(Ls + qt * 32 + 8 * jj + 4 * half);
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(Ds + qt * 32 + 8 * jj + 4 * half);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sc[4 * jj + e] = l4[e];
                        dp[4 * jj + e] = d4[e];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa[s], kf[s], sc, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s], vf[s], dp, 0, 0, 0);
                }
                // the transposed fragments of the second products land while the matrix pipe and the exponentials run
                bf16x8 oc[2][2], qc[2][2];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        oc[blk][dt] = frag_cols(Os, qt * 32 + blk * 16, dt * 32, lane);
                        qc[blk][dt] = frag_cols(Qs, qt * 32 + blk * 16, dt * 32, lane);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // (no key mask: a padded key only feeds its own dK / dV rows, which are never stored; padded queries
                    // have zero Q / dO rows and a huge Ls)
                    const float p = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(sc[r] * (0.125f * kLog2e)), 0.f, 1.f);  // (clamp: free on v_exp_f32)
                    sc[r] = p;           // P
                    dp[r] = p * dp[r];   // dS = P (dP - D)
                }
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const bf16x8 pa = acc_to_a(sc, blk), dsa = acc_to_a(dp, blk);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, oc[blk][dt], dv[dt], 0, 0, 0);
                        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, qc[blk][dt], dk[dt], 0, 0, 0);
                    }
                }
            }
            // this wave's own query rows for the second pass, while the Q / dO images are still there
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                qf[s] = frag_rows(Qs, own, s, half);
                gf[s] = frag_rows(Os, own, s, half);
            }
            pin(qf);
            pin(gf);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // K / V images (issued a whole pass ago), before the stores below
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t so = (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)(rs * 2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    store_bf16(rd, dvo, so + (uint32_t)(D * 2) + dt * 64, dk[dt][r] * 0.125f);
                    store_bf16(rd, dvo, so + (uint32_t)(D * 4) + dt * 64, dv[dt][r]);
                }
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // K / V complete for every wave; nobody reads the Q / dO / O images any more
        const int nbh = bh + (int)gridDim.x;
        if (nbh < BH) {
            issue_x(nbh);
            prefetch(nbh);
        }
        if (wave >= NT) continue;

        // ---- second pass: this wave's QUERY tile; scores transposed (lane = query, regs = keys) -> dQ
        const float lq = Ls[own], dq_row = Ds[own];  // (-8 lse, -D)
        f32x16 dq[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
#pragma unroll 1
        for (int kt = 0; kt < NT; ++kt) {
            bf16x8 ka[4], va[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ka[s] = frag_rows(Ks, kt * 32 + l31, s, half);
                va[s] = frag_rows(Vs, kt * 32 + l31, s, half);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = lq;
                dp[r] = dq_row;
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[s], qf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[s], gf[s], dp, 0, 0, 0);
            }
            bf16x8 kc[2][2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) kc[blk][dt] = frag_cols(Ks, kt * 32 + blk * 16, dt * 32, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                // No key mask, but P is clamped to [0, 1]: the K rows of padded keys are zeros, so whatever FINITE dS
                // they get adds nothing to dQ -- their score is 0, though, and with lse < -88 (scores that are all
                // strongly negative) exp(0 - lse) overflows to inf, inf * 0 = NaN (found by a diverging ViT run,
                // profiles/r3_convergence_vit.md).  For real keys P <= 1 up to rounding; the clamp rides on v_exp_f32.
                const float p = __builtin_amdgcn_fmed3f(__builtin_amdgcn_exp2f(sc[r] * (0.125f * kLog2e)), 0.f, 1.f);
                dp[r] = p * dp[r];
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 dsa = acc_to_a(dp, blk);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, kc[blk][dt], dq[dt], 0, 0, 0);
            }
        }
        // 32 stores per wave, issued for every row (rows past S are dropped by the descriptor): the counted wait at the top
        // of the loop needs at least that many operations behind the next head's DMA
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t so = (uint32_t)((r & 3) + 8 * (r >> 2)) * (uint32_t)(rs * 2);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) store_bf16(rd, dvo, so + dt * 64, dq[dt][r] * 0.125f);
        }
    }
}

// ============================================================================= any length, head_dim 32 / 64 / 128
// The kernels above keep a head's whole K / V (and its score row) on chip, which stops at 256 tokens and is built for
// head_dim 64.  The kernels below are the general path (ViT at 384 x 384: 577 tokens; ViT-Tiny: 3 heads of 64 ...
// head_dim 32 / 128 models): a workgroup owns 128 queries (or keys) of one head, one 32-row tile per wave, and the
// OTHER axis streams through LDS in blocks of 64 rows, so the sequence length is unbounded and LDS use is fixed
// (<= 32 KiB + 512 B).  Same MFMA dataflow as above (scores as accumulators feeding the next MFMA from registers).
//  * forward in two sweeps over the keys: sweep 1 = running row max + rescaled row sum (scalars per lane, no
//    accumulator rescaling, no data-dependent branch), sweep 2 = P = exp(s - lse) already normalised, O += P V.
//    1.5x the matrix work of a one-sweep online softmax, on a path whose job is generality.
//  * backward as two kernels (key-owner: dK, dV; query-owner: dQ) with D[q] = sum_d dO * O from a small pre-pass.
//  * the softmax scale is applied to the fp32 scores (1/sqrt(32) is not a bf16 number).
template <int HDT>
struct AttnGeo {
    static constexpr int ROWBYTES = HDT * 2, NCH = HDT / 8, STEPS = HDT / 16, DT = HDT / 32;
    static __device__ __forceinline__ int swz(int row, int chunk) {
        const int x = ((((row >> 1) & 1) << 2) | ((row >> 2) & 3)) & (NCH - 1);
        return row * ROWBYTES + ((chunk ^ x) << 4);
    }
    static __device__ __forceinline__ bf16x8 rows(const char* img, int row, int step, int half) {
        return *reinterpret_cast<const bf16x8*>(img + swz(row, 2 * step + half));
    }
    static __device__ __forceinline__ bf16x8 cols(const char* img, int r0, int colbase, int lane) {
        const int q4 = (lane & 15) >> 2, p = lane & 3, h16 = (lane >> 4) & 1, half = lane >> 5;
        const int col = colbase + h16 * 16 + p * 4;
        const int ra = r0 + 4 * half + q4, rb = ra + 8;
        const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + swz(ra, col >> 3) + (col & 7) * 2));
        const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(img + swz(rb, col >> 3) + (col & 7) * 2));
        return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // rows [r0, r0 + 64) of a [*, row_stride] tensor -> LDS image rows 0..63 (zeros past S); 256 threads
    static __device__ __forceinline__ void stage64(char* img, const bf16* src, long row_stride, int r0, int S, int tid) {
        // every load of the block is issued before the first LDS store (a load -> wait -> store loop pays the memory
        // latency once per chunk); rows past S read row S - 1 and are zeroed by a select, so no load sits behind a branch
        constexpr int IT = 64 * NCH / 256;
        static_assert(IT >= 1 && 64 * NCH % 256 == 0, "a block is a whole number of chunks per thread");
        bf16x8 v[IT];
#pragma unroll
        for (int t = 0; t < IT; ++t) {
            const int i = tid + 256 * t, row = i / NCH, ch = i % NCH;
            const int rr = r0 + row < S ? r0 + row : S - 1;
            v[t] = *reinterpret_cast<const bf16x8*>(src + (long)rr * row_stride + ch * 8);
        }
#pragma unroll
        for (int t = 0; t < IT; ++t) {
            const int i = tid + 256 * t, row = i / NCH, ch = i % NCH;
            if (r0 + row >= S) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[t][k] = (bf16)0.f;
            }
            *reinterpret_cast<bf16x8*>(img + swz(row, ch)) = v[t];
        }
    }
    // a wave's own 32 rows as MFMA fragments straight from global memory (lane = row, 8 elements per step and half)
    static __device__ __forceinline__ void own_rows(bf16x8* f, const bf16* src, long row_stride, int row, int S, int half) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
#pragma unroll
            for (int k = 0; k < 8; ++k) f[s][k] = (bf16)0.f;
            if (row < S) f[s] = *reinterpret_cast<const bf16x8*>(src + (long)row * row_stride + s * 16 + half * 8);
        }
    }
};

template <int HDT>
__global__ __launch_bounds__(256) void k_attn_fwd_kb(const bf16* __restrict__ qkv, bf16* __restrict__ out,
                                                     float* __restrict__ lse, int S, int H, float scale) {
    using G = AttnGeo<HDT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + 64 * G::ROWBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const long D = (long)H * HDT, rs = 3 * D;
    const bf16* base = qkv + (long)b * S * rs + h * HDT;
    const int q0 = blockIdx.x * 128 + wave * 32, q = q0 + l31;
    bf16x8 bq[G::STEPS];
    G::own_rows(bq, base, rs, q, S, half);
    const int nkb = (S + 63) / 64;

    // sweep 1: row max and row sum
    float m = kNegBig, sum = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        G::stage64(Ks, base + D, rs, kb * 64, S, tid);
        __syncthreads();
        f32x16 st[2];
        float mb = kNegBig;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < G::STEPS; ++s)
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G::rows(Ks, kt * 32 + l31, s, half), bq[s], st[kt], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[kt][r] = kb * 64 + kt * 32 + acc_row(r, half) < S ? st[kt][r] * scale : kNegBig;
                mb = fmaxf(mb, st[kt][r]);
            }
        }
        mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
        const float mn = fmaxf(m, mb);
        float part = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) part += __expf(st[kt][r] - mn);
        sum = sum * __expf(m - mn) + part;
        m = mn;
    }
    sum += __shfl_xor(sum, 32, 64);
    const float lq = m + __logf(sum);
    if (q < S && half == 0) lse[((long)b * H + h) * S + q] = lq;

    // sweep 2: O = softmax(S) V with the final normaliser
    f32x16 o[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        G::stage64(Ks, base + D, rs, kb * 64, S, tid);
        G::stage64(Vs, base + 2 * D, rs, kb * 64, S, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 st;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
            for (int s = 0; s < G::STEPS; ++s)
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G::rows(Ks, kt * 32 + l31, s, half), bq[s], st, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                st[r] = kb * 64 + kt * 32 + acc_row(r, half) < S ? __expf(st[r] * scale - lq) : 0.f;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 pa = acc_to_a(st, blk);
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, G::cols(Vs, kt * 32 + blk * 16, dt * 32, lane), o[dt], 0, 0, 0);
            }
        }
    }
    bf16* orow = out + ((long)b * S * H + h) * HDT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = q0 + acc_row(r, half);
        if (qq < S) {
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) orow[(long)qq * D + dt * 32 + l31] = (bf16)o[dt][r];
        }
    }
}

// D[b, h, q] = sum_d dO * O; one thread per (b, q, h)
template <int HDT>
__global__ __launch_bounds__(256) void k_attn_dsum(const bf16* __restrict__ out, const bf16* __restrict__ dout,
                                                   float* __restrict__ dsum, long total, int S, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (b * S + q) * H + h
    if (i >= total) return;
    const int h = (int)(i % H);
    const long bq = i / H;
    const int q = (int)(bq % S);
    const long b = bq / S;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < HDT / 8; ++c) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(out + i * HDT + c * 8);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dout + i * HDT + c * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) d += (float)a[k] * (float)g[k];
    }
    dsum[(b * H + h) * S + q] = d;
}

// a wave owns 32 KEYS; queries stream through LDS -> dK, dV
template <int HDT>
__global__ __launch_bounds__(256) void k_attn_bwd_kv(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                     const float* __restrict__ lse, const float* __restrict__ dsum,
                                                     bf16* __restrict__ dqkv, int S, int H, float scale) {
    using G = AttnGeo<HDT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Qs = smem;
    char* Os = smem + 64 * G::ROWBYTES;
    float* Ls = reinterpret_cast<float*>(Os + 64 * G::ROWBYTES);
    float* Ds = Ls + 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const long D = (long)H * HDT, rs = 3 * D;
    const bf16* base = qkv + (long)b * S * rs + h * HDT;
    const bf16* dobase = dout + ((long)b * S * H + h) * HDT;
    const float* lrow = lse + ((long)b * H + h) * S;
    const float* drow = dsum + ((long)b * H + h) * S;
    const int k0 = blockIdx.x * 128 + wave * 32, key = k0 + l31;
    bf16x8 kf[G::STEPS], vf[G::STEPS];
    G::own_rows(kf, base + D, rs, key, S, half);
    G::own_rows(vf, base + 2 * D, rs, key, S, half);
    f32x16 dk[G::DT], dv[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            dk[dt][r] = 0.f;
            dv[dt][r] = 0.f;
        }
    const int nqb = (S + 63) / 64;
    for (int qb = 0; qb < nqb; ++qb) {
        __syncthreads();
        G::stage64(Qs, base, rs, qb * 64, S, tid);
        G::stage64(Os, dobase, D, qb * 64, S, tid);
        if (tid < 64) {
            const int q = qb * 64 + tid;
            Ls[tid] = q < S ? lrow[q] : -kNegBig;  // huge on padding rows: P = 0 there
            Ds[tid] = q < S ? drow[q] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = 0.f;
                dp[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < G::STEPS; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G::rows(Qs, qt * 32 + l31, s, half), kf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G::rows(Os, qt * 32 + l31, s, half), vf[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = qt * 32 + acc_row(r, half);
                const float p = key < S ? __expf(sc[r] * scale - Ls[ql]) : 0.f;
                sc[r] = p;
                dp[r] = p * (dp[r] - Ds[ql]);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 pa = acc_to_a(sc, blk), dsa = acc_to_a(dp, blk);
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt) {
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, G::cols(Os, qt * 32 + blk * 16, dt * 32, lane), dv[dt], 0, 0, 0);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, G::cols(Qs, qt * 32 + blk * 16, dt * 32, lane), dk[dt], 0, 0, 0);
                }
            }
        }
    }
    bf16* dbase = dqkv + (long)b * S * rs + h * HDT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kk = k0 + acc_row(r, half);
        if (kk < S) {
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) {
                dbase[(long)kk * rs + D + dt * 32 + l31] = (bf16)(dk[dt][r] * scale);
                dbase[(long)kk * rs + 2 * D + dt * 32 + l31] = (bf16)dv[dt][r];
            }
        }
    }
}

// a wave owns 32 QUERIES; keys stream through LDS -> dQ
template <int HDT>
__global__ __launch_bounds__(256) void k_attn_bwd_q(const bf16* __restrict__ qkv, const bf16* __restrict__ dout,
                                                    const float* __restrict__ lse, const float* __restrict__ dsum,
                                                    bf16* __restrict__ dqkv, int S, int H, float scale) {
    using G = AttnGeo<HDT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + 64 * G::ROWBYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
    const int b = blockIdx.y / H, h = blockIdx.y % H;
    const long D = (long)H * HDT, rs = 3 * D;
    const bf16* base = qkv + (long)b * S * rs + h * HDT;
    const bf16* dobase = dout + ((long)b * S * H + h) * HDT;
    const int q0 = blockIdx.x * 128 + wave * 32, q = q0 + l31;
    bf16x8 qf[G::STEPS], gf[G::STEPS];
    G::own_rows(qf, base, rs, q, S, half);
    G::own_rows(gf, dobase, D, q, S, half);
    const float lq = q < S ? lse[((long)b * H + h) * S + q] : -kNegBig;
    const float dq_row = q < S ? dsum[((long)b * H + h) * S + q] : 0.f;
    f32x16 dq[G::DT];
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    const int nkb = (S + 63) / 64;
    for (int kb = 0; kb < nkb; ++kb) {
        __syncthreads();
        G::stage64(Ks, base + D, rs, kb * 64, S, tid);
        G::stage64(Vs, base + 2 * D, rs, kb * 64, S, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            f32x16 sc, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sc[r] = 0.f;
                dp[r] = 0.f;
            }
#pragma unroll
            for (int s = 0; s < G::STEPS; ++s) {
                sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G::rows(Ks, kt * 32 + l31, s, half), qf[s], sc, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(G::rows(Vs, kt * 32 + l31, s, half), gf[s], dp, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kk = kb * 64 + kt * 32 + acc_row(r, half);
                const float p = kk < S ? __expf(sc[r] * scale - lq) : 0.f;
                dp[r] = p * (dp[r] - dq_row);
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 dsa = acc_to_a(dp, blk);
#pragma unroll
                for (int dt = 0; dt < G::DT; ++dt)
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dsa, G::cols(Ks, kt * 32 + blk * 16, dt * 32, lane), dq[dt], 0, 0, 0);
            }
        }
    }
    bf16* dbase = dqkv + (long)b * S * rs + h * HDT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int qq = q0 + acc_row(r, half);
        if (qq < S) {
#pragma unroll
            for (int dt = 0; dt < G::DT; ++dt) dbase[(long)qq * rs + dt * 32 + l31] = (bf16)(dq[dt][r] * scale);
        }
    }
}

template <int HDT>
void launch_fwd_kb(const void* qkv, void* out, float* lse, int B, int S, int H, hipStream_t s) {
    const size_t lds = (size_t)2 * 64 * HDT * 2;
    hipLaunchKernelGGL((k_attn_fwd_kb<HDT>), dim3((S + 127) / 128, B * H), dim3(256), lds, s, (const bf16*)qkv, (bf16*)out, lse,
                       S, H, 1.f / sqrtf((float)HDT));
}
template <int HDT>
void launch_bwd_kb(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dsum, int B,
                   int S, int H, hipStream_t s) {
    const long total = (long)B * S * H;
    const float scale = 1.f / sqrtf((float)HDT);
    hipLaunchKernelGGL((k_attn_dsum<HDT>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const bf16*)out,
                       (const bf16*)dout, dsum, total, S, H);
    const size_t lds_kv = (size_t)2 * 64 * HDT * 2 + 2 * 64 * sizeof(float), lds_q = (size_t)2 * 64 * HDT * 2;
    const dim3 grid((S + 127) / 128, B * H);
    hipLaunchKernelGGL((k_attn_bwd_kv<HDT>), grid, dim3(256), lds_kv, s, (const bf16*)qkv, (const bf16*)dout, lse,
                       (const float*)dsum, (bf16*)dqkv, S, H, scale);
    hipLaunchKernelGGL((k_attn_bwd_q<HDT>), grid, dim3(256), lds_q, s, (const bf16*)qkv, (const bf16*)dout, lse,
                       (const float*)dsum, (bf16*)dqkv, S, H, scale);
}

int g_pipelined = [] {  // PDT_ATTN_PIPE=0: the one-head-per-workgroup kernels for every whole-head launch
    const char* e = getenv("PDT_ATTN_PIPE");
    return e && e[0] == '0' ? 0 : 1;
}();
bool pipelined() { return g_pipelined != 0; }
int cu_count() {
    static const int n = [] {
        int dev = 0, v = 256;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) v = 256;
        return v;
    }();
    return n;
}

bool force_generic() {
    static const bool on = [] {
        const char* e = getenv("PDT_ATTN_GENERIC");
        return e && e[0] == '1';
    }();
    return on;
}

template <typename K>
void set_lds(K kern, size_t bytes) {
    PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)bytes));
}

}  // namespace

void set_attention_pipelined(bool on) { g_pipelined = on ? 1 : 0; }
bool attention_shape_ok(int S, int head_dim) {
    return (head_dim == 32 || head_dim == 64 || head_dim == 128) && S >= 1 && (long)S < (1L << 24);
}
// the whole-head kernels (K / V resident in LDS, one sweep) take what they were built for; everything else streams
static bool whole_head(int S, int head_dim) { return head_dim == HD && S <= 256 && !force_generic(); }
bool attention_needs_dsum(int S, int head_dim) { return !whole_head(S, head_dim); }

void launch_attention_fwd(const void* qkv, void* out, float* lse, int B, int S, int H, int head_dim, hipStream_t s) {
    if (!attention_shape_ok(S, head_dim) || (long)B * H > 65535) abort();  // (the binding checks both)
    if (!whole_head(S, head_dim)) {
        if (head_dim == 32) launch_fwd_kb<32>(qkv, out, lse, B, S, H, s);
        else if (head_dim == 64) launch_fwd_kb<64>(qkv, out, lse, B, S, H, s);
        else launch_fwd_kb<128>(qkv, out, lse, B, S, H, s);
        return;
    }
    const int nt = (S + 31) / 32;
    if (nt >= 5 && pipelined()) {  // 8-wave heads: persistent workgroups, LDS-DMA, the next head in flight
        const size_t lds2 = (size_t)4 * nt * 32 * ROWB;
        const int BH = B * H, grid = BH < cu_count() ? BH : cu_count();
#define PDT_ATTN_FWD_P(N)                                                                                     \
    case N: {                                                                                                 \
        static bool once = (set_lds(&k_attn_fwd_p<N, 8>, lds2), true);                                        \
        (void)once;                                                                                           \
        hipLaunchKernelGGL((k_attn_fwd_p<N, 8>), dim3(grid), dim3(512), lds2, s, (const bf16*)qkv, (bf16*)out, lse, S, H, BH); \
    } break;
        switch (nt) {
            PDT_ATTN_FWD_P(5) PDT_ATTN_FWD_P(6) PDT_ATTN_FWD_P(7) PDT_ATTN_FWD_P(8)
            default: abort();
        }
#undef PDT_ATTN_FWD_P
        return;
    }
    const size_t lds = (size_t)2 * nt * 32 * ROWB;
#define PDT_ATTN_FWD(N)                                                                                        \
    case N: {                                                                                                  \
        constexpr int NW = N > 4 ? 8 : 4;                                                                      \
        static bool once = (set_lds(&k_attn_fwd<N, NW>, lds), true);                                           \
        (void)once;                                                                                            \
        hipLaunchKernelGGL((k_attn_fwd<N, NW>), dim3(B * H), dim3(64 * NW), lds, s, (const bf16*)qkv, (bf16*)out, lse, S, H); \
    } break;
    switch (nt) {
        PDT_ATTN_FWD(1) PDT_ATTN_FWD(2) PDT_ATTN_FWD(3) PDT_ATTN_FWD(4) PDT_ATTN_FWD(5) PDT_ATTN_FWD(6) PDT_ATTN_FWD(7)
        PDT_ATTN_FWD(8)
        default: abort();
    }
#undef PDT_ATTN_FWD
}

void launch_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* dsum,
                          int B, int S, int H, int head_dim, hipStream_t s) {
    if (!attention_shape_ok(S, head_dim) || (long)B * H > 65535) abort();
    if (!whole_head(S, head_dim)) {
        if (dsum == nullptr) abort();
        if (head_dim == 32) launch_bwd_kb<32>(qkv, out, dout, lse, dqkv, dsum, B, S, H, s);
        else if (head_dim == 64) launch_bwd_kb<64>(qkv, out, dout, lse, dqkv, dsum, B, S, H, s);
        else launch_bwd_kb<128>(qkv, out, dout, lse, dqkv, dsum, B, S, H, s);
        return;
    }
    const int nt = (S + 31) / 32;
    if (nt >= 5 && nt <= 7 && pipelined()) {  // (8 tiles: five images do not fit the LDS)
        const size_t lds5 = (size_t)5 * nt * 32 * ROWB + (size_t)2 * nt * 32 * sizeof(float);
        const int BH = B * H, grid = BH < cu_count() ? BH : cu_count();
#define PDT_ATTN_BWD_P(N)                                                                                     \
    case N: {                                                                                                 \
        static bool once = (set_lds(&k_attn_bwd_p<N, 8>, lds5), true);                                        \
        (void)once;                                                                                           \
        hipLaunchKernelGGL((k_attn_bwd_p<N, 8>), dim3(grid), dim3(512), lds5, s, (const bf16*)qkv, (const bf16*)out, \
                           (const bf16*)dout, lse, (bf16*)dqkv, S, H, BH);                                    \
    } break;
        switch (nt) {
            PDT_ATTN_BWD_P(5) PDT_ATTN_BWD_P(6) PDT_ATTN_BWD_P(7)
            default: abort();
        }
#undef PDT_ATTN_BWD_P
        return;
    }
    const size_t lds = (size_t)4 * nt * 32 * ROWB + (size_t)2 * nt * 32 * sizeof(float);
#define PDT_ATTN_BWD(N)                                                                                      \
    case N: {                                                                                                \
        constexpr int NW = N > 4 ? 8 : 4;                                                                    \
        static bool once = (set_lds(&k_attn_bwd<N, NW>, lds), true);                                         \
        (void)once;                                                                                          \
        hipLaunchKernelGGL((k_attn_bwd<N, NW>), dim3(B * H), dim3(64 * NW), lds, s, (const bf16*)qkv, (const bf16*)out, \
                           (const bf16*)dout, lse, (bf16*)dqkv, S, H);                                       \
    } break;
    switch (nt) {
        PDT_ATTN_BWD(1) PDT_ATTN_BWD(2) PDT_ATTN_BWD(3) PDT_ATTN_BWD(4) PDT_ATTN_BWD(5) PDT_ATTN_BWD(6) PDT_ATTN_BWD(7)
        PDT_ATTN_BWD(8)
        default: abort();
    }
#undef PDT_ATTN_BWD
}

}  // namespace pdt
