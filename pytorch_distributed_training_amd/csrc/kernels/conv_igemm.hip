// Implicit-GEMM convolution on the gfx950 matrix cores.
//
//   Out[b, i*os+ooh, j*os+oow, n] = sum_t sum_c In[b, i*is+dh[t], j*is+dw[t], c] * Wt[n][wslot[t]][c]
//
// One kernel serves the forward conv (In = x, taps = (r,s), is = stride) and the data gradient
// (In = dY, Wt = transposed weights, taps = the (r,s) that hit this output-parity class), see the
// host wrappers.  GEMM view: M = B*nOH*nOW output pixels, N = output channels, K = taps*C.
//
// Design for CDNA4 (wave64, 4 SIMDs/CU, 160 KiB LDS, MFMA 32x32x16 bf16):
//  * NHWC bf16 activations and [N][taps][C] weights make BOTH operands K-contiguous: every lane
//    moves 16 B (8 channels) per load, a tile row is one 128 B line (BK = 64 channels).
//  * zero padding comes from the buffer descriptor's range check (out-of-image taps get an
//    out-of-range offset and load zeros): no branches in the load path.
//  * 256 threads = 4 waves; block tile BM x BN (128x128 or 128x64), each wave owns a grid of
//    32x32 MFMA tiles with fp32 accumulators in registers.
//  * global -> registers -> LDS staging, double buffered: the loads of K-step k+1 are issued
//    before the MFMAs of step k and written to the other LDS buffer after them; one barrier
//    per K-step.  LDS rows are 128 B with the 16-B chunk XOR-swizzled by (row>>1)&7 so the
//    ds_read_b128 fragment reads are bank-conflict free.
//  * epilogue: accumulators are rounded to bf16, per-channel [sum, sum_sq] of the rounded
//    values are reduced in registers/LDS and written to a per-m-tile slab (BatchNorm
//    statistics for free), and the tile is staged through LDS so global stores are 16 B per
//    lane on full rows.
//  * block -> tile map is XCD-aware: the 8 XCDs get contiguous tile ranges, n-tiles fastest,
//    so blocks sharing an activation tile hit the same 4 MiB L2.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

// LDS rows hold BK channels (ROWB = 2*BK bytes).  A ds_read_b128 lane group covers 16 rows at one
// chunk; rows are spread over the 256-B bank row by XOR-ing the chunk with (row / rows_per_bank_row).
template <int BK>
__device__ __forceinline__ int lds_off(int row, int chunk) {
    constexpr int ROWB = BK * 2, CPR = ROWB / 16, RPB = 256 / ROWB;
    return row * ROWB + ((chunk ^ ((row / RPB) & (CPR - 1))) << 4);
}

// NS == 0: operand tiles staged through registers, two LDS buffers (prefetch distance one K-step).
// NS >= 3: LDS-DMA staging (buffer_load ... lds: no VGPR round trip, no ds_write) into a ring of NS stage
//          buffers with NS-1 K-steps in flight: the loads of step k+NS-1 are issued before the MFMAs of step k
//          and only a COUNTED s_waitcnt vmcnt retires step k+1 before the (raw) barrier, so global / L2 latency
//          spans several K-steps instead of one.  The DMA writes LDS lane-linearly (wave base + 16 B * lane), so
//          the XOR swizzle is applied on the SOURCE side: the lane that fills physical chunk p of a row fetches
//          logical chunk p ^ swz(row).
template <int BM, int BN, int WM, int WN, int BK, int MINW, int NS = 0>
__global__ __launch_bounds__(256, MINW) void k_conv_igemm(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                       bf16* __restrict__ y, const ConvGeom g,
                                                       float* __restrict__ stats_slab, int m_tiles, int n_tiles,
                                                       uint32_t x_bytes, uint32_t w_bytes, int xcd_group) {
    static_assert(WM * WN == 4, "4 waves per block");
    constexpr int ROWB = BK * 2;           // bytes per LDS row
    constexpr int CPR = ROWB / 16;         // 16-B chunks per row = threads staging one row
    constexpr int RPP = 256 / CPR;         // rows staged per pass
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PA = BM / RPP, PB = BN / RPP;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                  // [2][BM][128 B]
    char* Bs = smem + 2 * A_BYTES;    // [2][BN][128 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int lrow = tid / CPR, chunk = tid % CPR;

    // XCD-aware, bijective block -> tile map (blocks b and b+8 share an XCD)
    const int total = m_tiles * n_tiles;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        if (xcd_group > 0) {
            // groups of xcd_group consecutive tiles dealt round-robin to the XCDs: blocks that share operand
            // rows stay on one L2, and the chip as a whole walks the tensors front to back (one DRAM window)
            const int full = total / (8 * xcd_group) * (8 * xcd_group);
            tile = bid < full ? ((slot / xcd_group) * 8 + xcd) * xcd_group + slot % xcd_group : bid;
        } else {
            const int q = total >> 3, r = total & 7;
            tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        }
    }
    const int tile_n = tile % n_tiles, tile_m = tile / n_tiles;
    const long M = (long)g.B * g.nOH * g.nOW;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(x, x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(w, w_bytes);

    // per-row gather state of the A rows this thread stages
    int rowpix[PA], rih[PA], riw[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const long m = (long)tile_m * BM + p * RPP + lrow;
        if (m < M) {
            const uint32_t t1 = fdiv((uint32_t)m, g.div_now);
            const int j = (int)((uint32_t)m - t1 * (uint32_t)g.nOW);
            const uint32_t bq = fdiv(t1, g.div_noh);
            const int i = (int)(t1 - bq * (uint32_t)g.nOH);
            const int b = (int)bq;
            rowpix[p] = b * g.Hin * g.Win;
            rih[p] = i * g.is;
            riw[p] = j * g.is;
        } else {
            rowpix[p] = 0;
            rih[p] = -100000;  // every tap lands outside the image -> zeros
            riw[p] = 0;
        }
    }
    uint32_t wrow[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int n = tile_n * BN + p * RPP + lrow;
        wrow[p] = (n < g.N) ? (uint32_t)n * (uint32_t)g.wtaps : 0xffffffffu;
    }

    const int pix_elems = g.pix_elems ? g.pix_elems : g.C;  // memory elements per input pixel
    bf16x8 ra[PA], rb[PB];
    auto gload = [&](int t, int c0) {
        const int tp = g.tap[__builtin_amdgcn_readfirstlane(t)];
        const int dh = (int)(int8_t)(tp & 0xff), dw = (int)(int8_t)((tp >> 8) & 0xff), ws = (tp >> 16) & 0xff;
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int hi = rih[p] + dh, wi = riw[p] + dw;
            const bool ok = (unsigned)hi < (unsigned)g.Hin && (unsigned)wi < (unsigned)g.Win;
            const uint32_t off = (uint32_t)(((rowpix[p] + hi * g.Win + wi) * pix_elems + c0 + chunk * 8) * 2);
            ra[p] = buffer_load_bf16x8(rx, ok ? off : kOOB);
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const uint32_t off = ((wrow[p] + (uint32_t)ws) * (uint32_t)g.C + (uint32_t)(c0 + chunk * 8)) * 2u;
            rb[p] = buffer_load_bf16x8(rw, wrow[p] != 0xffffffffu ? off : kOOB);
        }
    };
    auto lds_store = [&](int buf) {
#pragma unroll
        for (int p = 0; p < PA; ++p)
            *reinterpret_cast<bf16x8*>(As + buf * A_BYTES + lds_off<BK>(p * RPP + lrow, chunk)) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p)
            *reinterpret_cast<bf16x8*>(Bs + buf * B_BYTES + lds_off<BK>(p * RPP + lrow, chunk)) = rb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int cpt = g.C / BK;  // K-steps per tap
    const int nks = g.ntaps * cpt;
    const int arow0 = wm * (BM / WM) + (lane & 31);
    const int brow0 = wn * (BN / WN) + (lane & 31);
    const int khalf = lane >> 5;
    auto mfma_step = [&](const char* Ab, const char* Bb) {
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(Ab + lds_off<BK>(arow0 + i * 32, 2 * s + khalf));
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + lds_off<BK>(brow0 + j * 32, 2 * s + khalf));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
    };

    if constexpr (NS == 0) {
        int t_next = 0, c_next = 0;
        gload(0, 0);
        lds_store(0);
        __syncthreads();
        for (int ks = 0; ks < nks; ++ks) {
            const int cur = ks & 1;
            const bool more = ks + 1 < nks;
            if (more) {
                c_next += BK;
                if (c_next == g.C) {
                    c_next = 0;
                    ++t_next;
                }
                gload(t_next, c_next);
            }
            mfma_step(As + cur * A_BYTES, Bs + cur * B_BYTES);
            if (more) lds_store(cur ^ 1);
            __syncthreads();
        }
    } else {
        constexpr int STAGE = A_BYTES + B_BYTES;
        constexpr int RPB = 256 / ROWB;
        constexpr int LOADS = PA + PB;               // LDS-DMA instructions per thread and stage
        constexpr int INFLIGHT = (NS - 2) * LOADS;   // what may still be outstanding once step k+1 must have landed
        static_assert(INFLIGHT < 64, "vmcnt field");
        const i32x4 rxw = make_rsrc_words(x, x_bytes), rww = make_rsrc_words(w, w_bytes);
        const uint32_t lds0 = lds_address(smem) + (uint32_t)__builtin_amdgcn_readfirstlane(wave) * 1024u;
        int ti = 0, ci = 0, issued = 0;
        // Per-lane byte offsets of this thread's 16-B chunks at channel 0 of the CURRENT tap (kOOB: the row is
        // padding / past the end -> the DMA writes zeros).  They change only when the tap does; inside a tap the
        // K loop advances the scalar SOFFSET of the loads, so a K-step costs no vector address arithmetic.
        uint32_t aoff[PA], boff[PB];
        auto enter_tap = [&]() {
            // the tap index must stay in a scalar register: a vector index into the kernel-argument table would turn
            // this read into a global load whose compiler-inserted vmcnt wait also drains the DMA ring
            const int tp = g.tap[__builtin_amdgcn_readfirstlane(ti)];
            const int dh = (int)(int8_t)(tp & 0xff), dw = (int)(int8_t)((tp >> 8) & 0xff), ws = (tp >> 16) & 0xff;
#pragma unroll
            for (int p = 0; p < PA; ++p) {
                const int row = p * RPP + lrow;
                const int lchunk = chunk ^ ((row / RPB) & (CPR - 1));
                const int hi = rih[p] + dh, wi = riw[p] + dw;
                const bool ok = (unsigned)hi < (unsigned)g.Hin && (unsigned)wi < (unsigned)g.Win;
                aoff[p] = ok ? (uint32_t)(((rowpix[p] + hi * g.Win + wi) * pix_elems + lchunk * 8) * 2) : kOOB;
            }
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const int row = p * RPP + lrow;
                const int lchunk = chunk ^ ((row / RPB) & (CPR - 1));
                boff[p] = wrow[p] != 0xffffffffu
                              ? ((wrow[p] + (uint32_t)ws) * (uint32_t)g.C + (uint32_t)(lchunk * 8)) * 2u
                              : kOOB;
            }
        };
        // issue the DMA loads of K-step `issued` into ring slot issued % NS
        auto gissue = [&]() {
            const uint32_t sa = lds0 + (uint32_t)((issued % NS) * STAGE);
            if (ci == 0) enter_tap();
            const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(ci * 2);
#pragma unroll
            for (int p = 0; p < PA; ++p) lds_dma16s(rxw, aoff[p], soff, sa + (uint32_t)(p * RPP * ROWB));
#pragma unroll
            for (int p = 0; p < PB; ++p) lds_dma16s(rww, boff[p], soff, sa + (uint32_t)(A_BYTES + p * RPP * ROWB));
            ++issued;
            ci += BK;
            if (ci == g.C) {
                ci = 0;
                ++ti;
            }
        };
        // retire the oldest stage still in flight: `ahead` = stages issued AFTER the one that must have landed
        // (counted waits need literal operands; the tail of the K loop walks down the chain)
        auto retire = [&](int ahead) {
            if (NS >= 5 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * LOADS) : "memory");
            else if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        static_assert(NS <= 5 && 3 * LOADS < 64, "retire() chain / vmcnt field");
        (void)INFLIGHT;
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nks) gissue();
        retire(issued - 1);
        __builtin_amdgcn_s_barrier();
        for (int ks = 0; ks < nks; ++ks) {
            // refill the slot whose reads finished before the barrier that ended step ks-1
            if (issued < nks) gissue();
            const char* st = smem + (ks % NS) * STAGE;
            mfma_step(st, st + A_BYTES);
            retire(issued - ks - 2);       // this wave's share of step ks+1 has landed ...
            __builtin_amdgcn_s_barrier();  // ... and everybody else's
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing may be in flight when the epilogue reuses LDS
    }

    // ---------------------------------------------------------------- epilogue
    // LDS reuse: out tile [BM][BN] bf16 at 0, stats scratch [WM][BN][2] fp32 after it.
    constexpr int OPITCH = BN * 2;
    char* Os = smem;
    float* Ss = reinterpret_cast<float*>(smem + BM * OPITCH);
    const bool odd = lane & 1;
    const bool bnr = g.bnr_x != nullptr;               // fused BN-backward reduction (data-gradient launches)
    const bool fwd_stats = stats_slab != nullptr && !bnr;
    constexpr int CH = BN / 8;          // 16-B chunks per output row
    constexpr int ORPP = 256 / CH;      // output rows per pass
    constexpr int NPASS = BM / ORPP;
    const int orow = tid / CH, och = tid % CH;
    const int ncol = tile_n * BN + och * 8;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        // Lanes l and l^1 hold columns (n, n+1) of the same rows r, r+1: one DPP exchange (a VALU move, no LDS
        // permute) gives the even lane row r and the odd lane row r+1 of BOTH columns as fp32, which one
        // v_cvt_pk_bf16_f32 rounds and packs into the dword the LDS tile takes.  The statistics are kept per
        // (row parity, column) and the two lanes swap halves once at the end.
        float slo = 0.f, shi = 0.f, qlo = 0.f, qhi = 0.f;
        const int col = wn * (BN / WN) + j * 32 + (lane & 31);
        const int c2 = odd ? col - 1 : col;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rbase = wm * (BM / WM) + i * 32 + 4 * khalf + (odd ? 1 : 0);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float a0 = acc[i][j][r], a1 = acc[i][j][r + 1];
                const float recv = dpp_xor1(odd ? a0 : a1);
                const bf16x2 pk = {(bf16)(odd ? recv : a0), (bf16)(odd ? a1 : recv)};
                const uint32_t packed = __builtin_bit_cast(uint32_t, pk);
                const float f0 = __builtin_bit_cast(float, packed << 16);
                const float f1 = __builtin_bit_cast(float, packed & 0xffff0000u);
                slo += f0;
                shi += f1;
                qlo += f0 * f0;
                qhi += f1 * f1;
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                *reinterpret_cast<uint32_t*>(Os + row * OPITCH + c2 * 2) = packed;
            }
        }
        // column n = even lane's lo + odd lane's lo, column n+1 = the two hi's
        float csum = (odd ? shi : slo) + dpp_xor1(odd ? slo : shi);
        float csq = (odd ? qhi : qlo) + dpp_xor1(odd ? qlo : qhi);
        if (fwd_stats) {
            csum += __shfl_xor(csum, 32, 64);
            csq += __shfl_xor(csq, 32, 64);
            if (lane < 32) {
                Ss[(wm * BN + col) * 2 + 0] = csum;
                Ss[(wm * BN + col) * 2 + 1] = csq;
            }
        }
    }
    __syncthreads();

    if (fwd_stats && tid < BN) {
        const int n = tile_n * BN + tid;
        if (n < g.N) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int k = 0; k < WM; ++k) {
                s0 += Ss[(k * BN + tid) * 2 + 0];
                s1 += Ss[(k * BN + tid) * 2 + 1];
            }
            float* dst = stats_slab + (long)tile_m * 2 * g.N;
            dst[n] = s0;
            dst[g.N + n] = s1;
        }
    }

    const bool dense = (g.os == 1 && g.nOH == g.Hout && g.nOW == g.Wout);
    // Everything the store loop reads from global memory (the addend with its gate, the BN input and mask of the
    // fused reduction) is fetched HERE for all passes at once: the accumulators are in LDS by now, so their
    // registers are free, and the thread has 2 * NPASS independent 16-B loads in flight instead of a
    // load -> use round trip per pass (these launches move 3-4 activation-sized tensors with a K loop of two
    // steps: they are streaming kernels and were latency-bound at ~3 TB/s).  Both extras only exist on dense
    // stride-1 launches, where pixel == m.
    constexpr bool PREFETCH = NPASS <= 16;  // (the 256-row tile: 2 x 16 x 4 registers, which its 2-workgroups-per-CU budget has once the accumulators are staged)
    constexpr int NPF = PREFETCH ? NPASS : 1;
    const bool has_add = g.addend != nullptr;
    bf16x8 bx[NPF], ad[NPF];
    unsigned bbits[NPF], abits[NPF];
    if (PREFETCH && (bnr || has_add)) {
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const long m = (long)tile_m * BM + p * ORPP + orow;
            bbits[p] = 0u;
            abits[p] = 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                bx[p][k] = (bf16)0.f;
                ad[p][k] = (bf16)0.f;
            }
            if (m < M && ncol < g.N) {
                if (bnr) {
                    bx[p] = __builtin_bit_cast(bf16x8, ld_stream(static_cast<const bf16*>(g.bnr_x) + m * g.N + ncol));
                    bbits[p] = g.bnr_mask[(m * g.N + ncol) >> 3];
                }
                if (has_add && g.addend_stride == 2) {
                    // compact stride-2 addend: only even (h, w) receive a term
                    const uint32_t t_ = fdiv((uint32_t)m, g.div_wout);
                    const int w_ = (int)((uint32_t)m - t_ * (uint32_t)g.Wout);
                    const uint32_t bq_ = fdiv(t_, g.div_hout);
                    const int h_ = (int)(t_ - bq_ * (uint32_t)g.Hout);
                    if (((w_ | h_) & 1) == 0) {
                        const long b_ = (long)bq_;
                        const long cp = (b_ * ((g.Hout + 1) >> 1) + (h_ >> 1)) * ((g.Wout + 1) >> 1) + (w_ >> 1);
                        ad[p] = *reinterpret_cast<const bf16x8*>(static_cast<const bf16*>(g.addend) + cp * g.N + ncol);
                        abits[p] = 0xffu;
                    }
                } else if (has_add) {
                    ad[p] = __builtin_bit_cast(bf16x8, ld_stream(static_cast<const bf16*>(g.addend) + m * g.N + ncol));
                    abits[p] = g.addend_mask != nullptr ? g.addend_mask[(m * g.N + ncol) >> 3] : 0xffu;
                }
            }
        }
    }
    float r0[8], r1[8], mu[8], is[8];  // bnr: this thread's 8 channels, summed over the rows it stores
    if (bnr) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            r0[k] = 0.f;
            r1[k] = 0.f;
            mu[k] = ncol < g.N ? g.bnr_mean[ncol + k] : 0.f;
            is[k] = ncol < g.N ? g.bnr_invstd[ncol + k] : 0.f;
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int row = p * ORPP + orow;
        const long m = (long)tile_m * BM + row;
        if (m < M && ncol < g.N) {
            long pix;
            if (dense) {
                pix = m;
            } else {
                const uint32_t t1 = fdiv((uint32_t)m, g.div_now);
                const int j = (int)((uint32_t)m - t1 * (uint32_t)g.nOW);
                const uint32_t bq = fdiv(t1, g.div_noh);
                const int i = (int)(t1 - bq * (uint32_t)g.nOH);
                const long b = (long)bq;
                pix = (b * g.Hout + (i * g.os + g.ooh)) * g.Wout + (j * g.os + g.oow);
            }
            bf16x8 v = *reinterpret_cast<const bf16x8*>(Os + row * OPITCH + och * 16);
            if constexpr (PREFETCH) {
                if (has_add) {
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        v[k] = (bf16)((float)v[k] + (((abits[p] >> k) & 1u) ? (float)ad[p][k] : 0.f));
                }
            } else if (g.addend != nullptr && g.addend_stride == 2) {
                // compact stride-2 addend (dense stride-1 launch: pix == m): only even (h, w) receive a term
                const uint32_t t_ = fdiv((uint32_t)m, g.div_wout);
                const int w_ = (int)((uint32_t)m - t_ * (uint32_t)g.Wout);
                const uint32_t bq_ = fdiv(t_, g.div_hout);
                const int h_ = (int)(t_ - bq_ * (uint32_t)g.Hout);
                if (((w_ | h_) & 1) == 0) {
                    const long b_ = (long)bq_;
                    const long cp = (b_ * ((g.Hout + 1) >> 1) + (h_ >> 1)) * ((g.Wout + 1) >> 1) + (w_ >> 1);
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(static_cast<const bf16*>(g.addend) + cp * g.N + ncol);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] + (float)a[k]);
                }
            } else if (g.addend != nullptr) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, ld_stream(static_cast<const bf16*>(g.addend) + pix * g.N + ncol));
                const unsigned abits = g.addend_mask != nullptr ? g.addend_mask[(pix * g.N + ncol) >> 3] : 0xffu;
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] + (((abits >> k) & 1u) ? (float)a[k] : 0.f));
            }
            if (bnr) {
                bf16x8 xv;
                unsigned bits;
                if constexpr (PREFETCH) {
                    xv = bx[p];
                    bits = bbits[p];
                } else {
                    xv = __builtin_bit_cast(bf16x8, ld_stream(static_cast<const bf16*>(g.bnr_x) + pix * g.N + ncol));
                    bits = g.bnr_mask[(pix * g.N + ncol) >> 3];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gg = ((bits >> k) & 1u) ? (float)v[k] : 0.f;
                    r0[k] += gg;
                    r1[k] += gg * ((float)xv[k] - mu[k]) * is[k];
                }
            }
            if (g.stream_out) st_stream(y + pix * g.N + ncol, __builtin_bit_cast(u32x4, v));
            else *reinterpret_cast<bf16x8*>(y + pix * g.N + ncol) = v;
        }
    }
    if (bnr) {
        // lanes och, och + CH, ... of a wave own the same 8 channels: butterfly over them, then the 4 waves via LDS
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int o = CH; o < 64; o <<= 1) {
                r0[k] += __shfl_xor(r0[k], o, 64);
                r1[k] += __shfl_xor(r1[k], o, 64);
            }
        }
        if (lane < CH) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                Ss[(wave * BN + och * 8 + k) * 2 + 0] = r0[k];
                Ss[(wave * BN + och * 8 + k) * 2 + 1] = r1[k];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int n = tile_n * BN + tid;
            if (n < g.N) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    s0 += Ss[(k * BN + tid) * 2 + 0];
                    s1 += Ss[(k * BN + tid) * 2 + 1];
                }
                float* dst = stats_slab + (long)tile_m * 2 * g.N;
                dst[n] = s0;
                dst[g.N + n] = s1;
            }
        }
    }
}

// tiles per round-robin XCD group (0 = contiguous tile range per XCD); PDT_IGEMM_XCD_GROUP = m-tiles per group.
// Measured at B=1024 with groups of 1 / 4 / 16 / 64 m-tiles: within +-0.5 % of the contiguous ranges on the sum
// of all ResNet-50 layers (two pointwise data gradients gain 5 %), so the default stays 0: unlike the streaming BN
// kernels, the convolutions are not limited by how many DRAM windows the chip has open.
int xcd_group_tiles(const ConvGeom& g, int n_tiles) {
    static const int env = [] {
        const char* e = std::getenv("PDT_IGEMM_XCD_GROUP");
        return e ? std::atoi(e) : 0;
    }();
    (void)g;
    return env > 0 ? env * n_tiles : 0;
}

template <int BM, int BN, int WM, int WN, int BK, int MINW, int NS = 0>
void launch_cfg(const void* x, const void* w, void* y, const ConvGeom& g, float* stats_slab, hipStream_t s) {
    const long M = (long)g.B * g.nOH * g.nOW;
    const int m_tiles = (int)((M + BM - 1) / BM), n_tiles = (g.N + BN - 1) / BN;
    size_t lds = (size_t)(NS > 0 ? NS : 2) * (BM + BN) * BK * 2;       // operand stage buffers
    const size_t epi = (size_t)BM * BN * 2 + (size_t)4 * BN * 2 * 4;  // staged output tile + stats scratch (4 waves)
    if (lds < epi) lds = epi;
    const uint32_t xb = (uint32_t)((long)g.B * g.Hin * g.Win * (g.pix_elems ? g.pix_elems : g.C) * 2);
    const uint32_t wb = (uint32_t)((long)g.N * g.wtaps * g.C * 2);
    static bool attr_set = false;
    if (!attr_set) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_igemm<BM, BN, WM, WN, BK, MINW, NS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    ConvGeom gd = g;
    gd.div_now = make_fastdiv((uint32_t)g.nOW);
    gd.div_noh = make_fastdiv((uint32_t)g.nOH);
    gd.div_wout = make_fastdiv((uint32_t)g.Wout);
    gd.div_hout = make_fastdiv((uint32_t)g.Hout);
    gd.stream_out = stream_nt((long)g.B * g.Hout * g.Wout * g.N * 2) ? 1 : 0;
    if (M >= (1L << 31)) {
        fprintf(stderr, "conv_igemm: %ld output pixels exceed the 31-bit index space\n", M);
        abort();
    }
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, WM, WN, BK, MINW, NS>), dim3(m_tiles * n_tiles), dim3(256), lds, s, (const bf16*)x,
                       (const bf16*)w, (bf16*)y, gd, stats_slab, m_tiles, n_tiles, xb, wb, xcd_group_tiles(g, n_tiles));
}

// A/B knob: PDT_IGEMM_BNR_FAT=1 keeps the fused-BN-reduce data gradients off the 128-register build
bool bnr_fat() {
    static const bool v = [] {
        const char* e = std::getenv("PDT_IGEMM_BNR_FAT");
        return e && e[0] == '1';
    }();
    return v;
}

// tile / staging configuration: an explicit argument of the launch (tests force every build per case) on top of the
// process-wide PDT_IGEMM_CFG experiment knob; 0 = the tuned per-shape heuristic
int igemm_cfg_override(int cfg_arg = 0) {
    static int v = [] {
        const char* e = std::getenv("PDT_IGEMM_CFG");
        return e ? std::atoi(e) : 0;
    }();
    return cfg_arg != 0 ? cfg_arg : v;
}

}  // namespace

namespace {
// 256x128 tile (each wave 128x64: 0.75 LDS fragment reads and 25 % fewer staged bytes per MFMA than 128x128) pays on
// long-K GEMMs once there are enough tiles to fill the chip: measured +15-25 % on the 256-channel 3x3s and the
// K >= 1024 pointwise layers, a loss on short-K / few-tile shapes (profiles/r1_igemm_glds_vs_regstage.md)
bool use_tall_tile(const ConvGeom& g) {
    static const bool enabled = [] {
        const char* e = std::getenv("PDT_IGEMM_TALL");
        return !(e && e[0] == '0');
    }();
    if (!enabled) return false;
    const long M = (long)g.B * g.nOH * g.nOW;
    return g.N % 128 == 0 && g.C % 64 == 0 && (long)g.ntaps * g.C >= 1024 && (M / 256) * (g.N / 128) >= 256;
}
// 64-wide outputs: a 256x64 tile (each wave 64x64) reads 1 instead of 1.5 LDS fragments per MFMA and stages the
// weights half as often; B=1024: 64-channel 3x3 484 -> 453 us, 256->64 pointwise forward 411 -> 373 us, a loss on
// K = 64 (nothing to amortise)
bool use_tall64(const ConvGeom& g) {
    const long M = (long)g.B * g.nOH * g.nOW;
    return g.N % 128 != 0 && (g.C % 64 == 0 || g.C == 32) && (long)g.ntaps * g.C >= 224 && M / 256 >= 512;
}
// rows of the output tile the launch heuristic below picks for this geometry (the slab of per-m-tile partial
// rows the bindings allocate has one row per tile)
int igemm_bm(const ConvGeom& g, int cfg_arg) {
    const int cfg = igemm_cfg_override(cfg_arg);
    const bool n128 = g.N % 128 == 0;
    if (cfg == 10) return n128 ? 256 : 128;
    if (cfg == 15) return n128 ? 128 : 256;
    if (cfg != 0) return 128;
    return (use_tall_tile(g) || use_tall64(g)) ? 256 : 128;
}
}  // namespace

int conv_igemm_m_tiles(const ConvGeom& g, int* bm_out, int cfg) {
    const long M = (long)g.B * g.nOH * g.nOW;
    const int bm = igemm_bm(g, cfg);
    if (bm_out) *bm_out = bm;
    return (int)((M + bm - 1) / bm);
}

void launch_conv_igemm(const void* x, const void* w, void* y, const ConvGeom& g, float* stats_slab,
                       hipStream_t s, int cfg) {
    const bool n128 = g.N % 128 == 0;
    if (g.C % 64 != 0) {  // 32-channel K rows (the packed stem): only the BK = 32 tiles apply
        if (n128) launch_cfg<128, 128, 2, 2, 32, 3, 3>(x, w, y, g, stats_slab, s);
        else if (igemm_bm(g, cfg) == 256) launch_cfg<256, 64, 4, 1, 32, 2, 3>(x, w, y, g, stats_slab, s);
        else launch_cfg<128, 64, 4, 1, 32, 3, 3>(x, w, y, g, stats_slab, s);
        return;
    }
    switch (igemm_cfg_override(cfg)) {  // explicit argument / experiment knob; 0 = tuned default
        case 9:  // the register-staged choice of the first round-1 builds (for A/B runs)
            if (g.ntaps == 1 && g.C <= 1024) {
                if (n128) launch_cfg<128, 128, 2, 2, 32, 3>(x, w, y, g, stats_slab, s);
                else launch_cfg<128, 64, 4, 1, 32, 3>(x, w, y, g, stats_slab, s);
            } else {
                if (n128) launch_cfg<128, 128, 2, 2, 64, 2>(x, w, y, g, stats_slab, s);
                else launch_cfg<128, 64, 4, 1, 64, 2>(x, w, y, g, stats_slab, s);
            }
            return;
        case 5:  // BK=64 everywhere (the round-1 baseline)
            if (n128) launch_cfg<128, 128, 2, 2, 64, 2>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 64, 2>(x, w, y, g, stats_slab, s);
            return;
        case 10:  // LDS-DMA ring, 256x128 tile (each wave 128x64), BK=32, 3 stages (72 KB: two workgroups per CU)
            if (n128) launch_cfg<256, 128, 2, 2, 32, 2, 3>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 32, 3, 3>(x, w, y, g, stats_slab, s);
            return;
        case 15:  // 256x64 tile for 64-wide outputs (each wave 64x64: 1 instead of 1.5 fragment reads per MFMA)
            if (n128) launch_cfg<128, 128, 2, 2, 32, 3, 3>(x, w, y, g, stats_slab, s);
            else launch_cfg<256, 64, 4, 1, 32, 2, 3>(x, w, y, g, stats_slab, s);
            return;
        case 11:  // double-buffered DMA (2 stages), 128-register build: four workgroups per CU for short-K layers
            if (n128) launch_cfg<128, 128, 2, 2, 32, 4, 2>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 32, 4, 2>(x, w, y, g, stats_slab, s);
            return;
        case 12:  // ring of 3, 128-register build
            if (n128) launch_cfg<128, 128, 2, 2, 32, 4, 3>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 32, 4, 3>(x, w, y, g, stats_slab, s);
            return;
        case 6:  // LDS-DMA ring, BK=32, 4 stages (64 KB: two workgroups per CU)
            if (n128) launch_cfg<128, 128, 2, 2, 32, 2, 4>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 32, 2, 4>(x, w, y, g, stats_slab, s);
            return;
        case 7:  // LDS-DMA ring, BK=64, 3 stages (96 KB: one workgroup per CU)
            if (n128) launch_cfg<128, 128, 2, 2, 64, 1, 3>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 64, 1, 3>(x, w, y, g, stats_slab, s);
            return;
        case 8:  // LDS-DMA ring, BK=32, 3 stages
            if (n128) launch_cfg<128, 128, 2, 2, 32, 3, 3>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 32, 3, 3>(x, w, y, g, stats_slab, s);
            return;
        case 1:  // BK=32, 3 waves/SIMD
            if (n128) launch_cfg<128, 128, 2, 2, 32, 3>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 32, 3>(x, w, y, g, stats_slab, s);
            return;
        case 2:  // BK=32, 4 waves/SIMD (register-capped at 128)
            if (n128) launch_cfg<128, 128, 2, 2, 32, 4>(x, w, y, g, stats_slab, s);
            else launch_cfg<128, 64, 4, 1, 32, 4>(x, w, y, g, stats_slab, s);
            return;
        case 3:  // BK=64 but always the narrow 128x64 tile (more blocks for small-M layers)
            launch_cfg<128, 64, 4, 1, 64, 2>(x, w, y, g, stats_slab, s);
            return;
        case 4:  // BK=32 narrow tile, 4 waves/SIMD
            launch_cfg<128, 64, 4, 1, 32, 4>(x, w, y, g, stats_slab, s);
            return;
        default:
            // measured per ResNet-50 layer shape on MI355X at B=256 (profiles/r1_igemm_cfg_sweep.md,
            // profiles/r1_igemm_glds_vs_regstage.md): the LDS-DMA ring (BK=32, 3 stages, 3 workgroups per CU) wins
            // on every pointwise layer and on the 128-channel 3x3s; long-K 3x3s with >= 256 channels want the
            // 4-stage ring at 2 workgroups per CU; the 64-channel 3x3 (9 short taps) keeps the register-staged
            // BK=64 loop
            // B=1024 re-sweep (bench/resnet50_roofline.py): short-K launches (K = taps x C <= 512: most pointwise
            // layers, the parity launches of stride-2 data gradients) are latency-bound, so they take the
            // 128-register build with a 2-stage DMA buffer = FOUR workgroups per CU (-5..-13 %); the 64-channel
            // 3x3 takes the 128-register ring of 3 (-4 %).
            if (use_tall_tile(g)) {
                launch_cfg<256, 128, 2, 2, 32, 2, 3>(x, w, y, g, stats_slab, s);
            } else if (use_tall64(g)) {
                launch_cfg<256, 64, 4, 1, 32, 2, 3>(x, w, y, g, stats_slab, s);
            } else if ((long)g.ntaps * g.C <= 512 && (g.bnr_x == nullptr || !bnr_fat())) {
                if (n128) launch_cfg<128, 128, 2, 2, 32, 4, 2>(x, w, y, g, stats_slab, s);
                else launch_cfg<128, 64, 4, 1, 32, 4, 2>(x, w, y, g, stats_slab, s);
            } else if (g.ntaps > 1 && g.C == 64) {
                if (n128) launch_cfg<128, 128, 2, 2, 32, 4, 3>(x, w, y, g, stats_slab, s);
                else launch_cfg<128, 64, 4, 1, 32, 4, 3>(x, w, y, g, stats_slab, s);
            } else if (g.ntaps > 1 && g.C >= 256) {
                if (n128) launch_cfg<128, 128, 2, 2, 32, 2, 4>(x, w, y, g, stats_slab, s);
                else launch_cfg<128, 64, 4, 1, 32, 2, 4>(x, w, y, g, stats_slab, s);
            } else {
                if (n128) launch_cfg<128, 128, 2, 2, 32, 3, 3>(x, w, y, g, stats_slab, s);
                else launch_cfg<128, 64, 4, 1, 32, 3, 3>(x, w, y, g, stats_slab, s);
            }
    }
}

}  // namespace pdt
