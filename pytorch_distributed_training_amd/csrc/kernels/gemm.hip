// bf16 GEMM on the gfx950 matrix cores for Linear layers and the whole ViT path.
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )        A, B row-major with the reduction axis contiguous
//
// which is Linear forward (A = x, B = W), Linear data gradient (A = dy, B = W^T kept as [K][N] by the
// multi-tensor transpose) and a 1x1 convolution in NHWC.  Reference: /root/reference/main.py:40 builds a model
// whose head is this GEMM; BASELINE.json config 5 (ViT-B/16) is ~70 % this kernel.
//
// Design for CDNA4 (one workgroup per CU, 8 waves = 2 per SIMD, 160 KiB LDS):
//  * 256 x 256 output tile, BK = 64, two LDS stage buffers of 64 KiB, each cut into four 16 KiB half-tiles
//    (A rows 0-127 / 128-255, B rows 0-127 / 128-255).  The bigger the tile the fewer bytes go through the
//    L2 -> LDS path per MFMA, and that path (~70 GB/s per CU), not the matrix pipe, limits a 256x128 tile.
//  * operands arrive by LDS-DMA (buffer_load_dwordx4 ... lds): no VGPR round trip, no ds_write; the XOR swizzle
//    that makes the ds_read_b128 fragment reads conflict-free is applied on the SOURCE address.  One half-tile
//    (2 instructions per thread) is issued per phase, and only a counted s_waitcnt vmcnt(2) at the end of a
//    K-tile retires them, so loads stay in flight across the raw s_barriers.
//  * the 8 waves are 2 (M) x 4 (N); each owns 128 x 64 of the tile = 8 x 4 MFMA 16x16x32 accumulators.
//    Waves 0-3 and 4-7 are SIMD partners and run ONE PHASE APART (waves 4-7 execute one extra barrier up front):
//    while one group issues its 16 MFMAs the other issues its LDS fragment reads and its DMA, then they swap.
//    The matrix pipe of every SIMD is fed continuously although each wave alternates between the two roles.
//  * per K-tile a wave runs 4 phases (row half x k-step of 32); fragment reads per phase 8 / 8 / 4 / 4.
//  * the MFMA operands are swapped (D^T = B . A^T) so that a lane's 4 accumulator registers are 4 CONSECUTIVE
//    output columns: one packed 8-byte LDS store per 16x16 tile in the epilogue, no cross-lane exchange.
//  * epilogue: + bias in fp32, round to bf16, stage the tile in LDS (row pitch 528 B), then a coalesced pass of
//    16 B per lane over full rows that can add a residual, write the pre-activation, apply GELU or multiply
//    by GELU'(u) (the data gradient of the layer in front of a GELU), and reduce per-column sums.
//  * ragged M / N / K come from the buffer descriptors' range check (zeros), and the descriptors are re-based per
//    tile with 64-bit arithmetic, so there is no 2 GiB limit on the operands.
//  * tail: tiles are dealt to the XCDs in contiguous ranges (n fastest) for L2 reuse of the A panel.
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int BM = 256, BN = 256, BK = 64;
constexpr int HALF_BYTES = 128 * BK * 2;       // 16 KiB: 128 rows of 128 B
constexpr int STAGE_BYTES = 4 * HALF_BYTES;    // A-lo, A-hi, B-lo, B-hi
// LDS map (all 160 KiB of the CU): [stage 0: 64 KiB][extra: 32 KiB][stage 1: 64 KiB].  The epilogue stages half
// an output tile (128 rows) in "extra" plus the stage buffer that does NOT receive the next tile's first K-tile.
constexpr int EXTRA_OFF = STAGE_BYTES, STAGE1_OFF = STAGE_BYTES + 32768;
constexpr int LDS_BYTES = STAGE1_OFF + STAGE_BYTES;  // 163840
// Staged output rows are 512 B with NO padding; bank conflicts are avoided by an XOR swizzle instead.  A padded
// pitch must stay a multiple of 16 B for the ds_read_b128 row pass, i.e. a multiple of 4 banks: with 528 B the
// 16 lanes of a ds_write_b64 group (16 consecutive rows, one 8-B column granule) fell on banks 4r, 4r+1 (mod 32):
// rows r and r+8 collided and half the banks idled: 2-way conflicts on all 32 staging stores of a wave, 18-19 % of
// the kernel's LDS cycles on the short-K convolution launches (profiles/r2_pmc_b1024_summary.md).  Swizzle: the
// 16-B chunk index is XORed with (row & 7) and the 8-B half inside the chunk with bit 3 of the row, so the 16 rows
// of a store group hit 16 different 8-B bank pairs, and the 16 lanes of a ds_read_b128 group (one row, 16 chunks
// that are distinct mod 16) stay conflict-free.  The reader's row has bit 3 == bit 3 of (tid >> 5): wave-uniform,
// waves 4-7 swap the two halves of what they read.
constexpr int OPITCH = BN * 2;                       // staged output row pitch (bytes)
static_assert(128 * OPITCH <= STAGE_BYTES + 32768, "half-tile staging fits");

__device__ __forceinline__ float gelu_erf_part(float u, float* pdf_out) {
    // 0.5 * (1 + erf(u / sqrt(2))) by Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7), and exp(-u^2/2) / sqrt(2 pi)
    const float x = fabsf(u) * 0.70710678118654752f;
    const float t = 1.f / (1.f + 0.3275911f * x);
    const float e = __expf(-x * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.f - poly * e;
    *pdf_out = e * 0.3989422804014327f;
    return 0.5f * (1.f + copysignf(erf_abs, u));
}
__device__ __forceinline__ float gelu_f(float u) {
    float pdf;
    return u * gelu_erf_part(u, &pdf);
}
__device__ __forceinline__ float gelu_grad_f(float u) {
    float pdf;
    const float cdf = gelu_erf_part(u, &pdf);
    return cdf + u * pdf;
}

__device__ __forceinline__ int stage_off(int s) { return s ? STAGE1_OFF : 0; }

// Persistent: gridDim.x workgroups (one per CU) walk the tile list.  Blocks b and b+8 share an XCD; every XCD label
// gets a contiguous range of tiles so that the workgroups that run together on one L2 share operand panels.
// EXTRA: 0 none, 1 residual added, 2 multiplied by GELU'(gelu_in).  ACT: GELU, the pre-activation is written too.
// CONV (the GEMM is a pointwise convolution in NHWC, ld == channels):
//   1  forward: per-column sum and sum of squares of the ROUNDED outputs (BatchNorm statistics) -> slab[m_tile][2][N]
//   2  data gradient: + addend gated by a bit mask (the residual branch's gradient through a ReLU), and the two
//      BatchNorm-backward sums of the result (sum g, sum g * xhat with g = result gated by the BN's ReLU mask)
//      -> slab[m_tile][2][N]; which of the two parts run is a launch argument.
// Compile-time so that the K loop does not carry a dozen wave-uniform conditions in scalar registers.
// (A variant in which waves 0-3 issued all operand DMA and waves 4-7 all output stores, so that no wave's counted
// wait on its DMA also waits for its own older stores, measured within 2 % of this one on every pointwise shape of
// ResNet-50 and on 4096^3: the store drain is not what short reductions wait for, and it was dropped.)
// TAPS: the A rows are the pixels of an NHWC image and the reduction also runs over the taps of a spatial filter
// (3x3, stride 1, same-size output): tap (dh, dw) reads the A row of pixel (y + dh, x + dw), which is the SAME row
// pattern shifted by (dh * W + dw) rows -- a per-tap constant that goes into the DMA's scalar offset -- and only the
// zero padding is per lane: 9 validity bits per staged row, computed once per tile.
// DUAL (CONV == 2 only): a second BatchNorm takes the same gated gradient (kernels.h, bnr_x2): one more once-read operand
// in the row pass and a third per-column sum; its own build so that the common one keeps its registers.
// BNT: tile width, 256 or 128.  The 128-wide build has no B-hi half-tile (DMA slot 3 stays empty), its waves own 128 x 32
// of the tile (8 x 2 accumulators, 8 MFMAs per phase) and the store pass covers 32 rows per pass.  It moves 48 KiB
// through the L2 -> LDS path per K-tile of 256 x 128 x 64 (1.5x the bytes per FLOP of the square tile), so it is for
// outputs that ARE 128 wide (the 128-channel 3x3 and pointwise layers) and for tile counts that leave the last round
// of square tiles mostly empty (ViT at small batch: 297 tiles on 256 CUs).
template <int EXTRA, bool ACT, bool COLSUM, bool RAGGED, int CONV, bool TAPS, bool DUAL = false, int BNT = 256>
__global__ __launch_bounds__(512, 2) void k_gemm_nt(const GemmArgs a) {
    static_assert(BNT == 256 || BNT == 128, "tile width");
    constexpr int NJ = BNT / 64;            // 16-column accumulator groups per wave
    constexpr int CPR = BNT / 8;            // 16-B chunks per staged output row
    constexpr int RPP = 512 / CPR;          // rows per store pass
    constexpr int NPASS = 128 / RPP;        // store passes per half tile
    constexpr int OPITCH_T = BNT * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    const int total = a.m_tiles * a.n_tiles;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int tq = total >> 3, trem = total & 7;
    const int range_start = xcd < trem ? xcd * (tq + 1) : trem * (tq + 1) + (xcd - trem) * tq;
    const int range_count = tq + (xcd < trem ? 1 : 0);
    const int wg_per_xcd = ((int)gridDim.x - xcd + 7) >> 3;
    int ti = slot;
    if (ti >= range_count) return;

    // ---- per-tile operand state (A/B side), recomputed for the NEXT tile while the current one is being stored
    i32x4 ra, rb;
    long m0;
    int n0, rows_m, rows_n, tile_m;
    // DMA source offsets: in every phase each thread fetches 2 pieces of one half-tile.  Piece pc = wave * 2 + j
    // covers rows pc*8 .. pc*8+7 (1 KiB); lane -> row pc*8 + lane/8, physical chunk lane%8, which holds logical
    // chunk (lane%8) ^ ((row >> 1) & 7).  The per-lane part of the address is the same for every tile and K-tile
    // (2 + 2 registers); the half (128 rows) and the K-tile go into the instruction's scalar offset, which the
    // range check of the descriptor does not see: rows past the end of a ragged tile are masked explicitly.
    int lch[2];
    uint32_t vA[2], vB[2];
    uint32_t tapmask[2][2];  // TAPS: bit t = tap t of staged A row (half h, piece j) reads inside the image
    const int halo = TAPS ? a.W + 1 : 0;  // rows the A descriptor starts ahead of the tile (negative tap shifts)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + (lane >> 3);
        lch[j] = (lane & 7) ^ ((row >> 1) & 7);
        vA[j] = (uint32_t)row * (uint32_t)(a.lda * 2) + (uint32_t)(lch[j] * 16);
        vB[j] = (uint32_t)row * (uint32_t)(a.ldb * 2) + (uint32_t)(lch[j] * 16);
    }
    auto setup = [&](int idx) {
        // grouped order: bands of a.group_m tile rows, m fastest inside a band (group_m = 1: n fastest)
        // (divisions by launch constants through multiply-high: two integer divisions by run-time values cost ~1 000
        // shader clocks here, once per tile, a tenth of a short-K tile)
        const int band_tiles = a.group_m * a.n_tiles;
        const int band = (int)fdiv((uint32_t)idx, a.div_band), rem = idx - band * band_tiles;
        const bool last = band == a.last_band;
        const int band_rows = last ? a.last_rows : a.group_m;
        const int tn = (int)(last ? fdiv((uint32_t)rem, a.div_last) : fdiv((uint32_t)rem, a.div_gm));
        tile_m = band * a.group_m + (rem - tn * band_rows);
        m0 = (long)tile_m * BM;
        n0 = tn * BNT;
        rows_m = (int)((a.M - m0) < BM ? (a.M - m0) : BM);
        rows_n = (a.N - n0) < BNT ? (a.N - n0) : BNT;
        // descriptors re-based at the tile's first row: offsets inside stay far below 2^31
        // (TAPS: the window is widened by the halo on both sides in case the range check sees the scalar offset;
        // every lane whose shifted row leaves the image, and with it the tensor, is masked explicitly)
        ra = make_rsrc_words(static_cast<const char*>(a.A) + (m0 - halo) * a.lda * 2,
                             (uint32_t)((long)(rows_m + 2 * halo) * a.lda * 2));
        rb = make_rsrc_words(static_cast<const char*>(a.B) + (long)n0 * a.ldb * 2, (uint32_t)((long)rows_n * a.ldb * 2));
        if (TAPS) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const uint32_t m = (uint32_t)(m0 + h * 128 + (wave * 2 + j) * 8 + (lane >> 3));  // < 2^31 pixels
                    const uint32_t img = fdiv(m, a.div_hw);
                    const uint32_t pix = m - img * a.div_hw.d;
                    const int y = (int)fdiv(pix, a.div_w);
                    const int x = (int)pix - y * (int)a.div_w.d;
                    uint32_t bits = 0u;
                    for (int t = 0; t < a.ntaps; ++t) {
                        const int dh = (int)(int8_t)(a.tap[t] & 0xff), dw = (int)(int8_t)((a.tap[t] >> 8) & 0xff);
                        const bool ok = (unsigned)(y + dh) < (unsigned)a.H && (unsigned)(x + dw) < (unsigned)a.W;
                        bits |= ok ? (1u << t) : 0u;
                    }
                    tapmask[h][j] = bits;
                }
        }
    };
    const uint32_t piece_base = lds_address(smem) + (uint32_t)wave * 2048u;  // + j * 1024

    const int nt = TAPS ? a.ntaps * (a.K / BK) : (a.K + BK - 1) / BK;
    // slot: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi; t = K-tile of the current operand state; par = its stage buffer
    auto issue = [&](int hslot, int t, int par) {
        if (BNT == 128 && hslot == 3) return;  // (no B-hi half: the counted waits below only assume "2 per issue")
        if (a.debug_nostore & 8) return;  // (timing diagnostics: no operand DMA)
        uint32_t dst = piece_base + (uint32_t)(stage_off(par) + hslot * HALF_BYTES);
        if (BNT == 128) dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)dst);  // (this build loses track of its uniformity)
        const bool isB = hslot >= 2;
        const int h = hslot & 1;
        const int ld2 = (isB ? a.ldb : a.lda) * 2;
        const int rows = isB ? rows_n : rows_m;
        int koff = t * BK * 2, ti = 0;  // byte offset along the row, tap index
        if (TAPS) {
            ti = (int)fdiv((uint32_t)t, a.div_kt);
            const int kc = t - ti * (int)a.div_kt.d;
            const int32_t tp = a.tap[ti];
            if (isB) koff = ((tp >> 16) * a.K + kc * BK) * 2;
            else koff = kc * BK * 2 + ((int)(int8_t)(tp & 0xff) * a.W + (int)(int8_t)((tp >> 8) & 0xff) + halo) * ld2;
        }
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(koff + h * 128 * ld2);
        const bool cut0 = RAGGED && t == nt - 1 && t * BK + lch[0] * 8 >= a.K;  // chunks past the end of the
        const bool cut1 = RAGGED && t == nt - 1 && t * BK + lch[1] * 8 >= a.K;  // reduction axis read zeros
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = h * 128 + (wave * 2 + j) * 8 + (lane >> 3);
            uint32_t v = isB ? vB[j] : vA[j];
            if (r >= rows || (j ? cut1 : cut0)) v = kOOB;
            if (TAPS && !isB && !((tapmask[h][j] >> ti) & 1u)) v = kOOB;
            if (!TAPS && !isB && a.a_stream) lds_dma16s_stream(ra, v, soff, dst + (uint32_t)(j * 1024));
            else lds_dma16s(isB ? rb : ra, v, soff, dst + (uint32_t)(j * 1024));
        }
    };

    // ---- fragment read addresses (bytes inside a stage buffer)
    const int l15 = lane & 15, lq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int lo = l15 * 128 + (((ks * 4 + lq) ^ (l15 >> 1)) << 4);
        a_off[ks] = wr * HALF_BYTES + lo;
        b_off[ks] = BNT == 256 ? 2 * HALF_BYTES + (wc >> 1) * HALF_BYTES + (wc & 1) * 64 * 128 + lo
                               : 2 * HALF_BYTES + wc * 32 * 128 + lo;
    }
    const int och = tid & (CPR - 1), orow = tid / CPR;  // store pass: CPR chunks of 8 columns per row, RPP rows per pass
    constexpr int PB0 = CONV == 2 ? (DUAL ? 2 : 4) : 8;  // passes whose global reads are in flight together
    constexpr int PB = PB0 < NPASS ? PB0 : NPASS;

    // ---- prologue of the first tile: K-tile 0
    int gk = 0;  // K-tiles consumed so far: K-tile t of the current tile lives in stage (gk + t) & 1
    setup(range_start + ti);
    issue(0, 0, 0);
    issue(1, 0, 0);
    issue(2, 0, 0);
    issue(3, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    while (true) {
        f32x4_t acc[8][NJ];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one phase behind their SIMD partners
        bf16x8 bfrag[2][NJ];
        for (int t = 0; t < nt; ++t) {
            const int par = (gk + t) & 1;
            const char* st = smem + stage_off(par);
            const bool more1 = t + 1 < nt, more2 = t + 2 < nt;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int mh = ph >> 1, ks = ph & 1;
                // ---------------- load segment (the partner wave is in its MFMA segment)
                bf16x8 afrag[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    afrag[i] = *reinterpret_cast<const bf16x8*>(st + a_off[ks] + (mh * 64 + i * 16) * 128);
                if (mh == 0) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        bfrag[ks][j] = *reinterpret_cast<const bf16x8*>(st + b_off[ks] + j * 16 * 128);
                }
                if (ph == 0) {
                    if (more1) {
                        if (t == 0) issue(2, 1, par ^ 1);  // (later tiles: issued two K-tiles ahead, below)
                        issue(3, t + 1, par ^ 1);
                    }
                } else if (ph == 1) {
                    if (more1) issue(0, t + 1, par ^ 1);
                } else if (ph == 2) {
                    if (more1) issue(1, t + 1, par ^ 1);
                } else {
                    if (more2) {
                        issue(2, t + 2, par);
                        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // everything of K-tile t+1 has landed
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __builtin_amdgcn_s_barrier();
                // ---------------- MFMA segment
                __builtin_amdgcn_s_setprio(1);
                if (!(a.debug_nostore & 4)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[mh * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[ks][j], afrag[i],
                                                                                          acc[mh * 4 + i][j], 0, 0, 0);
                } else {
                    acc[mh * 4][0][0] += (float)afrag[0][0] + (float)bfrag[ks][0][0];  // keep the fragment reads alive
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        if (wr == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups: every fragment read is done
        gk += nt;

        // -------------------------------------------------------------------------------- epilogue
        const long om0 = m0;
        const int on0 = n0, orows_m = rows_m, otile_m = tile_m;
        ti += wg_per_xcd;
        const bool has_next = ti < range_count;
        const int npar = gk & 1;  // stage that receives the next tile's first K-tile
        if (has_next) {
            setup(range_start + ti);
            issue(0, 0, npar);
            issue(1, 0, npar);
            issue(2, 0, npar);
            issue(3, 0, npar);
        }
        char* Os = smem + (npar == 0 ? EXTRA_OFF : 0);
        const int ncol = on0 + och * 8;
        const bool col_ok = ncol < a.N;  // N % 8 == 0
        const uint32_t cbytes = (a.debug_nostore & 1) ? 0u : (uint32_t)((long)orows_m * a.ldc * 2);
        const __amdgpu_buffer_rsrc_t rc = make_rsrc(static_cast<char*>(a.C) + om0 * a.ldc * 2, cbytes);
        const __amdgpu_buffer_rsrc_t rp = make_rsrc(ACT ? static_cast<char*>(a.preact_out) + om0 * a.ldc * 2 : nullptr, ACT ? cbytes : 0u);
        // one optional extra bf16 operand: the residual / gated addend (added) or the GELU input (its derivative
        // multiplies); the fused BN-backward sums read one more (the BN's input) and the two bit masks
        const bool has_add = CONV == 2 && a.addend != nullptr, has_amask = CONV == 2 && a.addend_mask != nullptr;
        const bool bnr = CONV == 2 && a.bnr_x != nullptr;
        const void* extra = CONV == 2 ? a.addend : (EXTRA == 1 ? a.residual : a.gelu_in);
        const int lde = CONV == 2 ? a.ldc : (EXTRA == 1 ? a.ldr : a.ldu);
        const bool use_extra = EXTRA != 0 || has_add;
        const __amdgpu_buffer_rsrc_t re = make_rsrc(use_extra ? static_cast<const char*>(extra) + om0 * lde * 2 : nullptr,
                                                    use_extra ? (uint32_t)((long)orows_m * lde * 2) : 0u);
        const __amdgpu_buffer_rsrc_t rx = make_rsrc(bnr ? static_cast<const char*>(a.bnr_x) + om0 * a.ldc * 2 : nullptr,
                                                    bnr ? (uint32_t)((long)orows_m * a.ldc * 2) : 0u);
        const __amdgpu_buffer_rsrc_t rx2 = make_rsrc(DUAL ? static_cast<const char*>(a.bnr_x2) + om0 * a.ldc * 2 : nullptr,
                                                     DUAL ? (uint32_t)((long)orows_m * a.ldc * 2) : 0u);
        // masks: one bit per element, [pixel][C/8] bytes (ldc == N for the convolution modes)
        const uint8_t* amask = has_amask ? a.addend_mask + ((om0 * a.ldc) >> 3) : nullptr;
        const uint8_t* bmask = bnr ? a.bnr_mask + ((om0 * a.ldc) >> 3) : nullptr;
        constexpr bool SUMS2 = CONV != 0;  // two per-column sums instead of one
        float csum[8], csq[SUMS2 ? 8 : 1], mu[CONV == 2 ? 8 : 1], is[CONV == 2 ? 8 : 1];
        float csq2[DUAL ? 8 : 1], mu2[DUAL ? 8 : 1], is2[DUAL ? 8 : 1];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            csum[k] = 0.f;
            if (SUMS2) csq[k] = 0.f;
            if (CONV == 2) {
                mu[k] = bnr && col_ok ? a.bnr_mean[ncol + k] : 0.f;
                is[k] = bnr && col_ok ? a.bnr_invstd[ncol + k] : 0.f;
            }
            if (DUAL) {
                csq2[k] = 0.f;
                mu2[k] = col_ok ? a.bnr_mean2[ncol + k] : 0.f;
                is2[k] = col_ok ? a.bnr_invstd2[ncol + k] : 0.f;
            }
        }

        // The tile leaves in two halves of 128 rows (64 KiB staged at a time).  A half is the rows EVERY wave holds in
        // four of its eight row groups (tile rows 64 hh .. + 63 and 128 + 64 hh .. + 63), so all 8 waves convert and
        // stage at once; with "half = the rows of waves 0-3 / 4-7" the four other waves waited through each staging
        // (PDT_GEMM_STAGE_BY_WAVE builds that older form).  grow(): tile row of staged row lr.
#ifdef PDT_GEMM_STAGE_BY_WAVE
        constexpr bool BY_I = false;
#else
        constexpr bool BY_I = true;
#endif
        auto grow = [&](int hh, int lr) { return BY_I ? hh * 64 + (lr & 63) + ((lr >> 6) << 7) : hh * 128 + lr; };
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            if (a.debug_nostore & 2) break;  // (timing diagnostics: no staging, no row pass)
            if (BY_I || wr == hh) {
                // lane holds, for 16x16 tile (i, j): row i*16 + lane%16, columns j*16 + (lane/16)*4 + 0..3
                const int wcol = wc * (BNT / 4) + lq * 4;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
                    const int n = on0 + wcol + j * 16;
                    if (CONV == 0 && a.bias != nullptr && n < a.N) bv = *reinterpret_cast<const f32x4_t*>(a.bias + n);  // N % 4 == 0
#pragma unroll
                    for (int ii = 0; ii < (BY_I ? 4 : 8); ++ii) {
                        const int i = BY_I ? hh * 4 + ii : ii;                  // accumulator row group
                        const int srow = (BY_I ? wr * 64 : 0) + ii * 16 + l15;  // staged row
                        const f32x4_t v = acc[i][j] + bv;
                        const bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                        *reinterpret_cast<bf16x4*>(Os + srow * OPITCH_T +
                                                   (((wc * (BNT / 32) + j * 2 + (lq >> 1)) ^ (l15 & 7)) << 4) +
                                                   (((lq & 1) ^ (l15 >> 3)) << 3)) = pk;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // coalesced pass over 128 rows: 8 passes of 16 rows, 16 B per lane, in batches of PB passes.  Everything
            // read from global memory is fetched for a whole batch first; every thread issues one store per pass
            // (masked ones are dropped by the descriptor's range check).
#pragma unroll
            for (int pb = 0; pb < NPASS; pb += PB) {
                bf16x8 ev[PB], bx[CONV == 2 ? PB : 1], bx2[DUAL ? PB : 1];
                unsigned abits[CONV == 2 ? PB : 1], bbits[CONV == 2 ? PB : 1];
                uint32_t coff[PB];
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int row = grow(hh, (pb + p) * RPP + orow);
                    coff[p] = col_ok ? (uint32_t)row * (uint32_t)(a.ldc * 2) + (uint32_t)(ncol * 2) : kOOB;
                }
                if (EXTRA != 0 || CONV == 2) {
#pragma unroll
                    for (int p = 0; p < PB; ++p) {
                        const int row = grow(hh, (pb + p) * RPP + orow);
                        const uint32_t eo = col_ok ? (uint32_t)row * (uint32_t)(lde * 2) + (uint32_t)(ncol * 2) : kOOB;
                        if (EXTRA != 0 || has_add) ev[p] = buffer_load_bf16x8_stream(re, eo);
                        if (CONV == 2) {
                            const bool in = col_ok && row < orows_m;
                            abits[p] = has_add ? 0xffu : 0u;
                            if (has_amask) abits[p] = in ? amask[eo >> 4] : 0u;
                            bbits[p] = 0u;
                            if (bnr) {
                                bx[p] = buffer_load_bf16x8_stream(rx, eo);
                                if (DUAL) bx2[p] = buffer_load_bf16x8_stream(rx2, eo);
                                bbits[p] = in ? bmask[eo >> 4] : 0u;
                            }
                        }
                    }
                }
#pragma unroll
                for (int p = 0; p < PB; ++p) {
                    const int lrow = (pb + p) * RPP + orow;
                    bf16x8 v = *reinterpret_cast<const bf16x8*>(Os + lrow * OPITCH_T + ((och ^ (orow & 7)) << 4));
                    if ((orow >> 3) & 1) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);  // rows with bit 3 set (wave-uniform)
                    if (ACT)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rp, coff[p], 0, kAuxStream);
                    if (ACT) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (bf16)gelu_f((float)v[k]);
                    }
                    if (EXTRA == 2) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] * gelu_grad_f((float)ev[p][k]));
                    }
                    if (EXTRA == 1) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] + (float)ev[p][k]);
                    }
                    if (COLSUM && grow(hh, lrow) < orows_m) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) csum[k] += (float)v[k];
                    }
                    if (CONV == 1) {  // (rows past M are zeros: their A rows were)
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const float f = (float)v[k];
                            csum[k] += f;
                            csq[k] += f * f;
                        }
                    }
                    if (CONV == 2) {
                        if (has_add) {
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                v[k] = (bf16)((float)v[k] + (((abits[p] >> k) & 1u) ? (float)ev[p][k] : 0.f));
                        }
                        if (bnr) {
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                const float gg = ((bbits[p] >> k) & 1u) ? (float)v[k] : 0.f;
                                csum[k] += gg;
                                csq[k] += gg * ((float)bx[p][k] - mu[k]) * is[k];
                                if (DUAL) csq2[k] += gg * ((float)bx2[p][k] - mu2[k]) * is2[k];
                            }
                        }
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rc, coff[p], 0, kAuxStream);
                }
            }
            if (hh == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();  // the staged rows are read before the other half overwrites them
            }
        }
        // The next tile's first K-tile was issued before this thread's 16 (32 with the pre-activation) stores and
        // vmcnt retires in issue order: at most that many operations outstanding means the DMA has landed.
        if (BNT == 256) {
            if (ACT) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {  // (2 halves x 4 passes: 8 stores per thread, 16 with the pre-activation)
            if (ACT) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        const bool want_sums = COLSUM || CONV == 1 || (CONV == 2 && bnr);
        if (want_sums) {
            // per-tile column sums of what was stored: threads och + 32 * r own the same 8 columns: lanes l and
            // l^32 in a wave, then the 8 waves through LDS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            float* Ss = reinterpret_cast<float*>(Os);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                csum[k] += __shfl_xor(csum[k], 32, 64);
                if (SUMS2) csq[k] += __shfl_xor(csq[k], 32, 64);
                if (DUAL) csq2[k] += __shfl_xor(csq2[k], 32, 64);
                if (BNT == 128) {  // 16 chunks per row: lanes l, l^16, l^32, l^48 own the same columns
                    csum[k] += __shfl_xor(csum[k], 16, 64);
                    if (SUMS2) csq[k] += __shfl_xor(csq[k], 16, 64);
                    if (DUAL) csq2[k] += __shfl_xor(csq2[k], 16, 64);
                }
            }
            if (lane < CPR) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    Ss[wave * BNT + och * 8 + k] = csum[k];
                    if (SUMS2) Ss[(8 + wave) * BNT + och * 8 + k] = csq[k];
                    if (DUAL) Ss[(16 + wave) * BNT + och * 8 + k] = csq2[k];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < BNT && on0 + tid < a.N) {
                float s = 0.f, q = 0.f, q2 = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) {
                    s += Ss[w * BNT + tid];
                    if (SUMS2) q += Ss[(8 + w) * BNT + tid];
                    if (DUAL) q2 += Ss[(16 + w) * BNT + tid];
                }
                if (DUAL) {  // same row layout as stats_slab: [sum g (shared), sum g * xhat2]
                    float* dst2 = a.stats_slab2 + (long)otile_m * 2 * a.N;
                    dst2[on0 + tid] = s;
                    dst2[a.N + on0 + tid] = q2;
                }
                if (SUMS2) {
                    float* dst = a.stats_slab + (long)otile_m * 2 * a.N;
                    dst[on0 + tid] = s;
                    dst[a.N + on0 + tid] = q;
                } else {
                    a.colsum_slab[(long)otile_m * a.N + on0 + tid] = s;
                }
            }
        }
        if (!has_next) break;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // staging reads done, the next K-tile 0 is visible to every wave
    }
}

}  // namespace

int gemm_m_tiles(long M) { return (int)((M + BM - 1) / BM); }

template <int BNT, int EXTRA, bool ACT, bool COLSUM, bool RAGGED, int CONV, bool TAPS = false, bool DUAL = false>
static void launch_variant(const GemmArgs& a, int grid, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt<EXTRA, ACT, COLSUM, RAGGED, CONV, TAPS, DUAL, BNT>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL((k_gemm_nt<EXTRA, ACT, COLSUM, RAGGED, CONV, TAPS, DUAL, BNT>), dim3(grid), dim3(512), LDS_BYTES, s, a);
}

// Tile width for a launch (GemmArgs::bn = 0): 128 only when the output is no wider than that.  Measured (round 3,
// profiles/r3_gemm_bn128.md): a 128-wide tile takes 0.62-0.65 of a square tile's time, not 0.5 (1.5x the operand
// bytes per FLOP through L2 -> LDS), so it loses even where it removes a mostly empty last round of square tiles
// (ViT-B/16 at B = 128, N = 768: 297 square tiles on 256 CUs, 59.0 us square vs 66.6 us narrow; the step 23.3 vs 24.5 ms)
// and on ResNet-50's 128-channel layers against the implicit-GEMM builds (15 285 vs 15 082 img/s at B = 1280).
// PDT_GEMM_BN=256|128 or set_gemm_tile_width() forces one.
static int g_forced_bn = [] {
    const char* e = std::getenv("PDT_GEMM_BN");
    return e ? std::atoi(e) : 0;
}();
void set_gemm_tile_width(int bn) { g_forced_bn = bn; }

static int pick_tile_width(const GemmArgs& a, int cus) {
    const int forced = g_forced_bn;
    if (a.bn == 128 || a.bn == 256) return a.bn;
    if (forced == 128 || forced == 256) return forced;
    (void)cus;
    return a.N <= 128 ? 128 : 256;
}

template <int BNT>
static void launch_width(GemmArgs& a, int cus, hipStream_t s) {
    a.n_tiles = (a.N + BNT - 1) / BNT;
    // (K-tiles are 64 elements = whole 128-B lines.  The same hint on the implicit-GEMM kernel's 64-B K-steps LOST 2.7 %
    // of the step: the second half of every line came from memory again.)  PDT_GEMM_A_STREAM=0: off.
    static const bool a_stream_on = [] {
        const char* e = getenv("PDT_GEMM_A_STREAM");
        return !(e && e[0] == '0');
    }();
    a.a_stream = a_stream_on && a.n_tiles == 1 && a.ntaps == 0 ? 1 : 0;
    // 32 tiles run together on one XCD: with many tile columns walk bands of 4 tile rows (4 A panels + 8 B panels
    // per K-step instead of 1 + 32)
    a.group_m = a.n_tiles > 12 ? 4 : 1;
    a.last_band = (a.m_tiles - 1) / a.group_m;
    a.last_rows = a.m_tiles - a.last_band * a.group_m;
    a.div_band = make_fastdiv((uint32_t)(a.group_m * a.n_tiles));
    a.div_gm = make_fastdiv((uint32_t)a.group_m);
    a.div_last = make_fastdiv((uint32_t)a.last_rows);
    const int total = a.m_tiles * a.n_tiles;
    const int grid = total < cus ? total : cus;
    const bool ragged = a.K % BK != 0, colsum = a.colsum_slab != nullptr;
    const int extra = a.residual ? 1 : (a.gelu_in ? 2 : 0);
    const bool act = a.act == 1;
    if (a.ntaps > 0) {
        // spatial filter: dense NHWC rows, K = channels per tap (a multiple of the K-tile), taps in the weight rows
        const bool fused = a.addend || a.bnr_x;
        if (a.bnr_x2 || ragged || extra || act || colsum || a.bias || a.ldc != a.N || a.lda != a.K || a.ntaps > 9 || a.H <= 0 ||
            a.W <= 0 || a.W > 126 || a.M % ((long)a.H * a.W) != 0 || a.M >= (1L << 31) || (a.conv_mode == 1 && !a.stats_slab) ||
            (a.conv_mode == 2 && (!fused || (a.bnr_x && !(a.stats_slab && a.bnr_mask && a.bnr_mean && a.bnr_invstd)) ||
                                  (a.addend_mask && !a.addend))) ||
            a.conv_mode > 2) {
            fprintf(stderr, "launch_gemm_nt: unsupported spatial-filter launch\n");
            abort();
        }
        a.div_w = make_fastdiv((uint32_t)a.W);
        a.div_hw = make_fastdiv((uint32_t)(a.H * a.W));
        a.div_kt = make_fastdiv((uint32_t)(a.K / BK));
        if (a.conv_mode == 1) launch_variant<BNT, 0, false, false, false, 1, true>(a, grid, s);
        else if (a.conv_mode == 2) launch_variant<BNT, 0, false, false, false, 2, true>(a, grid, s);
        else launch_variant<BNT, 0, false, false, false, 0, true>(a, grid, s);
        return;
    }
    if (a.conv_mode != 0) {
        // pointwise convolution in NHWC: dense rows (ld == channels), no bias
        const bool fused = a.addend || a.bnr_x;
        if (ragged || extra || act || colsum || a.bias || a.ldc != a.N || (a.conv_mode == 1 && !a.stats_slab) ||
            (a.conv_mode == 2 && (!fused || (a.bnr_x && !(a.stats_slab && a.bnr_mask && a.bnr_mean && a.bnr_invstd)) ||
                                  (a.addend_mask && !a.addend))) ||
            a.conv_mode > 2) {
            fprintf(stderr, "launch_gemm_nt: unsupported convolution epilogue\n");
            abort();
        }
        if (a.bnr_x2 && !(a.conv_mode == 2 && a.bnr_x && a.bnr_mean2 && a.bnr_invstd2 && a.stats_slab2)) {
            fprintf(stderr, "launch_gemm_nt: second BatchNorm operand without the first / its statistics\n");
            abort();
        }
        if (a.conv_mode == 1) launch_variant<BNT, 0, false, false, false, 1>(a, grid, s);
        else if (a.bnr_x2) launch_variant<BNT, 0, false, false, false, 2, false, true>(a, grid, s);
        else launch_variant<BNT, 0, false, false, false, 2>(a, grid, s);
        return;
    }
    if ((a.residual && a.gelu_in) || (act && (extra || colsum || !a.preact_out)) || (!act && a.preact_out) ||
        (ragged && (extra || act))) {
        fprintf(stderr, "launch_gemm_nt: unsupported epilogue combination\n");
        abort();
    }
    if (act) launch_variant<BNT, 0, true, false, false, 0>(a, grid, s);
    else if (extra == 1) colsum ? launch_variant<BNT, 1, false, true, false, 0>(a, grid, s) : launch_variant<BNT, 1, false, false, false, 0>(a, grid, s);
    else if (extra == 2) colsum ? launch_variant<BNT, 2, false, true, false, 0>(a, grid, s) : launch_variant<BNT, 2, false, false, false, 0>(a, grid, s);
    else if (ragged) colsum ? launch_variant<BNT, 0, false, true, true, 0>(a, grid, s) : launch_variant<BNT, 0, false, false, true, 0>(a, grid, s);
    else colsum ? launch_variant<BNT, 0, false, true, false, 0>(a, grid, s) : launch_variant<BNT, 0, false, false, false, 0>(a, grid, s);
}

int gemm_tile_width(const GemmArgs& a) {
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        return n;
    }();
    return pick_tile_width(a, cus);
}

void launch_gemm_nt(const GemmArgs& args, hipStream_t s) {
    GemmArgs a = args;
    if (a.M <= 0 || a.N <= 0) return;
    a.m_tiles = gemm_m_tiles(a.M);
    // timing diagnostics only (results are wrong): PDT_GEMM_NOSTORE, or PDT_GEMM_DEBUG = bit mask of 1 drop the
    // output stores, 2 skip the epilogue, 4 skip the MFMAs, 8 skip the operand DMA; re-read at every launch
    const char* dbg = std::getenv("PDT_GEMM_DEBUG");
    a.debug_nostore = (std::getenv("PDT_GEMM_NOSTORE") != nullptr ? 1 : 0) | (dbg ? std::atoi(dbg) : 0);
    static const int cus = [] {
        int dev = 0, n = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
        const char* e = std::getenv("PDT_GEMM_GRID");
        return e ? std::atoi(e) : n;
    }();
    if (pick_tile_width(a, cus) == 128) launch_width<128>(a, cus, s);
    else launch_width<256>(a, cus, s);
}

}  // namespace pdt
