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
constexpr int OPITCH = BN * 2 + 16;            // staged output row pitch (bytes)
constexpr int LDS_BYTES = BM * OPITCH;         // 135168 >= 2 * STAGE_BYTES

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

// byte offset of 16-B chunk `chunk` of row `row` inside a half-tile of 128-B rows
__device__ __forceinline__ int half_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(512, 2) void k_gemm_nt(const GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // XCD-aware, bijective block -> tile map (blocks b and b+8 share an XCD): contiguous tile ranges per XCD
    const int total = a.m_tiles * a.n_tiles;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = total >> 3, r = total & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tile_n = tile % a.n_tiles, tile_m = tile / a.n_tiles;
    const long m0 = (long)tile_m * BM;
    const int n0 = tile_n * BN;
    const int rows_m = (int)((a.M - m0) < BM ? (a.M - m0) : BM);   // valid rows of this tile
    const int rows_n = (a.N - n0) < BN ? (a.N - n0) : BN;

    // descriptors re-based at the tile's first row: offsets inside stay far below 2^31
    const char* Abase = static_cast<const char*>(a.A) + m0 * a.lda * 2;
    const char* Bbase = static_cast<const char*>(a.B) + (long)n0 * a.ldb * 2;
    const i32x4 ra = make_rsrc_words(Abase, (uint32_t)((long)rows_m * a.lda * 2));
    const i32x4 rb = make_rsrc_words(Bbase, (uint32_t)((long)rows_n * a.ldb * 2));

    // ---- DMA source offsets: in every phase each thread issues 2 pieces (j = 0, 1) of one half-tile.
    // piece pc = wave * 2 + j covers rows pc*8 .. pc*8+7 (1 KiB); lane -> row pc*8 + lane/8, physical chunk lane%8,
    // which holds logical chunk (lane%8) ^ ((row >> 1) & 7).
    uint32_t voffA[2][2], voffB[2][2];
    int lch[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pc = wave * 2 + j;
        const int row = pc * 8 + (lane >> 3);
        lch[j] = (lane & 7) ^ ((row >> 1) & 7);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = h * 128 + row;
            voffA[h][j] = r < rows_m ? (uint32_t)r * (uint32_t)(a.lda * 2) + (uint32_t)(lch[j] * 16) : kOOB;
            voffB[h][j] = r < rows_n ? (uint32_t)r * (uint32_t)(a.ldb * 2) + (uint32_t)(lch[j] * 16) : kOOB;
        }
    }
    const uint32_t lds0 = lds_address(smem);
    const uint32_t piece_base = lds0 + (uint32_t)wave * 2048u;  // + j * 1024

    const int nt = (a.K + BK - 1) / BK;
    const bool ragged_k = (a.K % BK) != 0;
    // slot: 0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi
    auto issue = [&](int slot, int t) {
        const uint32_t dst = piece_base + (uint32_t)((t & 1) * STAGE_BYTES + slot * HALF_BYTES);
        const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(t * BK * 2);
        const bool isB = slot >= 2;
        const int h = slot & 1;
        uint32_t v0 = isB ? voffB[h][0] : voffA[h][0];
        uint32_t v1 = isB ? voffB[h][1] : voffA[h][1];
        if (ragged_k && t == nt - 1) {  // chunks past the end of the reduction axis read zeros
            if (t * BK + lch[0] * 8 >= a.K) v0 = kOOB;
            if (t * BK + lch[1] * 8 >= a.K) v1 = kOOB;
        }
        if (isB) {
            lds_dma16s(rb, v0, soff, dst);
            lds_dma16s(rb, v1, soff, dst + 1024u);
        } else {
            lds_dma16s(ra, v0, soff, dst);
            lds_dma16s(ra, v1, soff, dst + 1024u);
        }
    };

    // ---- fragment read addresses (bytes inside a stage buffer)
    const int l15 = lane & 15, lq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int lo = l15 * 128 + (((ks * 4 + lq) ^ (l15 >> 1)) << 4);
        a_off[ks] = wr * HALF_BYTES + lo;
        b_off[ks] = 2 * HALF_BYTES + (wc >> 1) * HALF_BYTES + (wc & 1) * 64 * 128 + lo;
    }

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: all of tile 0, and B-lo of tile 1
    issue(0, 0);
    issue(1, 0);
    issue(2, 0);
    issue(3, 0);
    if (nt > 1) {
        issue(2, 1);
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // waves 4-7 run one phase behind their SIMD partners

    bf16x8 bfrag[2][4];
    for (int t = 0; t < nt; ++t) {
        const char* st = smem + (t & 1) * STAGE_BYTES;
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
                for (int j = 0; j < 4; ++j)
                    bfrag[ks][j] = *reinterpret_cast<const bf16x8*>(st + b_off[ks] + j * 16 * 128);
            }
            if (ph == 0) {
                if (more1) issue(3, t + 1);
            } else if (ph == 1) {
                if (more1) issue(0, t + 1);
            } else if (ph == 2) {
                if (more1) issue(1, t + 1);
            } else {
                if (more2) {
                    issue(2, t + 2);
                    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");  // everything of tile t+1 has landed
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            __builtin_amdgcn_s_barrier();
            // ---------------- MFMA segment
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[mh * 4 + i][j] =
                        __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfrag[ks][j], afrag[i], acc[mh * 4 + i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // re-align the two groups: every fragment read is done after this

    // ------------------------------------------------------------------------------------ epilogue
    // lane holds, for 16x16 tile (i, j): row m = i*16 + lane%16, columns j*16 + (lane/16)*4 + 0..3
    {
        const int ncol = wc * 64 + lq * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4_t bv = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int n = n0 + ncol + j * 16;
            if (a.bias != nullptr && n < a.N) bv = *reinterpret_cast<const f32x4_t*>(a.bias + n);  // N % 4 == 0
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f32x4_t v = acc[i][j] + bv;
                const bf16x4 pk = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                const int row = wr * 128 + i * 16 + l15;
                *reinterpret_cast<bf16x4*>(smem + row * OPITCH + (ncol + j * 16) * 2) = pk;
            }
        }
    }
    __syncthreads();

    const int och = tid & 31, orow = tid >> 5;  // 32 chunks of 8 columns per row, 16 rows per pass
    const int ncol = n0 + och * 8;
    const bool col_ok = ncol < a.N;  // N % 8 == 0
    char* Cbase = static_cast<char*>(a.C) + m0 * a.ldc * 2;
    const char* Rbase = a.residual ? static_cast<const char*>(a.residual) + m0 * a.ldr * 2 : nullptr;
    const char* Ubase = a.gelu_in ? static_cast<const char*>(a.gelu_in) + m0 * a.ldu * 2 : nullptr;
    char* Pbase = a.preact_out ? static_cast<char*>(a.preact_out) + m0 * a.ldc * 2 : nullptr;
    float csum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) csum[k] = 0.f;
#pragma unroll 4
    for (int p = 0; p < BM / 16; ++p) {
        const int row = p * 16 + orow;
        if (row < rows_m && col_ok) {
            bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * OPITCH + och * 16);
            if (Pbase != nullptr) *reinterpret_cast<bf16x8*>(Pbase + ((long)row * a.ldc + ncol) * 2) = v;
            if (a.act == 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (bf16)gelu_f((float)v[k]);
            }
            if (Ubase != nullptr) {
                const bf16x8 u = *reinterpret_cast<const bf16x8*>(Ubase + ((long)row * a.ldu + ncol) * 2);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] * gelu_grad_f((float)u[k]));
            }
            if (Rbase != nullptr) {
                const bf16x8 r = *reinterpret_cast<const bf16x8*>(Rbase + ((long)row * a.ldr + ncol) * 2);
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (bf16)((float)v[k] + (float)r[k]);
            }
            if (a.colsum_slab != nullptr) {
#pragma unroll
                for (int k = 0; k < 8; ++k) csum[k] += (float)v[k];
            }
            *reinterpret_cast<bf16x8*>(Cbase + ((long)row * a.ldc + ncol) * 2) = v;
        }
    }
    if (a.colsum_slab != nullptr) {
        // per-tile column sums of what was stored (bias gradient of the layer that produced this gradient):
        // threads och + 32 * r own the same 8 columns: lanes l and l^32 in a wave, then the 8 waves through LDS
        __syncthreads();
        float* Ss = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 8; ++k) csum[k] += __shfl_xor(csum[k], 32, 64);
        if (lane < 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) Ss[wave * BN + och * 8 + k] = csum[k];
        }
        __syncthreads();
        if (tid < BN && n0 + tid < a.N) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += Ss[w * BN + tid];
            a.colsum_slab[(long)tile_m * a.N + n0 + tid] = s;
        }
    }
}

}  // namespace

int gemm_m_tiles(long M) { return (int)((M + BM - 1) / BM); }

void launch_gemm_nt(const GemmArgs& args, hipStream_t s) {
    GemmArgs a = args;
    a.m_tiles = gemm_m_tiles(a.M);
    a.n_tiles = (a.N + BN - 1) / BN;
    static bool attr_set = false;
    if (!attr_set) {
        PDT_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_nt),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    if (a.M <= 0 || a.N <= 0) return;
    hipLaunchKernelGGL(k_gemm_nt, dim3(a.m_tiles * a.n_tiles), dim3(512), LDS_BYTES, s, a);
}

}  // namespace pdt
