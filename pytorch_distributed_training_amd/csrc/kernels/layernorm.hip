// LayerNorm over the last dimension for token-major activations [M, D] (ViT: D = 768), bf16 in / bf16 out,
// fp32 statistics and affine parameters.
//
// Any width D % 4 == 0 up to 2048 (ViT-Tiny 192 ... ViT-Huge 1280 and beyond): the kernels are built per piece COUNT
// NP = ceil(D / 256) and mask the pieces past the end of a row, so D = 768 still runs the exact-fit code.
// One wave per row: D / 64 elements per lane as 8-byte (4 x bf16) pieces, so a row is read ONCE into
// registers, mean and variance are two in-register passes (no E[x^2] - E[x]^2 cancellation) and the
// normalised row is written straight back: 2 bytes read + 2 bytes written per element, against the
// cast -> fp32 LayerNorm -> cast sequence (14 bytes per element) it replaces.
// Backward: one pass for dx; the per-column dgamma / dbeta sums stay in registers across all rows a wave
// visits (a lane always owns the same columns), are combined per block in LDS and written as partial
// rows [blocks][2][D] that a fixed-order second launch sums (deterministic, no float atomics).
#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

constexpr int kMaxPieces = 8;  // D <= 2048

template <int NP>
__global__ __launch_bounds__(256) void k_ln_fwd(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, bf16* __restrict__ y,
                                                float* __restrict__ mean_out, float* __restrict__ rstd_out, long M,
                                                int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int np4 = D >> 2;  // pieces of the row; piece lane + 64 j exists iff it is below
    const float inv_d = 1.f / (float)D;
    const bf16x4* xr = reinterpret_cast<const bf16x4*>(x + row * D);
    float v[NP][4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        bf16x4 t = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        if (lane + 64 * j < np4) t = xr[lane + 64 * j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[j][k] = (float)t[k];
            s += v[j][k];
        }
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = lane + 64 * j < np4 ? v[j][k] - mean : 0.f;
            q += d * d;
        }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
    bf16x4* yr = reinterpret_cast<bf16x4*>(y + row * D);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        if (lane + 64 * j >= np4) break;
        const f32x4 g = reinterpret_cast<const f32x4*>(gamma)[lane + 64 * j];
        const f32x4 b = reinterpret_cast<const f32x4*>(beta)[lane + 64 * j];
        bf16x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (bf16)((v[j][k] - mean) * rstd * g[k] + b[k]);
        yr[lane + 64 * j] = o;
    }
    if (lane == 0) {
        mean_out[row] = mean;
        rstd_out[row] = rstd;
    }
}

template <int NP>
__global__ __launch_bounds__(256) void k_ln_bwd(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                const float* __restrict__ gamma, bf16* __restrict__ dx,
                                                float* __restrict__ slab /*[gridDim.x][2][D]*/, long M, int D,
                                                const bf16* __restrict__ addend /* or null: added to dx */) {
    constexpr int DP = NP * 256;
    __shared__ float red[3][2][DP];  // waves 1..3 park their column sums here
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np4 = D >> 2;
    const float inv_d = 1.f / (float)D;
    float g[NP][4], dgam[NP][4], dbet[NP][4];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
        if (lane + 64 * j < np4) t = reinterpret_cast<const f32x4*>(gamma)[lane + 64 * j];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            g[j][k] = t[k];
            dgam[j][k] = 0.f;
            dbet[j][k] = 0.f;
        }
    }
    // Two rows per iteration and wave (R = 2): their loads are all issued before the first reduction, so twice the bytes
    // are in flight per wave and the four wave reductions of the pair overlap (one row at a time ran at 2.8 TB/s on
    // [25216, 768], half of what the forward kernel reaches).
    constexpr int R = NP <= 4 ? 2 : 1;
    const long stride = (long)gridDim.x * 4 * R;
    for (long row0 = ((long)blockIdx.x * 4 + wave) * R; row0 < M; row0 += stride) {
        float xh[R][NP][4], gg[R][NP][4], s1[R], s2[R], rs_[R];
        bf16x4 ta[R][NP];  // the residual-path gradient, fetched with x and dy (not after the reductions)
        bool live[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long row = row0 + r;
            live[r] = row < M;
            const bf16x4* xr = reinterpret_cast<const bf16x4*>(x + row * D);
            const bf16x4* dr = reinterpret_cast<const bf16x4*>(dy + row * D);
            const bf16x4* ar = reinterpret_cast<const bf16x4*>(addend + row * D);
            const float mu = live[r] ? mean[row] : 0.f;
            rs_[r] = live[r] ? rstd[row] : 0.f;
            s1[r] = 0.f;
            s2[r] = 0.f;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const bool in = live[r] && lane + 64 * j < np4;
                bf16x4 tx = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, td = tx;
                ta[r][j] = tx;
                if (in) {
                    tx = xr[lane + 64 * j];
                    td = dr[lane + 64 * j];
                    if (addend) ta[r][j] = ar[lane + 64 * j];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float d = (float)td[k];
                    xh[r][j][k] = in ? ((float)tx[k] - mu) * rs_[r] : 0.f;
                    gg[r][j][k] = d * g[j][k];
                    s1[r] += gg[r][j][k];
                    s2[r] += gg[r][j][k] * xh[r][j][k];
                    dgam[j][k] += d * xh[r][j][k];
                    dbet[j][k] += d;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float c1 = wave_sum(s1[r]) * inv_d, c2 = wave_sum(s2[r]) * inv_d;
            if (!live[r]) continue;
            const long row = row0 + r;
            const float rs = rs_[r];
            bf16x4* ox = reinterpret_cast<bf16x4*>(dx + row * D);
            // the gradient that reaches the same tensor through the residual connection joins here (pre-LN
            // transformer block: d(x) = d(skip) + LN'(d(branch))) instead of in a separate elementwise add
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                if (lane + 64 * j >= np4) break;
                bf16x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o[k] = (bf16)(rs * (gg[r][j][k] - c1 - xh[r][j][k] * c2) + (float)ta[r][j][k]);  // (zeros without an addend)
                ox[lane + 64 * j] = o;
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                red[wave - 1][0][(lane + 64 * j) * 4 + k] = dbet[j][k];
                red[wave - 1][1][(lane + 64 * j) * 4 + k] = dgam[j][k];
            }
    }
    __syncthreads();
    if (wave == 0) {
        float* out = slab + (long)blockIdx.x * 2 * D;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            if (lane + 64 * j >= np4) break;
            f32x4 ob, og;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = (lane + 64 * j) * 4 + k;
                ob[k] = dbet[j][k] + red[0][0][c] + red[1][0][c] + red[2][0][c];
                og[k] = dgam[j][k] + red[0][1][c] + red[1][1][c] + red[2][1][c];
            }
            reinterpret_cast<f32x4*>(out)[lane + 64 * j] = ob;
            reinterpret_cast<f32x4*>(out + D)[lane + 64 * j] = og;
        }
    }
}

}  // namespace

bool layernorm_dim_ok(int D) { return D >= 4 && D % 4 == 0 && D <= kMaxPieces * 256; }

int layernorm_bwd_blocks(long M) {
    long b = (M + 7) / 8;  // (a wave takes two rows per iteration up to D = 1024)
    if (b < 1) b = 1;
    return (int)(b < 1024 ? b : 1024);
}

#define PDT_LN_CASE(N, ...) case N: { constexpr int NP = N; __VA_ARGS__; } break;
#define PDT_LN_DISPATCH(D, ...)                                                                                   \
    switch (((D) + 255) / 256) {                                                                                  \
        PDT_LN_CASE(1, __VA_ARGS__) PDT_LN_CASE(2, __VA_ARGS__) PDT_LN_CASE(3, __VA_ARGS__) PDT_LN_CASE(4, __VA_ARGS__) \
        PDT_LN_CASE(5, __VA_ARGS__) \
        PDT_LN_CASE(6, __VA_ARGS__) PDT_LN_CASE(7, __VA_ARGS__) PDT_LN_CASE(8, __VA_ARGS__) \
        default: abort();                                                                                         \
    }

void launch_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                          long M, int D, float eps, hipStream_t s) {
    const unsigned grid = (unsigned)((M + 3) / 4);
    PDT_LN_DISPATCH(D, hipLaunchKernelGGL((k_ln_fwd<NP>), dim3(grid), dim3(256), 0, s, (const bf16*)x, gamma, beta,
                                          (bf16*)y, mean, rstd, M, D, eps));
}

void launch_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                          void* dx, float* slab, int blocks, long M, int D, const void* addend, hipStream_t s) {
    PDT_LN_DISPATCH(D, hipLaunchKernelGGL((k_ln_bwd<NP>), dim3(blocks), dim3(256), 0, s, (const bf16*)dy,
                                          (const bf16*)x, mean, rstd, gamma, (bf16*)dx, slab, M, D, (const bf16*)addend));
}

}  // namespace pdt
