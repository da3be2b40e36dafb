// ResNet stem max-pool (3x3 / stride 2 / pad 1) and the global average pool, NHWC bf16
// (SURVEY 2.4 N9).  A thread owns 8 consecutive channels (16 B) of one output pixel.
// Forward records the winning tap (0..8, first maximum in scan order like ATen) so backward
// is a gather over the <= 4 windows that cover an input pixel: no atomics, deterministic.
#include <cstdlib>
#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

__device__ __forceinline__ bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }

// Blocks are dealt round-robin to the 8 XCDs; windows of neighbouring output rows share input rows, so give each
// XCD runs of 64 consecutive logical blocks (about 37 image rows of the stem pool): the shared rows are then hit in
// that XCD's L2 instead of being fetched by two of them, while the chip still walks the tensor front to back.
__device__ __forceinline__ uint32_t xcd_grouped_block() {
    constexpr uint32_t GRP = 64;
    const uint32_t bid = blockIdx.x, full = gridDim.x / (8 * GRP) * (8 * GRP);
    if (bid >= full) return bid;
    const uint32_t xcd = bid & 7, slot = bid >> 3;
    return ((slot / GRP) * 8 + xcd) * GRP + slot % GRP;
}

struct PoolDiv {
    FastDiv g, w, h;  // channel groups, width, height of the index space a thread id is decomposed over
};

__global__ __launch_bounds__(256) void k_maxpool_fwd(const bf16* __restrict__ x, bf16* __restrict__ y,
                                                     uint8_t* __restrict__ argmax, int B, int H, int W, int C, int Ho,
                                                     int Wo, PoolDiv dv) {
    // 32-bit indices and multiply-shift divisions: with 64-bit '/' and '%' by runtime values the index math alone
    // cost more than the memory traffic (2.6 TB/s); the host checks that the thread count fits 31 bits
    const uint32_t G = (uint32_t)C >> 3;
    const uint32_t total = (uint32_t)B * Ho * Wo * G;
    for (uint32_t i = xcd_grouped_block() * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t p = fdiv(i, dv.g);
        const int cg = (int)(i - p * G);
        uint32_t t = fdiv(p, dv.w);
        const int wo = (int)(p - t * dv.w.d);
        const uint32_t bb = fdiv(t, dv.h);
        const int ho = (int)(t - bb * dv.h.d);
        const int b = (int)bb;
        float best[8];
        uint8_t arg[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            best[k] = -INFINITY;
            arg[k] = 0;
        }
        bool any = false;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hi = ho * 2 - 1 + r;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int wi = wo * 2 - 1 + s;
                if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                const bf16x8 v = ld8(x + (uint32_t)(((b * H + hi) * W + wi) * C + cg * 8));
                const uint8_t tap = (uint8_t)(r * 3 + s);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float f = (float)v[k];
                    if (!any) arg[k] = tap;            // ATen: index starts at the first in-bounds tap
                    if (f > best[k] || f != f) {       // strict '>' keeps the first maximum; NaN propagates
                        best[k] = f;
                        arg[k] = tap;
                    }
                }
                any = true;
            }
        }
        const uint32_t off = (uint32_t)(((b * Ho + ho) * Wo + wo) * C + cg * 8);
        bf16x8 o;
        uint32_t plo = 0, phi = 0;  // (32-bit words: 64-bit shifts run at a quarter of the rate)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            plo |= (uint32_t)arg[k] << (8 * k);
            phi |= (uint32_t)arg[4 + k] << (8 * k);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16)best[k];
        *reinterpret_cast<bf16x8*>(y + off) = o;
        *reinterpret_cast<uint2*>(argmax + off) = make_uint2(plo, phi);
    }
}

constexpr int kPoolUnroll = 4;  // full-resolution pixels per thread in the backward kernel

// One workgroup = one contiguous chunk of kPoolUnroll * (256 / G) pixels (16 KiB of dx at C = 64); a thread owns one
// channel group of kPoolUnroll pixels, R apart, all of whose window loads are issued before the first use.
// (Measured at B=1024: 0.78 ms = 2.9 TB/s, the same with one pixel per thread, with or without the XCD grouping;
// 0.49 ms with the loads removed and 0.46 ms with the store removed, i.e. not a bandwidth limit: ~300 vector
// instructions per 16 B written is what is left to remove.)
__global__ __launch_bounds__(256) void k_maxpool_bwd(const bf16* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                     bf16* __restrict__ dx, int B, int H, int W, int C, int Ho,
                                                     int Wo, PoolDiv dv) {
    const uint32_t G = (uint32_t)C >> 3, R = 256u / G;
    const uint32_t cg = threadIdx.x % G, rl = threadIdx.x / G;
    const uint32_t M = (uint32_t)B * H * W;
    const uint32_t p0 = xcd_grouped_block() * (kPoolUnroll * R) + rl;
    // The windows that cover (hi, wi): rows hi/2 and (hi+1)/2 (one row when hi is even: its tap row is 1; two when
    // odd: tap rows 2 and 0), same for columns -> 1, 2 or 4 windows.  All slots of all pixels are LOADED up front
    // from clamped coordinates; a slot that does not exist gets a tap code no byte can match.
    bf16x8 gy[kPoolUnroll][4];
    uint2 am[kPoolUnroll][4];
    uint32_t code[kPoolUnroll][4];
#pragma unroll
    for (int u = 0; u < kPoolUnroll; ++u) {
        const uint32_t p = min(p0 + u * R, M - 1);
        const uint32_t t = fdiv(p, dv.w);
        const int wi = (int)(p - t * dv.w.d);
        const uint32_t bb = fdiv(t, dv.h);
        const int hi = (int)(t - bb * dv.h.d);
        const int b = (int)bb;
        const int ho2[2] = {hi >> 1, (hi + 1) >> 1}, wo2[2] = {wi >> 1, (wi + 1) >> 1};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ho = ho2[a], wo = wo2[c];
                const bool ok = (a == 0 || ho2[1] != ho2[0]) && (c == 0 || wo2[1] != wo2[0]) && ho < Ho && wo < Wo;
                const uint32_t off = (uint32_t)(((b * Ho + min(ho, Ho - 1)) * Wo + min(wo, Wo - 1)) * C) + cg * 8;
                gy[u][a * 2 + c] = ld8(dy + off);
                am[u][a * 2 + c] = *reinterpret_cast<const uint2*>(argmax + off);
                code[u][a * 2 + c] = ok ? (uint32_t)((hi - (ho * 2 - 1)) * 3 + (wi - (wo * 2 - 1))) : 0xffu;
            }
    }
#pragma unroll
    for (int u = 0; u < kPoolUnroll; ++u) {
        const uint32_t p = p0 + u * R;
        if (p >= M) break;
        float acc[8] = {0};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (((am[u][q].x >> (8 * k)) & 0xffu) == code[u][q]) acc[k] += (float)gy[u][q][k];
                if (((am[u][q].y >> (8 * k)) & 0xffu) == code[u][q]) acc[4 + k] += (float)gy[u][q][4 + k];
            }
        }
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16)acc[k];
        st_stream(dx + p * (uint32_t)C + cg * 8, __builtin_bit_cast(u32x4, o));
    }
}

// Even H and W: a thread owns a 2x2 block of full-resolution pixels (h = 2i + a, w = 2j + b) of one channel group.
// Exactly the four windows (i + da, j + dc) touch it, and each (window, tap) pair lands on one known pixel of the
// block: window (0,0) taps 4 / 5 / 7 / 8 -> pixels (0,0) / (0,1) / (1,0) / (1,1), window (0,1) taps 3 / 6 -> (0,1) /
// (1,1), window (1,0) taps 1 / 2 -> (1,0) / (1,1), window (1,1) tap 0 -> (1,1).  Four window loads and nine compares
// per channel for four pixels, where the per-pixel kernel above loads nine windows and compares sixteen times (it was
// instruction-bound, see its comment); the sums run in the same order, so the results are bit-identical.
__global__ __launch_bounds__(256) void k_maxpool_bwd_2x2(const bf16* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                         bf16* __restrict__ dx, int B, int H, int W, int C, int Ho,
                                                         int Wo, PoolDiv dv /* .w: W / 2, .h: H / 2 */) {
    const uint32_t G = (uint32_t)C >> 3, R = 256u / G;
    const uint32_t cg = threadIdx.x % G, rl = threadIdx.x / G;
    const uint32_t Hb = (uint32_t)H >> 1, Wb = (uint32_t)W >> 1;
    const uint32_t p = xcd_grouped_block() * R + rl;
    if (p >= (uint32_t)B * Hb * Wb) return;
    const uint32_t t = fdiv(p, dv.w);
    const int j = (int)(p - t * Wb);
    const uint32_t bb = fdiv(t, dv.h);
    const int i = (int)(t - bb * Hb);
    const int b = (int)bb;
    bf16x8 gy[4];
    uint2 am[4];
#pragma unroll
    for (int da = 0; da < 2; ++da)
#pragma unroll
        for (int dc = 0; dc < 2; ++dc) {
            const int ho = i + da, wo = j + dc;
            const bool ok = ho < Ho && wo < Wo;
            const uint32_t off = (uint32_t)(((b * Ho + min(ho, Ho - 1)) * Wo + min(wo, Wo - 1)) * C) + cg * 8;
            gy[da * 2 + dc] = ld8(dy + off);
            const uint2 a = *reinterpret_cast<const uint2*>(argmax + off);
            am[da * 2 + dc] = ok ? a : make_uint2(0xffffffffu, 0xffffffffu);  // (no tap code matches 0xff)
        }
    bf16x8 o[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int sh = 8 * (k & 3);
        uint32_t c[4];
        float g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            c[q] = ((k < 4 ? am[q].x : am[q].y) >> sh) & 0xffu;
            g[q] = (float)gy[q][k];
        }
        o[0][k] = (bf16)(c[0] == 4u ? g[0] : 0.f);
        o[1][k] = (bf16)((c[0] == 5u ? g[0] : 0.f) + (c[1] == 3u ? g[1] : 0.f));
        o[2][k] = (bf16)((c[0] == 7u ? g[0] : 0.f) + (c[2] == 1u ? g[2] : 0.f));
        o[3][k] = (bf16)((((c[0] == 8u ? g[0] : 0.f) + (c[1] == 6u ? g[1] : 0.f)) + (c[2] == 2u ? g[2] : 0.f)) +
                         (c[3] == 0u ? g[3] : 0.f));
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t pix = (uint32_t)((b * H + 2 * i + a) * W + 2 * j + e);
            st_stream(dx + pix * (uint32_t)C + cg * 8, __builtin_bit_cast(u32x4, o[a * 2 + e]));
        }
}

// y[b][c] = mean over HW; thread = (b, 8 channels)
__global__ __launch_bounds__(256) void k_avgpool_fwd(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int HW,
                                                     int C) {
    const int G = C >> 3;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * G) return;
    const int cg = (int)(i % G);
    const int b = (int)(i / G);
    float acc[8] = {0};
    const bf16* p = x + (long)b * HW * C + cg * 8;
    for (int r = 0; r < HW; ++r) {
        const bf16x8 v = ld8(p + (long)r * C);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
    }
    bf16x8 o;
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)(acc[k] * inv);
    *reinterpret_cast<bf16x8*>(y + (long)b * C + cg * 8) = o;
}

__global__ __launch_bounds__(256) void k_avgpool_bwd(const bf16* __restrict__ dy, bf16* __restrict__ dx, int B, int HW,
                                                     int C, PoolDiv dv) {
    const uint32_t G = (uint32_t)C >> 3;
    const uint32_t total = (uint32_t)B * HW * G;
    const float inv = 1.f / (float)HW;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t p = fdiv(i, dv.g);
        const int cg = (int)(i - p * G);
        const int b = (int)fdiv(p, dv.w);  // dv.w divides by HW here
        const bf16x8 g = ld8(dy + (long)b * C + cg * 8);
        bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16)((float)g[k] * inv);
        *reinterpret_cast<bf16x8*>(dx + (long)i * 8) = o;
    }
}

// One 4 KiB output chunk per block, blocks in address order: the chip walks the tensors front to back (a capped
// grid-stride loop keeps gridDim far-apart streams open and loses ~15 % of HBM bandwidth, see bn.hip).
inline int grid_cap(long total) {
    long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > (1L << 30) ? (1L << 30) : b));
}

inline void check_index_space(long total) {
    if (total >= (1L << 31)) {
        fprintf(stderr, "pool: %ld work items do not fit the 31-bit index space of the kernels\n", total);
        abort();
    }
}

}  // namespace

void launch_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int B, int H, int W, int C, int Ho, int Wo,
                             hipStream_t s) {
    const long total = (long)B * Ho * Wo * (C >> 3);
    check_index_space(total);
    const PoolDiv dv{make_fastdiv((uint32_t)(C >> 3)), make_fastdiv((uint32_t)Wo), make_fastdiv((uint32_t)Ho)};
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(grid_cap(total)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, argmax, B, H,
                       W, C, Ho, Wo, dv);
}
void launch_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int H, int W, int C, int Ho,
                             int Wo, hipStream_t s) {
    const long total = (long)B * H * W * (C >> 3);
    check_index_space(total);
    const PoolDiv dv{make_fastdiv((uint32_t)(C >> 3)), make_fastdiv((uint32_t)W), make_fastdiv((uint32_t)H)};
    if (256 % (C >> 3) != 0) {
        fprintf(stderr, "maxpool bwd: C / 8 must divide 256 (got C = %d)\n", C);
        abort();
    }
    static const bool blocked = [] {
        const char* e = getenv("PDT_POOL_BWD_2X2");
        return !(e && e[0] == '0');
    }();
    if (blocked && H % 2 == 0 && W % 2 == 0 && Ho == H / 2 && Wo == W / 2) {
        const PoolDiv d2{make_fastdiv((uint32_t)(C >> 3)), make_fastdiv((uint32_t)(W / 2)), make_fastdiv((uint32_t)(H / 2))};
        const long per = 256 / (C >> 3);
        const long nb = ((long)B * (H / 2) * (W / 2) + per - 1) / per;
        hipLaunchKernelGGL(k_maxpool_bwd_2x2, dim3((unsigned)nb), dim3(256), 0, s, (const bf16*)dy, argmax, (bf16*)dx, B,
                           H, W, C, Ho, Wo, d2);
        return;
    }
    const long per_block = (long)kPoolUnroll * (256 / (C >> 3));
    const long blocks = ((long)B * H * W + per_block - 1) / per_block;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)dy, argmax, (bf16*)dx, B, H,
                       W, C, Ho, Wo, dv);
}
void launch_global_avgpool_fwd(const void* x, void* y, int B, int HW, int C, hipStream_t s) {
    const long total = (long)B * (C >> 3);
    hipLaunchKernelGGL(k_avgpool_fwd, dim3((int)((total + 255) / 256)), dim3(256), 0, s, (const bf16*)x, (bf16*)y, B,
                       HW, C);
}
void launch_global_avgpool_bwd(const void* dy, void* dx, int B, int HW, int C, hipStream_t s) {
    const long total = (long)B * HW * (C >> 3);
    check_index_space(total);
    const PoolDiv dv{make_fastdiv((uint32_t)(C >> 3)), make_fastdiv((uint32_t)HW), make_fastdiv(1u)};
    hipLaunchKernelGGL(k_avgpool_bwd, dim3(grid_cap(total)), dim3(256), 0, s, (const bf16*)dy, (bf16*)dx, B, HW, C, dv);
}

}  // namespace pdt
