// 3-channel stem (7x7 / stride 2 / pad 3, Cin = 3): operand packing for the matrix-core path.
//
// K = 7*7*3 = 147 does not tile the MFMA K step, and 3-channel pixels (6 B) cannot be moved in 16-B
// lanes.  The stem therefore runs on a packed image
//     xp[b][h + 3][w + 4][0..3] = x[b][h][w][0..2], 0      (zero frame: 3 rows / 4 columns, +1 spare)
// in which the 8 pixels x 4 channels under one filter ROW of an output pixel are 64 contiguous,
// 16-B aligned bytes starting at padded column 2*wo.  With the weights packed the same way
//     wp[n][kh][slot][c] = w[n][kh][slot - 1][c]   (slot 0 and c = 3 are zero)
// the forward conv is the generic implicit-GEMM kernel with 7 "taps" of 32 contiguous channels
// (conv_igemm.hip, ConvGeom::pix_elems = 4) and the weight gradient is the window kernel's MODE 1
// (conv_wgrad_win.hip).  K is 224 instead of 147: the stem is bound by its 64-channel output
// stream (B*112*112*64 bf16), not by MFMA work.
#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

// one thread = two packed pixels (16 B store); source pixels 2j-4, 2j-3 are both inside or both outside
__global__ __launch_bounds__(256) void k_stem_pack_x(const uint32_t* __restrict__ x, u32x4* __restrict__ xp, int H,
                                                     int W, int Hp, int Wp2, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i % Wp2);
    const long t = i / Wp2;
    const int hp = (int)(t % Hp);
    const long b = t / Hp;
    const int h = hp - 3, w = 2 * j - 4;
    u32x4 v = {0u, 0u, 0u, 0u};
    if ((unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W) {
        // 6 bf16 = 3 dwords at a 4-B aligned address (w is even)
        const uint32_t* src = x + (((b * H + h) * W + w) * 3) / 2;
        const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];  // [r0 g0] [b0 r1] [g1 b1]
        v[0] = d0;
        v[1] = d1 & 0xffffu;
        v[2] = (d1 >> 16) | (d2 << 16);
        v[3] = d2 >> 16;
    }
    st_stream(xp + i, v);
}

__global__ __launch_bounds__(256) void k_stem_pack_w(const bf16* __restrict__ w, bf16* __restrict__ wp, int total) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = i & 3, slot = (i >> 2) & 7, row = i >> 5;  // row = n*7 + kh
    bf16 v = (bf16)0.f;
    if (slot >= 1 && c < 3) v = w[(row * 7 + (slot - 1)) * 3 + c];
    wp[i] = v;
}

}  // namespace

void launch_stem_pack_x(const void* x, void* xp, int B, int H, int W, hipStream_t s) {
    const int Hp = H + 6, Wp2 = (W + 8) / 2;
    const long total = (long)B * Hp * Wp2;
    hipLaunchKernelGGL(k_stem_pack_x, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const uint32_t*)x,
                       (u32x4*)xp, H, W, Hp, Wp2, total);
}

void launch_stem_pack_w(const void* w, void* wp, int N, hipStream_t s) {
    const int total = N * 7 * 32;
    hipLaunchKernelGGL(k_stem_pack_w, dim3((total + 255) / 256), dim3(256), 0, s, (const bf16*)w, (bf16*)wp, total);
}

}  // namespace pdt
