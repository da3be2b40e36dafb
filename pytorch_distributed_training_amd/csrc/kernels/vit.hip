// Small streaming kernels around the ViT encoder (BASELINE.json config 5): patch extraction, class token +
// position embedding.  Each replaces a chain of ATen reshape / cat / add kernels by one pass of 16 B per lane.
#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

// x NHWC bf16 [B][G*P][G*P][3] -> patches [B*G*G][P*P*3], element order (py, px, c): exactly the physical order
// of a channels_last conv weight [D][P][P][3], so the patch embedding is a plain GEMM on both operands.
// One thread moves 8 elements (16 B); a patch row (P*3 elements) is contiguous in the image.
__global__ __launch_bounds__(256) void k_patchify(const bf16* __restrict__ x, bf16* __restrict__ out, int G, int P,
                                                  long chunks /* total 16-B chunks of out */) {
    const int row_elems = P * 3, cpr = row_elems / 8;  // chunks per patch row
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunks) return;
    const int c = (int)(i % cpr);
    long r = i / cpr;  // (b, gy, gx, py)
    const int py = (int)(r % P);
    r /= P;
    const int gx = (int)(r % G);
    r /= G;
    const int gy = (int)(r % G);
    const long b = r / G;
    const long W = (long)G * P;
    const bf16* src = x + ((b * W + (long)gy * P + py) * W + (long)gx * P) * 3 + c * 8;
    *reinterpret_cast<bf16x8*>(out + i * 8) = *reinterpret_cast<const bf16x8*>(src);
}
// gradient is not needed (the image has no gradient)

// out[b][0] = cls + pos[0]; out[b][1 + t] = patch[b][t] + pos[1 + t]     (D % 8 == 0)
__global__ __launch_bounds__(256) void k_vit_embed_fwd(const bf16* __restrict__ patch, const float* __restrict__ cls,
                                                       const float* __restrict__ pos, bf16* __restrict__ out, int T,
                                                       int D, long chunks) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= chunks) return;
    const int cpd = D / 8;
    const int c = (int)(i % cpd);
    const long r = i / cpd;           // (b, token)
    const int tok = (int)(r % (T + 1));
    const long b = r / (T + 1);
    const float* pp = pos + (long)tok * D + c * 8;
    bf16x8 o;
    if (tok == 0) {
        const float* cp = cls + c * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16)(cp[k] + pp[k]);
    } else {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(patch + ((b * T + tok - 1) * (long)D) + c * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (bf16)((float)v[k] + pp[k]);
    }
    *reinterpret_cast<bf16x8*>(out + i * 8) = o;
}

// dx [B][T+1][D] -> dpatch [B][T][D] (copy of the token rows) and slab[s][T+1][D] = sum over the batch slice s of dx
// (position-embedding gradient; row 0 is also the class-token gradient).  One thread = 8 channels of one token,
// walking the images of its slice.
__global__ __launch_bounds__(256) void k_vit_embed_bwd(const bf16* __restrict__ dx, bf16* __restrict__ dpatch,
                                                       float* __restrict__ slab, int B, int T, int D, int slices) {
    const int cpd = D / 8;
    const long per = (long)(T + 1) * cpd;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= per * slices) return;
    const int s = (int)(i / per);
    const long j = i - (long)s * per;
    const int c = (int)(j % cpd), tok = (int)(j / cpd);
    const int b0 = (int)((long)B * s / slices), b1 = (int)((long)B * (s + 1) / slices);
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int b = b0; b < b1; ++b) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(dx + ((long)b * (T + 1) + tok) * D + c * 8);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (float)v[k];
        if (tok > 0) *reinterpret_cast<bf16x8*>(dpatch + ((long)b * T + tok - 1) * D + c * 8) = v;
    }
    float* o = slab + ((long)s * (T + 1) + tok) * D + c * 8;
    *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
}

}  // namespace

void launch_patchify(const void* x, void* out, long B, int G, int P, hipStream_t s) {
    const long chunks = B * G * G * P * (P * 3 / 8);
    if (chunks == 0) return;
    hipLaunchKernelGGL(k_patchify, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16*)x, (bf16*)out, G,
                       P, chunks);
}

void launch_vit_embed_fwd(const void* patch, const float* cls, const float* pos, void* out, long B, int T, int D,
                          hipStream_t s) {
    const long chunks = B * (T + 1) * (D / 8);
    if (chunks == 0) return;
    hipLaunchKernelGGL(k_vit_embed_fwd, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, (const bf16*)patch, cls,
                       pos, (bf16*)out, T, D, chunks);
}

void launch_vit_embed_bwd(const void* dx, void* dpatch, float* slab, int B, int T, int D, int slices, hipStream_t s) {
    const long threads = (long)(T + 1) * (D / 8) * slices;
    hipLaunchKernelGGL(k_vit_embed_bwd, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, (const bf16*)dx,
                       (bf16*)dpatch, slab, B, T, D, slices);
}

}  // namespace pdt
