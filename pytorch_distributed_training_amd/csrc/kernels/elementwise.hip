// Memory-bound multi-tensor kernels: bucket flatten (+1/W scale, +cast), fused Adam/AdamW,
// fused SGD-momentum, input layout cast, weight transpose, slab reduction.
//
// Reference behaviour being replaced: c10d's per-tensor copy_/div_ into the bucket and ATen's
// ~10 _foreach_* launches per Adam step.  Here ONE launch walks a
// device-resident table (cached on the host while the pointers stay the same), 16 B per lane,
// fp32 math, and the optimizer also refreshes the bf16 working copy of the weights.
#include "common.h"
#include "kernels.h"

namespace pdt {

namespace {

struct Row {
    int64_t p0, p1, p2, p3, p4, numel, chunk_start, flags;
};

__device__ __forceinline__ void locate(const int64_t* table, int ntensors, int chunk, const Row*& row, long& base) {
    const int64_t e = table[(long)ntensors * kMTCols + chunk];
    const int t = (int)(e >> 32);
    const int local = (int)(e & 0xffffffff);
    row = reinterpret_cast<const Row*>(table) + t;
    base = (long)local * kMTChunk;
}

template <typename T>
__device__ __forceinline__ void load4(const T* p, long i, long n, float (&v)[4]);

template <>
__device__ __forceinline__ void load4<float>(const float* p, long i, long n, float (&v)[4]) {
    if (i + 4 <= n) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(p + i);
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i + k < n) ? p[i + k] : 0.f;
    }
}
template <>
__device__ __forceinline__ void load4<bf16>(const bf16* p, long i, long n, float (&v)[4]) {
    if (i + 4 <= n) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(p + i);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (float)t[k];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (i + k < n) ? (float)p[i + k] : 0.f;
    }
}
__device__ __forceinline__ void store4(float* p, long i, long n, const float (&v)[4]) {
    if (i + 4 <= n) {
        f32x4 t = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4*>(p + i) = t;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < n) p[i + k] = v[k];
    }
}
__device__ __forceinline__ void store4(bf16* p, long i, long n, const float (&v)[4]) {
    if (i + 4 <= n) {
        bf16x4 t = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
        *reinterpret_cast<bf16x4*>(p + i) = t;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < n) p[i + k] = (bf16)v[k];
    }
}

// row: p0 = src, p1 = dst
template <typename S, typename D>
__global__ __launch_bounds__(256) void k_multi_flatten(const int64_t* __restrict__ table, int ntensors, float scale) {
    const Row* r;
    long base;
    locate(table, ntensors, blockIdx.x, r, base);
    const S* src = reinterpret_cast<const S*>(r->p0);
    D* dst = reinterpret_cast<D*>(r->p1);
    const long n = r->numel;
#pragma unroll
    for (int it = 0; it < kMTChunk / (256 * 4); ++it) {
        const long i = base + ((long)it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        float v[4];
        load4<S>(src, i, n, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= scale;
        store4(dst, i, n, v);
    }
}

// Same copy, but the pointer table travels in the kernel arguments (<= kFlatMax tensors per launch):
// gradient tensors are re-allocated every step, so a cached device table would miss each time and
// cost a synchronous H2D copy on the autograd thread.
template <typename S, typename D>
__global__ __launch_bounds__(256) void k_flatten_args(const FlatArgs a, float scale) {
    // binary search: largest t with chunk_start[t] <= blockIdx.x
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.chunk_start[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const S* src = reinterpret_cast<const S*>(a.src[lo]);
    D* dst = reinterpret_cast<D*>(a.dst[lo]);
    const long n = a.numel[lo];
    const long base = (long)((int)blockIdx.x - a.chunk_start[lo]) * kMTChunk;
#pragma unroll
    for (int it = 0; it < kMTChunk / (256 * 4); ++it) {
        const long i = base + ((long)it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        float v[4];
        load4<S>(src, i, n, v);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] *= scale;
        store4(dst, i, n, v);
    }
}

// row: p0 = param (f32), p1 = grad, p2 = exp_avg, p3 = exp_avg_sq, p4 = bf16 copy or 0
template <typename G>
__global__ __launch_bounds__(256) void k_multi_adam(const int64_t* __restrict__ table, int ntensors, float lr,
                                                    float beta1, float beta2, float eps, float wd, float bc1,
                                                    float bc2_sqrt, const float* __restrict__ step_dev, int adamw,
                                                    float gscale) {
    if (step_dev != nullptr) {  // graph-capturable form: the step count lives in device memory
        const double st = (double)*step_dev;
        bc1 = (float)(1.0 - pow((double)beta1, st));
        bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, st));
    }
    const Row* r;
    long base;
    locate(table, ntensors, blockIdx.x, r, base);
    float* p = reinterpret_cast<float*>(r->p0);
    const G* g = reinterpret_cast<const G*>(r->p1);
    float* m = reinterpret_cast<float*>(r->p2);
    float* v = reinterpret_cast<float*>(r->p3);
    bf16* lp = reinterpret_cast<bf16*>(r->p4);
    const long n = r->numel;
    const float step_size = lr / bc1;
#pragma unroll
    for (int it = 0; it < kMTChunk / (256 * 4); ++it) {
        const long i = base + ((long)it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        float pv[4], gv[4], mv[4], vv[4];
        load4<float>(p, i, n, pv);
        load4<G>(g, i, n, gv);
        load4<float>(m, i, n, mv);
        load4<float>(v, i, n, vv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gg = gv[k] * gscale;
            float pp = pv[k];
            if (wd != 0.f) {
                if (adamw) pp *= (1.f - lr * wd);
                else gg += wd * pp;
            }
            const float mm = beta1 * mv[k] + (1.f - beta1) * gg;
            const float v2 = beta2 * vv[k] + (1.f - beta2) * gg * gg;
            const float denom = sqrtf(v2) / bc2_sqrt + eps;
            pp -= step_size * (mm / denom);
            pv[k] = pp; mv[k] = mm; vv[k] = v2;
        }
        store4(p, i, n, pv);
        store4(m, i, n, mv);
        store4(v, i, n, vv);
        if (lp) store4(lp, i, n, pv);
    }
}

// row: p0 = param (f32), p1 = grad, p2 = momentum buffer (or 0), p4 = bf16 copy or 0
template <typename G>
__global__ __launch_bounds__(256) void k_multi_sgd(const int64_t* __restrict__ table, int ntensors, float lr,
                                                   float momentum, float dampening, float wd, int nesterov,
                                                   int first, float gscale) {
    const Row* r;
    long base;
    locate(table, ntensors, blockIdx.x, r, base);
    float* p = reinterpret_cast<float*>(r->p0);
    const G* g = reinterpret_cast<const G*>(r->p1);
    float* buf = reinterpret_cast<float*>(r->p2);
    bf16* lp = reinterpret_cast<bf16*>(r->p4);
    const long n = r->numel;
#pragma unroll
    for (int it = 0; it < kMTChunk / (256 * 4); ++it) {
        const long i = base + ((long)it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        float pv[4], gv[4], bv[4] = {0.f, 0.f, 0.f, 0.f};
        load4<float>(p, i, n, pv);
        load4<G>(g, i, n, gv);
        if (momentum != 0.f && !first) load4<float>(buf, i, n, bv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gg = gv[k] * gscale + wd * pv[k];
            if (momentum != 0.f) {
                const float b = first ? gg : momentum * bv[k] + (1.f - dampening) * gg;
                bv[k] = b;
                gg = nesterov ? gg + momentum * b : b;
            }
            pv[k] -= lr * gg;
        }
        store4(p, i, n, pv);
        if (momentum != 0.f) store4(buf, i, n, bv);
        if (lp) store4(lp, i, n, pv);
    }
}

// f32 NCHW -> bf16 NHWC through an LDS tile so both sides move full lines.
// block = (b, h) row: tile [C][W] read coalesced along W, written coalesced along C.
__global__ __launch_bounds__(256) void k_nchw_f32_to_nhwc_bf16(const float* __restrict__ src, bf16* __restrict__ dst,
                                                               int C, int H, int W, float scale, FastDiv div_c) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tile = reinterpret_cast<float*>(smem);  // [C][W+1]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int pitch = W + 1;
    // (no per-element '/' and '%' by runtime values: they cost more than the 12 bytes an element moves)
    for (int c = 0; c < C; ++c) {
        const float* row = src + (((long)b * C + c) * H + h) * W;
        for (int w = threadIdx.x; w < W; w += blockDim.x) tile[c * pitch + w] = row[w] * scale;
    }
    __syncthreads();
    bf16* out = dst + ((long)b * H + h) * W * C;
    for (int i = threadIdx.x; i < C * W; i += blockDim.x) {
        const int w = (int)fdiv((uint32_t)i, div_c), c = i - w * C;
        out[i] = (bf16)tile[c * pitch + w];
    }
}

// The 3-channel case (every image model's input): a thread converts 8 consecutive pixels of a row, two float4 loads
// per colour plane and three 16-B stores (48 contiguous bytes); needs W % 8 == 0.  The LDS-tile kernel above moves
// 4-B / 2-B words and spends a workgroup on 2.7 KB: 460 us for 1280 images of 224x224 where this one takes ~220.
__global__ __launch_bounds__(256) void k_nchw3_f32_to_nhwc_bf16(const float* __restrict__ src, bf16* __restrict__ dst,
                                                                long groups, int HW, float scale, FastDiv div_hw8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // (b, pixel group of 8)
    if (i >= groups) return;
    const uint32_t b = fdiv((uint32_t)i, div_hw8);
    const uint32_t g = (uint32_t)i - b * (uint32_t)(HW >> 3);
    const float* p = src + (long)b * 3 * HW + (long)g * 8;
    float v[3][8];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const f32x4 lo = __builtin_bit_cast(f32x4, ld_stream(p + (long)c * HW));
        const f32x4 hi = __builtin_bit_cast(f32x4, ld_stream(p + (long)c * HW + 4));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[c][k] = lo[k] * scale;
            v[c][4 + k] = hi[k] * scale;
        }
    }
    bf16x8 o[3];
#pragma unroll
    for (int e = 0; e < 24; ++e) o[e >> 3][e & 7] = (bf16)v[e % 3][e / 3];
    bf16x8* out = reinterpret_cast<bf16x8*>(dst + ((long)b * HW + (long)g * 8) * 3);
    st_stream(out, __builtin_bit_cast(u32x4, o[0]));
    st_stream(out + 1, __builtin_bit_cast(u32x4, o[1]));
    st_stream(out + 2, __builtin_bit_cast(u32x4, o[2]));
}

// u8 NHWC -> bf16 NHWC (pure elementwise)
__global__ __launch_bounds__(256) void k_u8_to_bf16(const uint8_t* __restrict__ src, bf16* __restrict__ dst, long n,
                                                    float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = (bf16)((float)src[i] * scale);
}

// w [N][T][C] -> wt [C][T][N]; one block per (t, 32x32 tile of (n, c))
__global__ __launch_bounds__(256) void k_transpose_ntc(const bf16* __restrict__ w, bf16* __restrict__ wt, int N, int T,
                                                       int C) {
    __shared__ bf16 tile[32][34];
    const int t = blockIdx.z;
    const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        tile[r][tx] = (n < N && c < C) ? w[((long)n * T + t) * C + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        if (n < N && c < C) wt[((long)c * T + t) * N + n] = tile[tx][r];
    }
}

// Many weights at once: w_i [N][T][C] -> wt_i [C][T][N], table in the kernel arguments.  One launch
// per optimizer step refreshes every transposed conv weight the dgrad kernels read.
__global__ __launch_bounds__(256) void k_multi_transpose_ntc(const TransposeArgs a) {
    __shared__ bf16 tile[32][34];
    int lo = 0, hi = a.n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.block_start[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const bf16* w = static_cast<const bf16*>(a.src[lo]);
    bf16* wt = static_cast<bf16*>(a.dst[lo]);
    const int N = a.N[lo], T = a.T[lo], C = a.C[lo];
    int b = (int)blockIdx.x - a.block_start[lo];
    const int ct = (C + 31) / 32, nt = (N + 31) / 32;
    const int c0 = (b % ct) * 32;
    b /= ct;
    const int n0 = (b % nt) * 32;
    const int t = b / nt;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, c = c0 + tx;
        tile[r][tx] = (n < N && c < C) ? w[((long)n * T + t) * C + c] : (bf16)0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, n = n0 + tx;
        if (n < N && c < C) wt[((long)c * T + t) * N + n] = tile[tx][r];
    }
}

__global__ __launch_bounds__(64) void k_write_words(const TableWords t, int n, int64_t* __restrict__ dst) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i < n) dst[i] = t.w[i];
}

// out[y][j] = sum over the parts of chunk y of slab[p*stride + j].  Block = 64 column quads x 4
// part lanes; grid = (column blocks of 256, part chunks).  Deterministic (no atomics).
__global__ __launch_bounds__(256) void k_reduce_slab(const float* __restrict__ slab, int nparts, long stride, int n,
                                                     int parts_per_chunk, float* __restrict__ out) {
    __shared__ float red[4][64][4];
    const int cq = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const long j = ((long)blockIdx.x * 64 + cq) * 4;
    const int p0 = blockIdx.y * parts_per_chunk;
    const int p1 = min(nparts, p0 + parts_per_chunk);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (j < n) {
        // eight rows per trip, all loads issued before the first add (same summation order as one row at a time:
        // a runtime-count loop of load -> add pays one memory round trip per row)
        int p = p0 + rl;
        for (; p + 28 < p1; p += 32) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) load4<float>(slab + (long)(p + 4 * u) * stride, j, n, v[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) a[k] += v[u][k];
        }
        for (; p < p1; p += 4) {
            float v[4];
            load4<float>(slab + (long)p * stride, j, n, v);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) red[rl][cq][k] = a[k];
    __syncthreads();
    if (rl == 0 && j < n) {
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = red[0][cq][k] + red[1][cq][k] + red[2][cq][k] + red[3][cq][k];
        store4(out + (long)blockIdx.y * n, j, n, a);
    }
}


// Column sums of a row-major bf16 matrix [M][N] (bias gradients of the Linear layers: dy summed over tokens) into
// fp32 partial rows slab[chunk][N].  A block owns 128 columns x one row chunk: 16 lanes x 16 B cover the 256
// contiguous bytes of a row, 16 row lanes walk the chunk with four independent loads in flight each.
__global__ __launch_bounds__(256) void k_colsum_bf16(const bf16* __restrict__ x, long M, int N, int rows_per_block,
                                                     float* __restrict__ slab) {
    __shared__ float red[16][128 + 4];
    const int tid = threadIdx.x, cx = tid & 15, ry = tid >> 4;
    const int col = blockIdx.x * 128 + cx * 8;
    const long r0 = (long)blockIdx.y * rows_per_block;
    const long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = 0.f;
    if (col < N) {
        const bf16* base = x + col;
        long r = r0 + ry;
        for (; r + 48 < r1; r += 64) {
            bf16x8 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(base + (r + 16 * u) * N);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < 8; ++k) a[k] += (float)v[u][k];
        }
        for (; r < r1; r += 16) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(base + r * N);
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] += (float)v[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) red[ry][cx * 8 + k] = a[k];
    __syncthreads();
    if (tid < 128) {
        const int c = blockIdx.x * 128 + tid;
        if (c < N) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s += red[q][tid];
            slab[(long)blockIdx.y * N + c] = s;
        }
    }
}
}  // namespace

void launch_multi_flatten(const int64_t* table, int ntensors, int nchunks, float scale, int src_dtype,
                          int dst_dtype, hipStream_t s) {
    if (nchunks <= 0) return;
    dim3 grid(nchunks), block(256);
    if (src_dtype == kF32 && dst_dtype == kF32)
        hipLaunchKernelGGL((k_multi_flatten<float, float>), grid, block, 0, s, table, ntensors, scale);
    else if (src_dtype == kF32 && dst_dtype == kBF16)
        hipLaunchKernelGGL((k_multi_flatten<float, bf16>), grid, block, 0, s, table, ntensors, scale);
    else if (src_dtype == kBF16 && dst_dtype == kF32)
        hipLaunchKernelGGL((k_multi_flatten<bf16, float>), grid, block, 0, s, table, ntensors, scale);
    else
        hipLaunchKernelGGL((k_multi_flatten<bf16, bf16>), grid, block, 0, s, table, ntensors, scale);
}

void launch_write_words(const TableWords& t, int n, int64_t* dst, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_write_words, dim3((n + 63) / 64), dim3(64), 0, s, t, n, dst);
}

void launch_flatten_args(const FlatArgs& a, float scale, int src_dtype, int dst_dtype, hipStream_t s) {
    const int nchunks = a.chunk_start[a.n];
    if (a.n <= 0 || nchunks <= 0) return;
    dim3 grid(nchunks), block(256);
    if (src_dtype == kF32 && dst_dtype == kF32)
        hipLaunchKernelGGL((k_flatten_args<float, float>), grid, block, 0, s, a, scale);
    else if (src_dtype == kF32 && dst_dtype == kBF16)
        hipLaunchKernelGGL((k_flatten_args<float, bf16>), grid, block, 0, s, a, scale);
    else if (src_dtype == kBF16 && dst_dtype == kF32)
        hipLaunchKernelGGL((k_flatten_args<bf16, float>), grid, block, 0, s, a, scale);
    else
        hipLaunchKernelGGL((k_flatten_args<bf16, bf16>), grid, block, 0, s, a, scale);
}

void launch_multi_adam(const int64_t* table, int ntensors, int nchunks, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float bias_corr1, float bias_corr2_sqrt,
                       const float* step_dev, int adamw, float grad_scale, int grad_dtype, hipStream_t s) {
    if (nchunks <= 0) return;
    dim3 grid(nchunks), block(256);
    if (grad_dtype == kF32)
        hipLaunchKernelGGL((k_multi_adam<float>), grid, block, 0, s, table, ntensors, lr, beta1, beta2, eps,
                           weight_decay, bias_corr1, bias_corr2_sqrt, step_dev, adamw, grad_scale);
    else
        hipLaunchKernelGGL((k_multi_adam<bf16>), grid, block, 0, s, table, ntensors, lr, beta1, beta2, eps,
                           weight_decay, bias_corr1, bias_corr2_sqrt, step_dev, adamw, grad_scale);
}

void launch_multi_sgd(const int64_t* table, int ntensors, int nchunks, float lr, float momentum, float dampening,
                      float weight_decay, int nesterov, int first, float grad_scale, int grad_dtype,
                      hipStream_t s) {
    if (nchunks <= 0) return;
    dim3 grid(nchunks), block(256);
    if (grad_dtype == kF32)
        hipLaunchKernelGGL((k_multi_sgd<float>), grid, block, 0, s, table, ntensors, lr, momentum, dampening,
                           weight_decay, nesterov, first, grad_scale);
    else
        hipLaunchKernelGGL((k_multi_sgd<bf16>), grid, block, 0, s, table, ntensors, lr, momentum, dampening,
                           weight_decay, nesterov, first, grad_scale);
}

void launch_to_nhwc_bf16(const void* src, int src_dtype, void* dst, int B, int C, int H, int W, float scale,
                         hipStream_t s) {
    if (src_dtype == kF32 && C == 3 && (H * W) % 8 == 0 && (long)B * H * W / 8 < (1L << 31)) {
        const long groups = (long)B * H * W / 8;
        hipLaunchKernelGGL(k_nchw3_f32_to_nhwc_bf16, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s,
                           (const float*)src, (bf16*)dst, groups, H * W, scale, make_fastdiv((uint32_t)(H * W / 8)));
    } else if (src_dtype == kF32) {
        const size_t lds = (size_t)C * (W + 1) * sizeof(float);
        hipLaunchKernelGGL(k_nchw_f32_to_nhwc_bf16, dim3(B * H), dim3(256), lds, s, (const float*)src, (bf16*)dst, C,
                           H, W, scale, make_fastdiv((uint32_t)C));
    } else {
        const long n = (long)B * C * H * W;
        const int grid = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(k_u8_to_bf16, dim3(grid), dim3(256), 0, s, (const uint8_t*)src, (bf16*)dst, n, scale);
    }
}

void launch_transpose_ntc(const void* w, void* wt, int N, int T, int C, hipStream_t s) {
    dim3 grid((C + 31) / 32, (N + 31) / 32, T);
    hipLaunchKernelGGL(k_transpose_ntc, grid, dim3(256), 0, s, (const bf16*)w, (bf16*)wt, N, T, C);
}

void launch_multi_transpose(const TransposeArgs& a, hipStream_t s) {
    if (a.n <= 0) return;
    hipLaunchKernelGGL(k_multi_transpose_ntc, dim3(a.block_start[a.n]), dim3(256), 0, s, a);
}

void launch_reduce_slab(const float* slab, int nparts, long stride, int n, float* out, float* tmp,
                        hipStream_t s) {
    const int colblocks = (n + 255) / 256;
    if (nparts > 64 && tmp != nullptr) {
        // two deterministic levels: [nparts] -> [kReduceChunks] -> 1
        const int ppc = (nparts + kReduceChunks - 1) / kReduceChunks;
        const int chunks = (nparts + ppc - 1) / ppc;
        hipLaunchKernelGGL(k_reduce_slab, dim3(colblocks, chunks), dim3(256), 0, s, slab, nparts, stride, n, ppc, tmp);
        hipLaunchKernelGGL(k_reduce_slab, dim3(colblocks, 1), dim3(256), 0, s, (const float*)tmp, chunks, (long)n, n,
                           chunks, out);
    } else {
        hipLaunchKernelGGL(k_reduce_slab, dim3(colblocks, 1), dim3(256), 0, s, slab, nparts, stride, n, nparts, out);
    }
}

int launch_reduce_rows_l1(const float* slab, int nparts, long stride, int n, float* out, int max_chunks,
                          hipStream_t s) {
    const int colblocks = (n + 255) / 256;
    const int ppc = (nparts + max_chunks - 1) / max_chunks;
    const int chunks = (nparts + ppc - 1) / ppc;
    hipLaunchKernelGGL(k_reduce_slab, dim3(colblocks, chunks), dim3(256), 0, s, slab, nparts, stride, n, ppc, out);
    return chunks;
}

int launch_colsum_bf16(const void* x, long M, int N, float* slab, int max_chunks, hipStream_t s) {
    const int colblocks = (N + 127) / 128;
    long chunks = (M + 63) / 64;  // at least 64 rows per block
    const long want = (2048 + colblocks - 1) / colblocks;
    if (chunks > want) chunks = want;
    if (chunks > max_chunks) chunks = max_chunks;
    if (chunks < 1) chunks = 1;
    const int rpb = (int)((M + chunks - 1) / chunks);
    chunks = (M + rpb - 1) / rpb;
    hipLaunchKernelGGL(k_colsum_bf16, dim3(colblocks, (unsigned)chunks), dim3(256), 0, s, (const bf16*)x, M, N, rpb, slab);
    return (int)chunks;
}

}  // namespace pdt
