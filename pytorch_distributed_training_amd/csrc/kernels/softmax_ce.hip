// Fused softmax cross-entropy, forward + backward in one pass.
//
// Reference path: ATen log_softmax + nll_loss = 2 forward + 2 backward kernels over the
// [B, 1000] logits (main.py:79,101,103).  Here one wave (64 lanes) owns a row: row max,
// sum-exp, the row's loss and dlogits = (softmax - onehot) / B are produced from a single
// read of the row kept in registers; a second tiny launch averages the per-row losses
// (deterministic, no float atomics).  A label outside [0, N) makes the row's loss, and with it the mean, NaN
// (ATen raises a device-side assertion there): never an out-of-bounds read, never a silently wrong number.
#include "common.h"
#include "kernels.h"

namespace pdt {
namespace {

constexpr int kMaxPerLane = 32;  // rows up to 64*32 = 2048 classes stay in registers

template <typename T>
__global__ __launch_bounds__(256) void k_softmax_ce(const T* __restrict__ logits, const int64_t* __restrict__ labels,
                                                    int B, int N, float* __restrict__ row_loss,
                                                    T* __restrict__ dlogits) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;  // whole wave exits together
    const T* x = logits + (long)row * N;
    T* dx = dlogits + (long)row * N;
    const int label = (int)labels[row];
    const float invB = 1.f / (float)B;

    if (N <= 64 * kMaxPerLane) {
        float v[kMaxPerLane];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < kMaxPerLane; ++k) {
            const int j = lane + k * 64;
            v[k] = (j < N) ? (float)x[j] : -INFINITY;
            mx = fmaxf(mx, v[k]);
        }
        mx = wave_max(mx);
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < kMaxPerLane; ++k) {
            const int j = lane + k * 64;
            v[k] = (j < N) ? __expf(v[k] - mx) : 0.f;
            se += v[k];
        }
        se = wave_sum(se);
        const float inv = 1.f / se;
#pragma unroll
        for (int k = 0; k < kMaxPerLane; ++k) {
            const int j = lane + k * 64;
            if (j < N) dx[j] = (T)((v[k] * inv - (j == label ? 1.f : 0.f)) * invB);
        }
        if (lane == 0) row_loss[row] = __logf(se) + mx - ((label >= 0 && label < N) ? (float)x[label] : __builtin_nanf(""));
    } else {
        float mx = -INFINITY;
        for (int j = lane; j < N; j += 64) mx = fmaxf(mx, (float)x[j]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int j = lane; j < N; j += 64) se += __expf((float)x[j] - mx);
        se = wave_sum(se);
        const float inv = 1.f / se;
        for (int j = lane; j < N; j += 64)
            dx[j] = (T)((__expf((float)x[j] - mx) * inv - (j == label ? 1.f : 0.f)) * invB);
        if (lane == 0) row_loss[row] = __logf(se) + mx - ((label >= 0 && label < N) ? (float)x[label] : __builtin_nanf(""));
    }
}

__global__ __launch_bounds__(256) void k_mean(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1] + red[2] + red[3]) / (float)n;
}

}  // namespace

void launch_softmax_ce(const void* logits, int dtype, const int64_t* labels, int B, int N, float* row_loss,
                       void* dlogits, float* loss_out, hipStream_t s) {
    dim3 grid((B + 3) / 4), block(256);
    if (dtype == kBF16)
        hipLaunchKernelGGL((k_softmax_ce<bf16>), grid, block, 0, s, (const bf16*)logits, labels, B, N, row_loss,
                           (bf16*)dlogits);
    else
        hipLaunchKernelGGL((k_softmax_ce<float>), grid, block, 0, s, (const float*)logits, labels, B, N, row_loss,
                           (float*)dlogits);
    hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, s, (const float*)row_loss, B, loss_out);
}

}  // namespace pdt
