// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
// wave = 64 lanes, 16 B per lane is the widest (and preferred) global / LDS access.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastdiv.h"

namespace pdt {

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 128-bit buffer resource over [base, base+bytes): out-of-range offsets load zeros and drop
// stores, which is how the implicit-GEMM kernels get their zero padding without a branch.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), /*stride*/ 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ bf16x8 buffer_load_bf16x8(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return __builtin_bit_cast(bf16x8, v);
}
constexpr uint32_t kOOB = 0x80000000u;  // > any legal byte offset (tensors are < 2 GiB, checked on the host)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return (n + __umulhi(n, f.mlo)) >> f.shift;
}

#define PDT_HIP_CHECK(expr)                                                                       \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); \
            abort();                                                                              \
        }                                                                                         \
    } while (0)

}  // namespace pdt
