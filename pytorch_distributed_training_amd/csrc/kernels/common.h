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
// value of lane (l ^ 1): a DPP quad permute [1,0,3,2] -- one VALU move, no LDS permute, no address
__device__ __forceinline__ float dpp_xor1(float v) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
}

__device__ __forceinline__ bf16x8 buffer_load_bf16x8(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return __builtin_bit_cast(bf16x8, v);
}
// (an operand that is read exactly once: residual-gradient addends, BN inputs in the fused epilogues)
__device__ __forceinline__ bf16x8 buffer_load_bf16x8_stream(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off) {
#ifdef PDT_PLAIN_LDST
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
#else
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 2);
#endif
    return __builtin_bit_cast(bf16x8, v);
}
// ---- LDS-DMA (buffer_load_dwordx4 ... lds): 16 B per lane from a per-lane buffer offset straight into LDS at
// wave-uniform address + 16 * lane, no VGPR destination.  Issued through inline asm on purpose: for the builtin
// form the compiler cannot prove that a later ds_read does not alias the in-flight DMA and inserts its own
// s_waitcnt vmcnt before every LDS read, which collapses a multi-stage ring to one stage in flight.  Hand-issued,
// the DMA is invisible to the compiler's wait bookkeeping: the caller retires it with a counted
// `s_waitcnt vmcnt(N)` followed by a barrier before any lane reads the data (verified by bench/micro/glds_probe.hip:
// lane-linear placement, zeros for out-of-range lanes, destinations above 64 KiB).
typedef __attribute__((address_space(3))) char* lds_char_ptr;
__device__ __forceinline__ i32x4 make_rsrc_words(const void* base, uint32_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = (int)(uint32_t)a;
    r[1] = (int)((uint32_t)(a >> 32) & 0xffffu);
    r[2] = (int)bytes;
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return (uint32_t)(unsigned long long)(lds_char_ptr)(const_cast<void*>(p));
}
__device__ __forceinline__ void lds_dma16(i32x4 rsrc, uint32_t byte_off, uint32_t lds_addr_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(byte_off), "s"(rsrc), "s"(lds_addr_uniform)
        : "memory");
}

// the same for an operand that is streamed through once (non-temporal hint, see st_stream)
__device__ __forceinline__ void lds_dma16_stream(i32x4 rsrc, uint32_t byte_off, uint32_t lds_addr_uniform) {
#ifdef PDT_PLAIN_LDST
    lds_dma16(rsrc, byte_off, lds_addr_uniform);
#else
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen nt lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(byte_off), "s"(rsrc), "s"(lds_addr_uniform)
        : "memory");
#endif
}

// same with a scalar byte offset added to every lane's address (buffer SOFFSET): a K loop that only moves along the
// channel axis keeps its per-lane offsets and advances one SGPR
__device__ __forceinline__ void lds_dma16s(i32x4 rsrc, uint32_t byte_off, uint32_t soffset_uniform,
                                           uint32_t lds_addr_uniform) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(byte_off), "s"(rsrc), "s"(lds_addr_uniform), "s"(soffset_uniform)
        : "memory");
}

__device__ __forceinline__ void lds_dma16s_stream(i32x4 rsrc, uint32_t byte_off, uint32_t soffset_uniform,
                                                  uint32_t lds_addr_uniform) {
#ifdef PDT_PLAIN_LDST
    lds_dma16s(rsrc, byte_off, soffset_uniform, lds_addr_uniform);
#else
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen nt lds\n\ts_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(byte_off), "s"(rsrc), "s"(lds_addr_uniform), "s"(soffset_uniform)
        : "memory");
#endif
}

// Streams that are written once and read much later (activations, gigabytes per layer) go to memory with the
// non-temporal hint: measured +10 % on a streaming add (bench/micro/hbm_nt.hip).  -DPDT_PLAIN_LDST builds without.
#ifdef PDT_PLAIN_LDST
constexpr int kAuxStream = 0;
__device__ __forceinline__ void st_stream(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ u32x4 ld_stream(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
#else
constexpr int kAuxStream = 2;  // buffer instructions: the nt bit of the cache policy
__device__ __forceinline__ void st_stream(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }
__device__ __forceinline__ u32x4 ld_stream(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
#endif

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
