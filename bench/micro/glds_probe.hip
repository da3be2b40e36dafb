// Probe of the LDS-DMA buffer load (buffer_load_dwordx4 ... lds) on gfx950, builtin and inline-asm form:
//  * where do the 64 lanes' 16-B pieces land (expected: wave-uniform LDS base + 16 * lane),
//  * what lands for lanes whose offset is outside the buffer descriptor's range (expected: zeros),
//  * does the destination base reach beyond 64 KiB (a 160 KiB LDS needs more than M0[15:0]).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_char_ptr;

__device__ __forceinline__ void lds_dma16(i32x4 rsrc, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_addr)
                 : "memory");
}

constexpr int kHigh = 100 * 1024;  // byte offset of the second destination

__global__ void probe(const unsigned* src, unsigned nbytes, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* lds = reinterpret_cast<unsigned*>(smem);
    const int lane = threadIdx.x;
    for (int i = lane; i < 512; i += 64) {
        lds[i] = 0xdeadbeefu;
        lds[kHigh / 4 + i] = 0xdeadbeefu;
    }
    __syncthreads();
    const unsigned long long a = (unsigned long long)src;
    i32x4 rsrc;
    rsrc[0] = (int)(unsigned)a;
    rsrc[1] = (int)((unsigned)(a >> 32) & 0xffffu);
    rsrc[2] = (int)nbytes;
    rsrc[3] = 0x00020000;
    const unsigned base = (unsigned)(unsigned long long)(lds_char_ptr)smem;
    unsigned piece = (lane * 5 + 3) % 64;
    unsigned off = piece * 16;
    if (lane == 7 || lane == 40) off = 0x80000000u;
    lds_dma16(rsrc, off, base + 64);            // low destination, permuted + out-of-range lanes
    lds_dma16(rsrc, lane * 16, base + kHigh);   // destination beyond 64 KiB
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 512; i += 64) {
        out[i] = lds[i];
        out[512 + i] = lds[kHigh / 4 + i];
    }
}

int main() {
    const int n = 64 * 4;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = 1000 + i;
    unsigned *d, *o;
    hipMalloc(&d, n * 4);
    hipMalloc(&o, 1024 * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 128 * 1024, 0, d, (unsigned)(n * 4), o);
    std::vector<unsigned> r(1024);
    if (hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { puts("copy failed"); return 2; }
    int bad = 0;
    for (int i = 0; i < 16; ++i) bad += r[i] != 0xdeadbeefu;
    for (int l = 0; l < 64; ++l) {
        const unsigned piece = (l * 5 + 3) % 64;
        for (int k = 0; k < 4; ++k) {
            const unsigned want = (l == 7 || l == 40) ? 0u : 1000 + piece * 4 + k;
            const unsigned got = r[16 + l * 4 + k];
            if (got != want) { if (bad < 8) printf("low lane %d dword %d: got %u (0x%x) want %u\n", l, k, got, got, want); ++bad; }
            const unsigned got2 = r[512 + l * 4 + k];
            if (got2 != 1000u + l * 4 + k) { if (bad < 8) printf("high lane %d dword %d: got %u (0x%x)\n", l, k, got2, got2); ++bad; }
        }
    }
    for (int i = 16 + 256; i < 512; ++i) bad += r[i] != 0xdeadbeefu;
    printf("glds probe: %s (%d mismatches)\n", bad ? "FAIL" : "OK: lane-linear, out-of-range lanes write zeros, base above 64 KiB works", bad);
    return bad ? 1 : 0;
}
