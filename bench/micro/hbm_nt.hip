// Streaming bandwidth of c = a + b (bf16, 16 B per lane, one 16 KiB chunk per block in address order -- the shape of
// the BN kernels) with plain / non-temporal loads and stores.  hipcc --offload-arch=gfx950 -O3 hbm_nt.hip -o hbm_nt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int U>
__global__ __launch_bounds__(256) void k_add(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ c,
                                             long n) {
    const long base = (long)blockIdx.x * 256 * U + threadIdx.x;
    u32x4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i < n) {
            if (MODE & 1) {
                va[u] = __builtin_nontemporal_load(a + i);
                vb[u] = __builtin_nontemporal_load(b + i);
            } else {
                va[u] = a[i];
                vb[u] = b[i];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long i = base + u * 256;
        if (i < n) {
            const bf16x8 x = __builtin_bit_cast(bf16x8, va[u]), y = __builtin_bit_cast(bf16x8, vb[u]);
            bf16x8 z;
#pragma unroll
            for (int k = 0; k < 8; ++k) z[k] = (__bf16)((float)x[k] + (float)y[k]);
            const u32x4 o = __builtin_bit_cast(u32x4, z);
            if (MODE & 2) __builtin_nontemporal_store(o, c + i);
            else c[i] = o;
        }
    }
}

template <int MODE, int U>
float run(const u32x4* a, const u32x4* b, u32x4* c, long n) {
    const long blocks = (n + 256 * U - 1) / (256 * U);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_add<MODE, U>), dim3((unsigned)blocks), dim3(256), 0, 0, a, b, c, n);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_add<MODE, U>), dim3((unsigned)blocks), dim3(256), 0, 0, a, b, c, n);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / 10;
}

int main(int argc, char** argv) {
    const long bytes = (argc > 1 ? atol(argv[1]) : 2048L) << 20;  // per tensor
    const long n = bytes / 16;
    u32x4 *a, *b, *c;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&c, bytes) != hipSuccess) return 1;
    hipMemset(a, 0, bytes);
    hipMemset(b, 0, bytes);
    const char* names[4] = {"plain", "nt loads", "nt stores", "nt loads + stores"};
    for (int rep = 0; rep < 2; ++rep) {
        float t[4] = {run<0, 4>(a, b, c, n), run<1, 4>(a, b, c, n), run<2, 4>(a, b, c, n), run<3, 4>(a, b, c, n)};
        for (int m = 0; m < 4; ++m) printf("U=4 %-18s %.3f ms  %.2f TB/s\n", names[m], t[m], 3.0 * bytes / t[m] / 1e9);
        float t2[2] = {run<0, 2>(a, b, c, n), run<3, 2>(a, b, c, n)};
        printf("U=2 plain              %.3f ms  %.2f TB/s\nU=2 nt loads + stores  %.3f ms  %.2f TB/s\n", t2[0], 3.0 * bytes / t2[0] / 1e9, t2[1],
               3.0 * bytes / t2[1] / 1e9);
    }
    printf("U=8 nt loads + stores  %.3f ms  %.2f TB/s\n", run<3, 8>(a, b, c, n), 3.0 * bytes / run<3, 8>(a, b, c, n) / 1e9);
    printf("U=1 nt loads + stores  %.3f ms  %.2f TB/s\n", run<3, 1>(a, b, c, n), 3.0 * bytes / run<3, 1>(a, b, c, n) / 1e9);
    // relative placement of the three streams: b and c shifted by off and 2 * off bytes (all PyTorch blocks are 2 MiB
    // aligned, i.e. every stream of a kernel starts on the same memory channel)
    u32x4* big;
    const long pad = 64L << 20;
    if (hipMalloc(&big, 3 * bytes + 3 * pad) != hipSuccess) return 0;
    const long offs[] = {0, 256, 4096, 65536, (1L << 20) + 4096, (8L << 20) + 65536 + 256};
    for (long off : offs) {
        const u32x4* a2 = big;
        const u32x4* b2 = (const u32x4*)((const char*)big + bytes + pad + off);
        u32x4* c2 = (u32x4*)((char*)big + 2 * (bytes + pad) + 2 * off);
        const float t = run<3, 4>(a2, b2, c2, n);
        printf("streams shifted by %8ld B: %.3f ms  %.2f TB/s (nt loads + stores)\n", off, t, 3.0 * bytes / t / 1e9);
    }
    return 0;
}
