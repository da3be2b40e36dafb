// Does a consumer that walks a just-written tensor BACK TO FRONT find its tail in the Infinity Cache / L2?
// writer: fills S bytes front to back (plain or non-temporal stores); reader: sums it front to back or back to front
// (non-temporal or plain loads).  hipcc --offload-arch=gfx950 -O3 mall_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ c, long n, unsigned v) {
    const long base = (long)blockIdx.x * 1024 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256;
        if (i < n) {
            const u32x4 o = {v, v + 1, v + 2, (unsigned)i};
            if (NT) __builtin_nontemporal_store(o, c + i);
            else c[i] = o;
        }
    }
}
template <bool NT, bool REV>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ a, long n, unsigned* __restrict__ out) {
    const long blk = REV ? (long)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const long base = blk * 1024 + threadIdx.x;
    u32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256;
        v[u] = u32x4{0, 0, 0, 0};
        if (i < n) v[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
    }
    unsigned s = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) s += v[u][0] ^ v[u][1] ^ v[u][2] ^ v[u][3];
    if (s == 0x12345678u) out[0] = s;  // (never: keeps the loads alive)
}

template <bool WNT, bool RNT, bool REV>
float probe(u32x4* buf, long n, unsigned* out, u32x4* flush, long nflush) {
    const unsigned blocks = (unsigned)((n + 1023) / 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float tot = 0;
    for (int it = 0; it < 6; ++it) {
        hipLaunchKernelGGL((k_write<true>), dim3((unsigned)((nflush + 1023) / 1024)), dim3(256), 0, 0, flush, nflush, 7u);  // evict
        hipLaunchKernelGGL((k_write<WNT>), dim3(blocks), dim3(256), 0, 0, buf, n, (unsigned)it);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_read<RNT, REV>), dim3(blocks), dim3(256), 0, 0, buf, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) tot += ms;
    }
    return tot / 4;
}

int main() {
    const long maxb = 2048L << 20;
    u32x4 *buf, *flush;
    unsigned* out;
    if (hipMalloc(&buf, maxb) != hipSuccess || hipMalloc(&flush, 1024L << 20) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
    for (long mb : {64L, 128L, 256L, 512L, 1024L, 2048L}) {
        const long bytes = mb << 20, n = bytes / 16, nf = (1024L << 20) / 16;
        const float a = probe<false, false, false>(buf, n, out, flush, nf), b = probe<false, false, true>(buf, n, out, flush, nf);
        const float c = probe<true, true, false>(buf, n, out, flush, nf), d = probe<true, true, true>(buf, n, out, flush, nf);
        const float e = probe<false, true, true>(buf, n, out, flush, nf);
        printf("%5ld MiB  read after write, TB/s:  plain fwd %.2f  plain rev %.2f | nt fwd %.2f  nt rev %.2f | plain write + nt read rev %.2f\n", mb,
               bytes / a / 1e9, bytes / b / 1e9, bytes / c / 1e9, bytes / d / 1e9, bytes / e / 1e9);
    }
    return 0;
}
