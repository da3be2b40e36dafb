// Probe: is a 4-wave GEMM tile schedule (each wave 128 x 128 of a 256 x 256 tile, accumulators in the upper half of the
// unified register file, fragments double-buffered in VGPRs) faster than the 8-wave ping-pong of csrc/kernels/gemm.hip?
//
//   C[M,N] (bf16) = A[M,K] . B[N,K]^T      M, N multiples of 256, K a multiple of 32
//
// The production core reads 192 KiB of LDS fragments per 256 x 256 x 64 K-tile (A panel by 4 waves, B panel by 2) and
// sits at ~38 % MFMA-busy on ViT's shapes (profiles/r3_pmc_vit_b256_summary.md).  Here 4 waves read 128 KiB for the same
// FLOPs, and with 256 accumulator registers out of the way a wave can fetch the next k-step's fragments while the
// matrix pipe works on this one.  BK = 32, ring of 4 stages of 32 KiB (LDS-DMA, swizzle on the source side), one
// barrier per stage, counted vmcnt.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/gemm_v2_probe bench/micro/gemm_v2_probe.hip && /tmp/gemm_v2_probe 25088 768 768
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_char_ptr;

#define CHECK(e)                                                                      \
    do {                                                                              \
        hipError_t _e = (e);                                                          \
        if (_e != hipSuccess) {                                                       \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e)); \
            exit(1);                                                                  \
        }                                                                             \
    } while (0)

constexpr int BK = 32, STAGES = 4;
constexpr int OPER_BYTES = 256 * BK * 2;      // 16 KiB: 256 rows of 64 B
constexpr int STAGE_BYTES = 2 * OPER_BYTES;   // A, B
constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
constexpr uint32_t kOOB = 0x80000000u;

__device__ __forceinline__ void lds_dma16s(i32x4 rsrc, uint32_t byte_off, uint32_t soff, uint32_t lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(byte_off), "s"(rsrc), "s"(lds_addr), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ i32x4 rsrc_words(const void* base, uint32_t bytes) {
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((uint32_t)(a >> 32) & 0xffffu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
// 64-B rows (4 chunks of 16 B): chunk ^= (row >> 1) & 3 makes any 8 consecutive rows of one chunk column cover all 32 banks
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 3)) << 4); }

struct Args {
    const bf16* A;
    const bf16* B;
    bf16* C;
    int M, N, K;
    int m_tiles, n_tiles;
    int nostore;
};

__global__ __launch_bounds__(256) void k_gemm_v2(const Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const uint32_t lds0 = (uint32_t)(unsigned long long)(lds_char_ptr)smem;
    const int total = a.m_tiles * a.n_tiles;
    const int nk = a.K / BK;

    // DMA: a stage's operand is 16 pieces of 1 KiB (16 rows of 64 B); wave w issues pieces w, w+4, w+8, w+12 of A and of B.
    // lane -> row piece*16 + lane/4, physical chunk lane%4 holding logical chunk (lane%4) ^ ((row>>1)&3)
    uint32_t vA[4], vB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (wave + 4 * j) * 16 + (lane >> 2);
        const int lch = (lane & 3) ^ ((row >> 1) & 3);
        vA[j] = (uint32_t)row * (uint32_t)(a.K * 2) + (uint32_t)(lch * 16);
        vB[j] = vA[j];
    }
    // fragment read offsets inside an operand image: rows (wr|wc)*128 + i*32 + l31, chunk 2*ks + half
    int aoff[2], boff[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        aoff[ks] = swz(wr * 128 + l31, 2 * ks + half);
        boff[ks] = OPER_BYTES + swz(wc * 128 + l31, 2 * ks + half);
    }

    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        const int tm = tile / a.n_tiles, tn = tile - tm * a.n_tiles;  // n fastest: neighbours share the A panel
        const i32x4 ra = rsrc_words(a.A + (long)tm * 256 * a.K, (uint32_t)(256L * a.K * 2));
        const i32x4 rb = rsrc_words(a.B + (long)tn * 256 * a.K, (uint32_t)(256L * a.K * 2));
        auto issue = [&](int k) {  // K-step k of this tile -> stage k % STAGES (past the end: out of range, zeros nobody reads)
            const uint32_t dst = lds0 + (uint32_t)((k % STAGES) * STAGE_BYTES) + (uint32_t)wave * 1024u;
            const uint32_t soff = (uint32_t)__builtin_amdgcn_readfirstlane(k * BK * 2);
            const bool in = k < nk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                lds_dma16s(ra, in ? vA[j] : kOOB, soff, dst + (uint32_t)(j * 4096));
                lds_dma16s(rb, in ? vB[j] : kOOB, soff, dst + (uint32_t)(OPER_BYTES + j * 4096));
            }
        };
        f32x16 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

        __builtin_amdgcn_s_barrier();  // the previous tile's last stage reads are done
        issue(0);
        issue(1);
        issue(2);
        for (int k = 0; k < nk; ++k) {
            // stage k has landed once at most the two younger steps (16 DMA each... 8 per wave and step) are outstanding
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // ... for every wave; and everyone is done with stage (k - 1) % STAGES
            issue(k + 3);
            const char* st = smem + (k % STAGES) * STAGE_BYTES;
            bf16x8 af[2][4], bfr[2][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[0][i] = *reinterpret_cast<const bf16x8*>(st + aoff[0] + i * 32 * 64);
                bfr[0][i] = *reinterpret_cast<const bf16x8*>(st + boff[0] + i * 32 * 64);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        af[1][i] = *reinterpret_cast<const bf16x8*>(st + aoff[1] + i * 32 * 64);
                        bfr[1][i] = *reinterpret_cast<const bf16x8*>(st + boff[1] + i * 32 * 64);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)  // D^T = B . A^T: lane = output row, registers = 4-column groups
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[ks][j], af[ks][i], acc[i][j], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.nostore) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s += acc[i][j][0];
            if (s == 12345.678f) a.C[0] = (bf16)s;
            continue;
        }
        // direct stores: lane l31 = row, registers 4g..4g+3 = columns 8g + 4*half + 0..3 of the 32-column block
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long row = (long)tm * 256 + wr * 128 + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col0 = tn * 256 + wc * 128 + j * 32 + 4 * half;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = (bf16)acc[i][j][4 * g + e];
                    *reinterpret_cast<bf16x4*>(a.C + row * a.N + col0 + 8 * g) = o;
                }
            }
        }
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 8192, N = argc > 2 ? atoi(argv[2]) : 8192, K = argc > 3 ? atoi(argv[3]) : 8192;
    const int nostore = argc > 4 ? atoi(argv[4]) : 0;
    if (M % 256 || N % 256 || K % BK) {
        fprintf(stderr, "M, N multiples of 256, K of %d\n", BK);
        return 1;
    }
    std::vector<uint16_t> hA((size_t)M * K), hB((size_t)N * K);
    auto tobf = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    };
    auto tof = [](uint16_t h) {
        uint32_t u = (uint32_t)h << 16;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    uint32_t seed = 12345u;
    auto rnd = [&] {
        seed = seed * 1664525u + 1013904223u;
        return ((seed >> 8) & 0xffff) / 65536.f - 0.5f;
    };
    for (auto& v : hA) v = tobf(rnd());
    for (auto& v : hB) v = tobf(rnd());
    bf16 *dA, *dB, *dC;
    CHECK(hipMalloc(&dA, hA.size() * 2));
    CHECK(hipMalloc(&dB, hB.size() * 2));
    CHECK(hipMalloc(&dC, (size_t)M * N * 2));
    CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemset(dC, 0, (size_t)M * N * 2));
    Args a{dA, dB, dC, M, N, K, M / 256, N / 256, nostore};
    int cus = 256;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int total = a.m_tiles * a.n_tiles, grid = total < cus ? total : cus;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_v2), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(k_gemm_v2, dim3(grid), dim3(256), LDS_BYTES, 0, a);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(k_gemm_v2, dim3(grid), dim3(256), LDS_BYTES, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters;
    printf("M=%d N=%d K=%d nostore=%d: %.1f us  %.0f TFLOP/s  (%d tiles on %d CUs)\n", M, N, K, nostore, us, 2.0 * M * N * K / us / 1e6,
           total, grid);
    if (!nostore) {
        std::vector<uint16_t> hC((size_t)M * N);
        CHECK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 400; ++t) {
            seed = seed * 1664525u + 1013904223u;
            const int m = (seed >> 4) % M;
            seed = seed * 1664525u + 1013904223u;
            const int n = (seed >> 4) % N;
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)tof(hA[(size_t)m * K + k]) * tof(hB[(size_t)n * K + k]);
            const double got = tof(hC[(size_t)m * N + n]);
            const double err = fabs(got - ref) / (fabs(ref) + 0.05 * sqrt((double)K) * 0.083);
            if (err > worst) worst = err;
        }
        printf("  max relative error over 400 samples: %.3e %s\n", worst, worst < 2e-2 ? "ok" : "FAIL");
    }
    return 0;
}
