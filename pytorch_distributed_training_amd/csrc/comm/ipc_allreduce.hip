// One-shot all-reduce kernel over IPC-mapped mailboxes: see ipc_allreduce.h for the protocol.
#include <hip/hip_runtime.h>

#include "ipc_allreduce.h"

namespace pdt {

namespace {

constexpr int kIpcThreads = 1024;

__device__ __forceinline__ void store_word(uint64_t* p, uint64_t w) {
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // one aligned 64-bit vector store, sc0 sc1
}
__device__ __forceinline__ uint64_t load_word(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One workgroup: the messages are at most a few thousand floats, and a single group keeps the sequence word
// race-free (every thread reads it before the barrier, thread 0 advances it after).
__global__ __launch_bounds__(kIpcThreads) void k_ipc_allreduce(float* __restrict__ data, int n, IpcMailboxes mb) {
    uint32_t seq = *mb.seq + 1u;
    if (seq == 0u) seq = 2u;  // 0 is the flag of a never-written word; keep the parity alternating across the wrap
    const int set = (int)(seq & 1u);
    const size_t my_slot = ((size_t)set * mb.world + mb.rank) * mb.cap;

    // push: own mailbox last, peers starting from rank+1 so the ranks do not all hit the same target at once
    for (int i = threadIdx.x; i < n; i += kIpcThreads) {
        const uint64_t w = ((uint64_t)seq << 32) | (uint64_t)__float_as_uint(data[i]);
        for (int k = 1; k <= mb.world; ++k) {
            int p = mb.rank + k;
            if (p >= mb.world) p -= mb.world;
            store_word(mb.mail[p] + my_slot + i, w);
        }
    }

    // gather from the local mailbox, senders in rank order (identical rounding on every rank)
    const uint64_t* mine = mb.mail[mb.rank] + (size_t)set * mb.world * mb.cap;
    const long long t0 = wall_clock64();
    bool late = false;
    for (int i = threadIdx.x; i < n; i += kIpcThreads) {
        float acc = 0.f;
        for (int s = 0; s < mb.world; ++s) {
            const uint64_t* p = mine + (size_t)s * mb.cap + i;
            uint64_t w = load_word(p);
            while ((uint32_t)(w >> 32) != seq) {
                if (late || wall_clock64() - t0 > mb.timeout_ticks) {  // bounded: every wave leaves the loop
                    late = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                w = load_word(p);
            }
            acc += __uint_as_float((uint32_t)w);
        }
        data[i] = acc;
    }
    if (late) __hip_atomic_store(mb.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) *mb.seq = seq;
}

}  // namespace

void launch_ipc_allreduce(float* data, int n, const IpcMailboxes& mb, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(kIpcThreads), 0, s, data, n, mb);
}

}  // namespace pdt
