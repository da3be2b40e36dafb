// Device side of the IPC mailboxes (protocol: ipc_allreduce.h), usable from ANY kernel: the SyncBatchNorm kernels
// (kernels/bn.hip) exchange their per-channel statistics themselves instead of handing a packed tensor to a
// separate all-reduce launch (106 data-dependent exchanges per step that cannot be
// overlapped, only made cheap: zero extra launches, no packing / unpacking kernels).
//
//   const uint32_t seq = ipc_begin(mb);                     every thread of every block (reads the device word)
//   ipc_push(mb, seq, i, value);                            element i of this rank's message -> every mailbox
//   bool late = ipc_dead(mb);                               (an earlier exchange timed out: do not wait again)
//   v = ipc_wait(mb, seq, sender, i, &late, t0);            polls the OWN mailbox (local memory)
//   ipc_end(mb, seq, late);                                 every thread, at the end of the kernel
//
// Any number of blocks may take part: an element must be pushed by exactly one thread of the grid and may be
// waited for by any.  The sequence number advances when the LAST block of the grid is done (device counter), so
// the kernels stay replayable inside a hipGraph.
#pragma once
#include <hip/hip_runtime.h>

#include "ipc_allreduce.h"

namespace pdt {

__device__ __forceinline__ void ipc_store_word(uint64_t* p, uint64_t w) {
    __hip_atomic_store(p, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // one aligned 64-bit vector store, sc0 sc1
}
__device__ __forceinline__ uint64_t ipc_load_word(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__device__ __forceinline__ uint32_t ipc_begin(const IpcMailboxes& mb) {
    uint32_t seq = __hip_atomic_load(mb.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (seq == 0u) seq = 2u;  // 0 is the flag of a never-written word; keep the parity alternating across the wrap
    return seq;
}

// true once an earlier exchange of this job ran out of time: start with `bool late = ipc_dead(mb);` so that the
// kernels queued behind the failed one (the rest of a replayed hipGraph) do not each wait the time limit again
__device__ __forceinline__ bool ipc_dead(const IpcMailboxes& mb) {
    return __hip_atomic_load(mb.dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// own mailbox last, peers starting from rank+1 so the ranks do not all hit the same target at once
__device__ __forceinline__ void ipc_push_bits(const IpcMailboxes& mb, uint32_t seq, int i, uint32_t bits) {
    const size_t slot = ((size_t)(seq & 1u) * mb.world + mb.rank) * mb.cap + i;
    const uint64_t w = ((uint64_t)seq << 32) | (uint64_t)bits;
    for (int k = 1; k <= mb.world; ++k) {
        int p = mb.rank + k;
        if (p >= mb.world) p -= mb.world;
        ipc_store_word(mb.mail[p] + slot, w);
    }
}
__device__ __forceinline__ void ipc_push(const IpcMailboxes& mb, uint32_t seq, int i, float v) {
    ipc_push_bits(mb, seq, i, __float_as_uint(v));
}

// element i of `sender`'s message of operation seq; bounded: after the time limit (or once *late is set) the wait
// gives up, *late is raised and the value is whatever the word holds
__device__ __forceinline__ uint32_t ipc_wait_bits(const IpcMailboxes& mb, uint32_t seq, int sender, int i, bool* late,
                                                  long long t0) {
    const uint64_t* p = mb.mail[mb.rank] + ((size_t)(seq & 1u) * mb.world + sender) * mb.cap + i;
    uint64_t w = ipc_load_word(p);
    while ((uint32_t)(w >> 32) != seq) {
        if (*late || wall_clock64() - t0 > mb.timeout_ticks) {
            *late = true;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
        w = ipc_load_word(p);
    }
    return (uint32_t)w;
}
// the same wait on an explicit word of the own window (two-shot bucket all-reduce)
__device__ __forceinline__ uint32_t ipc_wait_word(const uint64_t* p, uint32_t seq, const IpcMailboxes& mb, bool* late,
                                                  long long t0) {
    uint64_t w = ipc_load_word(p);
    while ((uint32_t)(w >> 32) != seq) {
        if (*late || wall_clock64() - t0 > mb.timeout_ticks) {
            *late = true;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
        w = ipc_load_word(p);
    }
    return (uint32_t)w;
}
__device__ __forceinline__ float ipc_wait(const IpcMailboxes& mb, uint32_t seq, int sender, int i, bool* late,
                                          long long t0) {
    return __uint_as_float(ipc_wait_bits(mb, seq, sender, i, late, t0));
}

// every thread of the grid calls this once, after its last wait (contains a block barrier)
__device__ __forceinline__ void ipc_end(const IpcMailboxes& mb, uint32_t seq, bool late) {
    if (late) {
        __hip_atomic_store(mb.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(mb.dead, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
        const uint32_t blocks = gridDim.x * gridDim.y * gridDim.z;
        const uint32_t prev = __hip_atomic_fetch_add(mb.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1u == blocks) {  // the last block: nobody reads the sequence word of this operation any more
            __hip_atomic_store(mb.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mb.seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace pdt
