// One-shot all-reduce kernel over IPC-mapped mailboxes: see ipc_allreduce.h for the protocol.
#include <hip/hip_runtime.h>

#include "ipc_device.h"

namespace pdt {

namespace {

constexpr int kIpcThreads = 1024;

// One workgroup: the messages are at most a few thousand floats.
__global__ __launch_bounds__(kIpcThreads) void k_ipc_allreduce(float* __restrict__ data, int n, IpcMailboxes mb) {
    const uint32_t seq = ipc_begin(mb);
    for (int i = threadIdx.x; i < n; i += kIpcThreads) ipc_push(mb, seq, i, data[i]);
    // gather from the local mailbox, senders in rank order (identical rounding on every rank)
    const long long t0 = wall_clock64();
    bool late = ipc_dead(mb);
    for (int i = threadIdx.x; i < n; i += kIpcThreads) {
        float acc = 0.f;
        for (int s = 0; s < mb.world; ++s) acc += ipc_wait(mb, seq, s, i, &late, t0);
        data[i] = acc;
    }
    ipc_end(mb, seq, late);
}

}  // namespace

void launch_ipc_allreduce(float* data, int n, const IpcMailboxes& mb, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(kIpcThreads), 0, s, data, n, mb);
}

}  // namespace pdt
