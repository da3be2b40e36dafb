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

// ---------------------------------------------------------------------------------------------------------------
// Two-shot DIRECT all-reduce for gradient buckets (SURVEY 5.8 layer 4, 2.5 S5; /root/reference/main.py:83 delegates
// this to NCCL's rings).  xGMI is a full mesh of point-to-point links: a ring moves 2 (W-1)/W * S bytes of every
// rank over ONE link per direction, the direct form moves S/W to each of the W-1 peers at once, twice:
//
//   scatter   rank r stores shard j of its bucket into slot r of peer j's window (W-1 links busy in parallel)
//   reduce    rank r adds the W versions of shard r in RANK ORDER (own one from the bucket, the others from its
//             window), writes the result into its bucket ...
//   gather    ... and stores it into every peer's window; each rank then copies the W-1 foreign shards out of its
//             own window (local reads).  One rank computes each element, so all ranks hold identical bits.
//
// Same word format as the mailboxes: {32 payload bits, 32-bit sequence number} in one aligned 64-bit store, two of
// them per 16-B store; a word is valid when its sequence number is this operation's, so no fence orders payload
// against flag and nothing relies on how stores of different lanes arrive over a link.  That doubles the bytes on
// the links (2 * 2 * S / W per link against the ring's 2 (W-1)/W * S: still 1.75x fewer at W = 8) and keeps the
// protocol as simple as the one the SyncBN path has been running.  No double buffer is needed: a peer can start
// operation n+1 only after it has MY reduced shard of operation n, which I send after reading all of its words of
// operation n, and a gather word of n+1 needs my scatter words of n+1, which my NEXT kernel sends.
// A payload is one fp32 or two bf16 (added in fp32, rounded once).  Blocks never wait for blocks of their own
// grid, and every wait is for words that the peers store BEFORE their first wait, so the grid may be any size.
constexpr int kLargeThreads = 512;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ipc_store2(uint64_t* p, uint32_t seq, uint32_t a, uint32_t b) {
    // two self-validating words in one 16-B system-scope (write-through) store; either half may land first
    const u32x4_t w = {a, seq, b, seq};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
}
// both words of a pair out of the own window: one 16-B system-scope load per poll
__device__ __forceinline__ void ipc_wait2(const uint64_t* p, uint32_t seq, const IpcMailboxes& mb, bool* late, long long t0,
                                          uint32_t* a, uint32_t* b) {
    u32x4_t w;
    while (true) {
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
        if (w[1] == seq && w[3] == seq) break;
        if (*late || wall_clock64() - t0 > mb.timeout_ticks) {
            *late = true;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
    *a = w[0];
    *b = w[2];
}
__device__ __forceinline__ float2 unpack_pair(uint32_t v, bool bf16) {
    if (!bf16) return make_float2(__uint_as_float(v), 0.f);
    return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u));
}
__device__ __forceinline__ uint32_t f32_to_bf16_bits(float f) {
    const __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round to nearest even, a NaN stays a NaN
    return (uint32_t)__builtin_bit_cast(unsigned short, h);
}

template <bool BF16>
__global__ __launch_bounds__(kLargeThreads) void k_ipc_allreduce2(uint32_t* __restrict__ data, long nwords, long shard,
                                                                  IpcMailboxes mb) {
    const uint32_t seq = ipc_begin(mb);
    const int W = mb.world, r = mb.rank;
    const long tid = (long)blockIdx.x * kLargeThreads + threadIdx.x, nthr = (long)gridDim.x * kLargeThreads;
    // window of rank p: [W][cap] scatter slots, then [W][cap] gather slots (cap >= shard, shard even)
    const size_t gather_off = (size_t)W * mb.cap;
    // ---- scatter: my version of shard j -> slot r of peer j
    for (int k = 1; k < W; ++k) {
        int j = r + k;
        if (j >= W) j -= W;
        const long base = (long)j * shard;
        uint64_t* dst = mb.mail[j] + (size_t)r * mb.cap;
        for (long i = 2 * tid; i < shard; i += 2 * nthr) {
            const long g = base + i;
            const uint32_t a = g < nwords ? data[g] : 0u, b = g + 1 < nwords ? data[g + 1] : 0u;
            ipc_store2(dst + i, seq, a, b);
        }
    }
    // ---- reduce my shard in rank order, publish the result
    const long long t0 = wall_clock64();
    bool late = ipc_dead(mb);
    {
        const long base = (long)r * shard;
        const uint64_t* win = mb.mail[r];
        for (long i = 2 * tid; i < shard; i += 2 * nthr) {
            const long g = base + i;
            float2 s0 = make_float2(0.f, 0.f), s1 = make_float2(0.f, 0.f);
            for (int p = 0; p < W; ++p) {
                uint32_t a, b;
                if (p == r) {
                    a = g < nwords ? data[g] : 0u;
                    b = g + 1 < nwords ? data[g + 1] : 0u;
                } else {
                    ipc_wait2(win + (size_t)p * mb.cap + i, seq, mb, &late, t0, &a, &b);
                }
                const float2 fa = unpack_pair(a, BF16), fb = unpack_pair(b, BF16);
                s0.x += fa.x; s0.y += fa.y; s1.x += fb.x; s1.y += fb.y;
            }
            uint32_t ra, rb;
            if (BF16) {
                ra = f32_to_bf16_bits(s0.x) | (f32_to_bf16_bits(s0.y) << 16);
                rb = f32_to_bf16_bits(s1.x) | (f32_to_bf16_bits(s1.y) << 16);
            } else {
                ra = __float_as_uint(s0.x);
                rb = __float_as_uint(s1.x);
            }
            if (g < nwords) data[g] = ra;
            if (g + 1 < nwords) data[g + 1] = rb;
            for (int k = 1; k < W; ++k) {
                int j = r + k;
                if (j >= W) j -= W;
                ipc_store2(mb.mail[j] + gather_off + (size_t)r * mb.cap + i, seq, ra, rb);
            }
        }
    }
    // ---- gather: the other ranks' reduced shards out of my own window
    for (int k = 1; k < W; ++k) {
        int p = r + k;
        if (p >= W) p -= W;
        const long base = (long)p * shard;
        const uint64_t* src = mb.mail[r] + gather_off + (size_t)p * mb.cap;
        for (long i = 2 * tid; i < shard; i += 2 * nthr) {
            const long g = base + i;
            if (g >= nwords) break;
            uint32_t a, b;
            ipc_wait2(src + i, seq, mb, &late, t0, &a, &b);
            data[g] = a;
            if (g + 1 < nwords) data[g + 1] = b;
        }
    }
    ipc_end(mb, seq, late);
}

}  // namespace

void launch_ipc_allreduce_large(void* data, long nwords, bool bf16, const IpcMailboxes& mb, int blocks, hipStream_t s) {
    if (nwords <= 0 || mb.world <= 1) return;
    long shard = (nwords + mb.world - 1) / mb.world;
    shard += shard & 1;  // pairs of words travel together
    if (bf16) hipLaunchKernelGGL(k_ipc_allreduce2<true>, dim3(blocks), dim3(kLargeThreads), 0, s, static_cast<uint32_t*>(data), nwords, shard, mb);
    else hipLaunchKernelGGL(k_ipc_allreduce2<false>, dim3(blocks), dim3(kLargeThreads), 0, s, static_cast<uint32_t*>(data), nwords, shard, mb);
}

void launch_ipc_allreduce(float* data, int n, const IpcMailboxes& mb, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ipc_allreduce, dim3(1), dim3(kIpcThreads), 0, s, data, n, mb);
}

}  // namespace pdt
