// One-shot all-reduce for latency-bound messages over IPC-mapped peer memory (SURVEY 5.8 layer 3,
// 2.5 S4a/S4b: the 106 SyncBN all-reduces of 2C floats per step, C <= 2048).
//
// Every rank owns a mailbox in uncached device memory: [2 sets][W senders][cap] 64-bit words.  One
// all-reduce is ONE kernel on the caller's stream, no proxy thread and no stream hand-over:
//
//   push    each element is stored as {float bits, sequence number} in one 64-bit system-scope store
//           into slot `rank` of EVERY peer's mailbox (xGMI is a full mesh: the 7 peer stores of an
//           element leave on 7 different links) and into the rank's own;
//   gather  the rank polls its own mailbox (local HBM reads, nothing crosses a link) until the word
//           of sender s carries this operation's sequence number, and adds the W values in sender
//           order, so every rank produces bit-identical sums.
//
// The flag travels inside the same aligned 64-bit store as the payload, so no fence separates data
// from flag.  Two mailbox sets alternate with the sequence number's parity: a rank can only be one
// operation ahead of a peer (it needs the peer's words of operation n to finish n), so the words it
// overwrites for n+2 have been consumed.  The sequence number lives in device memory and is advanced
// by the kernel itself, which makes a launch replayable inside a hipGraph.
//
// Every wait is bounded by a wall-clock budget; a rank whose peer never arrives raises a flag in
// host-visible memory and returns instead of spinning forever.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace pdt {

constexpr int kIpcMaxRanks = 16;

struct IpcMailboxes {
    uint64_t* mail[kIpcMaxRanks];  // mail[p]: rank p's mailbox as mapped in this process (mail[rank] is local)
    int rank, world;
    int cap;                  // elements per slot
    uint32_t* seq;            // device word: sequence number of the last finished operation
    uint32_t* done;           // device word: blocks of the running multi-block operation that have finished
    uint32_t* status;         // host-pinned word: set to 1 by a wait that ran out of time
    uint32_t* dead;           // device copy of `status`: later kernels (a replayed hipGraph has up to 106 more
                              // exchanges queued behind the one that failed) read it and do not wait again
    long long timeout_ticks;  // wall_clock64 ticks (100 MHz)
};

// device launcher (ipc_allreduce.hip): in-place sum of n <= cap floats
void launch_ipc_allreduce(float* data, int n, const IpcMailboxes& mb, hipStream_t s);
// two-shot direct all-reduce of a bucket of 32-bit words (one fp32 or two bf16 each), in place; the window of every
// rank holds [2][W][cap] words with cap >= ceil(nwords / W) rounded up to even (ipc_allreduce.hip has the protocol)
void launch_ipc_allreduce_large(void* data, long nwords, bool bf16, const IpcMailboxes& mb, int blocks, hipStream_t s);

class IpcAllReduce {
   public:
    IpcAllReduce(int rank, int world, int device, int cap_elems, double timeout_s);
    ~IpcAllReduce();
    IpcAllReduce(const IpcAllReduce&) = delete;
    IpcAllReduce& operator=(const IpcAllReduce&) = delete;

    std::string handle() const;                          // this rank's hipIpcMemHandle_t bytes
    void open_peers(const std::vector<std::string>& h);  // handles of all ranks, indexed by rank
    bool ready() const { return ready_; }
    int cap() const { return mb_.cap; }
    int world_size() const { return mb_.world; }
    bool timed_out() const { return host_status_ != nullptr && *host_status_ != 0; }
    void clear_error();
    // ordered teardown: device synchronised, peer mappings closed, mailbox freed; idempotent, the object is
    // unusable afterwards (dist.shutdown calls it before the process group goes away)
    // (an exporter must not free memory a peer still has mapped: a job closes in two phases with a barrier
    // between them, `close_peers()` on every rank, then `close()`; parallel/ipc.py does that)
    void close_peers();
    void close();
    bool closed() const { return closed_; }
    void set_timeout(double seconds) { mb_.timeout_ticks = (long long)(seconds * 100e6); }
    void all_reduce(float* data, int64_t n, hipStream_t s);
    // gradient buckets: two-shot direct all-reduce of `numel` fp32 or bf16 elements (bf16: an even count); an object
    // used for this must not also carry small messages (the two protocols share the sequence word, not the layout)
    void all_reduce_large(void* data, int64_t numel, bool bf16, hipStream_t s);
    int64_t large_capacity_words() const { return (int64_t)mb_.world * (mb_.cap & ~1); }
    // for kernels that do the exchange themselves (ipc_device.h): the mailboxes, after the same checks as above
    const IpcMailboxes& mailboxes_for(int64_t n_elems);

   private:
    int device_;
    size_t bytes_ = 0;
    bool ready_ = false;
    bool closed_ = false;
    long calls_ = 0;
    IpcMailboxes mb_{};
    volatile uint32_t* host_status_ = nullptr;
    hipIpcMemHandle_t handle_{};
    std::vector<void*> opened_;
};

}  // namespace pdt
