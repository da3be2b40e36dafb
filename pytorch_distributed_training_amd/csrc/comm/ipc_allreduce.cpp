#include "ipc_allreduce.h"

#include <c10/util/Exception.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace pdt {

#define PDT_HIP(expr)                                                                \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        TORCH_CHECK(_e == hipSuccess, "HIP error: ", hipGetErrorString(_e), " (" #expr ")"); \
    } while (0)

IpcAllReduce::IpcAllReduce(int rank, int world, int device, int cap_elems, double timeout_s) : device_(device) {
    TORCH_CHECK(world >= 1 && world <= kIpcMaxRanks, "IpcAllReduce: world size ", world, " outside [1, ", kIpcMaxRanks, "]");
    TORCH_CHECK(rank >= 0 && rank < world, "IpcAllReduce: bad rank");
    TORCH_CHECK(cap_elems > 0 && cap_elems <= (1 << 26), "IpcAllReduce: bad capacity");
    PDT_HIP(hipSetDevice(device));
    mb_.rank = rank;
    mb_.world = world;
    mb_.cap = cap_elems;
    mb_.timeout_ticks = (long long)(timeout_s * 100e6);  // wall_clock64 runs at 100 MHz
    bytes_ = (size_t)2 * world * cap_elems * sizeof(uint64_t);
    // Peers store into this buffer over xGMI while the local kernel polls it: it must not be held in a
    // non-coherent cache.  PDT_IPC_MEM selects the allocation flavour for experiments on other hosts.
    const char* flavour = std::getenv("PDT_IPC_MEM");
    const std::string fl = flavour ? flavour : "uncached";
    void* base = nullptr;
    if (fl == "plain") {
        PDT_HIP(hipMalloc(&base, bytes_));
    } else {
        PDT_HIP(hipExtMallocWithFlags(&base, bytes_, fl == "finegrained" ? hipDeviceMallocFinegrained : hipDeviceMallocUncached));
    }
    PDT_HIP(hipMemset(base, 0, bytes_));
    for (auto& m : mb_.mail) m = nullptr;
    mb_.mail[rank] = static_cast<uint64_t*>(base);
    PDT_HIP(hipMalloc(reinterpret_cast<void**>(&mb_.seq), 4 * sizeof(uint32_t)));
    PDT_HIP(hipMemset(mb_.seq, 0, 4 * sizeof(uint32_t)));
    mb_.done = mb_.seq + 1;
    mb_.dead = mb_.seq + 2;
    void* hs = nullptr;
    PDT_HIP(hipHostMalloc(&hs, sizeof(uint32_t), hipHostMallocMapped));
    host_status_ = static_cast<volatile uint32_t*>(hs);
    *host_status_ = 0;
    void* ds = nullptr;
    PDT_HIP(hipHostGetDevicePointer(&ds, hs, 0));
    mb_.status = static_cast<uint32_t*>(ds);
    PDT_HIP(hipDeviceSynchronize());
    if (world > 1) PDT_HIP(hipIpcGetMemHandle(&handle_, base));
    ready_ = world == 1;
}

IpcAllReduce::~IpcAllReduce() { close(); }

void IpcAllReduce::close_peers() {
    if (closed_) return;
    ready_ = mb_.world == 1;
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();  // no kernel of this process still stores into a peer's mailbox or polls ours
    for (void* p : opened_) (void)hipIpcCloseMemHandle(p);
    opened_.clear();
    for (int p = 0; p < kIpcMaxRanks; ++p)
        if (p != mb_.rank) mb_.mail[p] = nullptr;
}

void IpcAllReduce::close() {
    if (closed_) return;
    close_peers();
    closed_ = true;
    ready_ = false;
    if (mb_.mail[mb_.rank]) (void)hipFree(mb_.mail[mb_.rank]);
    if (mb_.seq) (void)hipFree(mb_.seq);
    if (host_status_) (void)hipHostFree(const_cast<uint32_t*>(host_status_));
    for (auto& m : mb_.mail) m = nullptr;
    mb_.seq = mb_.done = mb_.dead = nullptr;
    host_status_ = nullptr;
}

std::string IpcAllReduce::handle() const {
    return std::string(reinterpret_cast<const char*>(&handle_), sizeof(handle_));
}

void IpcAllReduce::open_peers(const std::vector<std::string>& h) {
    TORCH_CHECK((int)h.size() == mb_.world, "IpcAllReduce: expected ", mb_.world, " handles, got ", h.size());
    TORCH_CHECK(!ready_ || mb_.world == 1, "IpcAllReduce: peers already opened");
    PDT_HIP(hipSetDevice(device_));
    for (int p = 0; p < mb_.world; ++p) {
        if (p == mb_.rank) continue;
        TORCH_CHECK(h[p].size() == sizeof(hipIpcMemHandle_t), "IpcAllReduce: bad handle size from rank ", p);
        hipIpcMemHandle_t hd;
        std::memcpy(&hd, h[p].data(), sizeof(hd));
        void* ptr = nullptr;
        PDT_HIP(hipIpcOpenMemHandle(&ptr, hd, hipIpcMemLazyEnablePeerAccess));
        opened_.push_back(ptr);
        mb_.mail[p] = static_cast<uint64_t*>(ptr);
    }
    ready_ = true;
}

void IpcAllReduce::clear_error() {
    if (closed_) return;
    (void)hipSetDevice(device_);
    (void)hipDeviceSynchronize();
    if (mb_.dead) (void)hipMemset(mb_.dead, 0, sizeof(uint32_t));
    if (host_status_) *host_status_ = 0;
}

const IpcMailboxes& IpcAllReduce::mailboxes_for(int64_t n) {
    TORCH_CHECK(!closed_, "IpcAllReduce: closed");
    TORCH_CHECK(ready_, "IpcAllReduce: open_peers() has not run");
    TORCH_CHECK(n >= 0 && n <= mb_.cap, "IpcAllReduce: ", n, " elements exceed the mailbox capacity ", mb_.cap);
    TORCH_CHECK(!timed_out(), "IpcAllReduce: an earlier exchange ran out of time waiting for a peer");
    return mb_;
}

void IpcAllReduce::all_reduce(float* data, int64_t n, hipStream_t s) {
    TORCH_CHECK(!closed_, "IpcAllReduce: closed");
    TORCH_CHECK(ready_, "IpcAllReduce: open_peers() has not run");
    TORCH_CHECK(n >= 0 && n <= mb_.cap, "IpcAllReduce: ", n, " elements exceed the mailbox capacity ", mb_.cap);
    TORCH_CHECK(!timed_out(), "IpcAllReduce: an earlier all-reduce ran out of time waiting for a peer");
    static const bool debug = std::getenv("PDT_IPC_DEBUG") != nullptr;
    if (debug) fprintf(stderr, "[ipc rank %d] call %ld n=%ld stream=%p\n", mb_.rank, (long)++calls_, (long)n, (void*)s);
    launch_ipc_allreduce(data, (int)n, mb_, s);
}

void IpcAllReduce::all_reduce_large(void* data, int64_t numel, bool bf16, hipStream_t s) {
    TORCH_CHECK(!closed_, "IpcAllReduce: closed");
    TORCH_CHECK(ready_, "IpcAllReduce: open_peers() has not run");
    TORCH_CHECK(!bf16 || numel % 2 == 0, "IpcAllReduce: a bf16 bucket needs an even element count");
    TORCH_CHECK(reinterpret_cast<uintptr_t>(data) % 4 == 0, "IpcAllReduce: bucket not 4-byte aligned");
    const int64_t nwords = bf16 ? numel / 2 : numel;
    TORCH_CHECK(nwords >= 0 && nwords <= large_capacity_words(), "IpcAllReduce: bucket of ", nwords,
                " words exceeds the windows (", large_capacity_words(), ")");
    TORCH_CHECK(!timed_out(), "IpcAllReduce: an earlier all-reduce ran out of time waiting for a peer");
    // enough workgroups to keep W-1 links busy, few enough to leave the CUs to backward (the kernel overlaps it)
    static const int blocks = [] {
        const char* e = std::getenv("PDT_IPC_BUCKET_BLOCKS");
        const int b = e ? std::atoi(e) : 48;
        return b < 1 ? 1 : (b > 256 ? 256 : b);
    }();
    const int64_t per_block = 2 * 512;
    int64_t want = (nwords / mb_.world + per_block - 1) / per_block;
    if (want < 1) want = 1;
    launch_ipc_allreduce_large(data, (long)nwords, bf16, mb_, (int)(want < blocks ? want : blocks), s);
}

}  // namespace pdt
