// Native RCCL communicator (SURVEY 5.8 layer 1).
//
// One communicator per process/GPU, created with ncclCommInitRank from a unique id that rank 0
// publishes through the c10d store (done by the Python wrapper).  Collectives are enqueued on
// streams WE own: a high-priority non-blocking HIP stream per communicator, fenced against the
// compute stream with events by the reducer.  A watchdog thread polls ncclCommGetAsyncError
// and aborts the communicator so a dead peer produces an exit instead of a hang.
#pragma once
#include <ATen/ATen.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <string>
#include <thread>

namespace pdt {

class RcclComm {
   public:
    RcclComm(int rank, int world_size, const std::string& unique_id, int device, bool high_priority);
    ~RcclComm();
    RcclComm(const RcclComm&) = delete;
    RcclComm& operator=(const RcclComm&) = delete;

    static std::string make_unique_id();

    int rank() const { return rank_; }
    int world_size() const { return world_; }
    hipStream_t stream() const { return stream_; }

    // in-place collectives on `s` (the communicator's own stream when s == nullptr)
    void all_reduce_sum(at::Tensor& t, hipStream_t s = nullptr);
    void broadcast(at::Tensor& t, int root, hipStream_t s = nullptr);
    void reduce_sum(at::Tensor& t, int root, hipStream_t s = nullptr);
    void all_gather(const at::Tensor& src, at::Tensor& dst, hipStream_t s = nullptr);

    // make `s` wait for everything enqueued so far on the communicator stream
    void fence_to(hipStream_t s);
    // make the communicator stream wait for everything enqueued so far on `s`
    void fence_from(hipStream_t s);
    void synchronize();
    std::string async_error();  // "" when healthy
    void abort();
    // Ordered teardown, callable before the destructor (and idempotent): 1. stop and join the watchdog thread,
    // 2. drain the communicator's stream, 3. ncclCommDestroy.  Every rank calls it after its last collective.
    void close();
    bool closed() const { return comm_ == nullptr; }

   private:
    void watchdog_loop();
    int rank_, world_, device_;
    ncclComm_t comm_ = nullptr;
    hipStream_t stream_ = nullptr;
    hipEvent_t ev_ = nullptr;
    std::atomic<bool> stop_{false};
    std::atomic<bool> aborted_{false};
    std::thread watchdog_;
};

}  // namespace pdt
