// C++ gradient reducer: the centrepiece of the data-parallel path.
//
//  * parameters are grouped into flat buckets (reverse registration order at first, observed
//    gradient-ready order after iteration 1; sizes chosen for xGMI, see parallel/buckets.py);
//  * a post hook on every parameter's AccumulateGrad node marks it ready -- no Python in the
//    hook path (hooks run on the autograd thread without the GIL);
//  * when every gradient of the next bucket in plan order is ready: ONE multi-tensor HIP launch
//    copies them into the flat bucket with the 1/W scale (and an optional bf16 cast) on the
//    compute stream, an event fences the communicator's own stream, and ncclAllReduce is
//    enqueued there, overlapping the rest of backward;
//  * buckets launch strictly in plan order on every rank (identical collective sequence);
//  * the finalize callback (queued on the autograd engine) makes the compute stream wait for
//    every bucket's completion event and points .grad at the bucket views, so the fused
//    optimizer reads gradients straight from the flat buffers (no unflatten copy).
// CPU tensors use the same logic with c10d (gloo) collectives: that is the plumbing/test path.
#pragma once
#include <ATen/ATen.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/distributed/c10d/ProcessGroup.hpp>

#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../comm/ipc_allreduce.h"
#include "../comm/rccl_comm.h"

#include "bucket_plan.h"

namespace pdt {

class Reducer {
   public:
    Reducer(std::vector<at::Tensor> params, std::vector<std::string> names, std::shared_ptr<RcclComm> rccl,
            c10::intrusive_ptr<c10d::ProcessGroup> pg, int64_t world_size, int64_t cap_bytes, int64_t first_cap_bytes,
            c10::optional<at::ScalarType> grad_dtype, bool gradient_as_bucket_view);
    ~Reducer();

    // Gradient buckets through the two-shot direct all-reduce over IPC windows (comm/ipc_allreduce.hip) instead of
    // ncclAllReduce: in the compute stream by default, on the communicator's stream (or on a stream of the reducer's
    // own when there is no RCCL communicator) with PDT_IPC_BUCKET_STREAM=comm; reducer.cpp says why.  Buckets that do
    // not fit the windows keep the other path.
    void set_bucket_ipc(std::shared_ptr<IpcAllReduce> ipc);
    int64_t ipc_bucket_launches() const { std::lock_guard<std::mutex> g(mu_); return ipc_launches_; }
    void prepare_for_backward();
    void set_require_sync(bool v) { std::lock_guard<std::mutex> g(mu_); require_sync_ = v; }
    // (every getter takes the mutex: the autograd thread appends to these vectors while hooks fire; found by the
    // AddressSanitizer build of csrc/tests/reducer_sanitize_test.cpp, which polls them from a second thread)
    bool rebuilt() const { std::lock_guard<std::mutex> g(mu_); return rebuilt_; }
    int64_t iteration() const { std::lock_guard<std::mutex> g(mu_); return iteration_; }
    std::vector<int64_t> ready_order() const { std::lock_guard<std::mutex> g(mu_); return ready_order_; }
    std::vector<int64_t> current_order() const { std::lock_guard<std::mutex> g(mu_); return order_; }
    void rebuild(const std::vector<int64_t>& order);  // call between iterations
    std::vector<std::vector<int64_t>> bucket_indices() const;
    std::vector<int64_t> bucket_numels() const;
    std::vector<at::Tensor> bucket_tensors() const;
    void remove_hooks();
    // (re)create the AccumulateGrad nodes and their post hooks.  A node is bound to the stream that is current
    // when it is created, so before a hipGraph capture this is called under the capture stream.
    void install_hooks();
    // Per-bucket all-reduce timing: two timed events per bucket on the communicator's stream and one
    // on the compute stream where backward ends.  Off by default; not usable while a hipGraph is being captured.
    void set_timing(bool on);
    // (per-bucket all-reduce ms, ms the compute stream had to wait after backward, ms from the first all-reduce's
    // start to the last one's end) of the most recent iteration; synchronises with the communicator's stream.
    std::tuple<std::vector<double>, double, double> comm_stats();
    // Debug mode ("the reducer is where races live"): every launch asserts that buckets go out in plan
    // order, exactly once, with every gradient slot marked ready (or the launch coming from finalize's zero-fill);
    // finalize asserts that every bucket was launched and, on the GPU path, that every completion event was
    // recorded this iteration and has really completed once the compute stream has waited for it.  Also keeps
    // the launch sequence of the last iteration for tests.  Costs one stream synchronisation per step.
    void set_debug(bool on);
    bool debug() const { std::lock_guard<std::mutex> g(mu_); return debug_; }
    std::vector<int64_t> launch_log() const { std::lock_guard<std::mutex> g(mu_); return launch_log_; }
    int64_t debug_checks() const { std::lock_guard<std::mutex> g(mu_); return debug_checks_; }
    // roctx range (rocprofv3 --marker-trace) around every bucket's flatten + all-reduce enqueue and around finalize
    void set_roctx(bool on);

   private:
    struct Bucket {
        BucketPlan plan;
        at::Tensor flat;
        std::vector<at::Tensor> views;  // shaped like the parameters, aliasing `flat`
        std::vector<char> ready;
        int64_t pending = 0;
        c10::intrusive_ptr<c10d::Work> work;
        at::Tensor host;  // GPU bucket on a group without RCCL: the CPU copy the collective runs on
        hipEvent_t done = nullptr;
        hipEvent_t t0 = nullptr, t1 = nullptr;  // timing mode only
        bool launched = false;
        int64_t done_iter = -1;  // iteration in which `done` was last recorded (debug mode)
    };
    void build(const std::vector<int64_t>& order);
    void destroy_buckets();
    void on_grad_ready(size_t idx);
    void launch_ready();
    void launch(Bucket& b);
    void finalize();

    std::vector<at::Tensor> params_;
    std::vector<std::string> names_;
    std::shared_ptr<RcclComm> rccl_;
    std::shared_ptr<IpcAllReduce> ipc_;      // optional: two-shot direct all-reduce of the buckets
    hipStream_t own_stream_ = nullptr;       // ipc_ without rccl_: the stream the bucket kernels run on
    hipEvent_t own_ev_ = nullptr;
    int64_t ipc_launches_ = 0;
    bool ipc_inline_ = true;                 // bucket kernels in the compute stream (PDT_IPC_BUCKET_STREAM=comm: side stream)
    bool device_comm() const { return is_cuda_ && (rccl_ || ipc_); }  // completion = an event on a comm stream
    hipStream_t comm_stream() const { return rccl_ ? rccl_->stream() : own_stream_; }
    bool ipc_fits(const Bucket& b) const;
    c10::intrusive_ptr<c10d::ProcessGroup> pg_;
    int64_t world_;
    int64_t cap_, first_cap_;
    at::ScalarType grad_dtype_;
    bool as_view_;
    bool is_cuda_;
    bool require_sync_ = true;
    bool rebuilt_ = false;
    bool callback_queued_ = false;
    bool timing_ = false, timed_ = false;
    bool debug_ = false, roctx_ = false, in_finalize_ = false;
    int64_t debug_checks_ = 0;
    std::vector<int64_t> launch_log_;
    hipEvent_t bwd_end_ = nullptr;
    int64_t iteration_ = 0;
    size_t next_bucket_ = 0;
    std::vector<int64_t> numels_, order_, ready_order_;
    std::vector<Bucket> buckets_;
    std::vector<std::pair<size_t, size_t>> where_;  // param -> (bucket, slot)
    std::vector<std::shared_ptr<torch::autograd::Node>> accumulators_;
    std::vector<uintptr_t> hook_keys_;
    mutable std::mutex mu_;
};

}  // namespace pdt
