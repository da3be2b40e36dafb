#include "reducer.h"

#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/functions/accumulate_grad.h>
#include <torch/csrc/autograd/utils/lambda_post_hook.h>
#include <torch/csrc/distributed/c10d/Types.hpp>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <ctime>

#include "../binding/binding.h"

namespace pdt {

namespace {
#define PDT_HIP(expr)                                                                \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        TORCH_CHECK(_e == hipSuccess, "HIP error: ", hipGetErrorString(_e), " (" #expr ")"); \
    } while (0)
// roctx through dlopen: the markers are optional, the extension must load on a box without the profiler SDK
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char* name : {"libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
            }
        }
        push = nullptr;
        pop = nullptr;
    }
};
Roctx& roctx() {
    static Roctx r;
    return r;
}
// PDT_TRACE_COLL=1: one stderr line per bucket launch / wait (which collectives are outstanding when a job stalls)
bool trace_coll() {
    static const bool on = [] {
        const char* e = std::getenv("PDT_TRACE_COLL");
        return e && e[0] == '1';
    }();
    return on;
}
double mono_s() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
struct RoctxRange {
    bool on;
    RoctxRange(bool enabled, const std::string& msg) : on(enabled && roctx().push != nullptr) {
        if (on) roctx().push(msg.c_str());
    }
    ~RoctxRange() {
        if (on) roctx().pop();
    }
};
}  // namespace

Reducer::Reducer(std::vector<at::Tensor> params, std::vector<std::string> names, std::shared_ptr<RcclComm> rccl,
                 c10::intrusive_ptr<c10d::ProcessGroup> pg, int64_t world_size, int64_t cap_bytes,
                 int64_t first_cap_bytes, c10::optional<at::ScalarType> grad_dtype, bool gradient_as_bucket_view)
    : params_(std::move(params)),
      names_(std::move(names)),
      rccl_(std::move(rccl)),
      pg_(std::move(pg)),
      world_(world_size),
      cap_(cap_bytes),
      first_cap_(first_cap_bytes),
      as_view_(gradient_as_bucket_view) {
    TORCH_CHECK(!params_.empty(), "Reducer: no parameters");
    is_cuda_ = params_[0].is_cuda();
    grad_dtype_ = grad_dtype.value_or(params_[0].scalar_type());
    for (const auto& p : params_) {
        TORCH_CHECK(p.device() == params_[0].device(), "Reducer: all parameters must live on one device");
        numels_.push_back(p.numel());
    }
    TORCH_CHECK(world_ == 1 || rccl_ || pg_, "Reducer: world_size > 1 needs a communicator");
    std::vector<int64_t> order;
    for (int64_t i = (int64_t)params_.size() - 1; i >= 0; --i) order.push_back(i);
    build(order);
    const char* dbg = std::getenv("PDT_DEBUG_REDUCER");
    debug_ = dbg && dbg[0] == '1';

    install_hooks();
}

void Reducer::set_debug(bool on) {
    std::lock_guard<std::mutex> g(mu_);
    debug_ = on;
}
void Reducer::set_roctx(bool on) {
    std::lock_guard<std::mutex> g(mu_);
    roctx_ = on;
}

void Reducer::install_hooks() {
    remove_hooks();
    // hooks on the AccumulateGrad nodes: fire after .grad has been written
    for (size_t i = 0; i < params_.size(); ++i) {
        if (!params_[i].requires_grad()) continue;
        auto acc = torch::autograd::impl::grad_accumulator(params_[i]);
        const uintptr_t key = acc->add_post_hook(std::make_unique<torch::autograd::utils::LambdaPostHook>(
            [this, i](const torch::autograd::variable_list& outputs, const torch::autograd::variable_list&) {
                this->on_grad_ready(i);
                return outputs;
            }));
        accumulators_.push_back(std::move(acc));
        hook_keys_.push_back(key);
    }
}

Reducer::~Reducer() {
    remove_hooks();
    destroy_buckets();
    if (bwd_end_) hipEventDestroy(bwd_end_);
    if (own_stream_) {
        hipStreamSynchronize(own_stream_);
        hipStreamDestroy(own_stream_);
    }
    if (own_ev_) hipEventDestroy(own_ev_);
}

void Reducer::set_bucket_ipc(std::shared_ptr<IpcAllReduce> ipc) {
    std::lock_guard<std::mutex> g(mu_);
    TORCH_CHECK(!callback_queued_, "set_bucket_ipc() must be called between iterations");
    TORCH_CHECK(!ipc || is_cuda_, "set_bucket_ipc(): GPU parameters only");
    TORCH_CHECK(!ipc || ipc->world_size() == world_, "set_bucket_ipc(): windows of ", ipc ? ipc->world_size() : 0,
                " ranks for a job of ", world_);
    ipc_ = std::move(ipc);
    // Where the bucket kernels run.  Default: IN the compute stream.  Every rank then enqueues ONE sequence of kernels that
    // wait for peers (SyncBN exchanges inside the BN kernels, bucket all-reduces), in the same order on every rank, so
    // no cycle can form whatever the GPU co-schedules; the price is that a bucket's transfer (~0.2 ms for 16 MiB at
    // W = 8 by the link model) is not hidden behind backward.  PDT_IPC_BUCKET_STREAM=comm puts them on the
    // communicator's stream (or a stream of the reducer's own) to overlap backward, like ncclAllReduce: measured
    // with two ranks SHARING one GPU, that variant stalled in 4 of 6 runs next to the in-kernel SyncBN exchange
    // (two streams per process full of spinning waves on one device; profiles/r3_w2_hang.md), so it is opt-in.
    const char* where = std::getenv("PDT_IPC_BUCKET_STREAM");
    ipc_inline_ = !(where && std::string(where) == "comm");
    if (ipc_ && !rccl_ && !own_stream_ && !ipc_inline_) {
        int lo = 0, hi = 0;
        PDT_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char* pr = std::getenv("PDT_IPC_BUCKET_PRIO");
        PDT_HIP(hipStreamCreateWithPriority(&own_stream_, hipStreamNonBlocking, (pr && pr[0] == '0') ? lo : hi));
        PDT_HIP(hipEventCreateWithFlags(&own_ev_, hipEventDisableTiming));
    }
}

bool Reducer::ipc_fits(const Bucket& b) const {
    if (!ipc_ || !is_cuda_ || world_ <= 1) return false;
    const bool bf16 = grad_dtype_ == at::kBFloat16;
    if (!bf16 && grad_dtype_ != at::kFloat) return false;
    if (bf16 && b.plan.numel % 2 != 0) return false;
    const int64_t words = bf16 ? b.plan.numel / 2 : b.plan.numel;
    return words <= ipc_->large_capacity_words();
}

void Reducer::remove_hooks() {
    std::lock_guard<std::mutex> g(mu_);
    for (size_t k = 0; k < accumulators_.size(); ++k) accumulators_[k]->del_post_hook(hook_keys_[k]);
    accumulators_.clear();
    hook_keys_.clear();
}

void Reducer::destroy_buckets() {
    for (auto& b : buckets_) {
        if (b.done) hipEventDestroy(b.done);
        if (b.t0) hipEventDestroy(b.t0);
        if (b.t1) hipEventDestroy(b.t1);
    }
    buckets_.clear();
    timed_ = false;
}

void Reducer::build(const std::vector<int64_t>& order_in) {
    std::vector<int64_t> order;
    for (int64_t i : order_in)
        if (params_[i].requires_grad()) order.push_back(i);
    destroy_buckets();
    const int64_t elem = (int64_t)c10::elementSize(grad_dtype_);
    auto plans = plan_buckets(numels_, order, elem, cap_, first_cap_);
    where_.assign(params_.size(), {SIZE_MAX, SIZE_MAX});
    auto opts = params_[0].options().dtype(grad_dtype_).requires_grad(false);
    for (size_t bi = 0; bi < plans.size(); ++bi) {
        Bucket b;
        b.plan = plans[bi];
        b.flat = at::zeros({b.plan.numel}, opts);
        for (size_t j = 0; j < b.plan.params.size(); ++j) {
            const auto& p = params_[b.plan.params[j]];
            at::Tensor v;
            if (p.is_non_overlapping_and_dense())
                v = at::as_strided(b.flat, p.sizes(), p.strides(), b.plan.offsets[j]);
            else
                v = b.flat.narrow(0, b.plan.offsets[j], p.numel()).view(p.sizes());
            b.views.push_back(v);
            where_[b.plan.params[j]] = {bi, j};
        }
        b.ready.assign(b.plan.params.size(), 0);
        b.pending = (int64_t)b.plan.params.size();
        if (is_cuda_) PDT_HIP(hipEventCreateWithFlags(&b.done, hipEventDisableTiming));
        buckets_.push_back(std::move(b));
    }
    order_ = order;
}

void Reducer::prepare_for_backward() {
    std::lock_guard<std::mutex> g(mu_);
    for (auto& b : buckets_) {
        std::fill(b.ready.begin(), b.ready.end(), 0);
        b.pending = (int64_t)b.plan.params.size();
        b.launched = false;
        b.work.reset();
    }
    next_bucket_ = 0;
    callback_queued_ = false;
    launch_log_.clear();
    if (!rebuilt_) ready_order_.clear();
}

void Reducer::on_grad_ready(size_t idx) {
    std::lock_guard<std::mutex> g(mu_);
    if (!require_sync_) return;
    if (!callback_queued_) {
        torch::autograd::Engine::get_default_engine().queue_callback([this] { this->finalize(); });
        callback_queued_ = true;
    }
    if (!rebuilt_) ready_order_.push_back((int64_t)idx);
    const auto [bi, j] = where_[idx];
    TORCH_CHECK(bi != SIZE_MAX, "gradient for a parameter that is not in any bucket: ", names_[idx]);
    Bucket& b = buckets_[bi];
    TORCH_CHECK(!b.ready[j], "parameter ", names_[idx],
                " received a second gradient in one backward; run the DDP forward once per backward");
    b.ready[j] = 1;
    --b.pending;
    launch_ready();
}

void Reducer::launch_ready() {
    while (next_bucket_ < buckets_.size() && buckets_[next_bucket_].pending == 0) {
        launch(buckets_[next_bucket_]);
        ++next_bucket_;
    }
}

void Reducer::launch(Bucket& b) {
    const size_t bidx = (size_t)(&b - buckets_.data());
    if (debug_) {
        // plan order, once, and only with every gradient in (the zero-fill of finalize is the one exception)
        TORCH_CHECK(bidx == next_bucket_, "reducer debug: bucket ", bidx, " launched out of plan order (expected ", next_bucket_, ")");
        TORCH_CHECK(!b.launched, "reducer debug: bucket ", bidx, " launched twice in one iteration");
        TORCH_CHECK(in_finalize_ || b.pending == 0, "reducer debug: bucket ", bidx, " launched with ", b.pending, " gradients missing");
        for (size_t j = 0; j < b.ready.size() && !in_finalize_; ++j)
            TORCH_CHECK(b.ready[j], "reducer debug: bucket ", bidx, " slot ", j, " (", names_[b.plan.params[j]], ") not ready");
        ++debug_checks_;
    }
    launch_log_.push_back((int64_t)bidx);
    RoctxRange range(roctx_, "pdt.bucket" + std::to_string(bidx) + ".allreduce[" +
                                 std::to_string(b.plan.numel * (int64_t)c10::elementSize(grad_dtype_)) + "B]");
    const double scale = 1.0 / (double)world_;
    std::vector<at::Tensor> srcs, dsts;
    for (size_t j = 0; j < b.plan.params.size(); ++j) {
        const at::Tensor& g = params_[b.plan.params[j]].grad();
        at::Tensor& v = b.views[j];
        if (!g.defined()) {  // unused in this graph: contributes zeros
            v.zero_();
            continue;
        }
        const bool aliased = g.data_ptr() == v.data_ptr() && g.strides() == v.strides() &&
                             g.scalar_type() == v.scalar_type();
        if (aliased && world_ == 1) continue;  // already in place, nothing to scale
        if (is_cuda_ && g.is_non_overlapping_and_dense() && g.strides() == v.strides() &&
            reinterpret_cast<uintptr_t>(g.data_ptr()) % 16 == 0 &&
            (g.scalar_type() == at::kFloat || g.scalar_type() == at::kBFloat16)) {
            srcs.push_back(g);
            dsts.push_back(v);
        } else if (aliased) {
            v.mul_(scale);
        } else {
            v.copy_(g);
            if (world_ != 1) v.mul_(scale);
        }
    }
    if (!srcs.empty()) multi_flatten_tensors(srcs, dsts, scale);  // one launch: copy + 1/W (+ cast)  (N2)

    if (ipc_fits(b)) {
        // two-shot direct all-reduce over the peers' IPC windows: same stream discipline as the RCCL branch
        hipStream_t compute = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(b.flat.device().index()).stream();
        hipStream_t cs = ipc_inline_ ? compute : comm_stream();
        if (ipc_inline_) {
            // (the kernel sits in the compute stream itself: no overlap with backward, no second stream)
        } else if (rccl_) {
            rccl_->fence_from(compute);
        } else {
            PDT_HIP(hipEventRecord(own_ev_, compute));
            PDT_HIP(hipStreamWaitEvent(own_stream_, own_ev_, 0));
        }
        if (timing_ && rccl_) {
            if (!b.t0) {
                PDT_HIP(hipEventCreate(&b.t0));
                PDT_HIP(hipEventCreate(&b.t1));
            }
            PDT_HIP(hipEventRecord(b.t0, cs));
        }
        ipc_->all_reduce_large(b.flat.data_ptr(), b.plan.numel, grad_dtype_ == at::kBFloat16, cs);
        if (timing_ && rccl_) PDT_HIP(hipEventRecord(b.t1, cs));
        PDT_HIP(hipEventRecord(b.done, cs));
        b.done_iter = iteration_;
        ++ipc_launches_;
    } else if (rccl_ || pg_) {  // (a size-1 communicator is legal: the GPU tests use it to run this exact path)
        if (rccl_ && is_cuda_) {
            hipStream_t compute = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(b.flat.device().index()).stream();
            rccl_->fence_from(compute);           // comm stream waits for the flatten
            if (timing_) {
                if (!b.t0) {
                    PDT_HIP(hipEventCreate(&b.t0));
                    PDT_HIP(hipEventCreate(&b.t1));
                }
                PDT_HIP(hipEventRecord(b.t0, rccl_->stream()));
            }
            rccl_->all_reduce_sum(b.flat);        // overlaps the rest of backward
            if (timing_) PDT_HIP(hipEventRecord(b.t1, rccl_->stream()));
            PDT_HIP(hipEventRecord(b.done, rccl_->stream()));
            b.done_iter = iteration_;
        } else {
            // GPU buckets on a c10d group without RCCL (gloo: the one-GPU rehearsal of W > 1) are staged through the
            // host HERE, with a synchronous copy on the compute stream, and the collective runs on the CPU copy.
            // Handing gloo the GPU tensor makes it stage on its own pool streams with worker threads that block
            // in hipStreamSynchronize while this thread blocks in the next small collective; with more streams
            // than hardware queues that stopped two ranks sharing a GPU (profiles/r3_w2_hang.md).
            if (is_cuda_) b.host = b.flat.to(at::kCPU);
            std::vector<at::Tensor> ts{is_cuda_ ? b.host : b.flat};
            c10d::AllreduceOptions opts;
            opts.reduceOp = c10d::ReduceOp::SUM;
            if (trace_coll()) fprintf(stderr, "[coll r%d %.4f] bucket%zu it%ld issue\n", pg_->getRank(), mono_s(), bidx, (long)iteration_);
            b.work = pg_->allreduce(ts, opts);
            if (trace_coll()) fprintf(stderr, "[coll r%d %.4f] bucket%zu it%ld issued\n", pg_->getRank(), mono_s(), bidx, (long)iteration_);
        }
    }
    b.launched = true;
}

void Reducer::finalize() {
    std::lock_guard<std::mutex> g(mu_);
    callback_queued_ = false;
    if (!require_sync_) return;
    RoctxRange range(roctx_, "pdt.reducer.finalize");
    // parameters that never produced a gradient are zero-filled and their buckets launched now
    in_finalize_ = true;
    while (next_bucket_ < buckets_.size()) {
        launch(buckets_[next_bucket_]);
        ++next_bucket_;
    }
    in_finalize_ = false;
    if (debug_) {
        for (size_t bi = 0; bi < buckets_.size(); ++bi) {
            TORCH_CHECK(buckets_[bi].launched, "reducer debug: bucket ", bi, " was never launched");
            if (rccl_ && is_cuda_)
                TORCH_CHECK(buckets_[bi].done_iter == iteration_, "reducer debug: bucket ", bi,
                            " has no completion event from this iteration");
        }
        TORCH_CHECK(launch_log_.size() == buckets_.size(), "reducer debug: ", launch_log_.size(), " launches for ",
                    buckets_.size(), " buckets");
        ++debug_checks_;
    }
    if (rccl_ || pg_ || ipc_) {
        if (device_comm()) {
            hipStream_t compute = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(params_[0].device().index()).stream();
            if (timing_) {  // where backward's own work ends: what the buckets finish after this is exposed
                if (!bwd_end_) PDT_HIP(hipEventCreate(&bwd_end_));
                PDT_HIP(hipEventRecord(bwd_end_, compute));
                timed_ = true;
            }
            for (auto& b : buckets_)
                if (b.done_iter == iteration_) PDT_HIP(hipStreamWaitEvent(compute, b.done, 0));
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            if (debug_ && hipStreamIsCapturing(compute, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
                // event order: once the compute stream has drained, every all-reduce it waited for must be complete
                PDT_HIP(hipStreamSynchronize(compute));
                for (size_t bi = 0; bi < buckets_.size(); ++bi)
                    TORCH_CHECK(hipEventQuery(buckets_[bi].done) == hipSuccess, "reducer debug: bucket ", bi,
                                " all-reduce still pending after the compute stream waited for it");
                ++debug_checks_;
            }
        }
        {
            for (auto& b : buckets_)
                if (b.work) {
                    const size_t bidx = (size_t)(&b - buckets_.data());
                    if (trace_coll()) fprintf(stderr, "[coll r%d %.4f] bucket%zu it%ld wait\n", pg_->getRank(), mono_s(), bidx, (long)iteration_);
                    b.work->wait();
                    b.work.reset();
                    if (b.host.defined()) {
                        b.flat.copy_(b.host);
                        b.host.reset();
                    }
                    if (trace_coll()) fprintf(stderr, "[coll r%d %.4f] bucket%zu it%ld done\n", pg_->getRank(), mono_s(), bidx, (long)iteration_);
                }
        }
    }
    for (auto& b : buckets_) {
        std::vector<at::Tensor> srcs, dsts;  // unflatten (N3): bucket -> .grad in one multi-tensor launch (+ cast)
        for (size_t j = 0; j < b.plan.params.size(); ++j) {
            at::Tensor& p = params_[b.plan.params[j]];
            at::Tensor& v = b.views[j];
            if (as_view_ && grad_dtype_ == p.scalar_type()) {
                const at::Tensor& cur = p.grad();
                if (!cur.defined() || cur.data_ptr() != v.data_ptr()) p.mutable_grad() = v;
            } else {
                if (!p.grad().defined()) p.mutable_grad() = at::empty_like(p);
                at::Tensor& g = p.mutable_grad();
                const auto gt = g.scalar_type(), vt = v.scalar_type();
                if (is_cuda_ && g.is_non_overlapping_and_dense() && g.strides() == v.strides() &&
                    reinterpret_cast<uintptr_t>(g.data_ptr()) % 16 == 0 &&
                    reinterpret_cast<uintptr_t>(v.data_ptr()) % 16 == 0 && (gt == at::kFloat || gt == at::kBFloat16) &&
                    (vt == at::kFloat || vt == at::kBFloat16) && (dsts.empty() || dsts[0].scalar_type() == gt)) {
                    srcs.push_back(v);
                    dsts.push_back(g);
                } else {
                    g.copy_(v);
                }
            }
        }
        if (!srcs.empty()) multi_flatten_tensors(srcs, dsts, 1.0);
    }
    ++iteration_;
}

void Reducer::rebuild(const std::vector<int64_t>& order) {
    std::lock_guard<std::mutex> g(mu_);
    TORCH_CHECK(!callback_queued_, "rebuild() must be called between iterations");
    // keep the freshly reduced gradients: re-home them into the new buckets
    std::vector<at::Tensor> old(params_.size());
    for (size_t i = 0; i < params_.size(); ++i)
        if (params_[i].grad().defined()) old[i] = params_[i].grad();
    std::vector<Bucket> keep_alive;
    keep_alive.swap(buckets_);  // old flat buffers must outlive the copies below
    build(order);
    if (as_view_) {
        for (auto& b : buckets_)
            for (size_t j = 0; j < b.plan.params.size(); ++j) {
                const int64_t i = b.plan.params[j];
                if (old[i].defined() && grad_dtype_ == params_[i].scalar_type()) {
                    b.views[j].copy_(old[i]);
                    params_[i].mutable_grad() = b.views[j];
                }
            }
    }
    for (auto& b : keep_alive) {
        if (b.done) hipEventDestroy(b.done);
        if (b.t0) hipEventDestroy(b.t0);
        if (b.t1) hipEventDestroy(b.t1);
    }
    rebuilt_ = true;
}

void Reducer::set_timing(bool on) {
    std::lock_guard<std::mutex> g(mu_);
    TORCH_CHECK(!callback_queued_, "set_timing() must be called between iterations");
    timing_ = on && is_cuda_ && rccl_ != nullptr;
    timed_ = false;
}

std::tuple<std::vector<double>, double, double> Reducer::comm_stats() {
    std::lock_guard<std::mutex> g(mu_);
    TORCH_CHECK(timing_ && timed_, "comm_stats(): no timed iteration (set_timing(true), then run a backward; the "
                                   "timing needs the native RCCL communicator)");
    std::vector<double> per;
    hipEvent_t first = nullptr, last = nullptr;
    for (auto& b : buckets_) {
        TORCH_CHECK(b.t0 && b.t1 && b.launched, "comm_stats(): a bucket was not reduced in the last iteration");
        PDT_HIP(hipEventSynchronize(b.t1));
        float ms = 0.f;
        PDT_HIP(hipEventElapsedTime(&ms, b.t0, b.t1));
        per.push_back(ms);
        if (!first) first = b.t0;
        last = b.t1;
    }
    PDT_HIP(hipEventSynchronize(bwd_end_));
    float exposed = 0.f, span = 0.f;
    PDT_HIP(hipEventElapsedTime(&span, first, last));
    // negative when the last bucket finished before backward did: hipEventElapsedTime then reports an error or a
    // negative number depending on the runtime, both mean "nothing exposed"
    if (hipEventElapsedTime(&exposed, bwd_end_, last) != hipSuccess || exposed < 0.f) exposed = 0.f;
    return {per, (double)exposed, (double)span};
}

std::vector<std::vector<int64_t>> Reducer::bucket_indices() const {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<std::vector<int64_t>> out;
    for (const auto& b : buckets_) out.push_back(b.plan.params);
    return out;
}
std::vector<int64_t> Reducer::bucket_numels() const {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<int64_t> out;
    for (const auto& b : buckets_) out.push_back(b.plan.numel);
    return out;
}
std::vector<at::Tensor> Reducer::bucket_tensors() const {
    std::lock_guard<std::mutex> g(mu_);
    std::vector<at::Tensor> out;
    for (const auto& b : buckets_) out.push_back(b.flat);
    return out;
}

}  // namespace pdt
