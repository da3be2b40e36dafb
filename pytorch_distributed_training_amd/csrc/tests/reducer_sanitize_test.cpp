// The C++ gradient reducer (csrc/reducer/reducer.cpp) and the host logic of the RCCL communicator wrapper
// (csrc/comm/rccl_comm.cpp) under AddressSanitizer + UBSan, and separately under ThreadSanitizer ("the
// reducer is where races live": autograd-thread hooks vs the finalize callback vs API calls from the main thread).
//
// No GPU and no Python: every "rank" is a thread with its own parameters and its own Reducer on CPU tensors; the
// collectives go through an in-process c10d::ProcessGroup whose allreduce sums across the rank threads (barrier +
// shared scratch), so the exact code path of the CPU/gloo tests runs (hooks -> bucket launch in plan order ->
// allreduce Work -> finalize -> rebuild from the observed order), but in ONE instrumented process.  The HIP-only
// branches of the reducer stay behind `is_cuda_`; they link (libamdhip64 / librccl are present) and never run.
//
// Built and run by tests/test_native_host.py:
//   hipcc -x c++ -std=c++17 -O1 -g -fsanitize=address,undefined  (and -fsanitize=thread)  ... -ltorch -ltorch_cpu -lc10 ...
#include <torch/torch.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <thread>

#include "../binding/binding.h"
#include "../reducer/reducer.h"

// the IPC window launchers are device code (comm/ipc_allreduce.hip): this host-only build never reaches them (no
// reducer here gets windows), the symbols only have to resolve
namespace pdt {
void launch_ipc_allreduce(float*, int, const IpcMailboxes&, hipStream_t) { std::abort(); }
void launch_ipc_allreduce_large(void*, long, bool, const IpcMailboxes&, int, hipStream_t) { std::abort(); }
}  // namespace pdt

#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            std::_Exit(1);                                                               \
        }                                                                                \
    } while (0)

namespace pdt {
// the GPU multi-tensor flatten lives in ops.cpp (kernel launches); CPU tensors never reach it
void multi_flatten_tensors(std::vector<at::Tensor>&, std::vector<at::Tensor>&, double) {
    std::fprintf(stderr, "multi_flatten_tensors called on the CPU path\n");
    std::_Exit(2);
}
}  // namespace pdt

namespace {

// ---- in-process "communicator": W rank threads meet at a reusable barrier ----------------------------------
struct Shared {
    explicit Shared(int w) : world(w), slots(w) {}
    int world;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long generation = 0;
    std::vector<at::Tensor> slots;
    at::Tensor sum;
    void barrier(const std::function<void()>& last) {
        std::unique_lock<std::mutex> lk(mu);
        const long gen = generation;
        if (++arrived == world) {
            if (last) last();
            arrived = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen; });
        }
    }
};

class DoneWork : public c10d::Work {
   public:
    bool isCompleted() override { return true; }
    bool isSuccess() const override { return true; }
    bool wait(std::chrono::milliseconds) override { return true; }
};

class ThreadGroup : public c10d::ProcessGroup {
   public:
    ThreadGroup(int rank, std::shared_ptr<Shared> sh) : c10d::ProcessGroup(rank, sh->world), sh_(std::move(sh)) {}
    c10::intrusive_ptr<c10d::Work> allreduce(std::vector<at::Tensor>& tensors, const c10d::AllreduceOptions&) override {
        CHECK(tensors.size() == 1);
        at::Tensor& t = tensors[0];
        {
            std::lock_guard<std::mutex> g(sh_->mu);
            sh_->slots[getRank()] = t;
        }
        sh_->barrier([&] {  // the last arriver sums, in rank order (every rank then reads identical bits)
            at::Tensor s = sh_->slots[0].clone();
            for (int r = 1; r < sh_->world; ++r) {
                CHECK(sh_->slots[r].numel() == s.numel());  // identical bucket plans on every rank
                s.add_(sh_->slots[r]);
            }
            sh_->sum = s;
        });
        t.copy_(sh_->sum);
        sh_->barrier(nullptr);  // nobody overwrites `sum` before everybody has read it
        ++calls;
        return c10::make_intrusive<DoneWork>();
    }
    const std::string getBackendName() const override { return "threads"; }
    std::atomic<int> calls{0};

   private:
    std::shared_ptr<Shared> sh_;
};

struct Result {
    std::vector<at::Tensor> grads;
    int allreduces = 0;
    int64_t checks = 0;
    std::vector<std::vector<int64_t>> buckets;
};

Result run_rank(int rank, const std::shared_ptr<Shared>& sh, bool as_view, int iters) {
    torch::manual_seed(7);  // identical parameters on every rank
    std::vector<at::Tensor> params;
    std::vector<std::string> names;
    const std::vector<std::vector<int64_t>> shapes = {{64, 3, 3, 3}, {64}, {128, 64}, {128}, {5, 128}, {5}, {300, 7}};
    for (size_t i = 0; i < shapes.size(); ++i) {
        params.push_back(torch::randn(shapes[i]).requires_grad_(true));
        names.push_back("p" + std::to_string(i));
    }
    params.push_back(torch::randn({9}).requires_grad_(true));  // never used in the graph: zero-filled by finalize
    names.push_back("unused");
    auto pg = c10::make_intrusive<ThreadGroup>(rank, sh);
    // tiny buckets: several launches per backward, rebuilt after the first iteration
    auto red = std::make_shared<pdt::Reducer>(params, names, nullptr, pg, sh->world, /*cap*/ 4096, /*first*/ 1024,
                                              c10::nullopt, as_view);
    red->set_debug(true);
    Result out;
    for (int it = 0; it < iters; ++it) {
        for (auto& p : params) p.mutable_grad() = at::Tensor();  // zero_grad(set_to_none)
        // per-rank data: the mean over ranks is what the gradients must become
        at::Tensor loss = torch::zeros({});
        for (size_t i = 0; i + 1 < params.size(); ++i)
            loss = loss + (params[i] * (float)(rank + 1 + it)).sum() * (float)(i + 1);
        if (!red->rebuilt() && red->iteration() >= 1) {
            // what reducer_native.py does between iteration 1 and 2: adopt the observed ready order
            auto ready = red->ready_order();
            std::vector<int64_t> order = ready;
            for (int64_t i : red->current_order())
                if (std::find(ready.begin(), ready.end(), i) == ready.end()) order.push_back(i);
            red->rebuild(order);
        }
        red->prepare_for_backward();
        // API calls from ANOTHER thread while the autograd engine drives the hooks: the reducer's mutex must hold
        std::atomic<bool> stop{false};
        std::thread poker([&] {
            while (!stop) {
                (void)red->bucket_numels();
                (void)red->launch_log();
                (void)red->debug_checks();
                std::this_thread::yield();
            }
        });
        loss.backward();
        stop = true;
        poker.join();
        float mean_scale = 0.f;
        for (int r = 0; r < sh->world; ++r) mean_scale += (float)(r + 1 + it);
        mean_scale /= (float)sh->world;
        for (size_t i = 0; i + 1 < params.size(); ++i) {
            const at::Tensor& g = params[i].grad();
            CHECK(g.defined());
            const float want = mean_scale * (float)(i + 1);
            CHECK((g - want).abs().max().item<float>() < 1e-4f * (1.f + want));
        }
        CHECK(params.back().grad().defined() && params.back().grad().abs().max().item<float>() == 0.f);
        const auto log = red->launch_log();
        for (size_t k = 0; k < log.size(); ++k) CHECK(log[k] == (int64_t)k);  // plan order
        CHECK(log.size() == red->bucket_numels().size());
    }
    CHECK(red->rebuilt());
    for (auto& p : params) out.grads.push_back(p.grad().clone());
    out.allreduces = pg->calls.load();
    out.checks = red->debug_checks();
    out.buckets = red->bucket_indices();
    red->remove_hooks();
    return out;
}

}  // namespace

int main() {
    at::set_num_threads(1);
    for (int world : {1, 2, 3})
        for (bool as_view : {true, false}) {
            auto sh = std::make_shared<Shared>(world);
            std::vector<Result> res(world);
            std::vector<std::thread> ts;
            for (int r = 0; r < world; ++r) ts.emplace_back([&, r] { res[r] = run_rank(r, sh, as_view, 4); });
            for (auto& t : ts) t.join();
            for (int r = 1; r < world; ++r) {
                CHECK(res[r].buckets == res[0].buckets);
                CHECK(res[r].allreduces == res[0].allreduces);
                for (size_t i = 0; i < res[0].grads.size(); ++i) CHECK(torch::equal(res[r].grads[i], res[0].grads[i]));
            }
            CHECK(res[0].buckets.size() >= 3 && res[0].checks > 0);
            if (world > 1) CHECK(res[0].allreduces >= 4 * 3);
        }
    // communicator wrapper: argument validation that needs no device
    try {
        (void)std::make_shared<pdt::RcclComm>(0, 1, std::string("short"), 0, true);
        CHECK(false && "a malformed unique id must be refused");
    } catch (const c10::Error&) {
    }
    std::printf("OK\n");
    std::fflush(stdout);
    // leave without the static destructors of the (uninstrumented) ROCm libraries: their teardown is not under test
    std::_Exit(0);
}
