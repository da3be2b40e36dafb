// Python module `pytorch_distributed_training_amd._C`: ops, native RCCL communicator, reducer.
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include "../comm/ipc_allreduce.h"
#include "../comm/rccl_comm.h"
#include "../reducer/reducer.h"
#include "binding.h"

namespace py = pybind11;

PYBIND11_MODULE(_C, m) {
    m.doc() = "MI355X (gfx950) native kernels, RCCL communicator and gradient reducer";
    m.attr("arch") = "gfx950";
    pdt::bind_ops(m);

    py::class_<pdt::RcclComm, std::shared_ptr<pdt::RcclComm>>(m, "RcclComm")
        // unique_id arrives as std::string (converted from bytes while the GIL is still held); the
        // blocking ncclCommInitRank then runs without the GIL
        .def(py::init([](int rank, int world, const std::string& id, int device, bool high_priority) {
                 return std::make_shared<pdt::RcclComm>(rank, world, id, device, high_priority);
             }),
             py::arg("rank"), py::arg("world_size"), py::arg("unique_id"), py::arg("device"),
             py::arg("high_priority") = true, py::call_guard<py::gil_scoped_release>())
        .def_static("make_unique_id", [] { return py::bytes(pdt::RcclComm::make_unique_id()); })
        .def_property_readonly("rank", &pdt::RcclComm::rank)
        .def_property_readonly("world_size", &pdt::RcclComm::world_size)
        .def("stream_handle", [](pdt::RcclComm& c) { return reinterpret_cast<uintptr_t>(c.stream()); })
        // synchronous-with-the-current-stream convenience collectives (tests, broadcasts)
        .def("all_reduce_", [](pdt::RcclComm& c, at::Tensor t) {
                 hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
                 c.fence_from(cur);
                 c.all_reduce_sum(t);
                 c.fence_to(cur);
             }, py::call_guard<py::gil_scoped_release>())
        .def("broadcast_", [](pdt::RcclComm& c, at::Tensor t, int root) {
                 hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
                 c.fence_from(cur);
                 c.broadcast(t, root);
                 c.fence_to(cur);
             }, py::call_guard<py::gil_scoped_release>())
        .def("reduce_", [](pdt::RcclComm& c, at::Tensor t, int root) {
                 hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
                 c.fence_from(cur);
                 c.reduce_sum(t, root);
                 c.fence_to(cur);
             }, py::call_guard<py::gil_scoped_release>())
        .def("all_gather_", [](pdt::RcclComm& c, at::Tensor src, at::Tensor dst) {
                 hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(src.device().index()).stream();
                 c.fence_from(cur);
                 c.all_gather(src, dst);
                 c.fence_to(cur);
             }, py::call_guard<py::gil_scoped_release>())
        .def("synchronize", &pdt::RcclComm::synchronize, py::call_guard<py::gil_scoped_release>())
        .def("async_error", &pdt::RcclComm::async_error)
        .def("abort", &pdt::RcclComm::abort)
        // ordered teardown: stop the watchdog, drain the communicator's stream, ncclCommDestroy (idempotent)
        .def("close", &pdt::RcclComm::close, py::call_guard<py::gil_scoped_release>())
        .def_property_readonly("closed", &pdt::RcclComm::closed);

    py::class_<pdt::IpcAllReduce, std::shared_ptr<pdt::IpcAllReduce>>(m, "IpcAllReduce")
        .def(py::init([](int rank, int world, int device, int cap, double timeout_s) {
                 return std::make_shared<pdt::IpcAllReduce>(rank, world, device, cap, timeout_s);
             }),
             py::arg("rank"), py::arg("world_size"), py::arg("device"), py::arg("capacity") = 16384,
             py::arg("timeout_s") = 30.0)
        .def("handle", [](pdt::IpcAllReduce& c) { return py::bytes(c.handle()); })
        .def("open_peers", [](pdt::IpcAllReduce& c, std::vector<std::string> handles) { c.open_peers(handles); })
        .def_property_readonly("ready", &pdt::IpcAllReduce::ready)
        .def_property_readonly("capacity", &pdt::IpcAllReduce::cap)
        .def_property_readonly("world_size", &pdt::IpcAllReduce::world_size)
        .def("timed_out", &pdt::IpcAllReduce::timed_out)
        .def("clear_error", &pdt::IpcAllReduce::clear_error)
        .def("set_timeout", &pdt::IpcAllReduce::set_timeout)
        .def("close_peers", &pdt::IpcAllReduce::close_peers, py::call_guard<py::gil_scoped_release>())
        .def("close", &pdt::IpcAllReduce::close, py::call_guard<py::gil_scoped_release>())
        .def_property_readonly("closed", &pdt::IpcAllReduce::closed)
        // in-place sum over the ranks of a contiguous fp32 GPU tensor, one kernel on the current stream
        .def("all_reduce_", [](pdt::IpcAllReduce& c, at::Tensor t) {
            TORCH_CHECK(t.is_cuda() && t.is_contiguous() && t.scalar_type() == at::kFloat,
                        "IpcAllReduce takes a contiguous fp32 GPU tensor");
            hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
            c.all_reduce(t.data_ptr<float>(), t.numel(), cur);
        })
        // gradient buckets: two-shot direct all-reduce of a contiguous fp32 / bf16 GPU tensor on the current stream
        .def("all_reduce_large_", [](pdt::IpcAllReduce& c, at::Tensor t) {
            TORCH_CHECK(t.is_cuda() && t.is_contiguous() && (t.scalar_type() == at::kFloat || t.scalar_type() == at::kBFloat16),
                        "IpcAllReduce.all_reduce_large_ takes a contiguous fp32 or bf16 GPU tensor");
            hipStream_t cur = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(t.device().index()).stream();
            c.all_reduce_large(t.data_ptr(), t.numel(), t.scalar_type() == at::kBFloat16, cur);
        })
        .def_property_readonly("large_capacity_words", &pdt::IpcAllReduce::large_capacity_words);

    py::class_<pdt::Reducer, std::shared_ptr<pdt::Reducer>>(m, "Reducer")
        .def(py::init([](std::vector<at::Tensor> params, std::vector<std::string> names,
                         std::shared_ptr<pdt::RcclComm> rccl, c10::intrusive_ptr<c10d::ProcessGroup> pg,
                         int64_t world, int64_t cap, int64_t first_cap, c10::optional<at::ScalarType> grad_dtype,
                         bool as_view) {
                 return std::make_shared<pdt::Reducer>(std::move(params), std::move(names), std::move(rccl),
                                                       std::move(pg), world, cap, first_cap, grad_dtype, as_view);
             }),
             py::arg("params"), py::arg("names"), py::arg("rccl"), py::arg("process_group"), py::arg("world_size"),
             py::arg("bucket_cap_bytes"), py::arg("first_bucket_bytes"), py::arg("grad_dtype"),
             py::arg("gradient_as_bucket_view"))
        .def("set_bucket_ipc", &pdt::Reducer::set_bucket_ipc)
        .def("ipc_bucket_launches", &pdt::Reducer::ipc_bucket_launches)
        .def("prepare_for_backward", &pdt::Reducer::prepare_for_backward)
        .def("set_require_sync", &pdt::Reducer::set_require_sync)
        .def_property_readonly("rebuilt", &pdt::Reducer::rebuilt)
        .def_property_readonly("iteration", &pdt::Reducer::iteration)
        .def("ready_order", &pdt::Reducer::ready_order)
        .def("current_order", &pdt::Reducer::current_order)
        .def("rebuild", &pdt::Reducer::rebuild)
        .def("bucket_indices", &pdt::Reducer::bucket_indices)
        .def("bucket_numels", &pdt::Reducer::bucket_numels)
        .def("bucket_tensors", &pdt::Reducer::bucket_tensors)
        .def("remove_hooks", &pdt::Reducer::remove_hooks)
        .def("install_hooks", &pdt::Reducer::install_hooks)
        .def("set_timing", &pdt::Reducer::set_timing)
        .def("set_debug", &pdt::Reducer::set_debug)
        .def("set_roctx", &pdt::Reducer::set_roctx)
        .def_property_readonly("debug", &pdt::Reducer::debug)
        .def("launch_log", &pdt::Reducer::launch_log)
        .def("debug_checks", &pdt::Reducer::debug_checks)
        .def("comm_stats", &pdt::Reducer::comm_stats, py::call_guard<py::gil_scoped_release>());

    m.def("plan_buckets", [](std::vector<int64_t> numels, std::vector<int64_t> order, int64_t elem, int64_t cap,
                             int64_t first_cap) {
        std::vector<std::tuple<std::vector<int64_t>, std::vector<int64_t>, int64_t>> out;
        for (auto& b : pdt::plan_buckets(numels, order, elem, cap, first_cap))
            out.emplace_back(b.params, b.offsets, b.numel);
        return out;
    });
}
