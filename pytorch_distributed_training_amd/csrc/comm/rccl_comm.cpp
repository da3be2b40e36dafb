#include "rccl_comm.h"

#include <c10/util/Exception.h>

#include <chrono>
#include <cstdlib>
#include <cstring>

namespace pdt {

#define PDT_NCCL(expr)                                                                      \
    do {                                                                                    \
        ncclResult_t _r = (expr);                                                           \
        TORCH_CHECK(_r == ncclSuccess || _r == ncclInProgress, "RCCL error: ", ncclGetErrorString(_r), \
                    " (" #expr ")");                                                        \
    } while (0)
#define PDT_HIP(expr)                                                                \
    do {                                                                             \
        hipError_t _e = (expr);                                                      \
        TORCH_CHECK(_e == hipSuccess, "HIP error: ", hipGetErrorString(_e), " (" #expr ")"); \
    } while (0)

namespace {
ncclDataType_t nccl_type(at::ScalarType st) {
    switch (st) {
        case at::kFloat: return ncclFloat32;
        case at::kBFloat16: return ncclBfloat16;
        case at::kHalf: return ncclFloat16;
        case at::kDouble: return ncclFloat64;
        case at::kLong: return ncclInt64;
        case at::kInt: return ncclInt32;
        case at::kByte: return ncclUint8;
        default: TORCH_CHECK(false, "unsupported dtype for RCCL: ", st);
    }
}
void check_tensor(const at::Tensor& t) {
    TORCH_CHECK(t.is_cuda() && t.is_contiguous(), "RCCL collectives take contiguous GPU tensors");
}
}  // namespace

std::string RcclComm::make_unique_id() {
    ncclUniqueId id;
    PDT_NCCL(ncclGetUniqueId(&id));
    return std::string(reinterpret_cast<const char*>(&id), sizeof(id));
}

RcclComm::RcclComm(int rank, int world_size, const std::string& unique_id, int device, bool high_priority)
    : rank_(rank), world_(world_size), device_(device) {
    TORCH_CHECK(unique_id.size() == sizeof(ncclUniqueId), "bad unique id size");
    PDT_HIP(hipSetDevice(device));
    int lo = 0, hi = 0;
    PDT_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));  // hi is the numerically lowest = highest priority
    PDT_HIP(hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, high_priority ? hi : lo));
    PDT_HIP(hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
    ncclUniqueId id;
    std::memcpy(&id, unique_id.data(), sizeof(id));
    PDT_NCCL(ncclCommInitRank(&comm_, world_size, id, rank));
    const char* wd = std::getenv("PDT_RCCL_WATCHDOG");
    if (!(wd && std::string(wd) == "0")) watchdog_ = std::thread([this] { watchdog_loop(); });
}

void RcclComm::close() {
    stop_ = true;
    if (watchdog_.joinable()) watchdog_.join();  // no thread may poll a communicator that is being destroyed
    if (comm_) {
        if (!aborted_) {
            hipStreamSynchronize(stream_);
            ncclCommDestroy(comm_);
        }
        comm_ = nullptr;
    }
}

RcclComm::~RcclComm() {
    close();
    if (ev_) hipEventDestroy(ev_);
    if (stream_) hipStreamDestroy(stream_);
}

void RcclComm::watchdog_loop() {
    while (!stop_) {
        for (int i = 0; i < 10 && !stop_; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        if (stop_ || aborted_) break;
        ncclResult_t async = ncclSuccess;
        if (ncclCommGetAsyncError(comm_, &async) != ncclSuccess) continue;
        if (async != ncclSuccess && async != ncclInProgress) {
            fprintf(stderr, "[pdt] rank %d: RCCL async error '%s': aborting the communicator and exiting\n", rank_,
                    ncclGetErrorString(async));
            aborted_ = true;
            ncclCommAbort(comm_);
            std::_Exit(134);
        }
    }
}

std::string RcclComm::async_error() {
    if (!comm_) return "communicator closed";
    ncclResult_t async = ncclSuccess;
    ncclResult_t r = ncclCommGetAsyncError(comm_, &async);
    if (r != ncclSuccess) return ncclGetErrorString(r);
    if (async != ncclSuccess && async != ncclInProgress) return ncclGetErrorString(async);
    return "";
}

void RcclComm::abort() {
    if (comm_ && !aborted_) {
        aborted_ = true;
        ncclCommAbort(comm_);
    }
}

void RcclComm::all_reduce_sum(at::Tensor& t, hipStream_t s) {
    check_tensor(t);
    TORCH_CHECK(comm_, "RcclComm: collective on a closed communicator");
    PDT_NCCL(ncclAllReduce(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_type(t.scalar_type()), ncclSum, comm_,
                           s ? s : stream_));
}
void RcclComm::broadcast(at::Tensor& t, int root, hipStream_t s) {
    check_tensor(t);
    TORCH_CHECK(comm_, "RcclComm: collective on a closed communicator");
    PDT_NCCL(ncclBroadcast(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_type(t.scalar_type()), root, comm_,
                           s ? s : stream_));
}
void RcclComm::reduce_sum(at::Tensor& t, int root, hipStream_t s) {
    check_tensor(t);
    TORCH_CHECK(comm_, "RcclComm: collective on a closed communicator");
    PDT_NCCL(ncclReduce(t.data_ptr(), t.data_ptr(), (size_t)t.numel(), nccl_type(t.scalar_type()), ncclSum, root,
                        comm_, s ? s : stream_));
}
void RcclComm::all_gather(const at::Tensor& src, at::Tensor& dst, hipStream_t s) {
    check_tensor(src);
    TORCH_CHECK(comm_, "RcclComm: collective on a closed communicator");
    check_tensor(dst);
    TORCH_CHECK(dst.numel() == src.numel() * world_ && dst.scalar_type() == src.scalar_type(), "all_gather size mismatch");
    PDT_NCCL(ncclAllGather(src.data_ptr(), dst.data_ptr(), (size_t)src.numel(), nccl_type(src.scalar_type()), comm_,
                           s ? s : stream_));
}

void RcclComm::fence_to(hipStream_t s) {
    PDT_HIP(hipEventRecord(ev_, stream_));
    PDT_HIP(hipStreamWaitEvent(s, ev_, 0));
}
void RcclComm::fence_from(hipStream_t s) {
    PDT_HIP(hipEventRecord(ev_, s));
    PDT_HIP(hipStreamWaitEvent(stream_, ev_, 0));
}
void RcclComm::synchronize() { PDT_HIP(hipStreamSynchronize(stream_)); }

}  // namespace pdt
