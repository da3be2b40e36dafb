#pragma once
#include <ATen/ATen.h>
#include <torch/extension.h>

#include <vector>

namespace pdt {

// dsts[i] <- srcs[i] * scale in one multi-tensor launch (GPU tensors, dense, equal strides, 16-B aligned)
void multi_flatten_tensors(std::vector<at::Tensor>& srcs, std::vector<at::Tensor>& dsts, double scale);

void bind_ops(pybind11::module& m);

}  // namespace pdt
