// refign_amd/torch_shim/correlation_shim.cpp -- the reference's native operator boundary as a BUILT artefact: a pybind11 /
// torch-extension module named `correlation` with exactly the two functions models/correlation_ops/correlation_sampler.cpp
// binds (:62-70 forward, :92-101 backward, :129-132 PYBIND11_MODULE), on top of the C ABI of librefign_hip.so
// (include/refign_hip.h: rfn_corr_fwd_* / rfn_corr_bwd_*).  Contract kept: contiguous NCHW inputs on one device
// (CHECK_CONTIGUOUS / CHECK_SAME_DEVICE, correlation_sampler.cpp:13-16 -> RuntimeError), freshly allocated results returned by
// value, float / double / half (AT_DISPATCH_FLOATING_TYPES_AND_HALF, correlation_cuda_kernel.cu:267).  One difference,
// on purpose: the kernels run on the CURRENT stream (the CUDA reference launches on the legacy default stream,
// correlation_cuda_kernel.cu:271) -- which is what torch code around it expects.
// Host-only C++: no HIP source here; built by refign_amd/torch_shim/build.py (torch.utils.cpp_extension, in-tree).
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/extension.h>

#include <vector>

#include "../../include/refign_hip.h"

namespace {

void check_inputs(const torch::Tensor& a, const torch::Tensor& b, const char* what) {
  TORCH_CHECK(a.is_contiguous() && b.is_contiguous(), what, ": inputs must be contiguous");
  TORCH_CHECK(a.device() == b.device(), what, ": inputs must be on the same device");
  TORCH_CHECK(a.is_cuda(), what, ": this build serves HIP tensors (the CPU path is the reference's own correlation.cpp)");
  TORCH_CHECK(a.dim() == 4 && b.sizes() == a.sizes(), what, ": inputs must be (B, C, H, W) of equal shape");
  TORCH_CHECK(a.scalar_type() == b.scalar_type() && (a.scalar_type() == torch::kFloat32 || a.scalar_type() == torch::kFloat64 ||
                                                     a.scalar_type() == torch::kFloat16),
              what, ": float32 / float64 / float16");
}

void check_rc(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", rfn_last_error()); }

torch::Tensor correlation_sample_forward(torch::Tensor input1, torch::Tensor input2, int kH, int kW, int patchH, int patchW,
                                         int padH, int padW, int dilationH, int dilationW, int dilation_patchH,
                                         int dilation_patchW, int dH, int dW) {
  check_inputs(input1, input2, "correlation.forward");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(input1.device());   // (ROCm torch presents its devices as "cuda")
  const int B = input1.size(0), C = input1.size(1), iH = input1.size(2), iW = input1.size(3);
  const int oH = (iH + 2 * padH - ((kH - 1) * dilationH + 1)) / dH + 1, oW = (iW + 2 * padW - ((kW - 1) * dilationW + 1)) / dW + 1;
  auto out = torch::empty({B, patchH, patchW, oH, oW}, input1.options());
  void* st = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(input1.device().index()).stream();
  if (input1.scalar_type() == torch::kFloat32)
    check_rc(rfn_corr_fwd_f32(input1.data_ptr<float>(), input2.data_ptr<float>(), out.data_ptr<float>(), B, C, iH, iW, kH, kW,
                              patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH, dilation_patchW, dH, dW, st),
             "correlation.forward");
  else if (input1.scalar_type() == torch::kFloat16)
    check_rc(rfn_corr_fwd_f16(input1.data_ptr(), input2.data_ptr(), out.data_ptr(), B, C, iH, iW, kH, kW, patchH, patchW, padH,
                              padW, dilationH, dilationW, dilation_patchH, dilation_patchW, dH, dW, st),
             "correlation.forward");
  else
    check_rc(rfn_corr_fwd_f64(input1.data_ptr<double>(), input2.data_ptr<double>(), out.data_ptr<double>(), B, C, iH, iW, kH, kW,
                              patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH, dilation_patchW, dH, dW, st),
             "correlation.forward");
  return out;
}

std::vector<torch::Tensor> correlation_sample_backward(torch::Tensor input1, torch::Tensor input2, torch::Tensor grad_output,
                                                       int kH, int kW, int patchH, int patchW, int padH, int padW, int dilationH,
                                                       int dilationW, int dilation_patchH, int dilation_patchW, int dH, int dW) {
  check_inputs(input1, input2, "correlation.backward");
  TORCH_CHECK(grad_output.device() == input1.device() && grad_output.scalar_type() == input1.scalar_type(),
              "correlation.backward: grad_output device / dtype");
  const c10::hip::HIPGuardMasqueradingAsCUDA guard(input1.device());   // (ROCm torch presents its devices as "cuda")
  auto go = grad_output.contiguous();
  const int B = input1.size(0), C = input1.size(1), iH = input1.size(2), iW = input1.size(3);
  auto g1 = torch::zeros_like(input1), g2 = torch::zeros_like(input2);
  void* st = (void*)c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(input1.device().index()).stream();
  if (input1.scalar_type() == torch::kFloat32)
    check_rc(rfn_corr_bwd_f32(input1.data_ptr<float>(), input2.data_ptr<float>(), go.data_ptr<float>(), g1.data_ptr<float>(),
                              g2.data_ptr<float>(), B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW,
                              dilation_patchH, dilation_patchW, dH, dW, st),
             "correlation.backward");
  else if (input1.scalar_type() == torch::kFloat16)
    check_rc(rfn_corr_bwd_f16(input1.data_ptr(), input2.data_ptr(), go.data_ptr(), g1.data_ptr(), g2.data_ptr(), B, C, iH, iW,
                              kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH, dilation_patchW, dH,
                              dW, st),
             "correlation.backward");
  else
    check_rc(rfn_corr_bwd_f64(input1.data_ptr<double>(), input2.data_ptr<double>(), go.data_ptr<double>(), g1.data_ptr<double>(),
                              g2.data_ptr<double>(), B, C, iH, iW, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW,
                              dilation_patchH, dilation_patchW, dH, dW, st),
             "correlation.backward");
  return {g1, g2};
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("forward", &correlation_sample_forward, "Spatial Correlation Sampler Forward (MI355X, librefign_hip.so)");
  m.def("backward", &correlation_sample_backward, "Spatial Correlation Sampler backward (MI355X, librefign_hip.so)");
}
