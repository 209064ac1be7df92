// qqq_torch.cpp -- compiled torch binding of the C-ABI (include/qqq_amd.h): the counterpart of the reference's pybind shim
// (csrc/pybind.cpp:3-5 -> qqq_gemm, csrc/qqq_gemm.cu:1048-1106).  Plain C++ against torch's own headers (c10/hip/HIPStream.h
// is what a ROCm build of torch ships; nothing here is hipified) -- it forwards data_ptr()s and the current HIP stream to
// libqqq_amd.so and carries no kernel code.  Two faces:
//   * pybind functions (`qqq_amd._torch_ext.qqq_gemm`, `.quantlinear_forward`, `.dynamic_quant`): the eager fast path -- no
//     dispatcher round trip, no ctypes marshalling (tools/host_overhead.py: 7.5 us -> ~2 us of host time per call);
//   * TORCH_LIBRARY ops `qqq_amd_native::{qqq_gemm, qqq_gemm_bias, qqq_gemm_w8, expand_int8, dynamic_quant}` for callers that want dispatcher-visible
//     native ops (the Python custom ops of ops.py stay the torch.compile path: they carry the fake kernels).
// Errors: the reference's own checks and messages (csrc/qqq_gemm.cu:1062-1075, :1096-1105) plus the ones it leaves undefined.
#include <c10/hip/HIPStream.h>
#include <torch/extension.h>
#include <torch/library.h>

#include "../../include/qqq_amd.h"

namespace {

struct Checked {
  int m, n, k, groupsize;
};

Checked check_common(const at::Tensor& A, const at::Tensor& B, const at::Tensor& C, const at::Tensor& D, const at::Tensor& s1,
                     const at::Tensor& s2, const at::Tensor& s3, const at::Tensor& workspace, int64_t max_par) {
  const int64_t m = A.size(0), n = C.size(1), k = A.size(1);
  const int64_t groupsize = s3.numel() == 0 ? -1 : k / s3.size(0);
  TORCH_CHECK(groupsize == -1 || groupsize * s3.size(0) == k, "k=", k, " not compatible with ", s3.size(0), " groups.");
  TORCH_CHECK(workspace.numel() >= n / 128 * max_par, "workspace must be of size at least ", n / 128 * max_par, ".");
  TORCH_CHECK(s1.scalar_type() == at::kFloat, "s1 dtype must be float32, but got ", s1.dtype(), ".");
  TORCH_CHECK(s2.scalar_type() == at::kFloat, "s2 dtype must be float32, but got ", s2.dtype(), ".");
  TORCH_CHECK(s3.scalar_type() == at::kHalf, "s3 dtype must be float16, but got ", s3.dtype(), ".");
  TORCH_CHECK(A.scalar_type() == at::kChar && B.scalar_type() == at::kInt && D.scalar_type() == at::kHalf && C.scalar_type() == at::kInt,
              "qqq_gemm: expected A int8, B int32, C int32, D float16");
  TORCH_CHECK(workspace.scalar_type() == at::kInt, "qqq_gemm: workspace must be int32");
  const at::Tensor* ts[] = {&A, &B, &C, &D, &s1, &s2, &workspace};
  const char* names[] = {"A", "B", "C", "D", "s1", "s2", "workspace"};
  for (int i = 0; i < 7; ++i) {
    TORCH_CHECK(ts[i]->is_contiguous(), "qqq_gemm: ", names[i], " must be contiguous");
    TORCH_CHECK(ts[i]->is_cuda() && ts[i]->device() == A.device(), "qqq_gemm: ", names[i],
                " must live on the same GPU as A (there is no CPU path)");
  }
  TORCH_CHECK(s3.numel() == 0 || (s3.is_contiguous() && s3.device() == A.device()), "qqq_gemm: s3 must be contiguous and on A's device");
  TORCH_CHECK(B.numel() == (k / 16) * (n * 2) && D.numel() == m * n, "qqq_gemm: B must be [k/16, 2n] and D [m, n]");
  TORCH_CHECK(s1.numel() == m && s2.numel() == n, "qqq_gemm: s1 must have m and s2 n elements");
  TORCH_CHECK(C.size(0) >= max_par * 64, "qqq_gemm: C must have at least max_par*64=", max_par * 64, " rows");
  return {(int)m, (int)n, (int)k, (int)groupsize};
}

void raise_for(int err, const Checked& c, int64_t thread_k, int64_t thread_n) {
  if (err == QQQ_OK) return;
  TORCH_CHECK(err != QQQ_ERR_PROB_SHAPE, "Problem (m=", c.m, ", n=", c.n, ", k=", c.k, ") not compatible with thread_k=", thread_k,
              ", thread_n=", thread_n, ".");
  TORCH_CHECK(err != QQQ_ERR_KERN_SHAPE, "No kernel implementation for thread_k=", thread_k, ", thread_n=", thread_n,
              ", groupsize=", c.groupsize, ".");
  TORCH_CHECK(false, "qqq_amd: error ", err, ": ", qqq_amd_last_error());
}

inline void* ptr(const at::Tensor& t) { return t.numel() ? t.data_ptr() : nullptr; }
inline void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void qqq_gemm(const at::Tensor& A, const at::Tensor& B, at::Tensor& C, at::Tensor& D, const at::Tensor& s1, const at::Tensor& s2,
              const at::Tensor& s3, at::Tensor& workspace, int64_t thread_k, int64_t thread_n, int64_t sms, int64_t max_par) {
  const Checked c = check_common(A, B, C, D, s1, s2, s3, workspace, max_par);
  const int err = qqq_w4a8_gemm(ptr(A), ptr(B), ptr(C), ptr(D), ptr(s1), ptr(s2), ptr(s3), c.m, c.n, c.k, ptr(workspace), c.groupsize,
                                A.device().index(), stream_of(A), (int)thread_k, (int)thread_n, (int)sms, (int)max_par);
  raise_for(err, c, thread_k, thread_n);
}

// the expanded int8 weights of a per-group layer (qqq_expand_int8): int8, k * n elements, on A's device
const void* checked_w8(const c10::optional<at::Tensor>& W8, int64_t k, int64_t n, int64_t groupsize, const c10::Device& dev) {
  if (!W8.has_value() || W8->numel() == 0) return nullptr;
  TORCH_CHECK(W8->scalar_type() == at::kChar && W8->numel() == k * n && W8->is_contiguous() && W8->device() == dev && (groupsize == 128 || groupsize == -1),
              "W8 must be the contiguous int8 [k * n] tensor of expand_int8 on A's device");
  return W8->data_ptr();
}

// qqq_gemm with the fused fp16 bias and / or the layer's expanded int8 weights (both optional)
void qqq_gemm_w8(const at::Tensor& A, const at::Tensor& B, at::Tensor& C, at::Tensor& D, const at::Tensor& s1, const at::Tensor& s2,
                 const at::Tensor& s3, at::Tensor& workspace, const c10::optional<at::Tensor>& bias, const c10::optional<at::Tensor>& W8,
                 int64_t max_par) {
  const Checked c = check_common(A, B, C, D, s1, s2, s3, workspace, max_par);
  const at::Tensor* b = bias.has_value() ? &bias.value() : nullptr;
  TORCH_CHECK(!b || (b->scalar_type() == at::kHalf && b->numel() == c.n && b->is_contiguous() && b->device() == A.device()),
              "bias must be a contiguous fp16 [n] tensor on A's device");
  const void* w8 = checked_w8(W8, c.k, c.n, c.groupsize, A.device());
  const int err = qqq_w4a8_gemm_ex2(ptr(A), ptr(B), ptr(C), ptr(D), ptr(s1), ptr(s2), ptr(s3), c.m, c.n, c.k, ptr(workspace), c.groupsize,
                                    A.device().index(), stream_of(A), -1, -1, -1, (int)max_par, nullptr, nullptr, b ? ptr(*b) : nullptr, w8);
  raise_for(err, c, -1, -1);
}

// load-time expansion of per-group weights: B [k/16, 2n] int32 + s_group [k/128, n] fp16 (stored order) -> W8 int8 [k * n]
at::Tensor expand_int8(const at::Tensor& B, const at::Tensor& s3) {
  TORCH_CHECK(B.scalar_type() == at::kInt && B.is_cuda() && B.is_contiguous() && B.dim() == 2, "expand_int8: B must be the packed int32 [k/16, 2n] weight on the GPU");
  const int64_t k = B.size(0) * 16, n = B.size(1) / 2;
  const bool grouped = s3.numel() != 0;
  TORCH_CHECK(!grouped || (s3.scalar_type() == at::kHalf && s3.is_contiguous() && s3.device() == B.device() && s3.dim() == 2 && s3.size(1) == n &&
                           s3.size(0) * 128 == k),
              "expand_int8: s_group must be the contiguous fp16 [k/128, n] tensor of a per-group layer on B's device (or empty: per-channel)");
  at::Tensor W8 = at::empty({k * n}, B.options().dtype(at::kChar));
  const int err = qqq_expand_int8(ptr(B), ptr(s3), ptr(W8), (int)k, (int)n, grouped ? 128 : -1, B.device().index(), stream_of(B));
  TORCH_CHECK(err == QQQ_OK, "qqq_amd: expand_int8 error ", err, ": ", qqq_amd_last_error());
  return W8;
}

void qqq_gemm_bias(const at::Tensor& A, const at::Tensor& B, at::Tensor& C, at::Tensor& D, const at::Tensor& s1, const at::Tensor& s2,
                   const at::Tensor& s3, at::Tensor& workspace, const at::Tensor& bias, int64_t max_par) {
  const Checked c = check_common(A, B, C, D, s1, s2, s3, workspace, max_par);
  TORCH_CHECK(bias.scalar_type() == at::kHalf && bias.numel() == c.n && bias.is_contiguous() && bias.device() == A.device(),
              "bias must be a contiguous fp16 [n] tensor on A's device");
  const int err = qqq_w4a8_gemm_ex(ptr(A), ptr(B), ptr(C), ptr(D), ptr(s1), ptr(s2), ptr(s3), c.m, c.n, c.k, ptr(workspace), c.groupsize,
                                   A.device().index(), stream_of(A), -1, -1, -1, (int)max_par, nullptr, nullptr, ptr(bias));
  raise_for(err, c, -1, -1);
}

std::tuple<at::Tensor, at::Tensor> dynamic_quant(const at::Tensor& x) {
  TORCH_CHECK(x.scalar_type() == at::kHalf && x.is_cuda() && x.dim() >= 1,
              "dynamic_quant: expected an fp16 tensor on the GPU (there is no CPU path)");
  const int64_t k = x.size(-1);
  const at::Tensor x2 = x.reshape({-1, k}).contiguous();
  const int64_t m = x2.size(0);
  at::Tensor xq = at::empty({m, k}, x.options().dtype(at::kChar));
  at::Tensor s1 = at::empty({m, 1}, x.options().dtype(at::kFloat));
  const int err = qqq_dynamic_quant(ptr(x2), ptr(xq), ptr(s1), (int)m, (int)k, x.device().index(), stream_of(x));
  TORCH_CHECK(err == QQQ_OK, "qqq_amd: dynamic_quant error ", err, ": ", qqq_amd_last_error());
  std::vector<int64_t> sh(x.sizes().begin(), x.sizes().end());
  at::Tensor xqr = xq.reshape(sh);
  sh.back() = 1;
  return {xqr, s1.reshape(sh)};
}

// QuantLinear.forward (qlinear_marlin.py:270-288) for a contiguous 2-D fp16 input: fused quantiser + GEMM (+ bias), one call
at::Tensor quantlinear_forward(const at::Tensor& x, const at::Tensor& B, at::Tensor& C, const at::Tensor& s2, const at::Tensor& s3,
                               at::Tensor& workspace, const c10::optional<at::Tensor>& bias, int64_t max_par,
                               const c10::optional<at::Tensor>& W8) {
  TORCH_CHECK(x.scalar_type() == at::kHalf && x.is_cuda() && x.dim() == 2 && x.is_contiguous(),
              "quantlinear_forward: expected a contiguous 2-D fp16 tensor on the GPU (there is no CPU path)");
  const int64_t m = x.size(0), k = x.size(1), n = C.size(1);
  const auto dev = x.device();
  TORCH_CHECK(B.size(0) * 16 == k && B.device() == dev && B.numel() == (k / 16) * (n * 2),
              "quantlinear_forward: B must be the packed [k/16, 2n] weight on x's device");
  const int64_t groupsize = s3.numel() == 0 ? -1 : k / s3.size(0);
  TORCH_CHECK(B.scalar_type() == at::kInt && C.scalar_type() == at::kInt && workspace.scalar_type() == at::kInt &&
                  s2.scalar_type() == at::kFloat && C.device() == dev && s2.device() == dev && workspace.device() == dev &&
                  B.is_contiguous() && C.is_contiguous() && s2.is_contiguous() && workspace.is_contiguous(),
              "quantlinear_forward: expected contiguous int32 B / C / workspace and float32 s2 on x's device");
  TORCH_CHECK(s2.numel() == n && C.size(0) >= max_par * 64 && workspace.numel() >= n / 128 * max_par, "quantlinear_forward: s2 needs n=", n,
              " elements, C max_par*64=", max_par * 64, " rows, workspace at least ", n / 128 * max_par, " entries");
  TORCH_CHECK(s3.numel() == 0 || (s3.scalar_type() == at::kHalf && s3.device() == dev && s3.is_contiguous() &&
                                  groupsize * s3.size(0) == k && s3.numel() == s3.size(0) * n),
              "quantlinear_forward: s3 must be a contiguous fp16 [k/groupsize, n] tensor on x's device");
  const at::Tensor* b = bias.has_value() ? &bias.value() : nullptr;
  TORCH_CHECK(!b || (b->scalar_type() == at::kHalf && b->numel() == n && b->device() == dev && b->is_contiguous()),
              "quantlinear_forward: bias must be a contiguous fp16 [n] tensor on x's device");
  at::Tensor xq = at::empty({m, k}, x.options().dtype(at::kChar));
  at::Tensor s1 = at::empty({m, 1}, x.options().dtype(at::kFloat));
  at::Tensor D = at::empty({m, n}, x.options());
  if (m == 0) return D;
  const void* w8 = checked_w8(W8, k, n, groupsize, dev);
  const int err = qqq_quantlinear_forward2(ptr(x), ptr(xq), ptr(s1), ptr(B), ptr(C), ptr(D), ptr(s2), ptr(s3), (int)m, (int)n, (int)k,
                                           ptr(workspace), (int)groupsize, dev.index(), stream_of(x), (int)max_par, b ? ptr(*b) : nullptr, w8);
  raise_for(err, Checked{(int)m, (int)n, (int)k, (int)groupsize}, -1, -1);
  return D;
}

}  // namespace

TORCH_LIBRARY(qqq_amd_native, m) {
  m.def("qqq_gemm(Tensor A, Tensor B, Tensor(a!) C, Tensor(b!) D, Tensor s1, Tensor s2, Tensor s3, Tensor(c!) workspace, int thread_k, "
        "int thread_n, int sms, int max_par) -> ()");
  m.def("qqq_gemm_bias(Tensor A, Tensor B, Tensor(a!) C, Tensor(b!) D, Tensor s1, Tensor s2, Tensor s3, Tensor(c!) workspace, Tensor bias, "
        "int max_par) -> ()");
  m.def("qqq_gemm_w8(Tensor A, Tensor B, Tensor(a!) C, Tensor(b!) D, Tensor s1, Tensor s2, Tensor s3, Tensor(c!) workspace, Tensor? bias, "
        "Tensor? W8, int max_par) -> ()");
  m.def("expand_int8(Tensor B, Tensor s_group) -> Tensor");
  m.def("dynamic_quant(Tensor x) -> (Tensor, Tensor)");
}

TORCH_LIBRARY_IMPL(qqq_amd_native, CUDA, m) {  // the HIP backend of a ROCm torch registers under the CUDA dispatch key
  m.impl("qqq_gemm", &qqq_gemm);
  m.impl("qqq_gemm_bias", &qqq_gemm_bias);
  m.impl("qqq_gemm_w8", &qqq_gemm_w8);
  m.impl("expand_int8", &expand_int8);
  m.impl("dynamic_quant", &dynamic_quant);
}

PYBIND11_MODULE(_torch_ext, m) {
  m.doc() = "compiled torch binding of libqqq_amd.so (include/qqq_amd.h)";
  m.def("qqq_gemm", &qqq_gemm);
  m.def("qqq_gemm_bias", &qqq_gemm_bias);
  m.def("dynamic_quant", &dynamic_quant);
  m.def("quantlinear_forward", &quantlinear_forward, pybind11::arg("x"), pybind11::arg("B"), pybind11::arg("C"), pybind11::arg("s2"), pybind11::arg("s3"),
        pybind11::arg("workspace"), pybind11::arg("bias"), pybind11::arg("max_par"), pybind11::arg("W8") = c10::optional<at::Tensor>());
  m.def("qqq_gemm_w8", &qqq_gemm_w8);
  m.def("expand_int8", &expand_int8);
  m.def("abi_version", []() { return qqq_amd_abi_version(); });
}
