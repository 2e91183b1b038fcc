// nfp_torch.cpp — C++ autograd nodes over the C ABI of libnfp_hip.so (include/nfp.h).
//
// The reference's NFP is a chain of ATen ops whose autograd graph PyTorch builds (nfp.py:132-159); here one
// forward and one backward kernel stand for it, and these two torch::autograd::Function classes are the graph
// nodes.  They do what functional.py's Python autograd.Functions do — allocate out / saved / grad_x, take torch's
// current stream, call nfp_forward / nfp_backward (nfp_pool_forward / nfp_pool_backward) — without the Python
// interpreter on the launch path: eager host cost per forward + backward drops from ~75 us to the autograd
// engine's own (scripts/host_overhead.py).  No device code in this file; plain pointers go across the ABI.
//
// Built by neighbour_feature_pooling_amd/build.py::build_torch_ext (g++ against libtorch + libnfp_hip.so);
// optional: functional.py falls back to its Python nodes when the module is absent.
#include <torch/extension.h>

#include <c10/hip/HIPStream.h>

#include "../../include/nfp.h"

namespace {

using torch::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::variable_list;

void check(int rc) {
  if (rc == NFP_OK) return;
  // the Python binding maps the prefix back onto NfpUnsupported / NfpError
  TORCH_CHECK(false, rc == NFP_E_UNSUPPORTED ? "libnfp_hip unsupported: " : "libnfp_hip error: ", nfp_last_error());
}

void* stream_of(const Tensor& x) { return (void*)c10::hip::getCurrentHIPStream(x.device().index()).stream(); }

Tensor empty_like_layout(const Tensor& x, bool nhwc) {
  return torch::empty(x.sizes(), x.options().memory_format(nhwc ? at::MemoryFormat::ChannelsLast : at::MemoryFormat::Contiguous));
}

// desc: CPU uint8 tensor holding one nfp_desc (the plan's descriptor; functional.py keeps it alive)
const nfp_desc* desc_of(const Tensor& desc) {
  TORCH_CHECK(desc.device().is_cpu() && desc.scalar_type() == torch::kUInt8 && desc.numel() == (int64_t)sizeof(nfp_desc),
              "descriptor tensor must be ", sizeof(nfp_desc), " bytes on the CPU");
  return (const nfp_desc*)desc.data_ptr();
}

struct NfpNode : torch::autograd::Function<NfpNode> {
  static Tensor forward(AutogradContext* ctx, Tensor x, Tensor desc, std::vector<int64_t> oshape, int64_t ns, bool nhwc) {
    c10::DeviceGuard guard(x.device());
    const nfp_desc* d = desc_of(desc);
    Tensor out = torch::empty(oshape, x.options().memory_format(at::MemoryFormat::Contiguous));
    Tensor saved = torch::empty({ns}, x.options().dtype(torch::kFloat32).memory_format(at::MemoryFormat::Contiguous));
    check(nfp_forward(d, x.data_ptr(), out.data_ptr(), ns > 0 ? saved.data_ptr<float>() : nullptr, stream_of(x)));
    ctx->save_for_backward({x, out, saved, desc});
    ctx->saved_data["nhwc"] = nhwc;
    return out;
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto sv = ctx->get_saved_variables();
    const Tensor &x = sv[0], &out = sv[1], &saved = sv[2], &desc = sv[3];
    c10::DeviceGuard guard(x.device());
    Tensor go = grads[0].contiguous();
    if (go.scalar_type() != x.scalar_type()) go = go.to(x.scalar_type());
    Tensor gx = empty_like_layout(x, ctx->saved_data["nhwc"].toBool());
    check(nfp_backward(desc_of(desc), x.data_ptr(), go.data_ptr(), out.data_ptr(),
                       saved.numel() ? saved.data_ptr<float>() : nullptr, gx.data_ptr(), stream_of(x)));
    return {gx, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

// want_gap: GAP(x) is part of the result (NFP_Pooling.py:27); false: the pooled NFP maps alone (texture_pooling.py:251-252)
// — an empty tensor stands in for it.  need_grad: a backward will follow — only then are the maps themselves stored.
struct NfpPoolNode : torch::autograd::Function<NfpPoolNode> {
  static variable_list forward(AutogradContext* ctx, Tensor x, Tensor desc, std::vector<int64_t> oshape, int64_t ns,
                               bool nhwc, bool want_gap, bool need_grad) {
    c10::DeviceGuard guard(x.device());
    const nfp_desc* d = desc_of(desc);
    const auto f32 = x.options().dtype(torch::kFloat32).memory_format(at::MemoryFormat::Contiguous);
    Tensor gap = torch::empty({want_gap ? oshape[0] : 0, x.size(1)}, f32), nfpm = torch::empty({oshape[0], oshape[1]}, f32);
    Tensor omap = need_grad ? torch::empty(oshape, x.options().memory_format(at::MemoryFormat::Contiguous))
                            : torch::empty({0}, x.options());
    Tensor saved = torch::empty({ns}, f32);
    check(nfp_pool_forward(d, x.data_ptr(), want_gap ? gap.data_ptr<float>() : nullptr, nfpm.data_ptr<float>(),
                           need_grad ? omap.data_ptr() : nullptr, ns > 0 ? saved.data_ptr<float>() : nullptr, stream_of(x)));
    ctx->save_for_backward({x, omap, saved, desc});
    ctx->saved_data["nhwc"] = nhwc;
    ctx->saved_data["want_gap"] = want_gap;
    return {gap, nfpm};
  }
  static variable_list backward(AutogradContext* ctx, variable_list grads) {
    const auto sv = ctx->get_saved_variables();
    const Tensor &x = sv[0], &omap = sv[1], &saved = sv[2], &desc = sv[3];
    c10::DeviceGuard guard(x.device());
    const auto f32 = x.options().dtype(torch::kFloat32);
    // a pooled output that took no part in the loss arrives undefined: its gradient is zero
    // (grad_gap = NULL: GAP(x) was not produced, or took no part in the loss — no adjoint of the mean, include/nfp.h)
    const bool has_gap = ctx->saved_data["want_gap"].toBool() && grads[0].defined();
    Tensor ggap = has_gap ? grads[0].contiguous().to(torch::kFloat32) : Tensor();
    Tensor gnfp = grads[1].defined() ? grads[1].contiguous().to(torch::kFloat32)
                                     : torch::zeros({omap.size(0), omap.size(1)}, f32);
    Tensor gx = empty_like_layout(x, ctx->saved_data["nhwc"].toBool());
    check(nfp_pool_backward(desc_of(desc), x.data_ptr(), has_gap ? ggap.data_ptr<float>() : nullptr, gnfp.data_ptr<float>(),
                            omap.data_ptr(), saved.numel() ? saved.data_ptr<float>() : nullptr, gx.data_ptr(), stream_of(x)));
    return {gx, Tensor(), Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor nfp_apply(Tensor x, Tensor desc, std::vector<int64_t> oshape, int64_t ns, bool nhwc) {
  return NfpNode::apply(x, desc, oshape, ns, nhwc);
}
std::vector<Tensor> nfp_pool_apply(Tensor x, Tensor desc, std::vector<int64_t> oshape, int64_t ns, bool nhwc, bool want_gap,
                                   bool need_grad) {
  return NfpPoolNode::apply(x, desc, oshape, ns, nhwc, want_gap, need_grad);
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("nfp_apply", &nfp_apply, "NFPPooling.forward as one autograd node (nfp_forward / nfp_backward)");
  m.def("nfp_pool_apply", &nfp_pool_apply, "the fused nfp_pooling tail as one autograd node (nfp_pool_forward / _backward)");
  m.attr("desc_bytes") = (int64_t)sizeof(nfp_desc);
}
