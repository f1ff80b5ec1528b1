#include <torch/extension.h>
using torch::autograd::AutogradContext; using torch::autograd::variable_list;
struct Triv : public torch::autograd::Function<Triv> {
  static variable_list forward(AutogradContext* ctx, const at::Tensor& x) {
    ctx->save_for_backward({x});
    return {at::empty_like(x)};
  }
  static variable_list backward(AutogradContext* ctx, variable_list g) {
    auto s = ctx->get_saved_variables();
    return {at::empty_like(s[0])};
  }
};
at::Tensor triv(const at::Tensor& x) { return Triv::apply(x)[0]; }
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("triv", &triv); }
