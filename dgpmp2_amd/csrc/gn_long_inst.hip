// gn_long_inst.hip -- the long-trajectory kernels (gn_long.h, n > 256) of every (dof, io dtype) and their launcher.
#include "gn_device.h"
#include <mutex>
#include <utility>
#include <vector>

namespace {

// more than 64 KB of dynamic LDS needs the attribute (gfx950: up to 160 KB per workgroup).  Raised ONCE per kernel (and device) to the largest
// block any trajectory length needs, not per launch: hipFuncSetAttribute is host time on every call and not a call to make under stream capture.
constexpr int kLongLdsMax = 160 * 1024;
template <typename K>
hipError_t ensure_lds_limit(K kernel) {
  static std::mutex mu;                               // (K is one function TYPE per signature, not per kernel: keyed on the kernel's address)
  static std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lk(mu);
  for (const auto& d : done) if (d.first == (const void*)kernel && d.second == dev) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLongLdsMax);
  if (e == hipSuccess) done.emplace_back((const void*)kernel, dev);
  return e;
}

template <typename K, typename... A>
hipError_t launch_dyn(K kernel, int lds_bytes, int B, hipStream_t s, const A&... args) {
  hipError_t e = ensure_lds_limit(kernel);
  if (e != hipSuccess) return e;
  dgp_host::LaunchEvents& le = dgp_host::launch_events();
  const hipEvent_t ev0 = (hipEvent_t)le.start, ev1 = (hipEvent_t)le.stop;
  le.start = le.stop = nullptr;
  if (ev0 && ev1) hipExtLaunchKernelGGL(kernel, dim3((unsigned)B), dim3(64), lds_bytes, s, ev0, ev1, 0, args...);
  else hipLaunchKernelGGL(kernel, dim3((unsigned)B), dim3(64), lds_bytes, s, args...);
  return hipGetLastError();
}

template <int DOF, typename IO>
hipError_t launch_long_typed(int mode, const dgp::GnParams& p, const dgp::GnGradParams* g, hipStream_t s) {
  const int lds = dgp::long_lds_bytes<2 * DOF>(p.n);
  if (mode == dgp::MODE_STEP) return launch_dyn(dgp_dev::gn_long_kernel<DOF, IO, dgp::MODE_STEP>, lds, p.B, s, p);
  if (mode == dgp::MODE_SOLVE) return launch_dyn(dgp_dev::gn_long_kernel<DOF, IO, dgp::MODE_SOLVE>, lds, p.B, s, p);
  if (mode == dgp::MODE_EVAL) return launch_dyn(dgp_dev::gn_long_kernel<DOF, IO, dgp::MODE_EVAL>, lds, p.B, s, p);
  if (mode == dgp_dev::MODE_BACKWARD && g) return launch_dyn(dgp_dev::gn_long_backward_kernel<DOF, IO>, lds, p.B, s, p, *g);
  return hipErrorInvalidValue;
}

}  // namespace

hipError_t dgp_launch_long(int dof, bool f64, int mode, const dgp::GnParams& p, const dgp::GnGradParams* g, hipStream_t s) {
  if (dof == 2) return f64 ? launch_long_typed<2, double>(mode, p, g, s) : launch_long_typed<2, float>(mode, p, g, s);
  return f64 ? launch_long_typed<3, double>(mode, p, g, s) : launch_long_typed<3, float>(mode, p, g, s);
}
