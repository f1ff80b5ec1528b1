// dgpmp2_hip.hip -- HIP kernels (gfx950 / CDNA4) and the C-ABI of include/dgpmp2_hip.h.
//
// One kernel launch == one batched Gauss-Newton step (PlanLayer.forward, plan_layer.py:87-99): factor
// evaluation, assembly of the block-tridiagonal normal equations and the per-trajectory block PCR solve are
// fused; see gn_lane.h for the per-lane program and DESIGN.md for the mapping and the roofline.
#include <hip/hip_runtime.h>

#define DGP_HD __host__ __device__ __forceinline__
#include "dgp_host.h"

namespace {

using dgp_host::fail;

// Device lane context: cross-lane fetches are ds_bpermute (any lane -> any lane inside the wavefront,
// no LDS memory is touched).
struct DevCtx {
  __device__ __forceinline__ int lane() const { return (int)(threadIdx.x & 63u); }
  __device__ __forceinline__ int wave() const { return (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)); }
  __device__ __forceinline__ int fetch_i(int v, int src) const { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
  __device__ __forceinline__ double fetch(double v, int src) const {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src << 2, lo);
    hi = __builtin_amdgcn_ds_bpermute(src << 2, hi);
    return __hiloint2double(hi, lo);
  }
  __device__ __forceinline__ bool any(bool pred) const { return __any(pred ? 1 : 0) != 0; }
  __device__ __forceinline__ void atomic_add(float* p, float v) const { atomicAdd(p, v); }
  __device__ __forceinline__ void atomic_add(double* p, double v) const { atomicAdd(p, v); }
};

template <int DOF, int LPT, int C, typename IO, int MODE>
__global__ void __launch_bounds__(64) gn_kernel(const dgp::GnParams p) {
  DevCtx cx;
  dgp::gn_lane_program<DOF, LPT, C, IO, MODE>(p, cx);
}

template <int DOF, int LPT, typename IO>
__global__ void __launch_bounds__(64) gn_backward_kernel(const dgp::GnParams p, const dgp::GnGradParams g) {
  DevCtx cx;
  dgp::gn_backward_lane_program<DOF, LPT, IO>(p, g, cx);
}

template <int DOF, int LPT, int C, typename IO>
hipError_t launch_mode(int mode, const dgp::GnParams& p, hipStream_t s) {
  constexpr int TPW = 64 / LPT;
  const unsigned grid = (unsigned)((p.B + TPW - 1) / TPW);
  switch (mode) {
    case dgp::MODE_STEP: hipLaunchKernelGGL((gn_kernel<DOF, LPT, C, IO, dgp::MODE_STEP>), dim3(grid), dim3(64), 0, s, p); break;
    case dgp::MODE_SOLVE: hipLaunchKernelGGL((gn_kernel<DOF, LPT, C, IO, dgp::MODE_SOLVE>), dim3(grid), dim3(64), 0, s, p); break;
    default: hipLaunchKernelGGL((gn_kernel<DOF, LPT, C, IO, dgp::MODE_EVAL>), dim3(grid), dim3(64), 0, s, p); break;
  }
  return hipGetLastError();
}

// every (LPT, C) of dgp_host::shape_supported
#define DGP_FOR_EACH_SHAPE(X) X(16, 1) X(32, 1) X(64, 1) X(16, 2) X(32, 2) X(64, 2) X(16, 4) X(32, 4) X(64, 4)

template <int DOF, typename IO>
hipError_t launch_shape(DgpShape sh, int mode, const dgp::GnParams& p, hipStream_t s) {
#define DGP_CASE(L, CC) if (sh.lpt == L && sh.c == CC) return launch_mode<DOF, L, CC, IO>(mode, p, s);
  DGP_FOR_EACH_SHAPE(DGP_CASE)
#undef DGP_CASE
  return hipErrorInvalidValue;
}

hipError_t launch(const DgpHandle* h, int mode, const dgp::GnParams& p, hipStream_t s) {
  const DgpShape sh = dgp_host::choose_shape(h, p.B);
  const bool f64 = h->cfg.io_dtype == DGP_F64;
  if (h->cfg.dof == 2) return f64 ? launch_shape<2, double>(sh, mode, p, s) : launch_shape<2, float>(sh, mode, p, s);
  return f64 ? launch_shape<3, double>(sh, mode, p, s) : launch_shape<3, float>(sh, mode, p, s);
}

template <int DOF, typename IO>
hipError_t launch_bwd_lpt(int lpt, const dgp::GnParams& p, const dgp::GnGradParams& g, hipStream_t s) {
  const int tpw = 64 / lpt;
  const unsigned grid = (unsigned)((p.B + tpw - 1) / tpw);
  switch (lpt) {
    case 16: hipLaunchKernelGGL((gn_backward_kernel<DOF, 16, IO>), dim3(grid), dim3(64), 0, s, p, g); break;
    case 32: hipLaunchKernelGGL((gn_backward_kernel<DOF, 32, IO>), dim3(grid), dim3(64), 0, s, p, g); break;
    default: hipLaunchKernelGGL((gn_backward_kernel<DOF, 64, IO>), dim3(grid), dim3(64), 0, s, p, g); break;
  }
  return hipGetLastError();
}

}  // namespace

extern "C" {

int dgp_abi_version(void) { return DGP_ABI_VERSION; }
const char* dgp_last_error(void) { return dgp_host::err_buf(); }
int dgp_create(const DgpConfig* cfg, DgpHandle** out) { return dgp_host::create(cfg, out); }
void dgp_destroy(DgpHandle* h) { delete h; }
int dgp_num_factor_rows(const DgpHandle* h) { return h ? h->M : fail(DGP_EINVAL, "null handle"); }

int dgp_gn_step(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                const DgpCovs* covs, void* dtheta, void* err, void* err_ext, int32_t* info, void* stream) {
  dgp::GnParams p;
  int rc = dgp_host::fill_step(h, batch, th, start, goal, sdf, covs, dtheta, err, err_ext, info, p);
  if (rc != DGP_OK) return rc;
  hipError_t e = launch(h, dgp::MODE_STEP, p, (hipStream_t)stream);
  if (e != hipSuccess) return fail(DGP_EHIP, "dgp_gn_step launch failed: %s", hipGetErrorString(e));
  return DGP_OK;
}

int dgp_gn_solve(const DgpHandle* h, int32_t batch, const void* th_init, const void* start, const void* goal, const DgpSdf* sdf,
                 const DgpCovs* covs, int32_t max_iters, double tol_delta, void* th_out, int32_t* iters, void* err_hist,
                 void* errext_hist, void* err_final, int32_t* info, void* stream) {
  dgp::GnParams p;
  int rc = dgp_host::fill_solve(h, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters, err_hist,
                                errext_hist, err_final, info, p);
  if (rc != DGP_OK) return rc;
  hipError_t e = launch(h, dgp::MODE_SOLVE, p, (hipStream_t)stream);
  if (e != hipSuccess) return fail(DGP_EHIP, "dgp_gn_solve launch failed: %s", hipGetErrorString(e));
  return DGP_OK;
}

int dgp_eval_errors(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                    const DgpCovs* covs, void* err, void* err_ext, void* unw_sg, void* unw_gp, void* unw_obs, void* stream) {
  dgp::GnParams p;
  int rc = dgp_host::fill_eval(h, batch, th, start, goal, sdf, covs, err, err_ext, unw_sg, unw_gp, unw_obs, p);
  if (rc != DGP_OK) return rc;
  hipError_t e = launch(h, dgp::MODE_EVAL, p, (hipStream_t)stream);
  if (e != hipSuccess) return fail(DGP_EHIP, "dgp_eval_errors launch failed: %s", hipGetErrorString(e));
  return DGP_OK;
}

int dgp_gn_step_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                         const DgpCovs* covs, const void* g_dtheta, const void* g_err_ext, void* g_th, void* g_start, void* g_goal,
                         void* g_sdf, int64_t g_sdf_batch_stride, void* g_qc_inv, void* g_obs_w, void* g_eps, void* stream) {
  dgp::GnParams p;
  dgp::GnGradParams g;
  int rc = dgp_host::fill_backward(h, batch, th, start, goal, sdf, covs, g_dtheta, g_err_ext, g_th, g_start, g_goal, g_sdf,
                                   g_sdf_batch_stride, g_qc_inv, g_obs_w, g_eps, p, g);
  if (rc != DGP_OK) return rc;
  if (!dgp::kBackwardImplemented) return fail(DGP_EUNSUPPORTED, "dgp_gn_step_backward is not implemented yet");
  const bool f64 = h->cfg.io_dtype == DGP_F64;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e;
  if (h->cfg.dof == 2) e = f64 ? launch_bwd_lpt<2, double>(64, p, g, s) : launch_bwd_lpt<2, float>(64, p, g, s);
  else e = f64 ? launch_bwd_lpt<3, double>(64, p, g, s) : launch_bwd_lpt<3, float>(64, p, g, s);
  if (e != hipSuccess) return fail(DGP_EHIP, "dgp_gn_step_backward launch failed: %s", hipGetErrorString(e));
  return DGP_OK;
}

}  // extern "C"
