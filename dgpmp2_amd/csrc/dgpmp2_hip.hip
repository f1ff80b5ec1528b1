// dgpmp2_hip.hip -- the C-ABI of include/dgpmp2_hip.h on top of the HIP kernels (gfx950 / CDNA4).
//
// One kernel launch == one batched Gauss-Newton step (PlanLayer.forward, plan_layer.py:87-99): factor evaluation,
// assembly of the block-tridiagonal normal equations and the per-trajectory block solve are fused; see gn_lane.h for
// the per-lane program, gn_backward.h for its adjoint and DESIGN.md for the mapping and the roofline.
#include "gn_device.h"

namespace {

using dgp_host::fail;

// a call that fails validation launches nothing: a pending dgp_time_next_launch request must not attach to an unrelated later launch
int drop_events(int rc) {
  dgp_host::LaunchEvents& le = dgp_host::launch_events();
  le.start = le.stop = nullptr;
  return rc;
}

// status of a launch.  kTiledTooLong is this file's PRIVATE code for "tiled grid beyond what the tiled translation units hold" (see launch()): not a value the HIP
// runtime returns, so a genuine hipErrorNotSupported of a launch is reported as what it is
static const hipError_t kTiledTooLong = (hipError_t)0x7f5d0001;
int hip_rc(hipError_t e, const char* what) {
  if (e == hipSuccess) return DGP_OK;
  if (e == kTiledTooLong)
    return fail(DGP_EUNSUPPORTED, "%s: tiled grids (DGP_SDF_TILED4) are implemented for num_states <= %d (launch shapes (16,4) and (32,4))", what, dgp_host::kMaxStatesTiled);
  return fail(DGP_EHIP, "%s launch failed: %s", what, hipGetErrorString(e));
}

hipError_t launch(const DgpHandle* h, int mode, const dgp::GnParams& p, const dgp::GnGradParams* g, hipStream_t s) {
  const bool tiled = p.sdf_layout != 0 && p.sdf != nullptr;
  if (tiled && p.n > dgp_host::kMaxStatesTiled) return kTiledTooLong;
  if (mode == dgp_host::kModeStepErrs) {          // dgp_gn_step_errors as ONE launch: the step kernels with the errors epilogue (gn_inst.hip with -DDGP_STEP_ERRS=1; host-checked: available)
    static const DgpLaunchFn etab[2][2][3] = {{{dgp_launch_2e_f32_g0, dgp_launch_2e_f32_g3, dgp_launch_2e_f32_g1}, {dgp_launch_2e_f64_g0, dgp_launch_2e_f64_g3, dgp_launch_2e_f64_g1}},
                                             {{dgp_launch_3e_f32_g0, dgp_launch_3e_f32_g3, nullptr}, {dgp_launch_3e_f64_g0, dgp_launch_3e_f64_g3, nullptr}}};
    const int grp = dgp_dev::launch_group(dgp::MODE_STEP, p);
    if (tiled || (grp != dgp_dev::GROUP_STATIC && grp != dgp_dev::GROUP_KRON && grp != dgp_dev::GROUP_GENERIC)) return hipErrorInvalidValue;
    if (grp == dgp_dev::GROUP_GENERIC && h->cfg.dof != 2) return hipErrorInvalidValue;      // (host-checked: dgp_host::gn_step_errors)
    const DgpShape she = dgp_host::choose_shape(h, p.B, dgp_host::shape_family(dgp::MODE_STEP, p), /*four states per lane=*/true);
    return etab[h->cfg.dof - 2][h->cfg.io_dtype == DGP_F64 ? 1 : 0][grp == dgp_dev::GROUP_KRON ? 1 : (grp == dgp_dev::GROUP_GENERIC ? 2 : 0)](she, dgp::MODE_STEP, p, g, s);
  }
  if (dgp_host::is_long(p.n)) return dgp_launch_long(h->cfg.dof, h->cfg.io_dtype == DGP_F64, mode, p, g, s);      // n > 256: gn_long.h
  const DgpShape sh = dgp_host::choose_shape(h, p.B, dgp_host::shape_family(mode, p), tiled);
  // [tiled][dof - 2][io dtype][kernel group] -> the translation unit that holds the kernel (gn_inst.hip; the tiled units: the same kernels compiled with -DDGP_TL=1)
  static const DgpLaunchFn table[2][2][2][dgp_dev::NUM_GROUPS] = {
      {{{dgp_launch_2_f32_g0, dgp_launch_2_f32_g1, dgp_launch_2_f32_g2, dgp_launch_2_f32_g3, dgp_launch_2_f32_g4},
        {dgp_launch_2_f64_g0, dgp_launch_2_f64_g1, dgp_launch_2_f64_g2, dgp_launch_2_f64_g3, dgp_launch_2_f64_g4}},
       {{dgp_launch_3_f32_g0, dgp_launch_3_f32_g1, dgp_launch_3_f32_g2, dgp_launch_3_f32_g3, dgp_launch_3_f32_g4},
        {dgp_launch_3_f64_g0, dgp_launch_3_f64_g1, dgp_launch_3_f64_g2, dgp_launch_3_f64_g3, dgp_launch_3_f64_g4}}},
      {{{dgp_launch_2t_f32_g0, dgp_launch_2t_f32_g1, dgp_launch_2t_f32_g2, dgp_launch_2t_f32_g3, dgp_launch_2t_f32_g4},
        {dgp_launch_2t_f64_g0, dgp_launch_2t_f64_g1, dgp_launch_2t_f64_g2, dgp_launch_2t_f64_g3, dgp_launch_2t_f64_g4}},
       {{dgp_launch_3t_f32_g0, dgp_launch_3t_f32_g1, dgp_launch_3t_f32_g2, dgp_launch_3t_f32_g3, dgp_launch_3t_f32_g4},
        {dgp_launch_3t_f64_g0, dgp_launch_3t_f64_g1, dgp_launch_3t_f64_g2, dgp_launch_3t_f64_g3, dgp_launch_3t_f64_g4}}}};
  const int f64 = h->cfg.io_dtype == DGP_F64 ? 1 : 0;
  return table[tiled ? 1 : 0][h->cfg.dof - 2][f64][dgp_dev::launch_group(mode, p)](sh, mode, p, g, s);
}

// dgp_sum_partial_grids: out[e] = scale * sum_c partial[c][e] -- the partial copies of a shared grid's gradient (dgp_gn_step_backward's g_sdf_copies) summed and cast in
// ONE pass at memory speed (torch's sum(0) of a (16, H W) tensor picks a reduction that takes 45 us for 8 MB on MI355X).  One element per lane and trip, the
// copies walked by every lane: consecutive lanes read consecutive elements of each copy.
template <typename TI, typename TO>
__global__ void __launch_bounds__(256) sum_partial_grids_kernel(const TI* __restrict__ in, int copies, int64_t elems, double scale, TO* __restrict__ out) {
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < elems; e += (int64_t)gridDim.x * 256) {
    double acc = 0.0;
    int c = 0;
    for (; c + 4 <= copies; c += 4) {                  // four independent loads in flight per lane (a serial chain over 16 copies ran at 1.5 TB/s)
      const double v0 = (double)in[(int64_t)c * elems + e], v1 = (double)in[(int64_t)(c + 1) * elems + e];
      const double v2 = (double)in[(int64_t)(c + 2) * elems + e], v3 = (double)in[(int64_t)(c + 3) * elems + e];
      acc += (v0 + v1) + (v2 + v3);
    }
    for (; c < copies; ++c) acc += (double)in[(int64_t)c * elems + e];
    out[e] = (TO)(acc * scale);
  }
}

// dgp_square_covariances[_backward]: one element of the module's output vector per lane and trip (rows of `width` values; consecutive lanes = consecutive columns)
template <typename T>
__global__ void __launch_bounds__(256) square_covs_kernel(const T* __restrict__ raw, int B, int W, int n_gp, int n, int learn_eps, int dof, T* __restrict__ s, T* __restrict__ blk,
                                                          T* __restrict__ ow, T* __restrict__ ep) {
  const int64_t total = (int64_t)B * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / W;
    const int c = (int)(e - b * W);
    const T v = raw[e];
    const T q = v * v;                                 // in the I/O type, as torch's v * v
    if (c < n_gp) {
      if (s) s[b * n_gp + c] = q;
      if (blk) {
        T* o = blk + (b * n_gp + c) * dof * dof;
        for (int i = 0; i < dof * dof; ++i) o[i] = (i % (dof + 1) == 0) ? q : (T)0;      // q * I, the 0 / 1 entries of the identity as exact values
      }
    } else if (c < n_gp + n) {
      if (ow) ow[b * n + (c - n_gp)] = q;
    } else if (learn_eps && c < n_gp + 2 * n) {
      if (ep) ep[b * n + (c - n_gp - n)] = q;
    }
  }
}
template <typename T>
__global__ void __launch_bounds__(256) square_covs_backward_kernel(const T* __restrict__ raw, int B, int W, int n_gp, int n, int learn_eps, int dof, const T* __restrict__ g_blk,
                                                                   const T* __restrict__ g_ow, const T* __restrict__ g_ep, T* __restrict__ g_raw) {
  const int64_t total = (int64_t)B * W;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t b = e / W;
    const int c = (int)(e - b * W);
    T g = (T)0;
    if (c < n_gp) {
      if (g_blk) {
        const T* o = g_blk + (b * n_gp + c) * dof * dof;
        for (int i = 0; i < dof; ++i) g += o[i * (dof + 1)];      // d (q I) / d q = I: the trace of the block's gradient
      }
    } else if (c < n_gp + n) {
      if (g_ow) g = g_ow[b * n + (c - n_gp)];
    } else if (learn_eps && c < n_gp + 2 * n) {
      if (g_ep) g = g_ep[b * n + (c - n_gp - n)];
    }
    g_raw[e] = (T)2 * raw[e] * g;
  }
}

int square_covs_check(const void* raw, int32_t dtype, int32_t batch, int32_t width, int32_t n_gp, int32_t n, int32_t learn_eps, int32_t dof) {
  if (!raw) return fail(DGP_EINVAL, "dgp_square_covariances: null raw");
  if (dtype != DGP_F32 && dtype != DGP_F64) return fail(DGP_EINVAL, "dgp_square_covariances: dtype %d", dtype);
  if (batch < 1 || n < 2 || (n_gp != 0 && n_gp != n - 1) || (dof != 2 && dof != 3)) return fail(DGP_EINVAL, "dgp_square_covariances: batch >= 1, num_states >= 2, n_gp in {0, num_states - 1}, dof in {2, 3}");
  if (width < n_gp + n * (learn_eps ? 2 : 1)) return fail(DGP_EINVAL, "dgp_square_covariances: width %d is less than the %d values the mode consumes", width, n_gp + n * (learn_eps ? 2 : 1));
  return DGP_OK;
}

}  // namespace

extern "C" {

int dgp_square_covariances(const void* raw, int32_t dtype, int32_t batch, int32_t width, int32_t n_gp, int32_t num_states, int32_t learn_eps, int32_t dof,
                           void* sq_scalars, void* sq_qc_inv, void* sq_obs_w, void* sq_eps, void* stream) {
  int rc = square_covs_check(raw, dtype, batch, width, n_gp, num_states, learn_eps, dof);
  if (rc != DGP_OK) return rc;
  const int64_t total = (int64_t)batch * width, blocks64 = (total + 255) / 256;
  const dim3 grid((unsigned)(blocks64 > 4096 ? 4096 : blocks64)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DGP_F64) hipLaunchKernelGGL((square_covs_kernel<double>), grid, block, 0, s, (const double*)raw, batch, width, n_gp, num_states, learn_eps, dof, (double*)sq_scalars,
                                           (double*)sq_qc_inv, (double*)sq_obs_w, (double*)sq_eps);
  else hipLaunchKernelGGL((square_covs_kernel<float>), grid, block, 0, s, (const float*)raw, batch, width, n_gp, num_states, learn_eps, dof, (float*)sq_scalars, (float*)sq_qc_inv,
                          (float*)sq_obs_w, (float*)sq_eps);
  const hipError_t e = hipGetLastError();
  return hip_rc(e, "dgp_square_covariances");
}

int dgp_square_covariances_backward(const void* raw, int32_t dtype, int32_t batch, int32_t width, int32_t n_gp, int32_t num_states, int32_t learn_eps, int32_t dof,
                                    const void* g_qc_inv, const void* g_obs_w, const void* g_eps, void* g_raw, void* stream) {
  int rc = square_covs_check(raw, dtype, batch, width, n_gp, num_states, learn_eps, dof);
  if (rc != DGP_OK) return rc;
  if (!g_raw) return fail(DGP_EINVAL, "dgp_square_covariances_backward: null g_raw");
  const int64_t total = (int64_t)batch * width, blocks64 = (total + 255) / 256;
  const dim3 grid((unsigned)(blocks64 > 4096 ? 4096 : blocks64)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == DGP_F64) hipLaunchKernelGGL((square_covs_backward_kernel<double>), grid, block, 0, s, (const double*)raw, batch, width, n_gp, num_states, learn_eps, dof,
                                           (const double*)g_qc_inv, (const double*)g_obs_w, (const double*)g_eps, (double*)g_raw);
  else hipLaunchKernelGGL((square_covs_backward_kernel<float>), grid, block, 0, s, (const float*)raw, batch, width, n_gp, num_states, learn_eps, dof, (const float*)g_qc_inv,
                          (const float*)g_obs_w, (const float*)g_eps, (float*)g_raw);
  const hipError_t e = hipGetLastError();
  return hip_rc(e, "dgp_square_covariances_backward");
}

int dgp_sum_partial_grids(const void* partial, int32_t partial_dtype, int32_t copies, int64_t elems, double scale, void* out, int32_t out_dtype, void* stream) {
  if (!partial || !out) return fail(DGP_EINVAL, "dgp_sum_partial_grids: null partial or out");
  if (copies < 1 || copies > 64 || elems < 1) return fail(DGP_EINVAL, "dgp_sum_partial_grids: copies must be in 1..64 and elems positive");
  if ((partial_dtype != DGP_F32 && partial_dtype != DGP_F64) || (out_dtype != DGP_F32 && out_dtype != DGP_F64)) return fail(DGP_EINVAL, "dgp_sum_partial_grids: dtypes must be DGP_F32 / DGP_F64");
  const int64_t blocks64 = (elems + 255) / 256;
  const dim3 grid((unsigned)(blocks64 > 4096 ? 4096 : blocks64)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (partial_dtype == DGP_F64) {
    if (out_dtype == DGP_F64) hipLaunchKernelGGL((sum_partial_grids_kernel<double, double>), grid, block, 0, s, (const double*)partial, copies, elems, scale, (double*)out);
    else hipLaunchKernelGGL((sum_partial_grids_kernel<double, float>), grid, block, 0, s, (const double*)partial, copies, elems, scale, (float*)out);
  } else {
    if (out_dtype == DGP_F64) hipLaunchKernelGGL((sum_partial_grids_kernel<float, double>), grid, block, 0, s, (const float*)partial, copies, elems, scale, (double*)out);
    else hipLaunchKernelGGL((sum_partial_grids_kernel<float, float>), grid, block, 0, s, (const float*)partial, copies, elems, scale, (float*)out);
  }
  const hipError_t e = hipGetLastError();
  return hip_rc(e, "dgp_sum_partial_grids");
}

int dgp_abi_version(void) { return DGP_ABI_VERSION; }
const char* dgp_last_error(void) { return dgp_host::err_buf(); }
int dgp_create(const DgpConfig* cfg, DgpHandle** out) { return dgp_host::create(cfg, out); }
void dgp_destroy(DgpHandle* h) { delete h; }
int dgp_num_factor_rows(const DgpHandle* h) { return h ? h->M : fail(DGP_EINVAL, "null handle"); }

int dgp_time_next_launch(void* start_event, void* stop_event) {
  dgp_host::LaunchEvents& le = dgp_host::launch_events();
  le.start = (start_event && stop_event) ? start_event : nullptr;
  le.stop = (start_event && stop_event) ? stop_event : nullptr;
  return DGP_OK;
}

int dgp_launch_shape(const DgpHandle* h, int32_t batch, int32_t* lpt, int32_t* c) {
  if (!h || batch <= 0) return fail(DGP_EINVAL, "null handle or non-positive batch");
  const DgpShape sh = dgp_host::choose_shape(h, batch, h->base.qc_diag == 0 ? dgp_host::FAM_GENERAL : dgp_host::FAM_STATIC);      // (static covariances; per-call q_full tensors may pick another shape)
  if (lpt) *lpt = sh.lpt;
  if (c) *c = sh.c;
  return DGP_OK;
}

int dgp_step_kernel_variant(const DgpHandle* h, int32_t batch) {
  if (!h || batch <= 0) return fail(DGP_EINVAL, "null handle or non-positive batch");
  return dgp_host::step_kernel_variant(h, batch);
}

int dgp_gn_step(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                const DgpCovs* covs, void* dtheta, void* err, void* err_ext, int32_t* info, void* stream) {
  dgp::GnParams p;
  int rc = dgp_host::fill_step(h, batch, th, start, goal, sdf, covs, dtheta, err, err_ext, info, p);
  if (rc != DGP_OK) return drop_events(rc);
  hipError_t e = launch(h, dgp::MODE_STEP, p, nullptr, (hipStream_t)stream);
  return hip_rc(e, "dgp_gn_step");
}

int dgp_gn_solve(const DgpHandle* h, int32_t batch, const void* th_init, const void* start, const void* goal, const DgpSdf* sdf,
                 const DgpCovs* covs, int32_t max_iters, double tol_delta, void* th_out, int32_t* iters, void* err_hist,
                 void* errext_hist, void* err_final, int32_t* info, void* stream) {
  dgp::GnParams p;
  int rc = dgp_host::fill_solve(h, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters, err_hist,
                                errext_hist, err_final, info, p);
  if (rc != DGP_OK) return drop_events(rc);
  hipError_t e = launch(h, dgp::MODE_SOLVE, p, nullptr, (hipStream_t)stream);
  return hip_rc(e, "dgp_gn_solve");
}

int dgp_eval_errors(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                    const DgpCovs* covs, void* err, void* err_ext, void* unw_sg, void* unw_gp, void* unw_obs, void* stream) {
  dgp::GnParams p;
  int rc = dgp_host::fill_eval(h, batch, th, start, goal, sdf, covs, err, err_ext, unw_sg, unw_gp, unw_obs, p);
  if (rc != DGP_OK) return drop_events(rc);
  hipError_t e = launch(h, dgp::MODE_EVAL, p, nullptr, (hipStream_t)stream);
  return hip_rc(e, "dgp_eval_errors");
}

int dgp_gn_step_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                         const DgpCovs* covs, const void* dtheta, const void* g_dtheta, const void* g_err_ext, void* g_th,
                         void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void* g_qc_inv,
                         void* g_obs_w, void* g_eps, void* stream) {
  dgp::GnParams p;
  dgp::GnGradParams g;
  int rc = dgp_host::fill_backward(h, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_th, g_start, g_goal, g_sdf,
                                   g_sdf_batch_stride, g_sdf_copies, g_qc_inv, g_obs_w, g_eps, p, g);
  if (rc != DGP_OK) return drop_events(rc);
  hipError_t e = launch(h, dgp_dev::MODE_BACKWARD, p, &g, (hipStream_t)stream);
  return hip_rc(e, "dgp_gn_step_backward");
}

int dgp_eval_errors_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                             const DgpCovs* covs, const void* g_err_ext, const void* g_unw_sg, const void* g_unw_gp, const void* g_unw_obs,
                             void* g_th, void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies,
                             void* g_eps, void* stream) {
  dgp::GnParams p;
  dgp::GnGradParams g;
  int rc = dgp_host::fill_eval_backward(h, batch, th, start, goal, sdf, covs, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th, g_start, g_goal,
                                        g_sdf, g_sdf_batch_stride, g_sdf_copies, g_eps, p, g);
  if (rc != DGP_OK) return drop_events(rc);
  hipError_t e = launch(h, dgp_dev::MODE_BACKWARD, p, &g, (hipStream_t)stream);
  return hip_rc(e, "dgp_eval_errors_backward");
}

// The round-4 entry points: their host logic (validation, the two launches behind one call) is shared with the test emulator, dgp_host.h
namespace {
struct HipLaunch {
  const DgpHandle* h; hipStream_t s; const char* what;
  int operator()(int mode, const dgp::GnParams& p, const dgp::GnGradParams* g) const {
    return hip_rc(launch(h, mode, p, g, s), what);
  }
};
}  // namespace

int dgp_gn_solve_traced(const DgpHandle* h, int32_t batch, const void* th_init, const void* start, const void* goal, const DgpSdf* sdf,
                        const DgpCovs* covs, int32_t max_iters, double tol_delta, void* th_out, int32_t* iters, void* err_hist,
                        void* errext_hist, void* err_final, int32_t* info, double* th_hist, void* stream) {
  int rc = dgp_host::gn_solve_traced(h, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters, err_hist, errext_hist, err_final, info,
                                     th_hist, HipLaunch{h, (hipStream_t)stream, "dgp_gn_solve_traced"});
  return rc == DGP_OK ? rc : drop_events(rc);
}

int dgp_gn_solve_backward(const DgpHandle* h, int32_t batch, const void* start, const void* goal, const DgpSdf* sdf, int32_t max_iters,
                          const double* th_hist, const void* th_out, const int32_t* iters, const void* g_th_out, void* g_th_init, void* g_start,
                          void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void* stream) {
  int rc = dgp_host::gn_solve_backward(h, batch, start, goal, sdf, max_iters, th_hist, th_out, iters, g_th_out, g_th_init, g_start, g_goal, g_sdf,
                                       g_sdf_batch_stride, g_sdf_copies, HipLaunch{h, (hipStream_t)stream, "dgp_gn_solve_backward"});
  return rc == DGP_OK ? rc : drop_events(rc);
}

int dgp_gn_step_errors(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                       const DgpCovs* covs, void* dtheta, void* err, void* err_ext, int32_t* info, void* unw_sg, void* unw_gp, void* unw_obs,
                       void* stream) {
  int rc = dgp_host::gn_step_errors(h, batch, th, start, goal, sdf, covs, dtheta, err, err_ext, info, unw_sg, unw_gp, unw_obs,
                                    HipLaunch{h, (hipStream_t)stream, "dgp_gn_step_errors"});
  return rc == DGP_OK ? rc : drop_events(rc);
}

int dgp_gn_step_errors_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                                const DgpCovs* covs, const void* dtheta, const void* g_dtheta, const void* g_err_ext, const void* g_unw_sg,
                                const void* g_unw_gp, const void* g_unw_obs, void* g_th, void* g_start, void* g_goal, void* g_sdf,
                                int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void* g_qc_inv, void* g_obs_w, void* g_eps, void* workspace,
                                void* stream) {
  int rc = dgp_host::gn_step_errors_backward(h, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th, g_start,
                                             g_goal, g_sdf, g_sdf_batch_stride, g_sdf_copies, g_qc_inv, g_obs_w, g_eps, workspace,
                                             HipLaunch{h, (hipStream_t)stream, "dgp_gn_step_errors_backward"});
  return rc == DGP_OK ? rc : drop_events(rc);
}

// Event helpers for dgp_time_next_launch: created / read through the HIP runtime THIS library is linked against (an event made by
// another copy of the runtime in the process is not portable to it).
int dgp_event_create(void** out) {
  if (!out) return fail(DGP_EINVAL, "null argument");
  hipEvent_t ev;
  hipError_t e = hipEventCreate(&ev);
  if (e != hipSuccess) return fail(DGP_EHIP, "hipEventCreate failed: %s", hipGetErrorString(e));
  *out = (void*)ev;
  return DGP_OK;
}
void dgp_event_destroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }
int dgp_event_elapsed_ms(void* start_event, void* stop_event, float* ms) {
  if (!start_event || !stop_event || !ms) return fail(DGP_EINVAL, "null argument");
  hipError_t e = hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event);
  if (e != hipSuccess) return fail(DGP_EHIP, "hipEventElapsedTime failed (synchronise first): %s", hipGetErrorString(e));
  return DGP_OK;
}

}  // extern "C"
