// dgp_host.h -- host-side logic behind the C-ABI (no HIP in here): config validation, the constants of
// PlanLayer.__init__ (plan_layer.py:14-81) and per-call argument marshalling into dgp::GnParams.
// Shared by dgpmp2_hip.hip (the product library) and tests/emul (the CPU wavefront emulator used by tests).
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <math.h>
#include <new>
#include "gn_lane.h"
#include "gn_backward.h"
#include "gn_long.h"
#include "../../include/dgpmp2_hip.h"

struct DgpHandle {
  DgpConfig cfg;
  int d;                // state_dim
  int M;                // plan_layer.py:43-45
  int force_lpt, force_c;   // >0: launch shape pinned by the environment variable DGP_FORCE_SHAPE="LPT,C" (tuning aid)
  dgp::GnParams base;   // constants filled once
};

// Launch shape: LPT lanes per trajectory x C consecutive states per lane, LPT*C >= n.
struct DgpShape { int lpt, c; };

namespace dgp_host {

inline char* err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}

// dgp_time_next_launch: the event pair the calling thread's next launch records its begin / end on (one variable per thread
// across all translation units of the library)
struct LaunchEvents { void* start; void* stop; };
inline LaunchEvents& launch_events() {
  static thread_local LaunchEvents ev = {nullptr, nullptr};
  return ev;
}

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// Shapes the kernels are instantiated for (see DGP_FOR_EACH_SHAPE in dgpmp2_hip.hip): LPT in {16,32,64} x C in {1,2,4}.
constexpr int kMaxStates = 256;
// Longer trajectories run the loop kernels of gn_long.h (one trajectory per wavefront, ceil(n / 64) rows per lane), whose limit is the
// wavefront's LDS block of (rows per lane - 1) parked (S_k^-1, z_k) slots: 16 rows per lane for d = 4 (105 KB), 10 for d = 6 (127 KB = dgp::long_lds_bytes<6>(640)).
constexpr int kMaxStatesLong4 = 1024, kMaxStatesLong6 = 640;
inline int max_states(int dof) { return dof == 3 ? kMaxStatesLong6 : kMaxStatesLong4; }
inline bool is_long(int n) { return n > kMaxStates; }
inline bool shape_supported(int lpt, int c) { return (lpt == 16 || lpt == 32 || lpt == 64) && (c == 1 || c == 2 || c == 4); }

// Pick (LPT, C) for n states and a batch of B trajectories: the cheapest supported shape under a two-parameter timing model
// fitted to profiles/tools/shape_grid.py on MI355X (static-covariance kernels, steady clocks):
//   time(shape, B) ~ fixed + T[shape] * max(1, 1.25 * wavefronts / 1024)
// T = per-wavefront time in us when every wavefront has a SIMD to itself (B = 256 column of the grid); a launch with more
// wavefronts than the chip has SIMDs (1024) runs them in turns and each then costs ~1.25 T (shared LDS crossbar / L2).
// More states per lane (larger C) means less arithmetic per trajectory (the local elimination is O(C) per lane while
// every PCR round costs the same whatever LPT is) but fewer, longer wavefronts.  d = 6 has its own table (T6), every entry from the
// round-2 sweep of the d = 6 kernels (profiles/r02_shape_sweep.txt: kernel time / turns at B = 4096, n = 16, 32, 64).
// `general`: the launch runs the general-covariance kernels (q_full tensors or a non-diagonal static Q_c_inv; dgp::QK_GENERAL).  For d = 6 they
// have their own table: the sweep behind T6 timed the static (Woodbury) kernels only, and the general (16,4) kernel -- 1.4 KB of scratch per
// lane -- is the slowest way to run q_full at B = 4096 (profiles/r03_shape_checks.txt: (16,4) 84.0 us, (32,2) 64.8 us, (32,4) 175 us, (64,1)
// 117 us; per-state Kronecker kernels: (16,4) 47.1 vs (32,2) 46.4 us, block elimination with velocity limits: 30.3 vs 42.5 us -- T6 holds).
// family 2 (FAM_KRON_BWD): the backward kernel for per-state Q_c^-1 tensors, d = 6: its four-states-per-lane instantiation keeps the three S_k^-1
// of the adjoint solve AND the covariance blocks alive into the chain rule -- (16,4) 98.3 us against (32,2) 65.6 us at B = 4096 (the forward step
// is indifferent, 47.4 vs 46.4 us, and the fused loop prefers (16,4), 76.0 vs 83.2 us per iteration, so only the backward changes family).
enum { FAM_STATIC = 0, FAM_GENERAL = 1, FAM_KRON_BWD = 2 };
// `tiled` (DgpSdf::layout = DGP_SDF_TILED4, and the step kernels with the errors epilogue): the twin translation units hold the shapes (16,4) and (32,4) only
// (num_states <= 128, host-checked)
inline bool tiled_shape_ok(int lpt, int c) { return c == 4 && (lpt == 16 || lpt == 32); }
constexpr int kMaxStatesTiled = 128;
inline DgpShape choose_shape(const DgpHandle* h, int B, int family = FAM_STATIC, bool tiled = false) {
  const bool general = family == FAM_GENERAL;
  if (is_long(h->cfg.num_states)) return DgpShape{64, (h->cfg.num_states + 63) / 64};      // gn_long.h: rows per lane reported as C
  if (h->force_lpt && (!tiled || tiled_shape_ok(h->force_lpt, h->force_c))) return DgpShape{h->force_lpt, h->force_c};
  if (tiled) return DgpShape{h->cfg.num_states <= 64 ? 16 : 32, 4};
  static const double T4[3][3] = {{1.5, 2.4, 4.9}, {2.9, 4.3, 6.6}, {5.7, 7.4, 10.6}};      // [LPT 16,32,64][C 1,2,4], us, d = 4
  static const double T6[3][3] = {{11.0, 14.2, 24.4}, {13.4, 17.1, 26.6}, {19.9, 26.1, 33.9}};  // d = 6
  static const double T6G[3][3] = {{11.0, 14.2, 67.2}, {13.4, 25.9, 70.1}, {23.5, 26.1, 89.0}};  // d = 6, general kernels: the C = 4 / (32,2) / (64,1) entries from the round-3 check (time / turns), the rest as T6
  const int n = h->cfg.num_states;
  DgpShape best{64, 4};
  double best_cost = 1e300;
  for (int li = 0; li < 3; ++li)
    for (int ci = 0; ci < 3; ++ci) {
      const int lpt = 16 << li, c = 1 << ci;
      if (lpt * c < n) continue;
      double t = h->cfg.dof == 3 ? (general ? T6G[li][ci] : T6[li][ci]) : T4[li][ci];
      if (h->cfg.dof == 3 && family == FAM_KRON_BWD) t = (ci == 2) ? t * 3.2 : t * 1.53;      // (16,4): 98.3 / 1.25 = 78.6 = 3.2 x 24.4;  (32,2): 65.6 / 2.5 = 26.2 = 1.53 x 17.1
      const double waves = (double)((B + (64 / lpt) - 1) / (64 / lpt));
      const double turns = 1.25 * waves / 1024.0;
      const double cost = t * (turns > 1.0 ? turns : 1.0);
      if (cost < best_cost) { best_cost = cost; best = DgpShape{lpt, c}; }
    }
  return best;
}

// kernel family of a launch for the shape choice (mode: dgp::MODE_* or 3 = backward)
inline int shape_family(int mode, const dgp::GnParams& p) {
  if (mode == dgp::MODE_EVAL) return FAM_STATIC;
  const int qk = dgp::kernel_variant(p);
  if (qk == dgp::QK_GENERAL) return FAM_GENERAL;
  if (qk == dgp::QK_KRON && mode == 3) return FAM_KRON_BWD;
  return FAM_STATIC;
}

// dgp_step_kernel_variant: the dgp::QK_* variant a static-covariance step of this batch size launches (mirrors launch_typed)
inline int step_kernel_variant(const DgpHandle* h, int B) {
  dgp::GnParams p = h->base;
  p.qc_mode = dgp::QC_STATIC;
  const DgpShape sh = choose_shape(h, B);
  if (is_long(p.n) || !dgp::use_static_kernels(p)) return dgp::QK_GENERAL;
  if (!dgp::wb_applies(p, sh.lpt, sh.c)) return dgp::QK_STATIC;
  return p.n == sh.lpt * sh.c ? dgp::QK_WB : dgp::QK_WBR;
}

// The constant blocks of a GP factor under the configured (static) Q_c_inv: Q^-1 exactly as dgp::fixed_Qinv builds it,
// U = -Phi^T Q^-1 (block (i,i+1) of Lambda) and Phi^T Q^-1 Phi (the factor's share of block (i,i)); Phi = [[I, dt I],[0, I]].
inline void fill_static_blocks(dgp::GnParams& p, int dof) {
  const int d = 2 * dof;
  double Q[6][6], U[6][6];
  for (int i = 0; i < dof; ++i)
    for (int j = 0; j < dof; ++j) {
      const double c = p.qc_fix[i * dof + j];
      if (j >= i) { Q[i][j] = Q[j][i] = p.qa * c; Q[dof + i][dof + j] = Q[dof + j][dof + i] = p.qc_ * c; }
      Q[i][dof + j] = Q[dof + j][i] = p.qb * c;
    }
  for (int a = 0; a < dof; ++a)
    for (int c = 0; c < d; ++c) { U[a][c] = -Q[a][c]; U[dof + a][c] = -(p.dt * Q[a][c] + Q[dof + a][c]); }
  p.qc_diag = 1;
  for (int i = 0; i < dof; ++i)
    for (int j = 0; j < dof; ++j)
      if (i != j && p.qc_fix[i * dof + j] != 0.0) p.qc_diag = 0;
  auto sidx = [d](int i, int j) { return i * d - (i * (i - 1)) / 2 + (j - i); };      // Sym<d>::idx for i <= j
  for (int a = 0; a < d; ++a)
    for (int c = 0; c < d; ++c) {
      p.u_fix[a * d + c] = U[a][c];
      if (c >= a) {
        p.q_fix[sidx(a, c)] = Q[a][c];
        // Phi^T Q Phi = -U Phi : columns pos = -U[:,pos], columns vel = -(dt U[:,pos] + U[:,vel])
        p.a_fix[sidx(a, c)] = (c < dof) ? -U[a][c] : -(p.dt * U[a][c - dof] + U[a][c]);
      }
    }
}

inline int create(const DgpConfig* cfg, DgpHandle** out) {
  if (!cfg || !out) return fail(DGP_EINVAL, "null argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(DgpConfig))
    return fail(DGP_EINVAL, "DgpConfig size mismatch: caller %u, library %zu", cfg->struct_size, sizeof(DgpConfig));
  if (cfg->dof != 2 && cfg->dof != 3) return fail(DGP_EUNSUPPORTED, "dof must be 2 or 3, got %d", cfg->dof);
  if (cfg->nlinks != 1) return fail(DGP_EUNSUPPORTED, "only nlinks == 1 (point robots) is implemented, got %d", cfg->nlinks);
  if (cfg->num_states < 2) return fail(DGP_EINVAL, "num_states must be >= 2, got %d", cfg->num_states);
  if (cfg->num_states > max_states(cfg->dof))
    return fail(DGP_EUNSUPPORTED, "num_states > %d is not implemented for dof %d (LDS capacity of the long-trajectory kernels), got %d", max_states(cfg->dof),
                cfg->dof, cfg->num_states);
  if (cfg->io_dtype != DGP_F32 && cfg->io_dtype != DGP_F64) return fail(DGP_EINVAL, "bad io_dtype %d", cfg->io_dtype);
  if ((cfg->flags & DGP_FLAG_NONHOLONOMIC) && cfg->dof != 3)
    return fail(DGP_EINVAL, "the non-holonomic factor needs the (x,y,theta) robot, dof == 3");
  if (cfg->flags & ~(DGP_FLAG_NONHOLONOMIC | DGP_FLAG_VEL_LIMITS)) return fail(DGP_EINVAL, "unknown flag bits 0x%x", cfg->flags);
  if (!(cfg->total_time_sec > 0.0)) return fail(DGP_EINVAL, "total_time_sec must be positive");
  if (!(cfg->x_lims[1] > cfg->x_lims[0]) || !(cfg->y_lims[1] > cfg->y_lims[0])) return fail(DGP_EINVAL, "empty x/y limits");
  if (!(cfg->K_s > 0.0) || !(cfg->K_g > 0.0) || !(cfg->cost_sigma > 0.0)) return fail(DGP_EINVAL, "K_s, K_g, cost_sigma must be positive");
  if ((cfg->flags & DGP_FLAG_NONHOLONOMIC) && !(cfg->K_d > 0.0)) return fail(DGP_EINVAL, "K_d must be positive");
  if ((cfg->flags & DGP_FLAG_VEL_LIMITS) && !(cfg->K_v > 0.0)) return fail(DGP_EINVAL, "K_v must be positive");
  DgpHandle* h = new (std::nothrow) DgpHandle();
  if (!h) return fail(DGP_EINVAL, "out of host memory");
  h->cfg = *cfg;
  const int n = cfg->num_states, dof = cfg->dof;
  h->d = 2 * dof;
  h->force_lpt = h->force_c = 0;
  if (const char* fs = is_long(n) ? nullptr : getenv("DGP_FORCE_SHAPE")) {      // (long trajectories have one shape)
    int l = 0, c = 0;
    if (sscanf(fs, "%d,%d", &l, &c) == 2 && shape_supported(l, c) && l * c >= n) { h->force_lpt = l; h->force_c = c; }
    else { delete h; return fail(DGP_EINVAL, "DGP_FORCE_SHAPE=%s is not a supported LPT,C pair covering n=%d", fs, n); }
  }
  h->M = h->d * ((n - 1) + 2) + n * cfg->nlinks;                       // plan_layer.py:43
  if (cfg->flags & DGP_FLAG_NONHOLONOMIC) h->M += n;                    // :44
  if (cfg->flags & DGP_FLAG_VEL_LIMITS) h->M += dof * n;                // :45
  dgp::GnParams& p = h->base;
  memset(&p, 0, sizeof(p));
  p.n = n;
  p.flags = cfg->flags;
  p.dt = cfg->total_time_sec * 1.0 / (double)(n - 1) * 1.0;             // plan_layer.py:31
  p.qa = 12.0 * pow(p.dt, -3.0);                                        // gp_factor.py:66-68
  p.qb = -6.0 * pow(p.dt, -2.0);
  p.qc_ = 4.0 * pow(p.dt, -1.0);
  p.e10 = p.dt * p.qa + p.qb;                                           // E = Phi2^T T, F = E Phi2 (QK_KRON kernels, gn_lane.h)
  p.e11 = p.dt * p.qb + p.qc_;
  p.f11 = p.dt * p.e10 + p.e11;
  p.w_s = 1.0 / pow(cfg->K_s, 2.0);                                     // plan_layer.py:64-65
  p.w_g = 1.0 / pow(cfg->K_g, 2.0);
  p.reg = cfg->reg;
  p.radius = cfg->sphere_radius;
  p.eps_static = cfg->epsilon_dist;
  p.obs_w_fix = 1.0 / pow(cfg->cost_sigma, 2.0);                        // plan_layer.py:74
  for (int k = 0; k < 9; ++k) p.qc_fix[k] = cfg->Q_c_inv[k];
  p.w_d = (cfg->flags & DGP_FLAG_NONHOLONOMIC) ? 1.0 / pow(cfg->K_d, 2.0) : 0.0;
  p.w_v = (cfg->flags & DGP_FLAG_VEL_LIMITS) ? 1.0 / pow(cfg->K_v, 2.0) : 0.0;
  p.vmax[0] = cfg->v_x; p.vmax[1] = cfg->v_y;
  p.M = (double)h->M;
  p.inv_M = 1.0 / p.M;
  fill_static_blocks(p, dof);
  dgp::wb_fill_table(p, dof);                                           // constants of the Woodbury kernels (gn_woodbury.h); wb_ok says whether they apply
  if (const char* nw = getenv("DGP_NO_WOODBURY")) { if (nw[0] == '1') p.wb_ok = 0; }      // tuning / A-B aid: keep the block elimination
  *out = h;
  return DGP_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Fill the per-call part of the kernel arguments; returns DGP_OK or an error code.
inline int fill_call(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                     const DgpCovs* covs, dgp::GnParams& p, bool sdf_optional = false, bool scalar_qc_ok = false) {
  if (!h) return fail(DGP_EINVAL, "null handle");
  if (batch <= 0) return fail(DGP_EINVAL, "batch must be positive, got %d", batch);
  if (!th || !start || !goal) return fail(DGP_EINVAL, "th/start/goal must be non-null device pointers");
  static const DgpSdf no_sdf = {nullptr, 2, 2, 0, DGP_SDF_ROWMAJOR, DGP_GSDF_DENSE, nullptr};      // dgp_eval_errors without obstacle outputs: no grid is read
  if (sdf_optional && (!sdf || !sdf->data)) sdf = &no_sdf;
  else if (!sdf || !sdf->data) return fail(DGP_EINVAL, "sdf must be non-null");
  if (sdf->rows < 1 || sdf->cols < 1) return fail(DGP_EINVAL, "sdf grid must be at least 1x1, got %dx%d", sdf->rows, sdf->cols);
  if (sdf->cols < 2) return fail(DGP_EUNSUPPORTED, "sdf grids with a single column are not implemented (the taps are fetched as column pairs)");
  if (sdf->batch_stride < 0) return fail(DGP_EINVAL, "negative sdf batch stride");
  if (((int64_t)sdf->rows + 3) * ((int64_t)sdf->cols + 3) >= ((int64_t)1 << 31)) return fail(DGP_EUNSUPPORTED, "sdf grid of %dx%d elements is too large", sdf->rows, sdf->cols);
  if (sdf->layout != DGP_SDF_ROWMAJOR && sdf->layout != DGP_SDF_TILED4) return fail(DGP_EINVAL, "bad DgpSdf::layout %d", sdf->layout);
  p = h->base;
  p.B = batch;
  p.th = th; p.start = start; p.goal = goal;
  p.sdf = sdf->data; p.sdf_rows = sdf->rows; p.sdf_cols = sdf->cols; p.sdf_bstride = sdf->batch_stride; p.sdf_layout = sdf->layout;
  // obstacle_cost.py:34 and sdf_utils.py:57-58, evaluated exactly as Python does (fp64)
  p.res = (h->cfg.x_lims[1] - h->cfg.x_lims[0]) / (double)sdf->cols;
  p.inv_res = 1.0 / p.res;
  p.orig_px = (0. - h->cfg.x_lims[0] / p.res);
  p.orig_py = (0. - h->cfg.y_lims[0] / p.res);
  p.qc_mode = DGP_QC_STATIC; p.qc = nullptr; p.obs_w = nullptr; p.eps = nullptr;
  p.vec_mu = (aligned16(start) && aligned16(goal)) ? 1 : 0;
  if (covs) {
    if (covs->qc_mode < DGP_QC_STATIC || covs->qc_mode > DGP_QC_SCALAR) return fail(DGP_EINVAL, "bad qc_mode %d", covs->qc_mode);
    if (covs->qc_mode == DGP_QC_SCALAR && !scalar_qc_ok)
      return fail(DGP_EUNSUPPORTED, "qc_mode DGP_QC_SCALAR is implemented by dgp_gn_step[_errors] and their backward only (pass the (B,n-1,dof,dof) tensors elsewhere)");
    if (covs->qc_mode == DGP_QC_SCALAR && h->base.qc_diag == 0)
      return fail(DGP_EUNSUPPORTED, "qc_mode DGP_QC_SCALAR needs a diagonal Q_c_inv in the configuration");
    if ((covs->qc_mode == DGP_QC_STATIC) != (covs->qc_inv == nullptr))
      return fail(DGP_EINVAL, "qc_inv must be NULL iff qc_mode == DGP_QC_STATIC");
    p.qc_mode = covs->qc_mode; p.qc = covs->qc_inv; p.obs_w = covs->obs_w; p.eps = covs->eps;
    p.vec_qc = (covs->qc_inv && aligned16(covs->qc_inv)) ? 1 : 0;
  }
  return DGP_OK;
}

inline int fill_step(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                     const DgpCovs* covs, void* dtheta, void* err, void* err_ext, int32_t* info, dgp::GnParams& p) {
  int rc = fill_call(h, batch, th, start, goal, sdf, covs, p, /*sdf_optional=*/false, /*scalar_qc_ok=*/true);
  if (rc != DGP_OK) return rc;
  if (p.qc_mode == dgp::QC_SCALAR && is_long(p.n)) return fail(DGP_EUNSUPPORTED, "qc_mode DGP_QC_SCALAR is not implemented for num_states > 256");
  if (!dtheta) return fail(DGP_EINVAL, "dtheta must be non-null");
  p.dtheta = dtheta; p.err = err; p.err_ext = err_ext; p.info = info;
  p.vec_io = (aligned16(th) && aligned16(dtheta)) ? 1 : 0;
  return DGP_OK;
}

inline int fill_solve(const DgpHandle* h, int32_t batch, const void* th_init, const void* start, const void* goal, const DgpSdf* sdf,
                      const DgpCovs* covs, int32_t max_iters, double tol_delta, void* th_out, int32_t* iters, void* err_hist,
                      void* errext_hist, void* err_final, int32_t* info, dgp::GnParams& p) {
  int rc = fill_call(h, batch, th_init, start, goal, sdf, covs, p);
  if (rc != DGP_OK) return rc;
  if (!th_out) return fail(DGP_EINVAL, "th_out must be non-null");
  if (max_iters < 1) return fail(DGP_EINVAL, "max_iters must be >= 1, got %d", max_iters);
  p.th_out = th_out; p.iters = iters; p.err_hist = err_hist; p.errext_hist = errext_hist; p.err_final = err_final;
  p.info = info; p.max_iters = max_iters; p.tol_delta = tol_delta;
  p.vec_io = (aligned16(th_init) && aligned16(th_out)) ? 1 : 0;
  return DGP_OK;
}

inline int fill_eval(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                     const DgpCovs* covs, void* err, void* err_ext, void* unw_sg, void* unw_gp, void* unw_obs, dgp::GnParams& p) {
  const bool no_grid = !sdf || !sdf->data;
  if (no_grid && (err || err_ext || unw_obs))
    return fail(DGP_EINVAL, "sdf may be NULL only when err, err_ext and unw_obs (the outputs that read the grid) are NULL");
  int rc = fill_call(h, batch, th, start, goal, sdf, covs, p, /*sdf_optional=*/true);
  if (rc != DGP_OK) return rc;
  p.err = err; p.err_ext = err_ext; p.unw_sg = unw_sg; p.unw_gp = unw_gp; p.unw_obs = unw_obs;
  p.vec_io = aligned16(th) ? 1 : 0;
  return DGP_OK;
}

// fields of GnGradParams that only the round-4 entry points set
inline void clear_extensions(dgp::GnGradParams& g) {
  g.accumulate = 0; g.g_th_new = nullptr; g.th_addend = nullptr; g.th_hist = nullptr; g.th_final = nullptr; g.iters = nullptr; g.chain_iters = 0; g.g_sdf_mode = dgp::GSDF_DENSE; g.g_sdf_idx = nullptr; g.g_sdf_passes = 1; g.g_sdf_pass0 = 0;
  g.f_unw_sg = g.f_unw_gp = g.f_unw_obs = nullptr; g.f_addend = nullptr;
}

// the destination of dL/d(sdf): validation shared by the four backward entry points (call after clear_extensions)
inline int fill_gsdf(const DgpHandle* h, const DgpSdf* sdf, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, const dgp::GnParams& p, dgp::GnGradParams& g) {
  if (g_sdf_batch_stride < 0) return fail(DGP_EINVAL, "negative g_sdf batch stride");
  if (g_sdf_copies < 1 || g_sdf_copies > 64) return fail(DGP_EINVAL, "g_sdf_copies must be in 1..64, got %d", g_sdf_copies);
  if (g_sdf_copies > 1 && g_sdf_batch_stride != 0) return fail(DGP_EINVAL, "partial SDF-gradient copies need a shared grid (stride 0)");
  g.g_sdf = g_sdf; g.g_sdf_bstride = g_sdf_batch_stride; g.g_sdf_copies = g_sdf_copies;
  g.g_sdf_mode = dgp::GSDF_DENSE; g.g_sdf_idx = nullptr; g.g_sdf_passes = 1; g.g_sdf_pass0 = 0;
  if (g_sdf && sdf) {
    if (sdf->grad_mode < DGP_GSDF_DENSE || sdf->grad_mode > DGP_GSDF_SPARSE) return fail(DGP_EINVAL, "bad DgpSdf::grad_mode %d", sdf->grad_mode);
    g.g_sdf_mode = sdf->grad_mode;
    if (sdf->grad_mode == DGP_GSDF_SPARSE) {
      if (!sdf->grad_indices) return fail(DGP_EINVAL, "DGP_GSDF_SPARSE needs DgpSdf::grad_indices");
      if (g_sdf_copies != 1) return fail(DGP_EINVAL, "DGP_GSDF_SPARSE takes no partial copies");
      if (is_long(p.n)) return fail(DGP_EUNSUPPORTED, "DGP_GSDF_SPARSE is not implemented for num_states > 256");
      if ((reinterpret_cast<uintptr_t>(g_sdf) & 15u) || (reinterpret_cast<uintptr_t>(sdf->grad_indices) & 15u))
        return fail(DGP_EINVAL, "DGP_GSDF_SPARSE needs 16-byte aligned value / index arrays (vector stores)");
      g.g_sdf_idx = sdf->grad_indices;
    }
    if (is_long(p.n) && sdf->grad_mode == DGP_GSDF_DENSE_F64 && h->cfg.io_dtype == DGP_F32) return fail(DGP_EUNSUPPORTED, "DGP_GSDF_DENSE_F64 is not implemented for num_states > 256");
  }
  return DGP_OK;
}

inline int fill_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                         const DgpCovs* covs, const void* dtheta, const void* g_dtheta, const void* g_err_ext, void* g_th,
                         void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void* g_qc_inv,
                         void* g_obs_w, void* g_eps, dgp::GnParams& p, dgp::GnGradParams& g) {
  int rc = fill_call(h, batch, th, start, goal, sdf, covs, p, /*sdf_optional=*/false, /*scalar_qc_ok=*/true);
  if (rc == DGP_OK && p.qc_mode == dgp::QC_SCALAR && is_long(p.n)) return fail(DGP_EUNSUPPORTED, "qc_mode DGP_QC_SCALAR is not implemented for num_states > 256");
  if (rc != DGP_OK) return rc;
  if (g_dtheta && !dtheta) return fail(DGP_EINVAL, "dtheta (the forward output) is needed with a g_dtheta cotangent");
  if (g_qc_inv && p.qc_mode == DGP_QC_STATIC) return fail(DGP_EINVAL, "g_qc_inv given but qc_mode is DGP_QC_STATIC");
  clear_extensions(g);
  rc = fill_gsdf(h, sdf, g_sdf, g_sdf_batch_stride, g_sdf_copies, p, g);
  if (rc != DGP_OK) return rc;
  g.dtheta = dtheta; g.g_dtheta = g_dtheta; g.g_err_ext = g_err_ext; g.g_th = g_th; g.g_start = g_start; g.g_goal = g_goal;
  g.g_unw_sg = g.g_unw_gp = g.g_unw_obs = nullptr;
  g.g_qc = g_qc_inv; g.g_obs_w = g_obs_w; g.g_eps = g_eps;
  p.vec_io = (aligned16(th) && aligned16(dtheta) && aligned16(g_dtheta) && aligned16(g_th)) ? 1 : 0;
  return DGP_OK;
}

// dgp_eval_errors_backward: the backward kernels without the adjoint solve (no dtheta cotangent).  Only eps of the covariance
// inputs enters err_ext / the unweighted errors (fixed GP / obstacle weights, plan_layer.py:318,330; unit weights, :374-388), so the
// static-covariance kernel is launched whatever qc_mode the caller's covs carry.
inline int fill_eval_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                              const DgpCovs* covs, const void* g_err_ext, const void* g_unw_sg, const void* g_unw_gp, const void* g_unw_obs,
                              void* g_th, void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies,
                              void* g_eps, dgp::GnParams& p, dgp::GnGradParams& g) {
  const bool no_grid = !sdf || !sdf->data;
  if (no_grid && (g_err_ext || g_unw_obs || g_sdf))
    return fail(DGP_EINVAL, "sdf may be NULL only when g_err_ext, g_unw_obs and g_sdf (what reads / writes the grid) are NULL");
  DgpCovs c = {DGP_QC_STATIC, nullptr, nullptr, covs ? covs->eps : nullptr};
  if (covs && (covs->qc_mode < DGP_QC_STATIC || covs->qc_mode > DGP_QC_SCALAR)) return fail(DGP_EINVAL, "bad qc_mode %d", covs->qc_mode);      // (whatever the mode: only eps is read)
  int rc = fill_call(h, batch, th, start, goal, sdf, &c, p, /*sdf_optional=*/true);
  if (rc != DGP_OK) return rc;
  if (no_grid) p.sdf = nullptr;
  if (g_eps && !c.eps) return fail(DGP_EINVAL, "g_eps given but covs->eps is NULL (static epsilon)");
  clear_extensions(g);
  rc = fill_gsdf(h, sdf, g_sdf, g_sdf_batch_stride, g_sdf_copies, p, g);
  if (rc != DGP_OK) return rc;
  g.dtheta = nullptr; g.g_dtheta = nullptr; g.g_err_ext = g_err_ext;
  g.g_unw_sg = g_unw_sg; g.g_unw_gp = g_unw_gp; g.g_unw_obs = g_unw_obs;
  g.g_th = g_th; g.g_start = g_start; g.g_goal = g_goal;
  g.g_qc = nullptr; g.g_obs_w = nullptr; g.g_eps = g_eps;
  p.vec_io = (aligned16(th) && aligned16(g_th)) ? 1 : 0;
  return DGP_OK;
}

// dgp_gn_solve_backward: the chain kernels (static covariances, n <= 256)
inline int fill_solve_backward(const DgpHandle* h, int32_t batch, const void* start, const void* goal, const DgpSdf* sdf, int32_t max_iters,
                               const double* th_hist, const void* th_out, const int32_t* iters, const void* g_th_out, void* g_th_init,
                               void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, dgp::GnParams& p,
                               dgp::GnGradParams& g) {
  if (!th_hist || !th_out || !iters || !g_th_out) return fail(DGP_EINVAL, "th_hist, th_out, iters and g_th_out must be non-null");
  int rc = fill_call(h, batch, th_out, start, goal, sdf, nullptr, p);
  if (rc != DGP_OK) return rc;
  if (is_long(p.n)) return fail(DGP_EUNSUPPORTED, "dgp_gn_solve_backward is not implemented for num_states > 256 (chain the per-step backward instead)");
  // (static covariances by construction: fill_call without covs.  A diagonal Q_c_inv runs the static / Woodbury chain kernels, any other the general ones -- round 6)
  if (max_iters < 1) return fail(DGP_EINVAL, "max_iters must be >= 1, got %d", max_iters);
  clear_extensions(g);
  rc = fill_gsdf(h, sdf, g_sdf, g_sdf_batch_stride, g_sdf_copies, p, g);
  if (rc != DGP_OK) return rc;
  g.dtheta = nullptr; g.g_dtheta = g_th_out; g.g_err_ext = nullptr; g.g_unw_sg = g.g_unw_gp = g.g_unw_obs = nullptr;
  g.g_th = g_th_init; g.g_start = g_start; g.g_goal = g_goal;
  g.g_qc = nullptr; g.g_obs_w = nullptr; g.g_eps = nullptr;
  g.th_hist = th_hist; g.th_final = th_out; g.iters = iters; g.chain_iters = max_iters; g.g_sdf_passes = max_iters;
  p.max_iters = max_iters;
  p.vec_io = (aligned16(th_out) && aligned16(g_th_out) && aligned16(g_th_init)) ? 1 : 0;
  return DGP_OK;
}

// ---- the round-4 entry points, generic over how a kernel is launched (HIP: dgpmp2_hip.hip; the test emulator: tests/emul) ------------------
// `launch(mode, p, g)` -> DGP_OK or an error code; mode: dgp::MODE_* / 3 = backward / 4 = chain backward.
enum { kModeBackward = 3, kModeChain = 4, kModeStepErrs = 5 };      // kModeStepErrs: MODE_STEP on the twin kernels that carry the errors epilogue (gn_lane.h: DGP_STEP_ERRS)

template <typename Launch>
int gn_solve_traced(const DgpHandle* h, int32_t batch, const void* th_init, const void* start, const void* goal, const DgpSdf* sdf, const DgpCovs* covs,
                    int32_t max_iters, double tol_delta, void* th_out, int32_t* iters, void* err_hist, void* errext_hist, void* err_final, int32_t* info,
                    double* th_hist, Launch&& launch) {
  dgp::GnParams p;
  int rc = fill_solve(h, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters, err_hist, errext_hist, err_final, info, p);
  if (rc != DGP_OK) return rc;
  if (th_hist && is_long(p.n)) return fail(DGP_EUNSUPPORTED, "th_hist is not implemented for num_states > 256");
  if (th_hist && !iters) return fail(DGP_EINVAL, "th_hist needs iters (rows at or past iters[b] are not written)");
  p.dtheta = th_hist;              // (the fused loop has no dtheta output: the field carries the history pointer, gn_lane.h)
  return launch(dgp::MODE_SOLVE, p, (const dgp::GnGradParams*)nullptr);
}

template <typename Launch>
int gn_solve_backward(const DgpHandle* h, int32_t batch, const void* start, const void* goal, const DgpSdf* sdf, int32_t max_iters, const double* th_hist,
                      const void* th_out, const int32_t* iters, const void* g_th_out, void* g_th_init, void* g_start, void* g_goal, void* g_sdf,
                      int64_t g_sdf_batch_stride, int32_t g_sdf_copies, Launch&& launch) {
  dgp::GnParams p;
  dgp::GnGradParams g;
  int rc = fill_solve_backward(h, batch, start, goal, sdf, max_iters, th_hist, th_out, iters, g_th_out, g_th_init, g_start, g_goal, g_sdf,
                               g_sdf_batch_stride, g_sdf_copies, p, g);
  if (rc != DGP_OK) return rc;
  return launch((int)kModeChain, p, &g);
}

template <typename Launch>
int gn_step_errors(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf, const DgpCovs* covs,
                   void* dtheta, void* err, void* err_ext, int32_t* info, void* unw_sg, void* unw_gp, void* unw_obs, Launch&& launch) {
  dgp::GnParams p;
  int rc = fill_step(h, batch, th, start, goal, sdf, covs, dtheta, err, err_ext, info, p);
  if (rc != DGP_OK) return rc;
  const bool errs = unw_sg || unw_gp || unw_obs;
  if (errs && is_long(p.n)) return fail(DGP_EUNSUPPORTED, "dgp_gn_step_errors is not implemented for num_states > 256");
  // ONE launch where the step kernels with the errors epilogue exist (round 5): row-major grid, up to 128 states (the two four-states-per-lane shapes); d = 4: every
  // covariance representation, d = 6: everything but the general family (q_full / a non-diagonal Q_c_inv: those twins were built and measured in round 6 -- exact, but
  // 92 us against 66 + 8 us for the two launches, profiles/r06_d6_general_twin.txt: not shipped); everything else: the error kernel stream-ordered behind the step, as in round 4.  (Round 5 excluded two more twins -- <3,16,4,float,STEP,static> and <2,32,4,float,STEP,general> -- after wrong
  // results on the GPU: the exec-join miscompile of profiles/r06_compiler_fault.md, which the build repairs since round 6.)
  const int qk = dgp::kernel_variant(p);
  const bool twin_ok = DGP_EXCLUDE_REPAIRED_TWINS ? (!(h->cfg.dof == 3 && (qk == dgp::QK_GENERAL || (qk == dgp::QK_STATIC && !p.wb_ok))) &&
                                                     !(qk == dgp::QK_GENERAL && h->cfg.io_dtype == DGP_F32 && p.n > 64))
                                                  : !(h->cfg.dof == 3 && qk == dgp::QK_GENERAL);
  if (errs && p.sdf_layout == 0 && p.n <= kMaxStatesTiled && twin_ok && !h->force_lpt) {
    p.unw_sg = unw_sg; p.unw_gp = unw_gp; p.unw_obs = unw_obs;
#if DGP_TWIN_REPRO      // reproducer builds (-DDGP_TWIN_REPRO=1): DGP_TWIN_NO_ERRS=1 launches the twin kernel with its epilogue switched off at run time (is the main body or the epilogue wrong?)
    if (getenv("DGP_TWIN_NO_ERRS")) p.unw_sg = p.unw_gp = p.unw_obs = nullptr;
#endif
    return launch((int)kModeStepErrs, p, (const dgp::GnGradParams*)nullptr);
  }
  rc = launch(dgp::MODE_STEP, p, (const dgp::GnGradParams*)nullptr);
  if (rc != DGP_OK || !errs) return rc;
  // the unweighted errors at th + dtheta: the error kernel with dtheta as addend, stream-ordered behind the step (only eps of the covariances enters them)
  DgpCovs c = {DGP_QC_STATIC, nullptr, nullptr, covs ? covs->eps : nullptr};
  rc = fill_eval(h, batch, th, start, goal, sdf, &c, nullptr, nullptr, unw_sg, unw_gp, unw_obs, p);
  if (rc != DGP_OK) return rc;
  p.dtheta = dtheta;               // MODE_EVAL: the addend (gn_lane.h)
  p.vec_io = (aligned16(th) && aligned16(dtheta)) ? 1 : 0;
  return launch(dgp::MODE_EVAL, p, (const dgp::GnGradParams*)nullptr);
}

template <typename Launch>
int gn_step_errors_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf, const DgpCovs* covs,
                            const void* dtheta, const void* g_dtheta, const void* g_err_ext, const void* g_unw_sg, const void* g_unw_gp,
                            const void* g_unw_obs, void* g_th, void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies,
                            void* g_qc_inv, void* g_obs_w, void* g_eps, void* workspace, Launch&& launch) {
  dgp::GnParams p;
  dgp::GnGradParams g;
  const bool errs = g_unw_sg || g_unw_gp || g_unw_obs;
  if (errs && !dtheta) return fail(DGP_EINVAL, "dgp_gn_step_errors_backward needs dtheta when an unweighted-error cotangent is given");
  if (errs && h && h->cfg.dof == 2 && !is_long(h->cfg.num_states)) {
    // ONE launch (round 5, d = 4): the errors' backward at th + dtheta runs as a prologue of the step's backward kernel (gn_backward.h: unweighted_errors_prologue) and
    // hands its trajectory gradient and its shares of the small gradients over in the lane's LDS slots -- no workspace
    int rc = fill_backward(h, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_th, g_start, g_goal, g_sdf, g_sdf_batch_stride,
                           g_sdf_copies, g_qc_inv, g_obs_w, g_eps, p, g);
    if (rc != DGP_OK) return rc;
    g.f_unw_sg = g_unw_sg; g.f_unw_gp = g_unw_gp; g.f_unw_obs = g_unw_obs; g.f_addend = dtheta;
    g.g_sdf_passes = 2;
    return launch((int)kModeBackward, p, &g);
  }
  if (errs) {
    // d = 6 (no prologue in its backward kernels: registers, LDS -- gn_backward.h) and long trajectories (the loop kernels of gn_long.h): two launches, as in round 4
    if (!workspace) return fail(DGP_EINVAL, "dgp_gn_step_errors_backward needs a (B,n,d) workspace for dof = 3 and for num_states > 256");
    // launch 1: backward of the unweighted errors at th + dtheta -> workspace (gradient w.r.t. th + dtheta), start / goal / eps / grid shares
    int rc = fill_eval_backward(h, batch, th, start, goal, sdf, covs, nullptr, g_unw_sg, g_unw_gp, g_unw_obs, workspace, g_start, g_goal, g_sdf,
                                g_sdf_batch_stride, g_sdf_copies, nullptr, p, g);
    if (rc != DGP_OK) return rc;
    g.g_eps = g_eps;                // (written by BOTH launches whatever the epsilon source: launch 2 adds to what this one stores)
    g.th_addend = dtheta;
    g.g_sdf_passes = 2; g.g_sdf_pass0 = 1;      // DGP_GSDF_SPARSE: the taps at th + dtheta are the second block
    p.vec_io = (p.vec_io && aligned16(dtheta)) ? 1 : 0;
    rc = launch((int)kModeBackward, p, &g);
    if (rc != DGP_OK) return rc;
  }
  // the step's backward (the only launch without error cotangents); behind launch 1 the workspace joins the dtheta cotangent and g_th,
  // and the small gradients are added to launch 1's
  int rc = fill_backward(h, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_th, g_start, g_goal, g_sdf, g_sdf_batch_stride,
                         g_sdf_copies, g_qc_inv, g_obs_w, g_eps, p, g);
  if (rc != DGP_OK) return rc;
  if (errs) {
    g.g_th_new = workspace; g.accumulate = 1; g.g_sdf_passes = 2;
    p.vec_io = (p.vec_io && aligned16(workspace)) ? 1 : 0;
  }
  return launch((int)kModeBackward, p, &g);
}

}  // namespace dgp_host
