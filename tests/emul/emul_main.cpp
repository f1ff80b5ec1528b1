// Wavefront emulator (TEST INFRASTRUCTURE ONLY -- never part of the product path).
// Compiles the very same per-lane program the HIP kernel runs (dgpmp2_amd/csrc/gn_lane.h) for the host and
// executes one wavefront as 64 lock-stepped threads; a cross-lane fetch is "publish, barrier, read, barrier".
// Lets the CPU-only test-suite check the kernel logic (factor evaluation, assembly, block PCR, modes)
// against the oracle before any GPU time is spent.  Built by tests/test_lane_emulator.py with g++.
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

#define DGP_HD inline
#define DGP_STEP_ERRS 1      // the step kernels' errors epilogue compiled in (behind its run-time test of GnParams::unw_*)
#define DGP_TL 2      // the emulator decides the grid layout at run time (GnParams::sdf_layout): one build for row-major and tiled grids (gn_lane.h)
#include "../../dgpmp2_amd/csrc/dgp_host.h"

namespace {

struct WaveShared {
  pthread_barrier_t bar;
  double slot[64];
  int islot[64];
  alignas(16) char lds[64 * (4 * 6 * 8 + 16) + 64 * 544];      // dgp::WaveStore staging block (largest chunk: C=4, d=6, f64) + dgp::SinvStash
  alignas(16) char wb[dgp::kWbLdsBytes];                        // LDS copy of the Woodbury constant table
  alignas(16) char chain[dgp::ChainSlots<4, 6>::kBytes];         // the chain backward's running cotangent + accumulated start / goal gradients (the largest layout)
  alignas(16) char longb[160 * 1024];                           // gn_long.h: the dynamic LDS block of the long-trajectory kernels (gfx950: 160 KB per workgroup)
};

pthread_mutex_t g_atomic_mutex = PTHREAD_MUTEX_INITIALIZER;

struct HostCtx {
  WaveShared* ws;
  int lane_, wave_;
  int lane() const { return lane_; }
  int wave() const { return wave_; }
  char* lds() { return ws->lds; }
  char* stash() { return ws->lds + 64 * (4 * 6 * 8 + 16); }
  char* wb_lds() { return ws->wb; }
  char* chain_lds() { return ws->chain; }
  char* long_lds() { return ws->longb; }
  void mem_sync() { pthread_barrier_wait(&ws->bar); }
  const double* wb_source(const dgp::GnParams& p) const { return p.wb_tab; }
  void lds_sync() { pthread_barrier_wait(&ws->bar); }
  double fetch(double v, int src) {
    ws->slot[lane_] = v;
    pthread_barrier_wait(&ws->bar);
    double r = ws->slot[src & 63];
    pthread_barrier_wait(&ws->bar);
    return r;
  }
  int fetch_i(int v, int src) {
    ws->islot[lane_] = v;
    pthread_barrier_wait(&ws->bar);
    int r = ws->islot[src & 63];
    pthread_barrier_wait(&ws->bar);
    return r;
  }
  // DPP row shifts of the device context: neighbour inside the 16-lane row, 0 where there is none
  template <int S>
  double row_from_lower(double v) {
    ws->slot[lane_] = v;
    pthread_barrier_wait(&ws->bar);
    double r = ((lane_ & 15) >= S) ? ws->slot[lane_ - S] : 0.0;
    pthread_barrier_wait(&ws->bar);
    return r;
  }
  template <int S>
  double row_from_upper(double v) {
    ws->slot[lane_] = v;
    pthread_barrier_wait(&ws->bar);
    double r = ((lane_ & 15) + S < 16) ? ws->slot[lane_ + S] : 0.0;
    pthread_barrier_wait(&ws->bar);
    return r;
  }
  template <int N>
  double row_rotate(double v) {       // DPP row_ror:N -- lane i reads lane (i - N) mod 16 of its 16-lane row
    ws->slot[lane_] = v;
    pthread_barrier_wait(&ws->bar);
    double r = ws->slot[(lane_ & ~15) | ((lane_ - N) & 15)];
    pthread_barrier_wait(&ws->bar);
    return r;
  }
  bool any(bool pred) {
    ws->islot[lane_] = pred ? 1 : 0;
    pthread_barrier_wait(&ws->bar);
    int r = 0;
    for (int k = 0; k < 64; ++k) r |= ws->islot[k];
    pthread_barrier_wait(&ws->bar);
    return r != 0;
  }
  uint64_t ballot(bool pred) {      // bit l = pred of lane l
    ws->islot[lane_] = pred ? 1 : 0;
    pthread_barrier_wait(&ws->bar);
    uint64_t r = 0;
    for (int k = 0; k < 64; ++k) r |= (uint64_t)(ws->islot[k] & 1) << k;
    pthread_barrier_wait(&ws->bar);
    return r;
  }
  int xcc_id() const { return wave_ & 7; }          // the emulator spreads "wavefronts" over 8 pretend XCDs
  template <typename T>
  void atomic_add(T* p, T v, bool = false) {
    pthread_mutex_lock(&g_atomic_mutex); *p += v; pthread_mutex_unlock(&g_atomic_mutex);
  }
};

template <int DOF, int LPT, int C, typename IO>
void run_wave(const dgp::GnParams& p, const dgp::GnGradParams* g, int mode, int wave) {
  WaveShared ws;
  pthread_barrier_init(&ws.bar, nullptr, 64);
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l) {
    th.emplace_back([&, l]() {
      HostCtx cx{&ws, l, wave};
      // same dispatch as dgp_dev::launch_typed: the kernel variant follows the covariance representation
      const int qk = dgp::kernel_variant(p);
      if constexpr (C == 4) {      // the Woodbury kernels, chosen exactly as dgp_dev::launch_typed does
        if (mode != dgp::MODE_EVAL && qk == dgp::QK_STATIC && dgp::wb_applies(p, LPT, C)) {
          if (p.n == LPT * C) {
            if (mode == dgp::MODE_STEP) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_STEP, dgp::QK_WB>(p, cx);
            else if (mode == dgp::MODE_SOLVE) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_SOLVE, dgp::QK_WB>(p, cx);
            else dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_WB>(p, *g, cx);
          } else {
            if (mode == dgp::MODE_STEP) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_STEP, dgp::QK_WBR>(p, cx);
            else if (mode == dgp::MODE_SOLVE) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_SOLVE, dgp::QK_WBR>(p, cx);
            else if (mode == 4) dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_WBR, true>(p, *g, cx);
            else dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_WBR>(p, *g, cx);
          }
          return;
        }
      }
      if (mode == dgp::MODE_STEP) {
        if (qk == dgp::QK_STATIC) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_STEP, dgp::QK_STATIC>(p, cx);
        else if (qk == dgp::QK_SCALED) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_STEP, dgp::QK_SCALED>(p, cx);
        else if (qk == dgp::QK_KRON) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_STEP, dgp::QK_KRON>(p, cx);
        else dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_STEP, dgp::QK_GENERAL>(p, cx);
      } else if (mode == dgp::MODE_SOLVE) {
        if (qk == dgp::QK_STATIC) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_SOLVE, dgp::QK_STATIC>(p, cx);
        else if (qk == dgp::QK_KRON) dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_SOLVE, dgp::QK_KRON>(p, cx);
        else dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_SOLVE, dgp::QK_GENERAL>(p, cx);
      } else if (mode == dgp::MODE_EVAL) {
        dgp::gn_lane_program<DOF, LPT, C, IO, dgp::MODE_EVAL, dgp::QK_GENERAL>(p, cx);
      } else if (mode == 4) {      // (dgp_gn_solve_backward: static covariances only, host-checked; a non-diagonal Q_c_inv runs the general-covariance chain kernel)
        if (qk == dgp::QK_STATIC) dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_STATIC, true>(p, *g, cx);
        else dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_GENERAL, true>(p, *g, cx);
      } else {
        if (qk == dgp::QK_STATIC) dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_STATIC>(p, *g, cx);
        else if (qk == dgp::QK_SCALED) dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_SCALED>(p, *g, cx);
        else if (qk == dgp::QK_KRON) dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_KRON>(p, *g, cx);
        else dgp::gn_backward_lane_program<DOF, LPT, C, IO, dgp::QK_GENERAL>(p, *g, cx);
      }
    });
  }
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&ws.bar);
}

#define EMUL_FOR_EACH_SHAPE(X) X(16, 1) X(32, 1) X(64, 1) X(16, 2) X(32, 2) X(64, 2) X(16, 4) X(32, 4) X(64, 4)

template <int DOF, typename IO>
void run_all(const dgp::GnParams& p, const dgp::GnGradParams* g, int mode, DgpShape sh) {
  const int tpw = 64 / sh.lpt;
  const int waves = (p.B + tpw - 1) / tpw;
  for (int w = 0; w < waves; ++w) {
#define EMUL_CASE(L, CC) if (sh.lpt == L && sh.c == CC) run_wave<DOF, L, CC, IO>(p, g, mode, w);
    EMUL_FOR_EACH_SHAPE(EMUL_CASE)
#undef EMUL_CASE
  }
}

// long trajectories (gn_long.h): one trajectory per wavefront, the same dispatch as gn_long_inst.hip
template <int DOF, typename IO>
void run_long(const dgp::GnParams& p, const dgp::GnGradParams* g, int mode) {
  if (dgp::long_lds_bytes<2 * DOF>(p.n) > (int)sizeof(WaveShared::longb)) abort();
  for (int w = 0; w < p.B; ++w) {
    WaveShared* ws = new WaveShared();
    pthread_barrier_init(&ws->bar, nullptr, 64);
    std::vector<std::thread> th;
    for (int l = 0; l < 64; ++l) {
      th.emplace_back([&, l]() {
        HostCtx cx{ws, l, w};
        if (mode == dgp::MODE_STEP) dgp::gn_long_program<DOF, IO, dgp::MODE_STEP>(p, cx);
        else if (mode == dgp::MODE_SOLVE) dgp::gn_long_program<DOF, IO, dgp::MODE_SOLVE>(p, cx);
        else if (mode == dgp::MODE_EVAL) dgp::gn_long_program<DOF, IO, dgp::MODE_EVAL>(p, cx);
        else dgp::gn_long_backward_program<DOF, IO>(p, *g, cx);
      });
    }
    for (auto& t : th) t.join();
    pthread_barrier_destroy(&ws->bar);
    delete ws;
  }
}

void run(const DgpHandle* h, const dgp::GnParams& p, const dgp::GnGradParams* g, int mode) {
  if (p.sdf_layout != 0 && p.sdf && p.n > dgp_host::kMaxStatesTiled) {
    fprintf(stderr, "emul: tiled grids are implemented for num_states <= 128 (the product library returns DGP_EUNSUPPORTED here)\n");
    abort();
  }
  if (dgp_host::is_long(p.n)) {
    const bool f64l = h->cfg.io_dtype == DGP_F64;
    if (h->cfg.dof == 2) { if (f64l) run_long<2, double>(p, g, mode); else run_long<2, float>(p, g, mode); }
    else { if (f64l) run_long<3, double>(p, g, mode); else run_long<3, float>(p, g, mode); }
    return;
  }
  const bool step_errs = mode == (int)dgp_host::kModeStepErrs;      // dgp_gn_step_errors in one launch: MODE_STEP with the errors epilogue, on the twin kernels' shapes
  if (step_errs) mode = dgp::MODE_STEP;
  const DgpShape sh = dgp_host::choose_shape(h, p.B, dgp_host::shape_family(mode, p), step_errs || (p.sdf_layout != 0 && p.sdf != nullptr));      // (the shape the product launches)
  const bool f64 = h->cfg.io_dtype == DGP_F64;
  if (h->cfg.dof == 2) { if (f64) run_all<2, double>(p, g, mode, sh); else run_all<2, float>(p, g, mode, sh); }
  else { if (f64) run_all<3, double>(p, g, mode, sh); else run_all<3, float>(p, g, mode, sh); }
}

}  // namespace

// Same entry points as include/dgpmp2_hip.h with the prefix emul_, on HOST pointers (stream ignored).
template <typename T>
static void emul_square(const T* raw, int B, int W, int n_gp, int n, int le, int dof, T* s, T* blk, T* ow, T* ep) {
  for (int64_t b = 0; b < B; ++b)
    for (int c = 0; c < W; ++c) {
      const T v = raw[b * W + c];
      const T q = v * v;
      if (c < n_gp) {
        if (s) s[b * n_gp + c] = q;
        if (blk) for (int i = 0; i < dof * dof; ++i) blk[(b * n_gp + c) * dof * dof + i] = (i % (dof + 1) == 0) ? q : (T)0;
      } else if (c < n_gp + n) { if (ow) ow[b * n + (c - n_gp)] = q; }
      else if (le && c < n_gp + 2 * n) { if (ep) ep[b * n + (c - n_gp - n)] = q; }
    }
}
template <typename T>
static void emul_square_bwd(const T* raw, int B, int W, int n_gp, int n, int le, int dof, const T* gb, const T* gw, const T* ge, T* graw) {
  for (int64_t b = 0; b < B; ++b)
    for (int c = 0; c < W; ++c) {
      T g = (T)0;
      if (c < n_gp) { if (gb) for (int i = 0; i < dof; ++i) g += gb[(b * n_gp + c) * dof * dof + i * (dof + 1)]; }
      else if (c < n_gp + n) { if (gw) g = gw[b * n + (c - n_gp)]; }
      else if (le && c < n_gp + 2 * n) { if (ge) g = ge[b * n + (c - n_gp - n)]; }
      graw[b * W + c] = (T)2 * raw[b * W + c] * g;
    }
}

extern "C" {

int emul_abi_version(void) { return DGP_ABI_VERSION; }
const char* emul_last_error(void) { return dgp_host::err_buf(); }
int emul_create(const DgpConfig* cfg, DgpHandle** out) { return dgp_host::create(cfg, out); }
void emul_destroy(DgpHandle* h) { delete h; }
int emul_num_factor_rows(const DgpHandle* h) { return h ? h->M : DGP_EINVAL; }

// dgp_sdf_2d is not a lane program: nothing to emulate (tests/test_sdf_edt.py checks the HIP kernels against scipy and oracle/edt_oracle.py)
size_t emul_sdf_2d_workspace_bytes(int32_t, int32_t, int32_t, int32_t) { return 0; }
int emul_sdf_2d(const void*, int32_t, int32_t, int32_t, int32_t, int32_t, double, void*, int32_t, int32_t, void*, size_t, void*) {
  return dgp_host::fail(DGP_EUNSUPPORTED, "the emulator covers the wavefront lane programs only");
}
int emul_time_next_launch(void*, void*) { return DGP_OK; }      // nothing to time: the emulator runs on the host

int emul_launch_shape(const DgpHandle* h, int32_t batch, int32_t* lpt, int32_t* c) {
  if (!h || batch <= 0) return DGP_EINVAL;
  const DgpShape sh = dgp_host::choose_shape(h, batch, h->base.qc_diag == 0 ? dgp_host::FAM_GENERAL : dgp_host::FAM_STATIC);
  if (lpt) *lpt = sh.lpt;
  if (c) *c = sh.c;
  return DGP_OK;
}

int emul_step_kernel_variant(const DgpHandle* h, int32_t batch) {
  if (!h || batch <= 0) return DGP_EINVAL;
  return dgp_host::step_kernel_variant(h, batch);
}

int emul_gn_step(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                 const DgpCovs* covs, void* dtheta, void* err, void* err_ext, int32_t* info, void*) {
  dgp::GnParams p;
  int rc = dgp_host::fill_step(h, batch, th, start, goal, sdf, covs, dtheta, err, err_ext, info, p);
  if (rc != DGP_OK) return rc;
  run(h, p, nullptr, dgp::MODE_STEP);
  return DGP_OK;
}

int emul_gn_solve(const DgpHandle* h, int32_t batch, const void* th_init, const void* start, const void* goal, const DgpSdf* sdf,
                  const DgpCovs* covs, int32_t max_iters, double tol_delta, void* th_out, int32_t* iters, void* err_hist,
                  void* errext_hist, void* err_final, int32_t* info, void*) {
  dgp::GnParams p;
  int rc = dgp_host::fill_solve(h, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters, err_hist,
                                errext_hist, err_final, info, p);
  if (rc != DGP_OK) return rc;
  run(h, p, nullptr, dgp::MODE_SOLVE);
  return DGP_OK;
}

int emul_eval_errors(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                     const DgpCovs* covs, void* err, void* err_ext, void* unw_sg, void* unw_gp, void* unw_obs, void*) {
  dgp::GnParams p;
  int rc = dgp_host::fill_eval(h, batch, th, start, goal, sdf, covs, err, err_ext, unw_sg, unw_gp, unw_obs, p);
  if (rc != DGP_OK) return rc;
  run(h, p, nullptr, dgp::MODE_EVAL);
  return DGP_OK;
}

int emul_gn_step_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                          const DgpCovs* covs, const void* dtheta, const void* g_dtheta, const void* g_err_ext, void* g_th,
                          void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void* g_qc_inv,
                          void* g_obs_w, void* g_eps, void*) {
  dgp::GnParams p;
  dgp::GnGradParams g;
  int rc = dgp_host::fill_backward(h, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_th, g_start, g_goal, g_sdf,
                                   g_sdf_batch_stride, g_sdf_copies, g_qc_inv, g_obs_w, g_eps, p, g);
  if (rc != DGP_OK) return rc;
  run(h, p, &g, 3);
  return DGP_OK;
}

int emul_eval_errors_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                              const DgpCovs* covs, const void* g_err_ext, const void* g_unw_sg, const void* g_unw_gp, const void* g_unw_obs,
                              void* g_th, void* g_start, void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies,
                              void* g_eps, void*) {
  dgp::GnParams p;
  dgp::GnGradParams g;
  int rc = dgp_host::fill_eval_backward(h, batch, th, start, goal, sdf, covs, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th, g_start, g_goal,
                                        g_sdf, g_sdf_batch_stride, g_sdf_copies, g_eps, p, g);
  if (rc != DGP_OK) return rc;
  run(h, p, &g, 3);
  return DGP_OK;
}

namespace {
struct EmulLaunch {
  const DgpHandle* h;
  int operator()(int mode, const dgp::GnParams& p, const dgp::GnGradParams* g) const { run(h, p, g, mode); return DGP_OK; }      // (kModeStepErrs: run() maps it to MODE_STEP on a four-states-per-lane shape)
};
}  // namespace

int emul_gn_solve_traced(const DgpHandle* h, int32_t batch, const void* th_init, const void* start, const void* goal, const DgpSdf* sdf,
                         const DgpCovs* covs, int32_t max_iters, double tol_delta, void* th_out, int32_t* iters, void* err_hist,
                         void* errext_hist, void* err_final, int32_t* info, double* th_hist, void*) {
  return dgp_host::gn_solve_traced(h, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters, err_hist, errext_hist, err_final, info, th_hist,
                                   EmulLaunch{h});
}

int emul_gn_solve_backward(const DgpHandle* h, int32_t batch, const void* start, const void* goal, const DgpSdf* sdf, int32_t max_iters,
                           const double* th_hist, const void* th_out, const int32_t* iters, const void* g_th_out, void* g_th_init, void* g_start,
                           void* g_goal, void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void*) {
  return dgp_host::gn_solve_backward(h, batch, start, goal, sdf, max_iters, th_hist, th_out, iters, g_th_out, g_th_init, g_start, g_goal, g_sdf,
                                     g_sdf_batch_stride, g_sdf_copies, EmulLaunch{h});
}

int emul_gn_step_errors(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                        const DgpCovs* covs, void* dtheta, void* err, void* err_ext, int32_t* info, void* unw_sg, void* unw_gp, void* unw_obs, void*) {
  return dgp_host::gn_step_errors(h, batch, th, start, goal, sdf, covs, dtheta, err, err_ext, info, unw_sg, unw_gp, unw_obs, EmulLaunch{h});
}

int emul_gn_step_errors_backward(const DgpHandle* h, int32_t batch, const void* th, const void* start, const void* goal, const DgpSdf* sdf,
                                 const DgpCovs* covs, const void* dtheta, const void* g_dtheta, const void* g_err_ext, const void* g_unw_sg,
                                 const void* g_unw_gp, const void* g_unw_obs, void* g_th, void* g_start, void* g_goal, void* g_sdf,
                                 int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void* g_qc_inv, void* g_obs_w, void* g_eps, void* workspace, void*) {
  return dgp_host::gn_step_errors_backward(h, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th, g_start,
                                           g_goal, g_sdf, g_sdf_batch_stride, g_sdf_copies, g_qc_inv, g_obs_w, g_eps, workspace, EmulLaunch{h});
}

int emul_sum_partial_grids(const void* partial, int32_t partial_dtype, int32_t copies, int64_t elems, double scale, void* out, int32_t out_dtype, void*) {
  if (!partial || !out || copies < 1 || copies > 64 || elems < 1) return DGP_EINVAL;
  for (int64_t e = 0; e < elems; ++e) {
    double acc = 0.0;
    for (int c = 0; c < copies; ++c) acc += partial_dtype == DGP_F64 ? ((const double*)partial)[(int64_t)c * elems + e] : (double)((const float*)partial)[(int64_t)c * elems + e];
    if (out_dtype == DGP_F64) ((double*)out)[e] = acc * scale; else ((float*)out)[e] = (float)(acc * scale);
  }
  return DGP_OK;
}

int emul_square_covariances(const void* raw, int32_t dtype, int32_t B, int32_t W, int32_t n_gp, int32_t n, int32_t le, int32_t dof, void* s, void* blk, void* ow, void* ep, void*) {
  if (!raw || B < 1 || W < n_gp + n * (le ? 2 : 1)) return DGP_EINVAL;
  if (dtype == DGP_F64) emul_square<double>((const double*)raw, B, W, n_gp, n, le, dof, (double*)s, (double*)blk, (double*)ow, (double*)ep);
  else emul_square<float>((const float*)raw, B, W, n_gp, n, le, dof, (float*)s, (float*)blk, (float*)ow, (float*)ep);
  return DGP_OK;
}
int emul_square_covariances_backward(const void* raw, int32_t dtype, int32_t B, int32_t W, int32_t n_gp, int32_t n, int32_t le, int32_t dof, const void* gb, const void* gw,
                                     const void* ge, void* graw, void*) {
  if (!raw || !graw || B < 1 || W < n_gp + n * (le ? 2 : 1)) return DGP_EINVAL;
  if (dtype == DGP_F64) emul_square_bwd<double>((const double*)raw, B, W, n_gp, n, le, dof, (const double*)gb, (const double*)gw, (const double*)ge, (double*)graw);
  else emul_square_bwd<float>((const float*)raw, B, W, n_gp, n, le, dof, (const float*)gb, (const float*)gw, (const float*)ge, (float*)graw);
  return DGP_OK;
}

int emul_event_create(void** out) { if (out) *out = nullptr; return DGP_OK; }      // nothing to time on the host
void emul_event_destroy(void*) {}
int emul_event_elapsed_ms(void*, void*, float* ms) { if (ms) *ms = 0.0f; return DGP_OK; }

}  // extern "C"
