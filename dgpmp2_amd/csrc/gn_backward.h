// gn_backward.h -- backward (adjoint) lane program of the fused Gauss-Newton step.
//
// Forward:  dtheta = Lambda^-1 eta,  Lambda = A^T K A + delta I,  eta = A^T K b   (plan_layer.py:214-228), and
//           err_ext = sum_f 1/2 e_f^T K_f^fix e_f / M                               (plan_layer.py:310-345).
// The reference differentiates this with torch autograd over its dense ops; here the adjoint is written out:
//   lambda = Lambda^-1 gbar                       (gbar = dL/d dtheta; same fused assembly + block solve, other rhs)
//   dL = lambda^T (d eta - d Lambda dtheta) + ebar d err_ext
// and, factor by factor (Jacobian H_f, error e_f, weight K_f; u_f = H_f lambda, rho_f = e_f - H_f dtheta):
//   dL_f = (dH_f lambda)^T K_f rho_f + u_f^T dK_f rho_f + u_f^T K_f (de_f - dH_f dtheta).
// Every lane recomputes the factors of its C states (nothing but dtheta is kept from the forward pass), owns the
// gradient rows of those states (no atomics for th / qc / obs_w / eps) and scatters the SDF gradient to the four
// bilinear taps with atomics.  Floor / clamp indices and the hinge / sign selections are piecewise constant, exactly
// as torch autograd treats them (sdf_utils.py:64-72, obstacle_cost.py:36-37, velocity_limit_factor.py:20-23).
#pragma once
#include "gn_lane.h"

#ifndef DGP_BWD_COLWISE
#define DGP_BWD_COLWISE(D) true      // order of the Woodbury Schur assembly in the adjoint solve (see gn_linear_solve_wb): column-wise for both robots (d = 6 backward 47.8 -> 40.2 us); tuning aid
#endif
namespace dgp {

enum { kMaxXcds = 8 };               // XCDs (each with its own, mutually non-coherent L2) on a gfx950 device

struct GnGradParams {
  const void *dtheta;                 // dtheta of the forward pass (B,n,d)
  const void *g_dtheta, *g_err_ext;   // cotangents; either may be null (= 0)
  const void *g_unw_sg, *g_unw_gp, *g_unw_obs;   // (B) cotangents of the unweighted errors (plan_layer.py:374-388; dgp_eval_errors_backward), null = 0
  void *g_th, *g_start, *g_goal, *g_sdf, *g_qc, *g_obs_w, *g_eps;
  int64_t g_sdf_bstride;
  int32_t g_sdf_copies;              // > 1: per-XCD partial grids of a shared SDF gradient (see include/dgpmp2_hip.h)
  // round 4 -- dgp_gn_step_errors_backward: the step's backward launched behind the errors' backward at th + dtheta
  int32_t accumulate;                // 1: g_start / g_goal / g_eps already hold the first launch's share -- add to them instead of overwriting
  const void* g_th_new;              // (B,n,d) or null: gradient w.r.t. th + dtheta left by the first launch; added to the dtheta cotangent AND to g_th
  const void* th_addend;             // (B,n,d) or null: evaluate at th + th_addend (the errors' backward of dgp_gn_step_errors_backward: th_addend = dtheta)
  // round 4 -- dgp_gn_solve_backward (chain kernels): the fused loop's trajectory history and iteration counts
  const double* th_hist;             // (max_iters,B,n,d) fp64: th_k of every iteration the forward loop ran (dgp_gn_solve's th_hist)
  const void* th_final;              // (B,n,d): th_out of the forward loop
  const int32_t* iters;              // (B): iterations each trajectory ran
  int32_t chain_iters;               // rows of th_hist (max_iters of the forward call)
  // round 5 -- how the grid gradient is delivered (DgpSdf::grad_mode): GSDF_DENSE: g_sdf = grids of the I/O type, accumulated with atomics;
  // GSDF_DENSE_F64: the same with FP64 grids whatever the I/O type (the partial copies of a shared grid: the sum over trajectories no longer depends
  // on the order of fp32 atomics); GSDF_SPARSE: no atomics, no zero-filled grid -- g_sdf = (passes, B, n, 4) tap VALUES and g_sdf_idx = the (4, nnz) int64
  // COO indices (b, 0, y, x) of a torch.sparse_coo_tensor of the grid tensor's shape (per-sample grids: the dense gradient is O(B H W) bytes of zeros)
  int32_t g_sdf_mode;
  int32_t g_sdf_passes, g_sdf_pass0; // GSDF_SPARSE: tap blocks the caller's arrays hold (1; max_iters for the chain kernels; 2 for dgp_gn_step_errors_backward) and the block this launch's first pass writes
  int64_t* g_sdf_idx;
  // round 5 -- dgp_gn_step_errors_backward as ONE launch (d = 4): the backward of the unweighted errors at th + f_addend runs as a prologue of this kernel
  // (unweighted_errors_prologue below) and hands its trajectory gradient and its shares of g_start / g_goal / g_eps to the main program in lane-private LDS
  const void *f_unw_sg, *f_unw_gp, *f_unw_obs;   // (B) cotangents of the three unweighted errors, null = 0
  const void* f_addend;                          // (B,n,d) dtheta of the forward pass; null: no prologue
};
enum { GSDF_DENSE = 0, GSDF_DENSE_F64 = 1, GSDF_SPARSE = 2 };

// ---------------------------------------------------------------------------------------------------
// Scatter-add of the SDF gradient.  An atomic instruction costs the memory pipeline one pass per DISTINCT cache line among
// its 64 lane addresses, not per lane (profiles/tools/atomic_probe.hip: 1024 wavefronts x 16 instructions take 51 us with 64
// scattered pixels per instruction, 27 us when lanes 2m, 2m+1 hit adjacent pixels; atomic scope makes no difference).  The two
// taps of a state in one grid row ARE adjacent pixels, so the taps are redistributed through LDS: lane s writes its pair
// {(i_x1, v_x1), (i_x2, v_x2)} of a row, and lane L of the half h re-reads entry L of lanes 32h .. 32h+31 -- lanes 2m, 2m+1 then
// carry the two adjacent pixels of lane 32h + m's state.  Same number of atomic instructions, half the line passes.
//
// Partial copies of a shared grid (g_sdf_copies > 1): XCD-local (workgroup-scope) atomics are only sound when no two XCDs
// share a copy -- gfx950 has at most 8 XCDs (XCC_ID 0..7), so with >= 8 copies every copy belongs to ONE XCD; with a multiple of
// 8 the wavefronts of an XCD spread over copies xcc, xcc + 8, ... (same-pixel updates serialise over copies / 8 lines of that
// L2 instead of one).  Fewer than 8 copies still spread the contention, but two non-coherent L2s may then target one copy ->
// device-scope atomics.
// ---------------------------------------------------------------------------------------------------
template <typename IO> struct __attribute__((aligned(2 * sizeof(IO) >= 8 ? 2 * sizeof(IO) : 8))) TapEntry { int32_t idx; IO val; };

template <int LPT, int C, typename IO, typename Ctx>
DGP_HD void sdf_scatter_pairs(const GnParams& p, const GnGradParams& gp, Ctx& cx, const int32_t (&tap_i)[C][4], const IO (&tap_v)[C][4]) {
  constexpr int TPW = 64 / LPT;
  typedef TapEntry<IO> E;
  const int lane = cx.lane();
  const bool local = gp.g_sdf_copies >= kMaxXcds;
  const bool wide = sizeof(IO) == 4 && gp.g_sdf_mode == GSDF_DENSE_F64;      // wave-uniform: fp64 grids behind fp32 I/O
  int64_t first = 0;
  if (gp.g_sdf_copies > 1) {
    const int xcc = cx.xcc_id();
    const int per_xcd = gp.g_sdf_copies / kMaxXcds;
    const int copy = (per_xcd >= 1 && gp.g_sdf_copies % kMaxXcds == 0) ? (xcc % kMaxXcds) + kMaxXcds * ((cx.wave() / kMaxXcds) % per_xcd) : xcc % gp.g_sdf_copies;      // (XCC_ID is a 4-bit field: never index past the copies)
    first = (int64_t)copy * grid_elems(p);
  }
  IO* base = (IO*)gp.g_sdf + first;
  double* base64 = (double*)gp.g_sdf + first;
  E* l = (E*)cx.lds();
#pragma unroll
  for (int k = 0; k < C; ++k)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      E e0, e1;
      e0.idx = tap_i[k][2 * r]; e0.val = tap_v[k][2 * r];
      e1.idx = tap_i[k][2 * r + 1]; e1.val = tap_v[k][2 * r + 1];
      l[2 * lane] = e0; l[2 * lane + 1] = e1;
      cx.lds_sync();
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const E e = l[h * 64 + lane];
        const int src = h * 32 + (lane >> 1);                                  // the lane whose state this tap belongs to
        const int64_t bs = (int64_t)cx.wave() * TPW + (src / LPT);             // ... and its trajectory (per-sample grids)
        if (e.idx >= 0) {
          if (wide) cx.atomic_add(base64 + bs * gp.g_sdf_bstride + e.idx, (double)e.val, local);
          else cx.atomic_add(base + bs * gp.g_sdf_bstride + e.idx, e.val, local);
        }
      }
      cx.lds_sync();
    }
}

// GSDF_SPARSE: the taps of the lane's C states as COO entries, position q = ((pass B + b) n + g) 4 + t of the value array and of each of the four index rows
// (b, 0, y, x).  Every entry of an existing state is written (value 0 where the hinge is inactive or the trajectory sat the pass out: explicit zeros are legal in
// an uncoalesced COO tensor; so are the duplicates of neighbouring states that share a pixel), nothing else: no atomics, no zero fill, no O(B H W) buffer.
// A lane's C states are 4 C consecutive entries: 16 C bytes of fp32 values and 32 C bytes per index row, written as 16-byte vectors.
template <int C, typename IO, typename Taps>
DGP_HD void sdf_emit_sparse(const GnParams& p, const GnGradParams& gp, int64_t b, int g0, bool traj_ok, int pass, int64_t nnz, const Taps& taps,
                            const int32_t (&tap_i)[C][4], const IO (&tap_v)[C][4]) {
  typedef long long i64x2 __attribute__((vector_size(16)));
  struct __attribute__((aligned(16))) V4 { IO v[4]; };
  IO* vals = (IO*)gp.g_sdf;
  int64_t* idx = gp.g_sdf_idx;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int g = g0 + k;
    if (!(traj_ok && g < p.n)) continue;
    const int64_t q = (((int64_t)pass * p.B + b) * p.n + g) * 4;
    const bool on = tap_i[k][0] >= 0;
    V4 v;
#pragma unroll
    for (int t = 0; t < 4; ++t) v.v[t] = on ? tap_v[k][t] : (IO)0;
    *(V4*)(vals + q) = v;
    const long long y1 = taps.oa[k].y1, y2 = taps.oa[k].y2, x1 = taps.oa[k].x1, x2 = taps.oa[k].x2;
    i64x2* r0 = (i64x2*)(idx + q); i64x2* r1 = (i64x2*)(idx + nnz + q); i64x2* r2 = (i64x2*)(idx + 2 * nnz + q); i64x2* r3 = (i64x2*)(idx + 3 * nnz + q);
    const i64x2 bb = {(long long)b, (long long)b}, zz = {0, 0};
    r0[0] = bb; r0[1] = bb; r1[0] = zz; r1[1] = zz;
#if DGP_TL != 0      // (preprocessor, not `if`: the standard units keep the exact code they were verified with -- even a constant-false branch moved 228 backward kernels by a few instructions)
    if (grid_is_tiled(p)) {
      // a tiled grid tensor (B,1,Ht,Wt,4,4): SIX index rows -- b, 0, tile row, tile column, row in tile, column in tile  (taps (x1,y1), (x2,y1), (x1,y2), (x2,y2))
      i64x2* r4 = (i64x2*)(idx + 4 * nnz + q); i64x2* r5 = (i64x2*)(idx + 5 * nnz + q);
      r2[0] = i64x2{y1 >> 2, y1 >> 2}; r2[1] = i64x2{y2 >> 2, y2 >> 2};
      r3[0] = i64x2{x1 >> 2, x2 >> 2}; r3[1] = i64x2{x1 >> 2, x2 >> 2};
      r4[0] = i64x2{y1 & 3, y1 & 3}; r4[1] = i64x2{y2 & 3, y2 & 3};
      r5[0] = i64x2{x1 & 3, x2 & 3}; r5[1] = i64x2{x1 & 3, x2 & 3};
      continue;
    }
#endif
    r2[0] = i64x2{y1, y1}; r2[1] = i64x2{y2, y2};                                  // taps (x1,y1), (x2,y1), (x1,y2), (x2,y2)
    r3[0] = i64x2{x1, x2}; r3[1] = i64x2{x1, x2};
  }
}

// LDS parking level of the adjoint solve's Woodbury elimination (gn_woodbury.h, PARK): the chain kernels also hold the running cotangent in LDS and
// leave L of S in registers, as the fused forward loop does -- four workgroups must fit the CU's 160 KB
template <int DOF, bool CHAIN> struct BwdParks {
  static constexpr int value = WbParks<DOF, MODE_BACKWARD_SOLVE>::value == 0 ? 0 : (CHAIN ? 2 : WbParks<DOF, MODE_BACKWARD_SOLVE>::value);
};

// The backward of DiffGPMP2Planner.unweighted_errors_batch at th + dtheta (plan_layer.py:374-388: start_goal_error = 1/2 |mu_s - x_0|^2 + 1/2 |mu_g - x_{n-1}|^2,
// gp_error = mean over the n - 1 factors of 1/2 |e|^2, obs_error = mean over the n states of 1/2 c^2), run as a PROLOGUE of the step's backward kernel: what
// round 4 launched as a kernel of its own in front of it (dgp_gn_step_errors_backward: +15 us of a 58 us replayed training iteration -- a second launch, the
// general chain rule evaluated with lambda = 0, a 4 MB workspace written and read back by the next launch).  No solve, no covariance weights, only the three
// cotangents: every lane hands the gradient rows of its states w.r.t. th + dtheta back in `gfold` (registers: the main program adds them to the dtheta cotangent in
// front of the adjoint solve, after which they are dead) AND parks them, with its shares of g_start / g_goal / g_eps, in the lane's LDS slots (FoldSlots, gn_lane.h):
// behind the solve, some 15 us later, the chain rule of the SAME lane adds them to its own rows.  The grid taps are scattered here (tap block g_sdf_pass0 + 1 of a
// sparse gradient).  Same formulas as the rows of gn_backward_lane_program with lambda = 0, ebar = 0.  d = 4 kernels only: the d = 6 backward kernels have neither
// the registers (the prologue's presence cost their scaled variant 5 of 57 us and put 23 more of them into the spill range of the known miscompiles) nor the LDS
// (34 KB already; four wavefronts per CU share 160 KB) -- the host gives d = 6 the two-launch form.  Hand-overs that were tried first: through g_th / g_start /
// g_goal / g_eps themselves -- every re-read in the chain rule then sits behind the previous row's stores, one exposed memory round trip per row (static backward
// with error cotangents 24.1 us against 17.1 us with the LDS slots); ordered by an agent-scope fence -- an L2 write-back per wavefront on gfx950: 57 instead of 36 us;
// by a workgroup-scope fence: 31 us.
template <int DOF, int LPT, int C, typename IO, typename Ctx>
DGP_HD void unweighted_errors_prologue(const GnParams& p, const GnGradParams& gp, Ctx& cx, const double (&th_rows)[C][2 * DOF], const double (&dq)[C][2 * DOF],
                                       const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF], double (&gfold)[C][2 * DOF]) {
  constexpr int D = 2 * DOF;
  constexpr int TPW = 64 / LPT;
  const int lane = cx.lane();
  const int j = lane_to_row<LPT>(lane & (LPT - 1));
  const int64_t b = (int64_t)cx.wave() * TPW + (lane / LPT);
  const int n = p.n;
  const bool traj_ok = b < p.B;
  const Nbr<LPT, 1, Ctx> nb(cx, j);
  const int g0 = j * C;
  const bool vec = p.vec_io != 0;
  double x[C][D];
  // (the th and dtheta rows and the means come from the main program, which loads them together with its own inputs: one memory round trip in front of the taps)
#pragma unroll
  for (int k = 0; k < C; ++k)
#pragma unroll
    for (int a = 0; a < D; ++a) x[k][a] = (double)(IO)((IO)th_rows[k][a] + (IO)dq[k][a]);      // the sum in the I/O type, as torch forms th_curr_b + dthetab
  const double gsg = (traj_ok && gp.f_unw_sg) ? ld<IO>(gp.f_unw_sg, b) : 0.0;
  const double ggp = (traj_ok && gp.f_unw_gp) ? ld<IO>(gp.f_unw_gp, b) / (double)(n - 1) : 0.0;
  const double gob = (traj_ok && gp.f_unw_obs) ? ld<IO>(gp.f_unw_obs, b) / (double)n : 0.0;
  const bool has_grid = p.sdf != nullptr && (gp.f_unw_obs != nullptr || gp.g_sdf != nullptr);      // wave-uniform
  LaneTaps<C, IO> taps;
  if (has_grid) lane_obstacle_loads<DOF, C, IO>(p, b, g0, traj_ok, x, taps);
  double x_prev[D], x_next[D];
#pragma unroll
  for (int a = 0; a < D; ++a) { x_prev[a] = nb.lo(x[C - 1][a]); x_next[a] = nb.hi(x[0][a]); }
  const double dt = p.dt;
  int32_t tap_i[C][4];
  IO tap_v[C][4];
#pragma unroll
  for (int k = 0; k < C; ++k) {
#pragma unroll
    for (int t = 0; t < 4; ++t) { tap_i[k][t] = -1; tap_v[k][t] = (IO)0; }
#pragma unroll
    for (int a = 0; a < D; ++a) gfold[k][a] = 0.0;
  }
  // nothing is stored to memory here: the rows and the shares of g_start / g_goal / g_eps go to the lane's LDS slots (FoldSlots) below
  double sh_s[D], sh_g[D], sh_e[D];
#pragma unroll
  for (int a = 0; a < D; ++a) { sh_s[a] = 0.0; sh_g[a] = 0.0; sh_e[a] = 0.0; }
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int g = g0 + k;
    if (!(traj_ok && g < n)) continue;
    const double* xk = x[k];
    const double* xm = (k == 0) ? x_prev : x[k > 0 ? k - 1 : 0];
    const double* xp = (k == C - 1) ? x_next : x[k < C - 1 ? k + 1 : 0];
    double gx[D];
#pragma unroll
    for (int a = 0; a < D; ++a) gx[a] = 0.0;
    if (g == 0 || g == n - 1) {                       // 1/2 |mu - x|^2
      const bool is_start = (g == 0);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double t = gsg * ((is_start ? mu_s[a] : mu_g[a]) - xk[a]);
        gx[a] -= t;
        if (is_start) sh_s[a] = t; else sh_g[a] = t;
      }
    }
    if (g < n - 1) {                                  // 1/2 |x_{g+1} - Phi x_g|^2 / (n - 1): this row's share is -Phi^T e
#pragma unroll
      for (int a = 0; a < DOF; ++a) {
        const double ep = ggp * (xp[a] - (xk[a] + dt * xk[DOF + a])), ev = ggp * (xp[DOF + a] - xk[DOF + a]);
        gx[a] -= ep;
        gx[DOF + a] -= dt * ep + ev;
      }
    }
    if (g > 0) {                                      // factor g-1 -> g: +e
#pragma unroll
      for (int a = 0; a < DOF; ++a) {
        gx[a] += ggp * (xk[a] - (xm[a] + dt * xm[DOF + a]));
        gx[DOF + a] += ggp * (xk[DOF + a] - xm[DOF + a]);
      }
    }
    double g_eps = 0.0;
    if (has_grid) {                                   // 1/2 c^2 / n, c = eps + r - dist (hinge)
      double c, hx, hy, d11, d21, d12, d22;
      ObsTaps tp;
      tap_values<C, IO>(taps, k, d11, d21, d12, d22);
      obstacle_finish(p, taps.oa[k], d11, d21, d12, d22, taps.eps[k], c, hx, hy, &tp);
      if (tp.act) {
        const double ga = gob * c;
        gx[0] -= ga * hx;
        gx[1] -= ga * hy;
        g_eps = ga;
        if (gp.g_sdf) {
          tap_i[k][0] = (int32_t)tp.i11; tap_v[k][0] = (IO)(-ga * (tp.wjc * tp.wja));
          tap_i[k][1] = (int32_t)tp.i21; tap_v[k][1] = (IO)(-ga * (tp.wjd * tp.wja));
          tap_i[k][2] = (int32_t)tp.i12; tap_v[k][2] = (IO)(-ga * (tp.wjc * tp.wjb));
          tap_i[k][3] = (int32_t)tp.i22; tap_v[k][3] = (IO)(-ga * (tp.wjd * tp.wjb));
        }
      }
    }
    sh_e[k] = g_eps;
#pragma unroll
    for (int a = 0; a < D; ++a) gfold[k][a] = gx[a];
  }
  static_assert(C <= D, "the epsilon shares of a lane's C states travel in one d-vector");
#pragma unroll
  for (int k = 0; k < C; ++k) fold_put<C, D>(cx.chain_lds(), lane, k, gfold[k]);
  fold_put<C, D>(cx.chain_lds(), lane, C, sh_s);
  fold_put<C, D>(cx.chain_lds(), lane, C + 1, sh_g);
  fold_put<C, D>(cx.chain_lds(), lane, C + 2, sh_e);
  if (gp.g_sdf && has_grid) {
    if (gp.g_sdf_mode == GSDF_SPARSE) sdf_emit_sparse<C, IO>(p, gp, b, g0, traj_ok, gp.g_sdf_pass0 + 1, (int64_t)gp.g_sdf_passes * p.B * p.n * 4, taps, tap_i, tap_v);
    else sdf_scatter_pairs<LPT, C, IO>(p, gp, cx, tap_i, tap_v);
  }
}

// CHAIN = false: the backward of ONE Gauss-Newton step (dgp_gn_step_backward, dgp_eval_errors_backward).
// CHAIN = true : the backward of the fused loop (dgp_gn_solve_backward): th_{k+1} = th_k + dtheta(th_k), k = 0 .. iters[b]-1, reversed inside the
//   kernel -- the running cotangent g (gradient w.r.t. th_{k+1}) stays in registers, every pass re-assembles Lambda(th_k) from the forward loop's
//   history (fp64 whatever the I/O type, so that dtheta_k = th_{k+1} - th_k is exact to rounding), solves the adjoint system, applies the per-factor
//   chain rule, g += J_k^T g; the start / goal gradients accumulate in registers, the grid gradient is scattered pass by pass.  Static covariances
//   only (what DiffGPMP2Planner.forward's fused path runs); a trajectory that stopped early sits out the passes it did not run (zero cotangent on
//   its final trajectory: the adjoint of a zero right-hand side is exactly zero).
template <int DOF, int LPT, int C, typename IO, int QK, bool CHAIN = false, typename Ctx>
DGP_HD void gn_backward_lane_program(const GnParams& p, const GnGradParams& gp, Ctx& cx) {
  constexpr int D = 2 * DOF;
  constexpr int TPW = 64 / LPT;
  const int lane = cx.lane();
  const int j = lane_to_row<LPT>(lane & (LPT - 1));
  const int64_t b = (int64_t)cx.wave() * TPW + (lane / LPT);
  const int n = p.n;
  const bool traj_ok = b < p.B;
  const Nbr<LPT, 1, Ctx> nb(cx, j);
  const int g0 = j * C;

  // QK_WB: the adjoint solve eliminates the interior rows through the Woodbury identity (gn_woodbury.h; same matrix as the forward
  // step, the cotangent as right-hand side); everything else of this program is the static-covariance (QK_STATIC) form
  struct NoStage {};
  typename std::conditional<is_wb(QK), WbStaged, NoStage>::type wbv;
  const bool vec = p.vec_io != 0;
  double x[C][D], gbar[C][D], lam[C][D], mu_s[D], mu_g[D];
  bool folded = false;
#ifndef DGP_BWD_FOLD
#define DGP_BWD_FOLD 1             // 0: compile the prologue out (A/B builds, profiles/tools/devbuild.py: what its presence costs the plain step backward)
#endif
  // the errors' prologue (dgp_gn_step_errors_backward in one launch) exists in the d = 4 single-step kernels; it hands its results to the main program through the
  // lane's LDS slots (FoldSlots)
  constexpr bool kFoldLds = !CHAIN && DOF == 2 && DGP_BWD_FOLD != 0;
  // d = 4: a fully populated wavefront block whose length fills the shape moves its row tensors (th, the dtheta cotangent, dtheta in;
  // g_th out) as full cache lines through the LDS staging block, as the forward step does (load / store_rows_through_lds; the output
  // write-through) instead of 16 bytes per lane at a 64-byte stride.  Wave-uniform.
  constexpr bool kBlockRows = WaveStore<IO, C, D>::kUsable && LPT != 32 && DOF == 2 && !CHAIN;
  bool block_rows = false;
  if constexpr (kBlockRows) block_rows = vec && n == LPT * C && ((int64_t)cx.wave() + 1) * TPW <= (int64_t)p.B;
  const int64_t wave_first_elem = (int64_t)cx.wave() * TPW * n * D;
  auto load_rows = [&](const void* src, double (&dst)[C][D]) {
    if (block_rows) {
      if constexpr (kBlockRows) load_rows_through_lds<IO, C, D>(cx, src, wave_first_elem, dst);
    } else {
      load_lane_rows<DOF, C, IO>(p, src, b, g0, traj_ok, vec, dst);
    }
  };
  if constexpr (!CHAIN) {
    load_rows(p.th, x);
    if (gp.th_addend) {             // wave-uniform: the point is th + dtheta, summed in the I/O type as torch forms th_curr_b + dthetab
      double dq[C][D];
      load_rows(gp.th_addend, dq);
#pragma unroll
      for (int k = 0; k < C; ++k)
#pragma unroll
        for (int a = 0; a < D; ++a) x[k][a] = (double)(IO)((IO)x[k][a] + (IO)dq[k][a]);
    }
  }
  ld_row<IO, D>(p.start, traj_ok ? b : 0, vec && p.vec_mu, mu_s);
  ld_row<IO, D>(p.goal, traj_ok ? b : 0, vec && p.vec_mu, mu_g);
  if constexpr (is_wb(QK)) wb_stage_issue<(QK == QK_WBR)>(p, cx, wbv);
  if constexpr (kFoldLds) {
    if (gp.f_addend) {             // wave-uniform: dgp_gn_step_errors_backward in one launch -- the errors' share of the dtheta cotangent arrives in gbar
      double dq[C][D], gd[C][D];
      load_rows(gp.f_addend, dq);
      const bool have_gd = gp.g_dtheta != nullptr;
      if (have_gd) load_rows(gp.g_dtheta, gd);      // (in flight under the prologue's arithmetic)
      unweighted_errors_prologue<DOF, LPT, C, IO>(p, gp, cx, x, dq, mu_s, mu_g, gbar);
      if (have_gd) {
#pragma unroll
        for (int k = 0; k < C; ++k)
#pragma unroll
          for (int a = 0; a < D; ++a) gbar[k][a] += gd[k][a];
      }
      folded = true;
    }
  }
#pragma unroll
  for (int k = 0; k < C; ++k)
#pragma unroll
    for (int a = 0; a < D; ++a) { if (!folded) gbar[k][a] = 0.0; lam[k][a] = 0.0; }
  if (gp.g_dtheta && !folded) load_rows(gp.g_dtheta, gbar);
  // dgp_gn_step_errors_backward, two-launch form (long trajectories: gn_long.h; kept here for callers that pass g_th_new without f_addend): the gradient
  // w.r.t. th + dtheta that the errors' backward left behind joins the dtheta cotangent (th + dtheta depends on dtheta with a unit Jacobian) and, further
  // down, the trajectory gradient (and on th likewise).  Both uses re-read it from memory behind a wave-uniform branch
  if constexpr (!CHAIN) {
    if (gp.g_th_new && !folded) {
      double gnew[C][D];
      load_rows(gp.g_th_new, gnew);
#pragma unroll
      for (int k = 0; k < C; ++k)
#pragma unroll
        for (int a = 0; a < D; ++a) gbar[k][a] += gnew[k][a];
    }
  }
  const bool have_gbar = gp.g_dtheta != nullptr || (!CHAIN && gp.g_th_new != nullptr) || folded;
  const bool lds_fold = kFoldLds && folded;      // wave-uniform
  LaneQ<D, C, QK> lq;              // generic covariances: Q^-1 of the lane's C + 1 GP factors, shared by the adjoint solve and the chain rule
  load_lane_Q<DOF, C, IO>(p, b, g0, traj_ok, lq);
  // ---- CHAIN: running cotangent (starts as the cotangent of th_final, loaded into gbar above), accumulated mean gradients, pass count
  // d = 6 chain kernels run register-lean: accumulated mean gradients in LDS, the means re-read per pass, the Woodbury table committed in front of the
  // loop, the trajectory rows re-read behind the solve (46.7 instead of 50.3 us per pass, 488 -> 188 B of scratch); d = 4 has no scratch either way and is
  // 0.5 us per pass faster with all of it in registers (15.9 against 16.4 us)
  constexpr bool kLean = CHAIN && D == 6;
  double gacc_s[(CHAIN && !kLean) ? D : 1], gacc_g[(CHAIN && !kLean) ? D : 1];
  int my_iters = 0, passes = 1;
  if constexpr (CHAIN && !kLean) {
#pragma unroll
    for (int a = 0; a < D; ++a) { gacc_s[a] = 0.0; gacc_g[a] = 0.0; }
  }
  if constexpr (CHAIN) {
    if constexpr (kLean) {         // the accumulated start / goal gradients live in the lane's LDS slots too (vectors C and C + 1)
      double z[D];
#pragma unroll
      for (int a = 0; a < D; ++a) z[a] = 0.0;
      chain_put<C, D>(cx.chain_lds(), lane, C, z); chain_put<C, D>(cx.chain_lds(), lane, C + 1, z);
    }
    my_iters = traj_ok ? gp.iters[b] : 0;
    my_iters = my_iters < 0 ? 0 : (my_iters > gp.chain_iters ? gp.chain_iters : my_iters);      // (device data the host cannot validate: never walk past the history)
    int mx = my_iters;                                     // passes = the most iterations any trajectory of this wavefront ran
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const int o = cx.fetch_i(mx, lane ^ m); mx = o > mx ? o : mx; }
    passes = mx;
  }
  // CHAIN: the running cotangent lives in lane-private LDS slots (cx.chain_lds()) between its two uses in a pass -- right-hand side of the adjoint
  // solve at the top, g += J^T g row by row at the bottom -- instead of in 2 C d registers across the solve and the chain rule
  if constexpr (CHAIN) {
#pragma unroll
    for (int k = 0; k < C; ++k) chain_put<C, D>(cx.chain_lds(), lane, k, gbar[k]);
  }
  if constexpr (kLean && is_wb(QK)) {
    // the Woodbury table goes to LDS once, in front of the passes (the single-step kernel commits it inside its solve, under the tap loads:
    // here that would keep the staged cells alive across the whole loop)
    wb_stage_commit<(QK == QK_WBR)>(cx, wbv);
  }
  bool first_pass = true;
#pragma unroll 1
  for (int it = passes - 1; it >= 0; --it) {
  bool pass_on = true;             // CHAIN: did this trajectory run iteration `it`
  auto load_th_it = [&]() {
    if constexpr (CHAIN) {
#pragma unroll
      for (int k = 0; k < C; ++k) {
        const bool valid = traj_ok && g0 + k < n;
        const int64_t row = valid ? b * n + g0 + k : 0;
        const double* h0 = gp.th_hist + ((int64_t)it * p.B * n + row) * D;
        double fin[D];
        ld_row<IO, D>(gp.th_final, row, vec, fin);
#pragma unroll
        for (int a = 0; a < D; ++a) x[k][a] = valid ? (pass_on ? h0[a] : fin[a]) : 0.0;
      }
    }
  };
  if constexpr (CHAIN) {
    if constexpr (kLean) {         // the start / goal means are re-read every pass (L2 hits) instead of being held in 4 d registers across the whole loop
      ld_row<IO, D>(p.start, traj_ok ? b : 0, vec && p.vec_mu, mu_s);
      ld_row<IO, D>(p.goal, traj_ok ? b : 0, vec && p.vec_mu, mu_g);
    }
    pass_on = it < my_iters;
    // th_it from the history; a trajectory that sits this pass out reads its FINAL trajectory instead (valid data; its cotangent is zeroed below)
    load_th_it();
  }
  // ---- lambda = Lambda^-1 gbar (skipped, wave-uniformly, when there is no dtheta cotangent)
  if (have_gbar) {
    ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
    SpdCheck<Ctx> ok = {&cx, 0};
    auto solve = [&](const double (&rhs)[C][D]) {
      if constexpr (is_wb(QK)) {
        static_assert(C == 4, "the Woodbury kernels are built for four states per lane");
        gn_linear_solve_wb<DOF, LPT, IO, true, DGP_BWD_COLWISE(D), (QK == QK_WBR), false, BwdParks<DOF, CHAIN>::value>(
            p, cx, b, j, traj_ok, x, mu_s, mu_g, rhs, lam, acc, ok, (!kLean && first_pass) ? &wbv : nullptr, [](const ErrAcc&) {});
      } else {
        gn_linear_solve<DOF, LPT, C, IO, true, QK, SinvStashBlocks<D, C, MODE_BACKWARD_SOLVE>::value>(p, cx, b, j, traj_ok, x, mu_s, mu_g, lq, rhs, lam, acc, ok);
      }
    };
    if constexpr (CHAIN) {
      double rhs[C][D];            // a trajectory that did not run iteration `it` contributes nothing: zero right-hand side, lambda == 0 exactly
#pragma unroll
      for (int k = 0; k < C; ++k) {
        chain_get<C, D>(cx.chain_lds(), lane, k, rhs[k]);
#pragma unroll
        for (int a = 0; a < D; ++a) rhs[k][a] = pass_on ? rhs[k][a] : 0.0;
      }
      solve(rhs);
    } else {
      solve(gbar);
    }
  }
  first_pass = false;
  // d = 6 block-elimination kernels (bit QK of DGP_BWD_RELOAD_D6; round 4): th and the start / goal means are read AGAIN behind the adjoint solve
  // (L2 hits) instead of being carried through it in 2 C d + 4 d registers.  Measured at B = 4096 (profiles/r04_kernel_variants.txt): static
  // covariances with velocity limits 64.0 -> 54.2 us (scratch 848 -> 408 B per lane), q_full tensors 122.9 -> 101.7 us; the per-state
  // (Kronecker) kernel 68.0 -> 69.3 and the Woodbury kernel 36.7 -> 37.6 us lose and keep their rows in registers.
#ifndef DGP_BWD_RELOAD_D6
#define DGP_BWD_RELOAD_D6 35     // QK_GENERAL | QK_STATIC | QK_SCALED
#endif
#ifndef DGP_BWD_RELOAD_D4
#define DGP_BWD_RELOAD_D4 0      // (measured: see profiles/r04_kernel_variants.txt)
#endif
  if constexpr (!CHAIN && !is_wb(QK) && ((((D == 6) ? (DGP_BWD_RELOAD_D6) : (DGP_BWD_RELOAD_D4)) >> QK) & 1) != 0) {
    if (have_gbar) {
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("" ::: "memory");
#endif
      load_rows(p.th, x);
      if (gp.th_addend) {
        double dq[C][D];
        load_rows(gp.th_addend, dq);
#pragma unroll
        for (int k = 0; k < C; ++k)
#pragma unroll
          for (int a = 0; a < D; ++a) x[k][a] = (double)(IO)((IO)x[k][a] + (IO)dq[k][a]);
      }
      ld_row<IO, D>(p.start, traj_ok ? b : 0, vec && p.vec_mu, mu_s);
      ld_row<IO, D>(p.goal, traj_ok ? b : 0, vec && p.vec_mu, mu_g);
    }
  }
  if constexpr (kLean) {
    // d = 6: the trajectory rows are read again behind the adjoint solve (L2 hits) instead of being carried through it in 2 C d registers
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
    load_th_it();
  }
  const double ebar = (traj_ok && gp.g_err_ext) ? ld<IO>(gp.g_err_ext, b) / p.M : 0.0;      // d L / d (M err_ext)
  // cotangents of the unweighted errors (dgp_eval_errors_backward): start_goal_error = 1/2 |mu_s - x_0|^2 + 1/2 |mu_g - x_{n-1}|^2
  // (plan_layer.py:384-388), gp_error = mean over the n-1 factors of 1/2 |e|^2 (:374-377), obs_error = mean over the n states of
  // 1/2 c^2 (:379-382) -- each the err_ext term of the same factor with the weight replaced by 1 (x the mean's 1/count)
  const double gsg = (traj_ok && gp.g_unw_sg) ? ld<IO>(gp.g_unw_sg, b) : 0.0;
  const double ggp = (traj_ok && gp.g_unw_gp) ? ld<IO>(gp.g_unw_gp, b) / (double)(n - 1) : 0.0;
  const double gob = (traj_ok && gp.g_unw_obs) ? ld<IO>(gp.g_unw_obs, b) / (double)n : 0.0;
  const bool has_grid = p.sdf != nullptr;      // wave-uniform; no grid (host-checked: nothing that reads it was requested): no obstacle factors

  // states / adjoints across the lane boundaries
  double x_prev[D], x_next[D], lam_prev[D], lam_next[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    x_prev[a] = nb.lo(x[C - 1][a]); x_next[a] = nb.hi(x[0][a]);
    lam_prev[a] = nb.lo(lam[C - 1][a]); lam_next[a] = nb.hi(lam[0][a]);
  }
  const double dt = p.dt;
  Sym<D> Qf;
  fixed_Qinv<DOF>(p, Qf);
  // everything the per-row chain rule reads from memory, loaded up front and branch-free: dtheta rows, epsilons / weights,
  // the four SDF taps of every state (one exposed memory round trip instead of one per divergent branch)
  double dthr[C][D], dth_next[D];
#pragma unroll
  for (int k = 0; k < C; ++k)
#pragma unroll
    for (int a = 0; a < D; ++a) dthr[k][a] = 0.0;
  if constexpr (CHAIN) {
    // dtheta_it = th_{it+1} - th_it, th_{it+1} = the next history row or, behind the last iteration, th_final (zero for a trajectory that sits the
    // pass out) -- loaded HERE, behind the solve, like the single-step kernel's dtheta rows
    const bool last = it + 1 >= my_iters;
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const bool valid = traj_ok && g0 + k < n;
      const int64_t row = valid ? b * n + g0 + k : 0;
      const int it1 = it + 1 < gp.chain_iters ? it + 1 : gp.chain_iters - 1;      // (never read past the last history row, not even speculatively: behind the last iteration th_final is used)
      const double* h1 = gp.th_hist + ((int64_t)it1 * p.B * n + row) * D;
      double fin[D];
      ld_row<IO, D>(gp.th_final, row, vec, fin);
#pragma unroll
      for (int a = 0; a < D; ++a) dthr[k][a] = valid ? ((pass_on && !last) ? h1[a] : fin[a]) - x[k][a] : 0.0;
    }
  } else {
    if (have_gbar) load_rows(gp.dtheta, dthr);
  }
#pragma unroll
  for (int a = 0; a < D; ++a) dth_next[a] = nb.hi(dthr[0][a]);
  LaneTaps<C, IO> taps;
  if (has_grid) lane_obstacle_loads<DOF, C, IO>(p, b, g0, traj_ok, x, taps);
  // SDF-gradient contributions of the lane's states: element offset inside the grid (-1: none -- hinge inactive or no such
  // row) and value for the four taps (x1,y1), (x2,y1), (x1,y2), (x2,y2)
  int32_t tap_i[C][4];
  IO tap_v[C][4];
  double gxs[kBlockRows ? C : 1][D];              // block_rows: the g_th rows, stored together behind the row loop
  double eold[D];                                 // the prologue's shares of the lane's epsilon gradients (LDS hand-over), else zero
#pragma unroll
  for (int a = 0; a < D; ++a) eold[a] = 0.0;
  if constexpr (kFoldLds) { if (lds_fold) fold_get<C, D>(cx.chain_lds(), lane, C + 2, eold); }
#pragma unroll
  for (int k = 0; k < C; ++k)
#pragma unroll
    for (int t = 0; t < 4; ++t) { tap_i[k][t] = -1; tap_v[k][t] = (IO)0; }

#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int g = g0 + k;
    if (!(traj_ok && g < n)) continue;
    const double* xk = x[k];
    const double* xm = (k == 0) ? x_prev : x[k > 0 ? k - 1 : 0];
    const double* xp = (k == C - 1) ? x_next : x[k < C - 1 ? k + 1 : 0];
    const double* lk = lam[k];
    const double* lm = (k == 0) ? lam_prev : lam[k > 0 ? k - 1 : 0];
    const double* lp = (k == C - 1) ? lam_next : lam[k < C - 1 ? k + 1 : 0];
    const double* dth = dthr[k];
    const double* dth_p = (k == C - 1) ? dth_next : dthr[k < C - 1 ? k + 1 : 0];      // rows past n-1 are zero
    double gx[D];
#pragma unroll
    for (int a = 0; a < D; ++a) gx[a] = 0.0;

    // ---- priors: e = mu - x, H = I, K = w I  ->  dL = w (lambda + ebar e)^T (dmu - dx)
    if (g == 0 || g == n - 1) {
      const bool is_start = (g == 0);
      void* gmu = is_start ? gp.g_start : gp.g_goal;
      const double w = is_start ? p.w_s : p.w_g;
      double tacc[kLean ? D : 1];
      if constexpr (kLean) chain_get<C, D>(cx.chain_lds(), lane, is_start ? C : C + 1, tacc);
      double told[D];              // the prologue's share of this mean's gradient (LDS hand-over), else zero
#pragma unroll
      for (int a = 0; a < D; ++a) told[a] = 0.0;
      if constexpr (kFoldLds) { if (lds_fold) fold_get<C, D>(cx.chain_lds(), lane, is_start ? C : C + 1, told); }
#pragma unroll
      for (int a = 0; a < D; ++a) {
        const double ea = (is_start ? mu_s[a] : mu_g[a]) - xk[a];
        const double t = w * (lk[a] + ebar * ea) + gsg * ea;
        gx[a] -= t;
        if constexpr (kLean) {
          tacc[a] += t;            // (zero for a trajectory that sits the pass out: lambda == 0, no error cotangents in the chain)
        } else if constexpr (CHAIN) {
          if (is_start) gacc_s[a] += t; else gacc_g[a] += t;
        } else {
          if (gmu) {
            if (gp.accumulate) st<IO>(gmu, b * D + a, t + ld<IO>(gmu, b * D + a));
            else st<IO>(gmu, b * D + a, t + told[a]);
          }
        }
      }
      if constexpr (kLean) chain_put<C, D>(cx.chain_lds(), lane, is_start ? C : C + 1, tacc);
    }
    // ---- GP factor (g -> g+1), owned by this row: e = x_{g+1} - Phi x_g, H = [Phi, -I], K = Q^-1
    if (g < n - 1) {
      Sym<D> Q;
      if constexpr (QK == QK_STATIC || QK == QK_SCALED || is_wb(QK)) fixed_Qinv<DOF>(p, Q);
      else if constexpr (QK == QK_KRON) kron_to_sym<DOF>(p, lq.c[k], Q);
      else Q = lq.q[k];
      double e[D], u[D], rho[D];
#pragma unroll
      for (int a = 0; a < DOF; ++a) {
        e[a] = xp[a] - (xk[a] + dt * xk[DOF + a]);
        e[DOF + a] = xp[DOF + a] - xk[DOF + a];
        u[a] = (lk[a] + dt * lk[DOF + a]) - lp[a];                         // u = Phi lambda_g - lambda_{g+1}
        u[DOF + a] = lk[DOF + a] - lp[DOF + a];
        rho[a] = e[a] - ((dth[a] + dt * dth[DOF + a]) - dth_p[a]);         // rho = e - (Phi dth_g - dth_{g+1})
        rho[DOF + a] = e[DOF + a] - (dth[DOF + a] - dth_p[DOF + a]);
      }
      // dL = u^T Q de + ebar e^T Qfix de,  de = dx_{g+1} - Phi dx_g   ->  this row's share: -Phi^T (Q u + ebar Qfix e)
      double t[D];
      if constexpr (QK == QK_STATIC || QK == QK_SCALED || is_wb(QK)) {
        // static covariances with a diagonal Q_c_inv: Q IS the fixed Q^-1 and entry (a, c) is zero unless a = c (mod dof) -- half the products
        // (two thirds at d = 6) vanish, and the two terms share the matrix
        double w[D];
        // (QK_SCALED: the factor's scalar weighs the system (u), not err_ext (ebar) or the unweighted error (ggp))
#pragma unroll
        for (int c = 0; c < D; ++c) {
          if constexpr (QK == QK_SCALED) w[c] = lq.s[k] * u[c] + ebar * e[c];
          else w[c] = u[c] + ebar * e[c];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) if (gp_nz<D>(a, c)) s += Q(a, c) * w[c];
          t[a] = s + ggp * e[a];
        }
      } else {
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) s += Q(a, c) * u[c] + ebar * Qf(a, c) * e[c];
          t[a] = s + ggp * e[a];
        }
      }
#pragma unroll
      for (int a = 0; a < DOF; ++a) {
        gx[a] -= t[a];
        gx[DOF + a] -= dt * t[a] + t[DOF + a];
      }
      // dL = u^T dQ^-1 rho  ->  gradient of the covariance input, element by element as autograd over gp_factor.py:65-73
      // gives it.  The reference solves through torch.cholesky, whose backward returns the SYMMETRISED gradient of
      // Lambda, so the Lambda part of dL/dQ^-1 is -(u v^T + v u^T)/2 (v = H dtheta = e - rho) and the eta part is u e^T.
      // For the symmetric covariances every mode of the reference produces this equals u rho^T in every directional
      // derivative; matching the convention makes the tensors agree entry by entry.
      if (gp.g_qc) {
        double Gm[D][D];
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int c = 0; c < D; ++c) {
            const double va = e[a] - rho[a], vc = e[c] - rho[c];
            Gm[a][c] = u[a] * e[c] - 0.5 * (u[a] * vc + va * u[c]);
          }
        if constexpr (QK == QK_SCALED) {
          // the gradient of the dof x dof blocks s_k I, entry by entry as for per-state tensors (it does not depend on Q)
          const int64_t base = (b * (n - 1) + g) * (DOF * DOF);
#pragma unroll
          for (int a = 0; a < DOF; ++a)
#pragma unroll
            for (int c = 0; c < DOF; ++c)
              st<IO>(gp.g_qc, base + a * DOF + c,
                     p.qa * Gm[a][c] + p.qb * (Gm[a][DOF + c] + Gm[DOF + a][c]) + p.qc_ * Gm[DOF + a][DOF + c]);
        } else if (p.qc_mode == QC_QFULL) {
          const int64_t base = (b * (n - 1) + g) * (D * D);
#pragma unroll
          for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) st<IO>(gp.g_qc, base + a * D + c, Gm[a][c]);
        } else if (p.qc_mode == QC_PERSTATE) {
          const int64_t base = (b * (n - 1) + g) * (DOF * DOF);
#pragma unroll
          for (int a = 0; a < DOF; ++a)
#pragma unroll
            for (int c = 0; c < DOF; ++c)
              st<IO>(gp.g_qc, base + a * DOF + c,
                     p.qa * Gm[a][c] + p.qb * (Gm[a][DOF + c] + Gm[DOF + a][c]) + p.qc_ * Gm[DOF + a][DOF + c]);
        }
      }
    }
    // ---- GP factor (g-1 -> g): this row's share is +(Q_{g-1} u_{g-1} + ebar Qfix e_{g-1})
    if (g > 0) {
      Sym<D> Q;
      if constexpr (QK == QK_STATIC || QK == QK_SCALED || is_wb(QK)) fixed_Qinv<DOF>(p, Q);
      else if constexpr (QK == QK_KRON) kron_to_sym<DOF>(p, (k == 0) ? lq.cm0 : lq.c[k > 0 ? k - 1 : 0], Q);
      else Q = (k == 0) ? lq.qm0 : lq.q[k > 0 ? k - 1 : 0];
      double e[D], u[D];
#pragma unroll
      for (int a = 0; a < DOF; ++a) {
        e[a] = xk[a] - (xm[a] + dt * xm[DOF + a]);
        e[DOF + a] = xk[DOF + a] - xm[DOF + a];
        u[a] = (lm[a] + dt * lm[DOF + a]) - lk[a];
        u[DOF + a] = lm[DOF + a] - lk[DOF + a];
      }
      if constexpr (QK == QK_STATIC || QK == QK_SCALED || is_wb(QK)) {
        double w[D];
#pragma unroll
        for (int c = 0; c < D; ++c) {
          if constexpr (QK == QK_SCALED) w[c] = ((k == 0) ? lq.sm0 : lq.s[k > 0 ? k - 1 : 0]) * u[c] + ebar * e[c];
          else w[c] = u[c] + ebar * e[c];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) if (gp_nz<D>(a, c)) s += Q(a, c) * w[c];
          gx[a] += s + ggp * e[a];
        }
      } else {
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) s += Q(a, c) * u[c] + ebar * Qf(a, c) * e[c];
          gx[a] += s + ggp * e[a];
        }
      }
    }
    // ---- obstacle factor: e = c, H = [hx, hy, 0..], K = omega
    if (!has_grid) {
      if (gp.g_eps && !gp.accumulate) st<IO>(gp.g_eps, b * n + g, 0.0);
      if (gp.g_obs_w) st<IO>(gp.g_obs_w, b * n + g, 0.0);
    } else {
      const double w = taps.ow[k];
      double c, hx, hy;
      ObsTaps tp;
      double d11, d21, d12, d22;
      tap_values<C, IO>(taps, k, d11, d21, d12, d22);
      obstacle_finish(p, taps.oa[k], d11, d21, d12, d22, taps.eps[k], c, hx, hy, &tp);
      double g_eps = 0.0, g_w = 0.0;
      if (tp.act) {
        const double u = hx * lk[0] + hy * lk[1];
        const double rho = c - (hx * dth[0] + hy * dth[1]);
        const double al = w * (rho * lk[0] - u * dth[0]);        // coefficient of d hx
        const double be = w * (rho * lk[1] - u * dth[1]);        // coefficient of d hy
        const double ga = u * w + (ebar * p.obs_w_fix + gob) * c;      // coefficient of d c  (c = eps + r - dist)
        // 1 / res: the host's value for d = 4 (a full fp64 division per state otherwise).  The d = 6 kernels keep the division: with the host's value their
        // register allocation moves and the static backward goes from 41.1 to 43.0 us (measured; the same bits either way)
        const double ir = (DOF == 2) ? p.inv_res : 1.0 / p.res;
        // hx = (wja (d21-d11) + wjb (d22-d12)) / res ; hy = -(wjc (d12-d11) + wjd (d22-d21)) / res ; px = ox + x/res ; py = oy - y/res
        gx[0] += be * (-tp.cross * ir * ir) - ga * hx;
        gx[1] += al * (-tp.cross * ir * ir) - ga * hy;
        g_eps = ga;
        g_w = u * rho;
        if (gp.g_sdf) {               // recorded here, scattered after the row loop (sdf_scatter_pairs below)
          const double wa = tp.wjc * tp.wja, wb = tp.wjd * tp.wja, wc = tp.wjc * tp.wjb, wd = tp.wjd * tp.wjb;
          tap_i[k][0] = (int32_t)tp.i11; tap_v[k][0] = (IO)(al * (-tp.wja * ir) + be * (tp.wjc * ir) - ga * wa);
          tap_i[k][1] = (int32_t)tp.i21; tap_v[k][1] = (IO)(al * (tp.wja * ir) + be * (tp.wjd * ir) - ga * wb);
          tap_i[k][2] = (int32_t)tp.i12; tap_v[k][2] = (IO)(al * (-tp.wjb * ir) + be * (-tp.wjc * ir) - ga * wc);
          tap_i[k][3] = (int32_t)tp.i22; tap_v[k][3] = (IO)(al * (tp.wjb * ir) + be * (-tp.wjd * ir) - ga * wd);
        }
      }
      if (gp.g_eps) {
        if (gp.accumulate) st<IO>(gp.g_eps, b * n + g, g_eps + ld<IO>(gp.g_eps, b * n + g));
        else st<IO>(gp.g_eps, b * n + g, g_eps + eold[k]);
      }
      if (gp.g_obs_w) st<IO>(gp.g_obs_w, b * n + g, g_w);
    }
    // ---- velocity limits: e = |v| - vmax, H = -sign(v) (piecewise constant), K = w_v
    if (p.flags & FLAG_VEL_LIMITS) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const double v = xk[DOF + a];
        const double av = fabs(v);
        if (av >= p.vmax[a]) {
          const double sg = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
          const double c = av - p.vmax[a];
          const double u = -sg * lk[DOF + a];
          gx[DOF + a] += p.w_v * (u + ebar * c) * sg;            // d c / d v = sign(v)
        }
      }
    }
    // ---- non-holonomic: e = vy cos - vx sin, H = [0,0,h2,h3,h4,0] as the reference writes it, K = w_d
    if constexpr (DOF == 3) if (p.flags & FLAG_NONHOLONOMIC) {
      const double th = xk[2], vx = xk[DOF], vy = xk[DOF + 1];
      const double sn = sin(th), cs = cos(th);
      const double e = vy * cs - vx * sn;
      const double h2 = -vy * sn + vx * cs, h3 = -sn, h4 = cs;
      const double u = h2 * lk[2] + h3 * lk[3] + h4 * lk[4];
      const double rho = e - (h2 * dth[2] + h3 * dth[3] + h4 * dth[4]);
      // dL = w_d [ sum_k (rho lambda_k - u dth_k) dh_k + (u + ebar e) de ]
      const double a2 = p.w_d * (rho * lk[2] - u * dth[2]);
      const double a3 = p.w_d * (rho * lk[3] - u * dth[3]);
      const double a4 = p.w_d * (rho * lk[4] - u * dth[4]);
      const double ae = p.w_d * (u + ebar * e);
      // d/dtheta: h2 -> -vy cos - vx sin, h3 -> -cos, h4 -> -sin, e -> -vy sin - vx cos
      gx[2] += a2 * (-vy * cs - vx * sn) + a3 * (-cs) + a4 * (-sn) + ae * (-vy * sn - vx * cs);
      gx[DOF] += a2 * cs + ae * (-sn);                             // d/dvx: h2 -> cos, e -> -sin
      gx[DOF + 1] += a2 * (-sn) + ae * cs;                         // d/dvy: h2 -> -sin, e -> cos
    }
    if constexpr (CHAIN) {
      // g_{th_it} = g_{th_{it+1}} + J_it^T g_{th_{it+1}}  (th_{it+1} = th_it + dtheta(th_it)); nothing to add for a pass the trajectory sat out
      double gk[D];
      chain_get<C, D>(cx.chain_lds(), lane, k, gk);
#pragma unroll
      for (int a = 0; a < D; ++a) gk[a] += pass_on ? gx[a] : 0.0;
      chain_put<C, D>(cx.chain_lds(), lane, k, gk);
    } else if (gp.g_th) {
      if (lds_fold) {                // (dgp_gn_step_errors_backward: the errors' share of the trajectory gradient -- from the lane's LDS slots, one launch, d = 4 ...
        if constexpr (kFoldLds) {
          double t[D];
          fold_get<C, D>(cx.chain_lds(), lane, k, t);
#pragma unroll
          for (int a = 0; a < D; ++a) gx[a] += t[a];
        }
      } else if (gp.g_th_new) {      //  ... or from memory: the second of two launches, d = 6 and long trajectories)
        double t[D];
        ld_row<IO, D>(gp.g_th_new, b * n + g, vec, t);
#pragma unroll
        for (int a = 0; a < D; ++a) gx[a] += t[a];
      }
      if (block_rows) {
        if constexpr (kBlockRows) {
#pragma unroll
          for (int a = 0; a < D; ++a) gxs[k][a] = gx[a];
        }
      } else {
        st_row<IO, D>(gp.g_th, b * n + g, vec, gx);
      }
    }
  }
  if constexpr (kBlockRows) {
    if (block_rows && gp.g_th) {
      store_rows_through_lds_wt<IO, C, D>(cx, gp.g_th, wave_first_elem, gxs);
#if !defined(__HIP_DEVICE_COMPILE__)
      cx.lds_sync();      // host emulator: its lanes are threads, and the pair staging below reuses the block the others may still be reading
#endif                    // (a wavefront issues its LDS instructions in order: nothing to wait for on the device)
    }
  }
  if (gp.g_sdf) {                                                           // wave-uniform
    if (gp.g_sdf_mode == GSDF_SPARSE) sdf_emit_sparse<C, IO>(p, gp, b, g0, traj_ok, gp.g_sdf_pass0 + it, (int64_t)gp.g_sdf_passes * p.B * p.n * 4, taps, tap_i, tap_v);
    else sdf_scatter_pairs<LPT, C, IO>(p, gp, cx, tap_i, tap_v);
  }
  }  // passes (one unless CHAIN)
  if constexpr (CHAIN) {
    // gradient w.r.t. the INITIAL trajectory, and the start / goal means' gradients summed over the passes
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const int g = g0 + k;
      double gk[D];
      chain_get<C, D>(cx.chain_lds(), lane, k, gk);
      if (traj_ok && g < n && gp.g_th) st_row<IO, D>(gp.g_th, b * n + g, vec, gk);
    }
    if (traj_ok && g0 == 0 && gp.g_start) {
      double t[D];
      if constexpr (kLean) chain_get<C, D>(cx.chain_lds(), lane, C, t);
      else { for (int a = 0; a < D; ++a) t[a] = gacc_s[a]; }
#pragma unroll
      for (int a = 0; a < D; ++a) st<IO>(gp.g_start, b * D + a, t[a]);
    }
    if (traj_ok && g0 <= n - 1 && n - 1 < g0 + C && gp.g_goal) {
      double t[D];
      if constexpr (kLean) chain_get<C, D>(cx.chain_lds(), lane, C + 1, t);
      else { for (int a = 0; a < D; ++a) t[a] = gacc_g[a]; }
#pragma unroll
      for (int a = 0; a < D; ++a) st<IO>(gp.g_goal, b * D + a, t[a]);
    }
  }
}

}  // namespace dgp
