// gn_long.h -- the lane program for LONG trajectories (n > 256 support states): the same Gauss-Newton step as gn_lane.h with the
// number of rows per lane a RUNTIME value instead of a template parameter.
//
// The unrolled kernels of gn_lane.h hold a lane's C <= 4 rows in registers, which caps n at 64 * 4.  The reference accepts any
// total_time_step (plan_layer.py:30: dense N = d n).  Here ONE wavefront owns ONE trajectory, lane j owns the Cn = ceil(n / 64)
// consecutive rows j Cn .. j Cn + Cn - 1 and walks them in a loop:
//   a. forward sweep over its Cn - 1 interior rows, one row at a time -- states, covariance blocks and SDF taps are read from memory
//      as the row comes up, factors evaluated with eval_state (the generic row of gn_lane.h: every factor type, every covariance
//      mode), the streamed block elimination of gn_linear_solve (S_k, G_k, z_k and the running P_0, N_0, W_0, Pi); what the
//      recovery needs of a row (S_k^-1, z_k) is parked in the wavefront's dynamic LDS block, (Cn - 1) slots per lane;
//   b. the separator row (the lane's last), reduced with its own last interior row and the next lane's (P_0, N_0, W_0);
//   c. block PCR over the 64 lanes (pcr_solve of gn_lane.h);
//   d. interior recovery: forward recursion w_k (q_k = z_k - w_k overwrites z_k in LDS), then the back substitution, each dtheta
//      row stored (MODE_STEP) or applied to the trajectory (MODE_SOLVE: the state lives in th_out) as it is produced.
// Nothing is unrolled over rows, so the kernels are small (no spills) whatever n is; they are ~3x slower per state than the
// unrolled ones (memory round trips per row instead of once per launch) -- a correctness path for lengths the benchmark
// configurations never reach, not a tuned one.  Limit: the LDS block, (Cn - 1) (d (d + 1) / 2 + d) doubles per lane:
// n <= 1024 for d = 4, n <= 640 for d = 6 (dgp_host::kMaxStatesLong*).
#pragma once
#include "gn_lane.h"
#include "gn_backward.h"

namespace dgp {

template <int D> struct LongSlot {
  static constexpr int kDoubles = D * (D + 1) / 2 + D;                    // S_k^-1 (packed symmetric) + z_k
  static constexpr int kBytes = ((kDoubles * 8 + 15) / 16) * 16;
};
// lane stride of the LDS block in bytes: an odd number of 16-byte cells (conflict-free 128-bit accesses)
template <int D> DGP_HD int long_lane_stride(int cn) {
  const int cells = (cn > 1 ? cn - 1 : 1) * (LongSlot<D>::kBytes / 16);
  return ((cells & 1) ? cells : cells + 1) * 16;
}
template <int D> DGP_HD int long_lds_bytes(int n) { return 64 * long_lane_stride<D>((n + 63) / 64); }

template <int D, typename Ctx>
DGP_HD void long_put(Ctx& cx, int stride, int slot, const Sym<D>& S, const double (&z)[D]) {
  double* l = (double*)(cx.long_lds() + cx.lane() * stride + slot * LongSlot<D>::kBytes);
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) l[i] = S.v[i];
#pragma unroll
  for (int a = 0; a < D; ++a) l[D * (D + 1) / 2 + a] = z[a];
}
template <int D, typename Ctx>
DGP_HD void long_get(Ctx& cx, int stride, int slot, Sym<D>& S, double (&z)[D]) {
  const double* l = (const double*)(cx.long_lds() + cx.lane() * stride + slot * LongSlot<D>::kBytes);
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) S.v[i] = l[i];
#pragma unroll
  for (int a = 0; a < D; ++a) z[a] = l[D * (D + 1) / 2 + a];
}
template <int D, typename Ctx>
DGP_HD void long_put_z(Ctx& cx, int stride, int slot, const double (&z)[D]) {
  double* l = (double*)(cx.long_lds() + cx.lane() * stride + slot * LongSlot<D>::kBytes);
#pragma unroll
  for (int a = 0; a < D; ++a) l[D * (D + 1) / 2 + a] = z[a];
}

// Q^-1 of GP factor f (clamped to an existing factor: rows without one multiply it by a zero mask)
template <int DOF, typename IO>
DGP_HD void long_Q(const GnParams& p, int64_t b, int f, Sym<2 * DOF>& Q) {
  if (p.qc_mode == QC_STATIC) fixed_Qinv<DOF>(p, Q);
  else load_Qinv<DOF, IO>(p, b, imin32(imax32(f, 0), p.n - 2), Q);
}
// U_g = block (g, g+1) = -Phi^T Q_g (gp_factor.py:100-110), zero where row g has no successor
template <int DOF>
DGP_HD void long_coupling(const GnParams& p, int g, const Sym<2 * DOF>& Q, Mat<2 * DOF>& U) {
  constexpr int D = 2 * DOF;
  const double m = (g >= 0 && g < p.n - 1) ? 1.0 : 0.0;
#pragma unroll
  for (int a = 0; a < DOF; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      U.v[a][c] = -m * Q(a, c);
      U.v[DOF + a][c] = -m * (p.dt * Q(a, c) + Q(DOF + a, c));
    }
}

// one row of the system: states g-1, g, g+1 and the covariances of the factors (g-1 -> g), (g -> g+1) from memory
// `rhs` (backward kernel: the adjoint system) replaces eta by that tensor's row, or null.
template <int DOF, typename IO, bool ASSEMBLE>
DGP_HD void long_row(const GnParams& p, const void* th, int64_t b, int g, const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF],
                     Sym<2 * DOF>& Dk, Mat<2 * DOF>& Uk, double (&rk)[2 * DOF], ErrAcc& acc, const void* rhs = nullptr) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  const bool valid = g < n;
  const bool vec = p.vec_io != 0;
  double x[D], xm[D], xp[D];
  const int gc = valid ? g : 0;
  ld_row<IO, D>(th, b * n + gc, vec, x);
  ld_row<IO, D>(th, b * n + imax32(gc - 1, 0), vec, xm);
  ld_row<IO, D>(th, b * n + imin32(gc + 1, n - 1), vec, xp);
  Sym<D> Qown, Qm;
  long_Q<DOF, IO>(p, b, gc, Qown);
  long_Q<DOF, IO>(p, b, gc - 1, Qm);
  double ow = p.obs_w_fix, eps = p.eps_static, oc = 0.0, ohx = 0.0, ohy = 0.0;
  if (p.eps) eps = ld<IO>(p.eps, b * n + gc);
  if (p.obs_w) ow = ld<IO>(p.obs_w, b * n + gc);
  if (p.sdf) obstacle_eval<IO>(p, (const IO*)p.sdf + b * p.sdf_bstride, x[0], x[1], eps, oc, ohx, ohy);
  else ow = 0.0;
  if (!valid) { ow = 0.0; oc = 0.0; ohx = 0.0; ohy = 0.0; }
  eval_state<DOF, IO, ASSEMBLE>(p, b, g, valid, x, xm, xp, mu_s, mu_g, Qown, Qm, ow, oc, ohx, ohy, Dk, Uk, rk, acc);
  if (rhs) {
    double v[D];
    ld_row<IO, D>(rhs, b * n + gc, vec, v);
#pragma unroll
    for (int a = 0; a < D; ++a) rk[a] = valid ? v[a] : 0.0;
  }
}

// One linear solve of the trajectory in `th` (rows j cn .. j cn + cn - 1 of this lane).  `row_done(g, dx)` receives every row's
// dtheta (valid rows only), `before_pcr(acc)` is called once all factors have been evaluated.
// KEEP_X: every interior row's solution is left in its LDS slot (in place of z_k) for the caller (backward kernel: lambda).
template <int DOF, typename IO, bool KEEP_X, typename Ctx, typename Hook, typename RowDone>
DGP_HD void gn_long_solve(const GnParams& p, Ctx& cx, const void* th, int64_t b, int j, int cn, const double (&mu_s)[2 * DOF],
                          const double (&mu_g)[2 * DOF], ErrAcc& acc, SpdCheck<Ctx>& ok, Hook&& before_pcr, RowDone&& row_done,
                          const void* rhs = nullptr, double* xs_out = nullptr) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  const int g0 = j * cn;
  const int stride = long_lane_stride<D>(cn);
  const Nbr<64, 1, Ctx> nb(cx, j);
  // loop-carried state of the streamed elimination (see gn_linear_solve): X = U_{k-1}^T G_{k-1} (symmetric), y = G_{k-1}^T z_{k-1},
  // the running product Pi_k and the accumulators N_0, P_0 -- G_k and U_k themselves do not survive their iteration
  Mat<D> Pi;
  Sym<D> N0, X;
  double P0[D], y[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    P0[a] = 0.0; y[a] = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) { Pi.v[a][c] = 0.0; if (c >= a) { N0(a, c) = 0.0; X(a, c) = 0.0; } }
  }
  // ---- a. forward sweep over the interior rows
#pragma unroll 1
  for (int k = 0; k < cn - 1; ++k) {
    Sym<D> Dk; Mat<D> Uk; double zk[D];
    long_row<DOF, IO, true>(p, th, b, g0 + k, mu_s, mu_g, Dk, Uk, zk, acc, rhs);
#pragma unroll
    for (int a = 0; a < D; ++a) {                  // S_k = D_k - U_{k-1}^T G_{k-1},  z_k = r_k - G_{k-1}^T z_{k-1}   (X = 0, y = 0 for k = 0)
      zk[a] -= y[a];
#pragma unroll
      for (int c = a; c < D; ++c) Dk(a, c) -= X(a, c);
    }
    Sym<D> Si;
    sym_inverse<D>(Dk, Si, ok);
    long_put<D>(cx, stride, k, Si, zk);
    Mat<D> G;
    sym_times_mat<D>(Si, Uk, G);                   // G_k = S_k^-1 U_k
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) t += G.v[q][a] * zk[q];
      y[a] = t;                                    // y = G_k^T z_k  (= U_k^T S_k^-1 z_k)
#pragma unroll
      for (int c = a; c < D; ++c) {
        double w = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) w += Uk.v[q][a] * G.v[q][c];
        X(a, c) = w;                               // X = U_k^T G_k
      }
    }
    if (k == 0) {
      N0 = Si;                                     // N_0 = Pi_0 S_0^-1 Pi_0^T, Pi_0 = I
      sym_times_vec<D>(Si, zk, P0);                // P_0 = S_0^-1 z_0
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) Pi.v[a][c] = -G.v[a][c];       // Pi_1 = -G_0
    } else {
      Mat<D> Mk;                                   // M_k = Pi_k S_k^-1
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double t = 0.0;
#pragma unroll
          for (int q = 0; q < D; ++q) t += Pi.v[a][q] * Si(q, c);
          Mk.v[a][c] = t;
        }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = a; c < D; ++c) {               // N_0 += M_k Pi_k^T
          double t = N0(a, c);
#pragma unroll
          for (int q = 0; q < D; ++q) t += Mk.v[a][q] * Pi.v[c][q];
          N0(a, c) = t;
        }
      add_A_v<D>(P0, Mk, zk);                      // P_0 += Pi_k S_k^-1 z_k
      neg_A_B<D>(Mk, Uk, Pi);                      // Pi_{k+1} = -Pi_k G_k = -M_k U_k
    }
  }
  // ---- b. separator row -> one row of the 64-row reduced system
  Sym<D> Ds; Mat<D> Us; double rs[D];
  long_row<DOF, IO, true>(p, th, b, g0 + cn - 1, mu_s, mu_g, Ds, Us, rs, acc, rhs);
  if (cn > 1) {
#pragma unroll
    for (int a = 0; a < D; ++a) {                  // D_s -= U_{cn-2}^T G_{cn-2},  r_s -= U_{cn-2}^T S_{cn-2}^-1 z_{cn-2}
      rs[a] -= y[a];
#pragma unroll
      for (int c = a; c < D; ++c) Ds(a, c) -= X(a, c);
    }
    // first interior row of the next lane: x'_0 = P'_0 - N'_0 U_s^T x_s - W'_0 x'_s with W_0 = Pi_{cn-2} G_{cn-2} = -Pi_{cn-1}
    // (U_s == 0 where there is no next row, so what a lane without a right neighbour fetches is multiplied away)
    Sym<D> Nn; Mat<D> Wn; double Pn[D];
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) Nn.v[i] = nb.hi(N0.v[i]);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      Pn[a] = nb.hi(P0[a]);
#pragma unroll
      for (int c = 0; c < D; ++c) Wn.v[a][c] = -nb.hi(Pi.v[a][c]);
    }
    Mat<D> T, Ur;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) t += Us.v[a][q] * Nn(q, c);
        T.v[a][c] = t;
      }
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = a; c < D; ++c) {
        double t = Ds(a, c);
#pragma unroll
        for (int q = 0; q < D; ++q) t -= T.v[a][q] * Us.v[c][q];
        Ds(a, c) = t;
      }
    sub_A_v<D>(rs, Us, Pn);
    neg_A_B<D>(Us, Wn, Ur);
    Us = Ur;
  }
  before_pcr(acc);
  // ---- c. block PCR over the 64 lanes
  double xs[D];
  pcr_solve<D, 64, false>(cx, j, Ds, Us, rs, xs, ok);
  if (g0 + cn - 1 < n) row_done(g0 + cn - 1, xs);
  if (xs_out) {
#pragma unroll
    for (int a = 0; a < D; ++a) xs_out[a] = xs[a];
  }
  // ---- d. interior rows
  if (cn > 1) {
    double xps[D], w[D];
#pragma unroll
    for (int a = 0; a < D; ++a) xps[a] = nb.lo(xs[a]);
    {
      Sym<D> Qm; Mat<D> Um;
      long_Q<DOF, IO>(p, b, g0 - 1, Qm);
      long_coupling<DOF>(p, (g0 < n) ? g0 - 1 : -1, Qm, Um);      // L_0 = U_{g0-1}^T (zero for the first row of the trajectory and for padding lanes)
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) t += Um.v[c][a] * xps[c];
        w[a] = t;
      }
    }
    // forward: q_k = z_k - w_k,  w_{k+1} = -U_k^T S_k^-1 w_k
#pragma unroll 1
    for (int k = 0; k < cn - 1; ++k) {
      Sym<D> Si; double zk[D], qk[D], t[D];
      long_get<D>(cx, stride, k, Si, zk);
#pragma unroll
      for (int a = 0; a < D; ++a) qk[a] = zk[a] - w[a];
      long_put_z<D>(cx, stride, k, qk);
      if (k < cn - 2) {
        sym_times_vec<D>(Si, w, t);
        Sym<D> Q; Mat<D> U;
        long_Q<DOF, IO>(p, b, g0 + k, Q);
        long_coupling<DOF>(p, g0 + k, Q, U);
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) s -= U.v[c][a] * t[c];
          w[a] = s;
        }
      }
    }
    // backward: x_k = S_k^-1 (q_k - U_k x_{k+1})
    double xn[D];
#pragma unroll
    for (int a = 0; a < D; ++a) xn[a] = xs[a];
#pragma unroll 1
    for (int k = cn - 2; k >= 0; --k) {
      Sym<D> Si; double qk[D];
      long_get<D>(cx, stride, k, Si, qk);
      Sym<D> Q; Mat<D> U;
      long_Q<DOF, IO>(p, b, g0 + k, Q);
      long_coupling<DOF>(p, g0 + k, Q, U);
      sub_A_v<D>(qk, U, xn);
      sym_times_vec<D>(Si, qk, xn);
      if (KEEP_X) long_put_z<D>(cx, stride, k, xn);
      if (g0 + k < n) row_done(g0 + k, xn);
    }
  }
}

// errors only: sums over the lane's rows
template <int DOF, typename IO, typename Ctx>
DGP_HD void gn_long_eval(const GnParams& p, Ctx& cx, const void* th, int64_t b, int j, int cn, const double (&mu_s)[2 * DOF],
                         const double (&mu_g)[2 * DOF], ErrAcc& acc) {
  constexpr int D = 2 * DOF;
  Sym<D> Dk; Mat<D> Uk; double rk[D];
#pragma unroll 1
  for (int k = 0; k < cn; ++k) long_row<DOF, IO, false>(p, th, b, j * cn + k, mu_s, mu_g, Dk, Uk, rk, acc);
}

template <int DOF, typename IO, int MODE, typename Ctx>
DGP_HD void gn_long_program(const GnParams& p, Ctx& cx) {
  constexpr int D = 2 * DOF;
  const int j = cx.lane();
  const int64_t b = cx.wave();                    // one trajectory per wavefront (grid = B)
  const int n = p.n;
  const int cn = (n + 63) / 64;
  const bool vec = p.vec_io != 0;
  double mu_s[D], mu_g[D];
  ld_row<IO, D>(p.start, b, vec && p.vec_mu, mu_s);
  ld_row<IO, D>(p.goal, b, vec && p.vec_mu, mu_g);

  if (MODE == MODE_EVAL) {
    ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
    gn_long_eval<DOF, IO>(p, cx, p.th, b, j, cn, mu_s, mu_g, acc);
    const double e = group_sum<64>(cx, acc.e), ee = group_sum<64>(cx, acc.eext);
    const double usg = group_sum<64>(cx, acc.usg), ugp = group_sum<64>(cx, acc.ugp), uobs = group_sum<64>(cx, acc.uobs);
    if (j == 0) {
      if (p.err) st<IO>(p.err, b, div_M(p, e));
      if (p.err_ext) st<IO>(p.err_ext, b, div_M(p, ee));
      if (p.unw_sg) st<IO>(p.unw_sg, b, usg);
      if (p.unw_gp) st<IO>(p.unw_gp, b, ugp / (double)(n - 1));
      if (p.unw_obs) st<IO>(p.unw_obs, b, uobs / (double)n);
    }
    return;
  }

  SpdCheck<Ctx> ok = {&cx, 0};
  if (MODE == MODE_STEP) {
    ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
    gn_long_solve<DOF, IO, false>(p, cx, p.th, b, j, cn, mu_s, mu_g, acc, ok,
                           [&](const ErrAcc& a) {
                             const double e = group_sum<64>(cx, a.e), ee = group_sum<64>(cx, a.eext);
                             if (j == 0) {
                               if (p.err) st<IO>(p.err, b, div_M(p, e));
                               if (p.err_ext) st<IO>(p.err_ext, b, div_M(p, ee));
                             }
                           },
                           [&](int g, const double (&dx)[D]) { st_row<IO, D>(p.dtheta, b * n + g, vec, dx); });
  } else {
    // fused GN loop (diff_gpmp2_planner.py:122-156): the state lives in th_out; one trajectory per wavefront, so the convergence
    // test is wave-uniform
#pragma unroll 1
    for (int k = 0; k < cn; ++k) {
      const int g = j * cn + k;
      if (g < n) { double x[D]; ld_row<IO, D>(p.th, b * n + g, vec, x); st_row<IO, D>(p.th_out, b * n + g, vec, x); }
    }
    cx.mem_sync();
    int my_iters = 0;
#pragma unroll 1
    for (int it = 0; it < p.max_iters; ++it) {
      ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
      double e = 0.0, ee = 0.0, s2 = 0.0;
      gn_long_solve<DOF, IO, false>(p, cx, p.th_out, b, j, cn, mu_s, mu_g, acc, ok,
                             [&](const ErrAcc& a) { e = group_sum<64>(cx, a.e); ee = group_sum<64>(cx, a.eext); },
                             [&](int g, const double (&dx)[D]) {
                               double x[D];
                               ld_row<IO, D>(p.th_out, b * n + g, vec, x);
#pragma unroll
                               for (int a = 0; a < D; ++a) { x[a] += dx[a]; s2 += dx[a] * dx[a]; }      // th_new = th_curr + dtheta (:144)
                               st_row<IO, D>(p.th_out, b * n + g, vec, x);
                             });
      cx.mem_sync();                              // the rows other lanes just wrote are read by the next sweep (x_{g-1}, x_{g+1})
      s2 = group_sum<64>(cx, s2);
      if (j == 0) {
        if (p.err_hist) st<IO>(p.err_hist, b * (int64_t)p.max_iters + it, div_M(p, e));
        if (p.errext_hist) st<IO>(p.errext_hist, b * (int64_t)p.max_iters + it, div_M(p, ee));
      }
      my_iters = it + 1;
      if (sqrt(s2) < p.tol_delta) break;          // planner_utils.py:4 (the last dtheta IS applied)
    }
    if (j == 0 && p.iters) p.iters[b] = my_iters;
    if (p.err_final) {
      ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
      gn_long_eval<DOF, IO>(p, cx, p.th_out, b, j, cn, mu_s, mu_g, acc);
      const double e = group_sum<64>(cx, acc.e);
      if (j == 0) st<IO>(p.err_final, b, div_M(p, e));
    }
  }
  if (p.info && j == 0) p.info[b] = ok.bad != 0 ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------
// Backward (dgp_gn_step_backward / dgp_eval_errors_backward) for long trajectories: the adjoint solve through gn_long_solve with
// the cotangent as right-hand side (lambda of every interior row is left in its LDS slot), then the per-factor chain rule of
// gn_backward.h row by row -- the same formulas (see the derivation there), with the states, dtheta rows and covariance blocks
// read from memory as the row comes up and the SDF gradient scattered with one atomic per tap.
// ---------------------------------------------------------------------------------------------------
template <int DOF, typename IO, typename Ctx>
DGP_HD void gn_long_backward_program(const GnParams& p, const GnGradParams& gp, Ctx& cx) {
  constexpr int D = 2 * DOF;
  const int j = cx.lane();
  const int64_t b = cx.wave();
  const int n = p.n;
  const int cn = (n + 63) / 64;
  const int g0 = j * cn;
  const int stride = long_lane_stride<D>(cn);
  const bool vec = p.vec_io != 0;
  const Nbr<64, 1, Ctx> nb(cx, j);
  double mu_s[D], mu_g[D];
  ld_row<IO, D>(p.start, b, vec && p.vec_mu, mu_s);
  ld_row<IO, D>(p.goal, b, vec && p.vec_mu, mu_g);
  // ---- lambda = Lambda^-1 gbar (skipped, wave-uniformly, when there is no dtheta cotangent)
  double lam_s[D];                              // lambda of the lane's last row (the separator); interior rows: LDS slots
#pragma unroll
  for (int a = 0; a < D; ++a) lam_s[a] = 0.0;
  const bool solve = gp.g_dtheta != nullptr;
  if (solve) {
    ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
    SpdCheck<Ctx> ok = {&cx, 0};
    gn_long_solve<DOF, IO, true>(p, cx, p.th, b, j, cn, mu_s, mu_g, acc, ok, [](const ErrAcc&) {}, [](int, const double (&)[D]) {}, gp.g_dtheta, lam_s);
  }
  auto lam_row = [&](int k, double (&l)[D]) {   // lambda of the lane's row k
    if (!solve) {
#pragma unroll
      for (int a = 0; a < D; ++a) l[a] = 0.0;
    } else if (k == cn - 1) {
#pragma unroll
      for (int a = 0; a < D; ++a) l[a] = lam_s[a];
    } else {
      Sym<D> dummy;
      long_get<D>(cx, stride, k, dummy, l);
    }
  };
  // lambda across the lane boundaries: the previous lane's last row, the next lane's first row
  double lam_first[D], lam_prev[D], lam_next[D];
  lam_row(0, lam_first);
#pragma unroll
  for (int a = 0; a < D; ++a) { lam_prev[a] = nb.lo(lam_s[a]); lam_next[a] = nb.hi(lam_first[a]); }
  if (j == 0) {
#pragma unroll
    for (int a = 0; a < D; ++a) lam_prev[a] = 0.0;
  }
  if (j == 63) {
#pragma unroll
    for (int a = 0; a < D; ++a) lam_next[a] = 0.0;
  }
  const double ebar = gp.g_err_ext ? ld<IO>(gp.g_err_ext, b) / p.M : 0.0;
  const double gsg = gp.g_unw_sg ? ld<IO>(gp.g_unw_sg, b) : 0.0;
  const double ggp = gp.g_unw_gp ? ld<IO>(gp.g_unw_gp, b) / (double)(n - 1) : 0.0;
  const double gob = gp.g_unw_obs ? ld<IO>(gp.g_unw_obs, b) / (double)n : 0.0;
  const bool has_grid = p.sdf != nullptr;
  const double dt = p.dt;
  Sym<D> Qf;
  fixed_Qinv<DOF>(p, Qf);
  // SDF-gradient target: partial copies exactly as sdf_scatter_pairs picks them
  IO* gs_base = (IO*)gp.g_sdf;
  const bool local = gp.g_sdf_copies >= kMaxXcds;
  if (gp.g_sdf) {
    if (gp.g_sdf_copies > 1) {
      const int xcc = cx.xcc_id();
      const int per_xcd = gp.g_sdf_copies / kMaxXcds;
      const int copy = (per_xcd >= 1 && gp.g_sdf_copies % kMaxXcds == 0) ? (xcc % kMaxXcds) + kMaxXcds * ((cx.wave() / kMaxXcds) % per_xcd) : xcc % gp.g_sdf_copies;
      gs_base += (int64_t)copy * grid_elems(p);
    }
    gs_base += b * gp.g_sdf_bstride;
  }
  double lm[D];                                 // lambda_{g-1}, carried from row to row
#pragma unroll
  for (int a = 0; a < D; ++a) lm[a] = lam_prev[a];
#pragma unroll 1
  for (int k = 0; k < cn; ++k) {
    const int g = g0 + k;
    double lk[D], lp[D];
    lam_row(k, lk);
    if (k == cn - 1) {
#pragma unroll
      for (int a = 0; a < D; ++a) lp[a] = lam_next[a];
    } else {
      lam_row(k + 1, lp);
    }
    if (g < n) {
      double xk[D], xm[D], xp[D], dth[D], dth_p[D];
      ld_row<IO, D>(p.th, b * n + g, vec, xk);
      ld_row<IO, D>(p.th, b * n + imax32(g - 1, 0), vec, xm);
      ld_row<IO, D>(p.th, b * n + imin32(g + 1, n - 1), vec, xp);
#pragma unroll
      for (int a = 0; a < D; ++a) { dth[a] = 0.0; dth_p[a] = 0.0; }
      if (gp.g_dtheta) {
        ld_row<IO, D>(gp.dtheta, b * n + g, vec, dth);
        if (g + 1 < n) ld_row<IO, D>(gp.dtheta, b * n + g + 1, vec, dth_p);
      }
      double gx[D];
#pragma unroll
      for (int a = 0; a < D; ++a) gx[a] = 0.0;
      // ---- priors: e = mu - x, H = I, K = w I  ->  dL = w (lambda + ebar e)^T (dmu - dx)
      if (g == 0 || g == n - 1) {
        const bool is_start = (g == 0);
        void* gmu = is_start ? gp.g_start : gp.g_goal;
        const double w = is_start ? p.w_s : p.w_g;
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const double ea = (is_start ? mu_s[a] : mu_g[a]) - xk[a];
          const double t = w * (lk[a] + ebar * ea) + gsg * ea;
          gx[a] -= t;
          if (gmu) st<IO>(gmu, b * D + a, t);
        }
      }
      // ---- GP factor (g -> g+1): e = x_{g+1} - Phi x_g, H = [Phi, -I], K = Q^-1
      if (g < n - 1) {
        Sym<D> Q;
        long_Q<DOF, IO>(p, b, g, Q);
        double e[D], u[D], rho[D], t[D];
#pragma unroll
        for (int a = 0; a < DOF; ++a) {
          e[a] = xp[a] - (xk[a] + dt * xk[DOF + a]);
          e[DOF + a] = xp[DOF + a] - xk[DOF + a];
          u[a] = (lk[a] + dt * lk[DOF + a]) - lp[a];
          u[DOF + a] = lk[DOF + a] - lp[DOF + a];
          rho[a] = e[a] - ((dth[a] + dt * dth[DOF + a]) - dth_p[a]);
          rho[DOF + a] = e[DOF + a] - (dth[DOF + a] - dth_p[DOF + a]);
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) s += Q(a, c) * u[c] + ebar * Qf(a, c) * e[c];
          t[a] = s + ggp * e[a];
        }
#pragma unroll
        for (int a = 0; a < DOF; ++a) {
          gx[a] -= t[a];
          gx[DOF + a] -= dt * t[a] + t[DOF + a];
        }
        if (gp.g_qc) {                          // symmetrised convention of torch.cholesky's backward: see gn_backward.h
          double Gm[D][D];
#pragma unroll
          for (int a = 0; a < D; ++a)
#pragma unroll
            for (int c = 0; c < D; ++c) {
              const double va = e[a] - rho[a], vc = e[c] - rho[c];
              Gm[a][c] = u[a] * e[c] - 0.5 * (u[a] * vc + va * u[c]);
            }
          if (p.qc_mode == QC_QFULL) {
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
                st<IO>(gp.g_qc, base + a * DOF + c, p.qa * Gm[a][c] + p.qb * (Gm[a][DOF + c] + Gm[DOF + a][c]) + p.qc_ * Gm[DOF + a][DOF + c]);
          }
        }
      }
      // ---- GP factor (g-1 -> g): this row's share is +(Q_{g-1} u_{g-1} + ebar Qfix e_{g-1})
      if (g > 0) {
        Sym<D> Q;
        long_Q<DOF, IO>(p, b, g - 1, Q);
        double e[D], u[D];
#pragma unroll
        for (int a = 0; a < DOF; ++a) {
          e[a] = xk[a] - (xm[a] + dt * xm[DOF + a]);
          e[DOF + a] = xk[DOF + a] - xm[DOF + a];
          u[a] = (lm[a] + dt * lm[DOF + a]) - lk[a];
          u[DOF + a] = lm[DOF + a] - lk[DOF + a];
        }
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double s = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) s += Q(a, c) * u[c] + ebar * Qf(a, c) * e[c];
          gx[a] += s + ggp * e[a];
        }
      }
      // ---- obstacle factor: e = c, H = [hx, hy, 0..], K = omega
      if (!has_grid) {
        if (gp.g_eps) st<IO>(gp.g_eps, b * n + g, 0.0);
        if (gp.g_obs_w) st<IO>(gp.g_obs_w, b * n + g, 0.0);
      } else {
        double w = p.obs_w_fix, eps = p.eps_static;
        if (p.eps) eps = ld<IO>(p.eps, b * n + g);
        if (p.obs_w) w = ld<IO>(p.obs_w, b * n + g);
        double c, hx, hy;
        ObsTaps tp;
        obstacle_eval<IO>(p, (const IO*)p.sdf + b * p.sdf_bstride, xk[0], xk[1], eps, c, hx, hy, &tp);
        double g_eps = 0.0, g_w = 0.0;
        if (tp.act) {
          const double u = hx * lk[0] + hy * lk[1];
          const double rho = c - (hx * dth[0] + hy * dth[1]);
          const double al = w * (rho * lk[0] - u * dth[0]);
          const double be = w * (rho * lk[1] - u * dth[1]);
          const double ga = u * w + (ebar * p.obs_w_fix + gob) * c;
          const double ir = 1.0 / p.res;
          gx[0] += be * (-tp.cross * ir * ir) - ga * hx;
          gx[1] += al * (-tp.cross * ir * ir) - ga * hy;
          g_eps = ga;
          g_w = u * rho;
          if (gp.g_sdf) {
            const double wa = tp.wjc * tp.wja, wb = tp.wjd * tp.wja, wc = tp.wjc * tp.wjb, wd = tp.wjd * tp.wjb;
            cx.atomic_add(gs_base + tp.i11, (IO)(al * (-tp.wja * ir) + be * (tp.wjc * ir) - ga * wa), local);
            cx.atomic_add(gs_base + tp.i21, (IO)(al * (tp.wja * ir) + be * (tp.wjd * ir) - ga * wb), local);
            cx.atomic_add(gs_base + tp.i12, (IO)(al * (-tp.wjb * ir) + be * (-tp.wjc * ir) - ga * wc), local);
            cx.atomic_add(gs_base + tp.i22, (IO)(al * (tp.wjb * ir) + be * (-tp.wjd * ir) - ga * wd), local);
          }
        }
        if (gp.g_eps) st<IO>(gp.g_eps, b * n + g, g_eps);
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
            gx[DOF + a] += p.w_v * (u + ebar * c) * sg;
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
        const double a2 = p.w_d * (rho * lk[2] - u * dth[2]);
        const double a3 = p.w_d * (rho * lk[3] - u * dth[3]);
        const double a4 = p.w_d * (rho * lk[4] - u * dth[4]);
        const double ae = p.w_d * (u + ebar * e);
        gx[2] += a2 * (-vy * cs - vx * sn) + a3 * (-cs) + a4 * (-sn) + ae * (-vy * sn - vx * cs);
        gx[DOF] += a2 * cs + ae * (-sn);
        gx[DOF + 1] += a2 * (-sn) + ae * cs;
      }
      if (gp.g_th) st_row<IO, D>(gp.g_th, b * n + g, vec, gx);
    }
#pragma unroll
    for (int a = 0; a < D; ++a) lm[a] = lk[a];
  }
}

}  // namespace dgp
