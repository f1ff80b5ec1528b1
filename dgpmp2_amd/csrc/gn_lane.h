// gn_lane.h -- the per-lane program of the fused Gauss-Newton kernel.
//
// Mapping (MI355X, wave64): one trajectory occupies LPT consecutive lanes of a wavefront
// (LPT = 16/32/64, the next power of two >= n), lane i of the group OWNS support state i.
// Every lane
//   1. loads its state x_i, gets x_{i-1}, x_{i+1} from its neighbours with cross-lane moves,
//   2. evaluates the factors touching state i and writes the i-th block row of the block-tridiagonal
//      normal equations  (D_i sym dxd, U_i dxd, eta_i)  straight into registers,
//   3. takes part in a block parallel-cyclic-reduction (PCR) solve: log2(LPT) rounds, in each round a lane
//      inverts its own D_i, fetches (D^-1, U, eta) of lanes i-s and i+s, and eliminates them.
// Lambda never exists in memory.  All arithmetic is fp64 (the reference is fp64-only).
//
// The program is written against a tiny "lane context" (cross-lane fetch + ids) so that the very same
// source also compiles for the host, where tests/emul runs 64 lock-stepped threads as a wavefront
// emulator to check the kernel logic without a GPU.  The product path only ever uses the device context.
//
// Reference math (paths relative to /root/reference/diff_gpmp2/):
//   GP factor        gpmp2/gp/gp_factor.py:31-37 (Phi), :65-73 (Q^-1), :100-110 (error, H1=Phi, H2=-I)
//   prior factor     gpmp2/gp/prior_factor.py:15-18;  weights gpmp2/plan_layer.py:64-68
//   obstacle factor  gpmp2/obstacle/obstacle_factor.py:35-40, obstacle_cost.py:29-38, utils/sdf_utils.py:38-107
//   velocity limit   gpmp2/custom_factors/velocity_limit_factor.py:17-29
//   non-holonomic    gpmp2/custom_factors/nonholonomic_factor.py:16-30
//   system / solve   gpmp2/plan_layer.py:152-234;  errors :273-345, :374-388
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef DGP_HD
#define DGP_HD __host__ __device__ __forceinline__
#endif

namespace dgp {

enum { MODE_STEP = 0, MODE_SOLVE = 1, MODE_EVAL = 2 };
enum { QC_STATIC = 0, QC_PERSTATE = 1, QC_QFULL = 2 };
enum { FLAG_NONHOLONOMIC = 1u, FLAG_VEL_LIMITS = 2u };

// Kernel arguments (plain data, passed by value).
struct GnParams {
  int32_t B, n;
  int32_t sdf_rows, sdf_cols;
  int64_t sdf_bstride;
  int32_t qc_mode;
  uint32_t flags;
  int32_t max_iters;
  int32_t pad_;
  const void *th, *start, *goal, *sdf, *qc, *obs_w, *eps;
  void *dtheta, *err, *err_ext;
  int32_t* info;
  // MODE_SOLVE
  void *th_out, *err_hist, *errext_hist, *err_final;
  int32_t* iters;
  // MODE_EVAL
  void *unw_sg, *unw_gp, *unw_obs;
  // constants
  double dt;
  double qa, qb, qc_;        // 12 dt^-3, -6 dt^-2, 4 dt^-1      (gp_factor.py:66-68)
  double w_s, w_g;           // 1/K_s^2, 1/K_g^2                 (plan_layer.py:64-65)
  double reg;                // delta                            (plan_layer.py:219)
  double radius;             // sphere radius                    (obstacle_cost.py:30)
  double eps_static;         // obs_params['epsilon_dist']
  double obs_w_fix;          // 1/cost_sigma^2                   (plan_layer.py:74)
  double qc_fix[9];          // gp_params['Q_c_inv']
  double res, orig_px, orig_py;   // sdf_utils.py:57-58, obstacle_cost.py:34
  double w_d, w_v, vmax[2];  // 1/K_d^2, 1/K_v^2, (v_x, v_y)
  double M;                  // plan_layer.py:43-45
  double tol_delta;
};

// ---------------------------------------------------------------------------------------------------
// tiny fixed-size linear algebra, fully unrolled so that everything lives in registers
// ---------------------------------------------------------------------------------------------------
template <int D>
struct Sym {                 // symmetric DxD, packed upper triangle
  double v[D * (D + 1) / 2];
  static constexpr DGP_HD int idx(int i, int j) {
    return (i <= j) ? (i * D - (i * (i - 1)) / 2 + (j - i)) : (j * D - (j * (j - 1)) / 2 + (i - j));
  }
  DGP_HD double& operator()(int i, int j) { return v[idx(i, j)]; }
  DGP_HD double operator()(int i, int j) const { return v[idx(i, j)]; }
};

template <int D>
struct Mat { double v[D][D]; };

// A^-1 of an SPD matrix through LDL^T; ok=false when a pivot is <= 0 or NaN.
template <int D>
DGP_HD void sym_inverse(const Sym<D>& A, Sym<D>& Ai, bool& ok) {
  double L[D][D];      // unit lower (strict part used)
  double dinv[D];
  double dd[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double v = A(j, j);
#pragma unroll
    for (int k = 0; k < j; ++k) v -= L[j][k] * L[j][k] * dd[k];
    dd[j] = v;
    ok = ok && (v > 0.0);
    dinv[j] = 1.0 / v;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double w = A(i, j);
#pragma unroll
      for (int k = 0; k < j; ++k) w -= L[i][k] * L[j][k] * dd[k];
      L[i][j] = w * dinv[j];
    }
  }
  // Mi = L^-1 (unit lower)
  double Mi[D][D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double w = -L[i][j];
#pragma unroll
      for (int k = j + 1; k < i; ++k) w -= L[i][k] * Mi[k][j];
      Mi[i][j] = w;
    }
  }
  // Ai = Mi^T diag(dinv) Mi
#pragma unroll
  for (int i = 0; i < D; ++i) {
#pragma unroll
    for (int j = i; j < D; ++j) {
      // k runs from j..D-1; Mi[k][k] = 1
      double w = (i == j) ? dinv[j] : Mi[j][i] * dinv[j];
#pragma unroll
      for (int k = j + 1; k < D; ++k) w += Mi[k][i] * dinv[k] * Mi[k][j];
      Ai(i, j) = w;
    }
  }
}

// x = A^-1 b through LDL^T
template <int D>
DGP_HD void sym_solve(const Sym<D>& A, const double (&b)[D], double (&x)[D], bool& ok) {
  double L[D][D];
  double dd[D], dinv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double v = A(j, j);
#pragma unroll
    for (int k = 0; k < j; ++k) v -= L[j][k] * L[j][k] * dd[k];
    dd[j] = v;
    ok = ok && (v > 0.0);
    dinv[j] = 1.0 / v;
#pragma unroll
    for (int i = j + 1; i < D; ++i) {
      double w = A(i, j);
#pragma unroll
      for (int k = 0; k < j; ++k) w -= L[i][k] * L[j][k] * dd[k];
      L[i][j] = w * dinv[j];
    }
  }
  double y[D];
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double w = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) w -= L[i][k] * y[k];
    y[i] = w;
  }
#pragma unroll
  for (int i = D - 1; i >= 0; --i) {
    double w = y[i] * dinv[i];
#pragma unroll
    for (int k = i + 1; k < D; ++k) w -= L[k][i] * x[k];
    x[i] = w;
  }
}

// ---------------------------------------------------------------------------------------------------
// typed loads / stores of the I/O element type
// ---------------------------------------------------------------------------------------------------
template <typename IO> DGP_HD double ld(const void* p, int64_t i) { return (double)((const IO*)p)[i]; }
template <typename IO> DGP_HD void st(void* p, int64_t i, double v) { ((IO*)p)[i] = (IO)v; }

// ---------------------------------------------------------------------------------------------------
// bilinear SDF lookup + hinge, bit-for-bit the reference's fp64 op order (no FMA contraction here, so
// the `dist <= eps + r` decision equals the CPU oracle's on identical inputs).
//   utils/sdf_utils.py:57-94, gpmp2/obstacle/obstacle_cost.py:30,36-37
// ---------------------------------------------------------------------------------------------------
template <typename IO>
DGP_HD void obstacle_eval(const GnParams& p, const IO* grid, double x, double y, double eps, double& cost, double& hx,
                          double& hy) {
#pragma clang fp contract(off)
  const double res = p.res;
  double px = p.orig_px + x / res;                        // :61
  double py = p.orig_py - y / res;                        // :62
  double fpx = floor(px), fpy = floor(py);
  // floor -> int64 -> clamp (:64-72); saturate first so that huge |px| cannot overflow the conversion
  const double big = 1.0e9;
  double cx = fpx < -big ? -big : (fpx > big ? big : fpx);
  double cy = fpy < -big ? -big : (fpy > big ? big : fpy);
  if (!(cx == cx)) cx = 0.0;                              // NaN coordinates: any in-range index (result is NaN anyway)
  if (!(cy == cy)) cy = 0.0;
  int64_t x1 = (int64_t)cx, y1 = (int64_t)cy;
  int64_t x2 = x1 + 1, y2 = y1 + 1;                       // :65,67 (before clamping)
  const int64_t W = p.sdf_cols, H = p.sdf_rows;
  x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1);
  x2 = x2 < 0 ? 0 : (x2 > W - 1 ? W - 1 : x2);
  y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1);
  y2 = y2 < 0 ? 0 : (y2 > H - 1 ? H - 1 : y2);
  double d11 = (double)grid[y1 * W + x1];                 // dx1y1 (:76)
  double d21 = (double)grid[y1 * W + x2];                 // dx2y1
  double d12 = (double)grid[y2 * W + x1];                 // dx1y2
  double d22 = (double)grid[y2 * W + x2];                 // dx2y2
  double fx1 = (double)x1, fx2 = (double)x2, fy1 = (double)y1, fy2 = (double)y2;
  double wa = (fx2 - px) * (fy2 - py);                    // :81-84
  double wb = (px - fx1) * (fy2 - py);
  double wc = (fx2 - px) * (py - fy1);
  double wd = (px - fx1) * (py - fy1);
  double dist = wa * d11 + wb * d21 + wc * d12 + wd * d22;   // :90
  double Jx = (-1.0 * ((fy2 - py) * (d21 - d11) + (py - fy1) * (d22 - d12))) / res;   // :93
  double Jy = ((fx2 - px) * (d12 - d11) + (px - fx1) * (d22 - d21)) / res;            // :94
  double eps_tot = eps + p.radius;                        // obstacle_cost.py:30
  bool act = dist <= eps_tot;                             // :36
  cost = act ? (eps_tot - dist) : 0.0;
  hx = act ? (-1.0 * Jx) : 0.0;                           // :37
  hy = act ? (-1.0 * Jy) : 0.0;
}

// Q^-1 of one GP factor (gp_factor.py:65-73 / plan_layer.py:90)
template <int DOF, typename IO>
DGP_HD void load_Qinv(const GnParams& p, int64_t b, int f, Sym<2 * DOF>& Q) {
  constexpr int D = 2 * DOF;
  if (p.qc_mode == QC_QFULL) {
    const int64_t base = (b * (p.n - 1) + f) * (D * D);
#pragma unroll
    for (int i = 0; i < D; ++i)
#pragma unroll
      for (int j = i; j < D; ++j) Q(i, j) = ld<IO>(p.qc, base + i * D + j);
    return;
  }
  double C[DOF][DOF];
  if (p.qc_mode == QC_PERSTATE) {
    const int64_t base = (b * (p.n - 1) + f) * (DOF * DOF);
#pragma unroll
    for (int i = 0; i < DOF; ++i)
#pragma unroll
      for (int j = 0; j < DOF; ++j) C[i][j] = ld<IO>(p.qc, base + i * DOF + j);
  } else {
#pragma unroll
    for (int i = 0; i < DOF; ++i)
#pragma unroll
      for (int j = 0; j < DOF; ++j) C[i][j] = p.qc_fix[i * DOF + j];
  }
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = 0; j < DOF; ++j) {
      if (j >= i) {
        Q(i, j) = p.qa * C[i][j];
        Q(DOF + i, DOF + j) = p.qc_ * C[i][j];
      }
      Q(i, DOF + j) = p.qb * C[i][j];      // upper-right block (all entries are in the packed upper triangle)
    }
}

template <int DOF>
DGP_HD void fixed_Qinv(const GnParams& p, Sym<2 * DOF>& Q) {
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = 0; j < DOF; ++j) {
      const double c = p.qc_fix[i * DOF + j];
      if (j >= i) {
        Q(i, j) = p.qa * c;
        Q(DOF + i, DOF + j) = p.qc_ * c;
      }
      Q(i, DOF + j) = p.qb * c;
    }
}

template <int D>
DGP_HD double quad(const Sym<D>& Q, const double (&e)[D]) {      // e^T Q e
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double t = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) t += Q(i, j) * e[j];
    s += e[i] * t;
  }
  return s;
}

// ---------------------------------------------------------------------------------------------------
// factor evaluation for the state owned by this lane -> block row (Dm, U, r) and error partial sums
// ---------------------------------------------------------------------------------------------------
template <int DOF>
struct LaneEval {
  static constexpr int D = 2 * DOF;
  Sym<D> Dm;          // diagonal block
  Mat<D> U;           // coupling to the next state: block (i, i+1) = -Phi^T Q_i^-1
  double r[D];        // eta_i
  double e, eext;     // partial sums of err / err_ext (un-normalised)
  double usg, ugp, uobs;   // unweighted partials (plan_layer.py:374-388)
};

template <int DOF, typename IO, bool ASSEMBLE>
DGP_HD void eval_state(const GnParams& p, int64_t b, int i, bool valid, const double (&x)[2 * DOF],
                       const double (&xm)[2 * DOF], const double (&xp)[2 * DOF], LaneEval<DOF>& o) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  o.e = 0.0; o.eext = 0.0; o.usg = 0.0; o.ugp = 0.0; o.uobs = 0.0;
  if (ASSEMBLE) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
      o.r[a] = 0.0;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        o.U.v[a][c] = 0.0;
        if (c >= a) o.Dm(a, c) = (a == c) ? 1.0 : 0.0;     // padding lanes: identity row, x = 0
      }
    }
  }
  if (!valid) return;
  if (ASSEMBLE) {
#pragma unroll
    for (int a = 0; a < D; ++a) o.Dm(a, a) = p.reg;        // delta I (plan_layer.py:219)
  }
  // ---- start / goal priors: e = mu - x, H = +I, weight I/K^2 (prior_factor.py:15-18; plan_layer.py:64-68)
  if (i == 0 || i == n - 1) {
    const bool is_start = (i == 0);
    // n == 1 cannot happen (n >= 2 enforced by the host); i==0 and i==n-1 are distinct lanes
    const void* mu = is_start ? p.start : p.goal;
    const double w = is_start ? p.w_s : p.w_g;
    double s2 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const double ea = ld<IO>(mu, b * D + a) - x[a];
      s2 += ea * ea;
      if (ASSEMBLE) { o.Dm(a, a) += w; o.r[a] += w * ea; }
    }
    o.e += 0.5 * w * s2; o.eext += 0.5 * w * s2; o.usg += 0.5 * s2;
  }
  // ---- GP factor (i -> i+1), owned by lane i: e = x_{i+1} - Phi x_i (gp_factor.py:105)
  const double dt = p.dt;
  if (i < n - 1) {
    Sym<D> Q;
    load_Qinv<DOF, IO>(p, b, i, Q);
    double e[D];
#pragma unroll
    for (int a = 0; a < DOF; ++a) {
      e[a] = xp[a] - (x[a] + dt * x[DOF + a]);
      e[DOF + a] = xp[DOF + a] - x[DOF + a];
    }
    const double q = quad<D>(Q, e);
    o.e += 0.5 * q;
    double s2 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) s2 += e[a] * e[a];
    o.ugp += 0.5 * s2;
    if (p.qc_mode == QC_STATIC) {
      o.eext += 0.5 * q;
    } else {
      Sym<D> Qf;
      fixed_Qinv<DOF>(p, Qf);
      o.eext += 0.5 * quad<D>(Qf, e);                      // plan_layer.py:318-321
    }
    if (ASSEMBLE) {
      // PQ = Phi^T Q : rows pos = Q[pos,:], rows vel = dt*Q[pos,:] + Q[vel,:]
      double PQ[D][D];
#pragma unroll
      for (int a = 0; a < DOF; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          PQ[a][c] = Q(a, c);
          PQ[DOF + a][c] = dt * Q(a, c) + Q(DOF + a, c);
        }
      // Dm += PQ Phi : cols pos = PQ[:,pos], cols vel = dt*PQ[:,pos] + PQ[:,vel]   (upper triangle only)
#pragma unroll
      for (int a = 0; a < D; ++a) {
#pragma unroll
        for (int c = 0; c < DOF; ++c) {
          if (c >= a) o.Dm(a, c) += PQ[a][c];
          if (DOF + c >= a) o.Dm(a, DOF + c) += dt * PQ[a][c] + PQ[a][DOF + c];
        }
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          o.U.v[a][c] = -PQ[a][c];                         // block (i,i+1) = H1^T Q^-1 H2 = -Phi^T Q^-1
          t += PQ[a][c] * e[c];
        }
        o.r[a] += t;                                       // H1^T Q^-1 e
      }
    }
  }
  // ---- GP factor (i-1 -> i): contributes Q_{i-1}^-1 to D_i and -Q_{i-1}^-1 e_{i-1} to eta_i
  if (ASSEMBLE && i > 0) {
    Sym<D> Q;
    load_Qinv<DOF, IO>(p, b, i - 1, Q);
    double e[D];
#pragma unroll
    for (int a = 0; a < DOF; ++a) {
      e[a] = x[a] - (xm[a] + dt * xm[DOF + a]);
      e[DOF + a] = x[DOF + a] - xm[DOF + a];
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double t = 0.0;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        if (c >= a) o.Dm(a, c) += Q(a, c);
        t += Q(a, c) * e[c];
      }
      o.r[a] -= t;
    }
  }
  // ---- obstacle factor (obstacle_factor.py:35-40): sphere centre = x[0:2], H = H_e H_fk, H_fk = I_d[0:2,:]
  {
    const double eps = p.eps ? ld<IO>(p.eps, b * n + i) : p.eps_static;
    const double w = p.obs_w ? ld<IO>(p.obs_w, b * n + i) : p.obs_w_fix;
    const IO* grid = (const IO*)p.sdf + b * p.sdf_bstride;
    double c, hx, hy;
    obstacle_eval<IO>(p, grid, x[0], x[1], eps, c, hx, hy);
    o.e += 0.5 * w * c * c;
    o.eext += 0.5 * p.obs_w_fix * c * c;                   // plan_layer.py:329-332 (fixed weight, current eps)
    o.uobs += 0.5 * c * c;
    if (ASSEMBLE) {
      o.Dm(0, 0) += w * hx * hx; o.Dm(0, 1) += w * hx * hy; o.Dm(1, 1) += w * hy * hy;
      o.r[0] += w * hx * c; o.r[1] += w * hy * c;
    }
  }
  // ---- velocity-limit factor (velocity_limit_factor.py:17-29): '>=' (not '>'), H = -sign(v) e_{dof+a}
  if (p.flags & FLAG_VEL_LIMITS) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const double v = x[DOF + a];
      const double av = fabs(v);
      const bool act = av >= p.vmax[a];
      const double c = act ? (av - p.vmax[a]) : 0.0;
      const double sg = (v > 0.0) ? 1.0 : ((v < 0.0) ? -1.0 : 0.0);
      const double h = act ? -sg : 0.0;
      o.e += 0.5 * p.w_v * c * c; o.eext += 0.5 * p.w_v * c * c;
      if (ASSEMBLE) { o.Dm(DOF + a, DOF + a) += p.w_v * h * h; o.r[DOF + a] += p.w_v * h * c; }
    }
  }
  // ---- non-holonomic factor (nonholonomic_factor.py:16-30), state [x,y,th,vx,vy,w]; H as the reference writes it
  if (DOF == 3 && (p.flags & FLAG_NONHOLONOMIC)) {
    const double th = x[2], vx = x[DOF], vy = x[DOF + 1];
    const double sn = sin(th), cs = cos(th);
    const double e = vy * cs - vx * sn;
    const double h[3] = {-vy * sn + vx * cs, -sn, cs};    // columns 2,3,4
    o.e += 0.5 * p.w_d * e * e; o.eext += 0.5 * p.w_d * e * e;
    if (ASSEMBLE) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = a; c < 3; ++c) o.Dm(2 + a, 2 + c) += p.w_d * h[a] * h[c];
        o.r[2 + a] += p.w_d * h[a] * e;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// block PCR over the LPT lanes of one trajectory.  On return x = Lambda^-1 eta (this lane's d entries).
//   row i:  U_{i-s}^T x_{i-s} + D_i x_i + U_i x_{i+s} = r_i
//   eliminate x_{i-s}, x_{i+s} with rows i-s, i+s:
//     T2 = U_{i-s}^T D_{i-s}^-1        T = U_i D_{i+s}^-1
//     D_i' = D_i - T2 U_{i-s} - T U_i^T ;  r_i' = r_i - T2 r_{i-s} - T r_{i+s} ;  U_i' = -T U_{i+s}
//   (the lower coupling stays the transpose of the upper one, so only U is carried).
// ---------------------------------------------------------------------------------------------------
template <int D, int LPT, typename Ctx>
DGP_HD void pcr_solve(Ctx& cx, int i, Sym<D>& Dm, Mat<D>& U, double (&r)[D], double (&x)[D], bool& ok) {
  const int lane = cx.lane();
#pragma unroll 1
  for (int s = 1; s < LPT; s <<= 1) {
    Sym<D> Di;
    sym_inverse<D>(Dm, Di, ok);
    const bool has_l = (i >= s);
    const bool last = (2 * s >= LPT);
    const int src_l = has_l ? lane - s : lane;
    const int src_r = (i + s < LPT) ? lane + s : lane;      // no right neighbour => own U is already 0
    double rn[D];
    // ---- left neighbour
    {
      Sym<D> DiL;
      Mat<D> UL;
      double rL[D];
#pragma unroll
      for (int k = 0; k < D * (D + 1) / 2; ++k) DiL.v[k] = cx.fetch(Di.v[k], src_l);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        rL[a] = cx.fetch(r[a], src_l);
#pragma unroll
        for (int c = 0; c < D; ++c) {
          const double u = cx.fetch(U.v[a][c], src_l);
          UL.v[a][c] = has_l ? u : 0.0;
        }
      }
      // T2 = UL^T DiL
      double T2[D][D];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double t = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) t += UL.v[k][a] * DiL(k, c);
          T2[a][c] = t;
        }
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double t = r[a];
#pragma unroll
        for (int k = 0; k < D; ++k) t -= T2[a][k] * rL[k];
        rn[a] = t;
#pragma unroll
        for (int c = a; c < D; ++c) {
          double w = Dm(a, c);
#pragma unroll
          for (int k = 0; k < D; ++k) w -= T2[a][k] * UL.v[k][c];
          Dm(a, c) = w;
        }
      }
    }
    // ---- right neighbour
    {
      Sym<D> DiR;
      double rR[D];
#pragma unroll
      for (int k = 0; k < D * (D + 1) / 2; ++k) DiR.v[k] = cx.fetch(Di.v[k], src_r);
#pragma unroll
      for (int a = 0; a < D; ++a) rR[a] = cx.fetch(r[a], src_r);
      // T = U DiR
      double T[D][D];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double t = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) t += U.v[a][k] * DiR(k, c);
          T[a][c] = t;
        }
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double t = rn[a];
#pragma unroll
        for (int k = 0; k < D; ++k) t -= T[a][k] * rR[k];
        rn[a] = t;
#pragma unroll
        for (int c = a; c < D; ++c) {
          double w = Dm(a, c);
#pragma unroll
          for (int k = 0; k < D; ++k) w -= T[a][k] * U.v[c][k];
          Dm(a, c) = w;
        }
      }
      if (!last) {
        Mat<D> UR;
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int c = 0; c < D; ++c) UR.v[a][c] = cx.fetch(U.v[a][c], src_r);
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int c = 0; c < D; ++c) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < D; ++k) t -= T[a][k] * UR.v[k][c];
            U.v[a][c] = t;
          }
      }
    }
#pragma unroll
    for (int a = 0; a < D; ++a) r[a] = rn[a];
  }
  sym_solve<D>(Dm, r, x, ok);
}

// sum over the LPT lanes of one trajectory (butterfly); every lane gets the total
template <int LPT, typename Ctx>
DGP_HD double group_sum(Ctx& cx, double v) {
  const int lane = cx.lane();
#pragma unroll
  for (int m = LPT / 2; m >= 1; m >>= 1) v += cx.fetch(v, lane ^ m);
  return v;
}
template <int LPT, typename Ctx>
DGP_HD int group_or(Ctx& cx, int v) {
  const int lane = cx.lane();
#pragma unroll
  for (int m = LPT / 2; m >= 1; m >>= 1) v |= cx.fetch_i(v, lane ^ m);
  return v;
}

// ---------------------------------------------------------------------------------------------------
// the lane program
// ---------------------------------------------------------------------------------------------------
template <int DOF, int LPT, typename IO, int MODE, typename Ctx>
DGP_HD void gn_lane_program(const GnParams& p, Ctx& cx) {
  constexpr int D = 2 * DOF;
  constexpr int TPW = 64 / LPT;                 // trajectories per wavefront
  const int lane = cx.lane();
  const int i = lane & (LPT - 1);               // state index owned by this lane
  const int64_t b = (int64_t)cx.wave() * TPW + (lane / LPT);
  const int n = p.n;
  const bool traj_ok = b < p.B;
  const bool valid = traj_ok && (i < n);
  const int src_m = (i >= 1) ? lane - 1 : lane;
  const int src_p = (i + 1 < LPT) ? lane + 1 : lane;

  double x[D];
#pragma unroll
  for (int a = 0; a < D; ++a) x[a] = valid ? ld<IO>(p.th, (b * n + i) * D + a) : 0.0;

  if (MODE == MODE_EVAL) {
    double xm[D], xp[D];
#pragma unroll
    for (int a = 0; a < D; ++a) { xm[a] = 0.0; xp[a] = cx.fetch(x[a], src_p); }
    LaneEval<DOF> ev;
    eval_state<DOF, IO, false>(p, b, i, valid, x, xm, xp, ev);
    const double e = group_sum<LPT>(cx, ev.e), ee = group_sum<LPT>(cx, ev.eext);
    const double usg = group_sum<LPT>(cx, ev.usg), ugp = group_sum<LPT>(cx, ev.ugp), uobs = group_sum<LPT>(cx, ev.uobs);
    if (traj_ok && i == 0) {
      if (p.err) st<IO>(p.err, b, e / p.M);
      if (p.err_ext) st<IO>(p.err_ext, b, ee / p.M);
      if (p.unw_sg) st<IO>(p.unw_sg, b, usg);
      if (p.unw_gp) st<IO>(p.unw_gp, b, ugp / (double)(n - 1));     // torch.mean over the n-1 factors
      if (p.unw_obs) st<IO>(p.unw_obs, b, uobs / (double)n);
    }
    return;
  }

  const int iters_max = (MODE == MODE_SOLVE) ? p.max_iters : 1;
  bool active = traj_ok;
  int my_iters = 0;
  int bad = 0;
#pragma unroll 1
  for (int it = 0; it < iters_max; ++it) {
    double xm[D], xp[D];
#pragma unroll
    for (int a = 0; a < D; ++a) { xm[a] = cx.fetch(x[a], src_m); xp[a] = cx.fetch(x[a], src_p); }
    LaneEval<DOF> ev;
    eval_state<DOF, IO, true>(p, b, i, valid, x, xm, xp, ev);
    const double e = group_sum<LPT>(cx, ev.e), ee = group_sum<LPT>(cx, ev.eext);
    double dx[D];
    bool ok = true;
    pcr_solve<D, LPT>(cx, i, ev.Dm, ev.U, ev.r, dx, ok);
    bad |= (valid && !ok) ? 1 : 0;
    if (MODE == MODE_STEP) {
      if (valid) {
#pragma unroll
        for (int a = 0; a < D; ++a) st<IO>(p.dtheta, (b * n + i) * D + a, dx[a]);
      }
      if (traj_ok && i == 0) {
        if (p.err) st<IO>(p.err, b, e / p.M);
        if (p.err_ext) st<IO>(p.err_ext, b, ee / p.M);
      }
    } else {
      double s2 = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) s2 += valid ? dx[a] * dx[a] : 0.0;
      s2 = group_sum<LPT>(cx, s2);
      if (active) {
        if (i == 0) {
          if (p.err_hist) st<IO>(p.err_hist, b * (int64_t)p.max_iters + it, e / p.M);
          if (p.errext_hist) st<IO>(p.errext_hist, b * (int64_t)p.max_iters + it, ee / p.M);
        }
#pragma unroll
        for (int a = 0; a < D; ++a) x[a] += valid ? dx[a] : 0.0;         // th_new = th_curr + dtheta (:144)
        my_iters = it + 1;
        if (sqrt(s2) < p.tol_delta) active = false;                      // planner_utils.py:4
      }
      if (!cx.any(active)) break;
    }
  }
  bad = group_or<LPT>(cx, bad);
  if (p.info && traj_ok && i == 0) p.info[b] = bad;
  if (MODE == MODE_SOLVE) {
    if (valid) {
#pragma unroll
      for (int a = 0; a < D; ++a) st<IO>(p.th_out, (b * n + i) * D + a, x[a]);
    }
    if (traj_ok && i == 0 && p.iters) p.iters[b] = my_iters;
    if (p.err_final) {
      double xm[D], xp[D];
#pragma unroll
      for (int a = 0; a < D; ++a) { xm[a] = 0.0; xp[a] = cx.fetch(x[a], src_p); }
      LaneEval<DOF> ev;
      eval_state<DOF, IO, false>(p, b, i, valid, x, xm, xp, ev);
      const double e = group_sum<LPT>(cx, ev.e);
      if (traj_ok && i == 0) st<IO>(p.err_final, b, e / p.M);
    }
  }
}

}  // namespace dgp
