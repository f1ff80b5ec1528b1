// gn_woodbury.h -- interior elimination of a lane's three interior rows through the Woodbury identity (kernel variant QK_WB).
//
// Included by gn_lane.h (uses its helpers).  Applies to static covariances with Q_c_inv = c I, no velocity-limit factors, C = 4
// states per lane and n >= 4 (BASELINE configs[1] and configs[3]; the reference YAML's n = 101).
//
// A lane's interior block (rows g0 .. g0+2, unknowns y; p = separator of the previous lane, s = own separator) is
//     Int y + Cp p + Cs s = r,      Int = K0 + sum_f w_f h_f h_f^T,
// where K0 -- delta I + the prior / GP terms of plan_layer.py:152-200 restricted to the three rows -- is the SAME constant
// matrix for every lane of every trajectory -- in up to four versions: ordinary rows; the first lane of a trajectory, whose row
// 0 carries the start prior and has no predecessor; the lane that holds the goal row n-1 as an interior row (when n is not a
// multiple of 4), padding rows behind it; lanes of padding rows only -- and only the single-state factors f (obstacle, obstacle_factor.py:35-40;
// non-holonomic, nonholonomic_factor.py:16-30) depend on the data: ONE rank-1 term per factor and state.  With Q_c_inv = c I,
// K0 does not couple different degrees of freedom and is identical for each: K = K0^-1 is a 6 x 6 matrix per dof
// (3 rows x {position, velocity}), inverted once on the host (dgp_host::create).  With hh_f = sqrt(w_f) h_f:
//     Int^-1 = K - K H M H^T K,     M = (I + H^T K H)^-1      (R x R, R = 3 factors (d = 4) or 6 (d = 6)).
// Everything the separator rows need follows from products with constants and ONE R x R inverse:
//     Epp = Cp^T Int^-1 Cp,  Eps = Cp^T Int^-1 Cs,  Ess = Cs^T Int^-1 Cs,  ep = Cp^T Int^-1 r,  es = Cs^T Int^-1 r,
// and after the PCR solve of the separators  y = Int^-1 (r - Cp p - Cs s).  Against the streamed block elimination of
// gn_linear_solve (three d x d inverses, the running products Pi_k, M_k, N_0, W_0) this is ~55 % fewer instructions in the
// local phases of the d = 4 kernel and ~65 % of the d = 6 kernel, and 24 instead of 42 (d = 6: 54 instead of 81) doubles kept
// across the PCR rounds.  The result differs from the block elimination by rounding only (measured: tests).
//
// The constants live in a table of WB_TYPES x WB_TYPE_DOUBLES doubles inside the kernel arguments; every wavefront copies it
// into LDS at kernel entry and a lane reads its version (first lane / other lanes) with ds_read -- LDS reads do not take
// VALU issue slots, which is what this kernel is short of.
#pragma once

namespace dgp {

// ---- host: the table (plain C++, no HIP) ----------------------------------------------------------------
// Per-dof blocks (2 x 2, index pv = 0 position, 1 velocity) of a GP factor under Q_c_inv = c I (gp_factor.py:31-37,65-73):
//   Q2 = c [[qa, qb],[qb, qc]],  U2 = -Phi2^T Q2 (block (i, i+1)),  A2 = Phi2^T Q2 Phi2,  Phi2 = [[1, dt],[0, 1]].
inline void wb_fill_table(GnParams& p, int dof) {
  p.wb_ok = 0;
  p.obs_w_sqrt = sqrt(p.obs_w_fix);
  p.w_d_sqrt = sqrt(p.w_d);
  for (int i = 0; i < WB_TYPES * WB_TYPE_DOUBLES; ++i) p.wb_tab[i] = 0.0;
  if (!p.qc_diag || (p.flags & FLAG_VEL_LIMITS)) return;
  const double c = p.qc_fix[0];
  for (int i = 1; i < dof; ++i)
    if (p.qc_fix[i * dof + i] != c) return;
  typedef long double L;
  const L dt = p.dt;
  const L Q2[2][2] = {{(L)p.qa * c, (L)p.qb * c}, {(L)p.qb * c, (L)p.qc_ * c}};
  L U2[2][2], A2[2][2];
  // U2 = -Phi2^T Q2 : row pos = -Q2[pos,:], row vel = -(dt Q2[pos,:] + Q2[vel,:])
  for (int k = 0; k < 2; ++k) { U2[0][k] = -Q2[0][k]; U2[1][k] = -(dt * Q2[0][k] + Q2[1][k]); }
  // A2 = -U2 Phi2 : col pos = -U2[:,pos], col vel = -(dt U2[:,pos] + U2[:,vel])
  for (int k = 0; k < 2; ++k) { A2[k][0] = -U2[k][0]; A2[k][1] = -(dt * U2[k][0] + U2[k][1]); }
  const int n = p.n;
  if (n < 4) return;                                                       // (the first lane would hold the goal row as an interior row)
  for (int type = 0; type < WB_TYPES; ++type) {
    // global index g0 of the lane's first row for this version (a representative: only g == 0, g vs n-1 and g >= n matter)
    int g0;
    if (type == WB_T_STD) g0 = 4;                                          // (whatever n is, the CONSTANTS are those of three ordinary rows)
    else if (type == WB_T_FIRST) g0 = 0;
    else if (type == WB_T_GOAL) g0 = ((n - 1) / 4) * 4;                    // the lane that holds row n-1
    else g0 = ((n + 3) / 4) * 4 + 4;                                       // beyond the trajectory
    const bool std_rows = (type == WB_T_STD);
    const bool first = (type == WB_T_FIRST);
    if (type == WB_T_GOAL && ((n - 1) % 4 == 3 || g0 == 0)) continue;      // the goal row is a separator (n a multiple of 4): version unused, left zero
    auto valid = [&](int k) { return std_rows || (g0 + k <= n - 1); };
    auto goal = [&](int k) { return !std_rows && (g0 + k == n - 1); };
    L K0[6][6] = {};
    for (int k = 0; k < 3; ++k) {
      const bool has_next = valid(k) && !goal(k);                          // GP factor (g -> g+1) exists: A2 on the diagonal, U2 to row k+1
      const bool has_prev = valid(k) && !(first && k == 0);                // GP factor (g-1 -> g) exists: Q2 on the diagonal
      for (int a = 0; a < 2; ++a)
        for (int e = 0; e < 2; ++e) {
          L v = 0;
          if (!valid(k)) v = (a == e) ? (L)1 : (L)0;                        // padding row: identity (static_diag: dbase = 1)
          else {
            if (has_next) v += A2[a][e];
            if (has_prev) v += Q2[a][e];
            if (a == e) v += (L)p.reg + ((first && k == 0) ? (L)p.w_s : (L)0) + (goal(k) ? (L)p.w_g : (L)0);   // delta I (plan_layer.py:219), priors (:64-65)
          }
          K0[2 * k + a][2 * k + e] = v;
          if (k < 2 && has_next) { K0[2 * k + a][2 * (k + 1) + e] = U2[a][e]; K0[2 * (k + 1) + e][2 * k + a] = U2[a][e]; }
        }
    }
    // K = K0^-1 : Gauss-Jordan with partial pivoting in long double
    L Aug[6][12];
    for (int i = 0; i < 6; ++i)
      for (int k = 0; k < 12; ++k) Aug[i][k] = (k < 6) ? K0[i][k] : ((k - 6 == i) ? (L)1 : (L)0);
    for (int col = 0; col < 6; ++col) {
      int piv = col;
      for (int i = col + 1; i < 6; ++i)
        if (fabsl(Aug[i][col]) > fabsl(Aug[piv][col])) piv = i;
      if (piv != col)
        for (int k = 0; k < 12; ++k) { const L tmp = Aug[col][k]; Aug[col][k] = Aug[piv][k]; Aug[piv][k] = tmp; }
      const L d = Aug[col][col];
      if (!(fabsl(d) > 0)) return;                                         // singular (cannot happen for valid configurations): keep the block elimination
      for (int k = 0; k < 12; ++k) Aug[col][k] /= d;
      for (int i = 0; i < 6; ++i)
        if (i != col) {
          const L f = Aug[i][col];
          for (int k = 0; k < 12; ++k) Aug[i][k] -= f * Aug[col][k];
        }
    }
    L K[6][6];
    for (int i = 0; i < 6; ++i)
      for (int k = 0; k < 6; ++k) K[i][k] = (L)0.5 * (Aug[i][6 + k] + Aug[k][6 + i]);
    // Cp = [L_0; 0; 0] with L_0 = U2^T (row 0 valid and not the first row of a trajectory), Cs = [0; 0; U2] (row 2 valid and not the goal)
    L Cp[6][2] = {}, Cs[6][2] = {};
    for (int a = 0; a < 2; ++a)
      for (int e = 0; e < 2; ++e) {
        if (valid(0) && !first) Cp[a][e] = U2[e][a];
        if (valid(2) && !goal(2)) Cs[4 + a][e] = U2[a][e];
      }
    L KCp[6][2], KCs[6][2];
    for (int i = 0; i < 6; ++i)
      for (int e = 0; e < 2; ++e) {
        L sp = 0, ss = 0;
        for (int k = 0; k < 6; ++k) { sp += K[i][k] * Cp[k][e]; ss += K[i][k] * Cs[k][e]; }
        KCp[i][e] = sp; KCs[i][e] = ss;
      }
    L Gpp[2][2], Gps[2][2], Gss[2][2];
    for (int a = 0; a < 2; ++a)
      for (int e = 0; e < 2; ++e) {
        L spp = 0, sps = 0, sss = 0;
        for (int k = 0; k < 6; ++k) { spp += Cp[k][a] * KCp[k][e]; sps += Cp[k][a] * KCs[k][e]; sss += Cs[k][a] * KCs[k][e]; }
        Gpp[a][e] = spp; Gps[a][e] = sps; Gss[a][e] = sss;
      }
    double* t = p.wb_tab + type * WB_TYPE_DOUBLES;
    for (int i = 0; i < 6; ++i)
      for (int k = 0; k < 6; ++k) t[WB_K6 + i * 6 + k] = (double)K[i][k];
    for (int i = 0; i < 6; ++i)
      for (int e = 0; e < 2; ++e) { t[WB_KCP + i * 2 + e] = (double)KCp[i][e]; t[WB_KCS + i * 2 + e] = (double)KCs[i][e]; }
    t[WB_GPP + 0] = (double)Gpp[0][0]; t[WB_GPP + 1] = (double)(0.5L * (Gpp[0][1] + Gpp[1][0])); t[WB_GPP + 2] = (double)Gpp[1][1];
    t[WB_GPS + 0] = (double)Gps[0][0]; t[WB_GPS + 1] = (double)Gps[0][1]; t[WB_GPS + 2] = (double)Gps[1][0]; t[WB_GPS + 3] = (double)Gps[1][1];
    t[WB_GSS + 0] = (double)Gss[0][0]; t[WB_GSS + 1] = (double)(0.5L * (Gss[0][1] + Gss[1][0])); t[WB_GSS + 2] = (double)Gss[1][1];
  }
  p.wb_ok = 1;
}

// The Woodbury kernels apply to a launch of shape (LPT, C) iff the handle's table is valid (which includes n >= 4) and C == 4.
DGP_HD bool wb_applies(const GnParams& p, int lpt, int c) { return p.wb_ok != 0 && p.qc_mode == QC_STATIC && c == 4 && p.n <= lpt * 4; }
// table version of the lane whose first row is g0 (WB_T_*): padding only / first lane / three ordinary rows / the goal row among them
DGP_HD int wb_lane_type(int g0, int n) { return g0 >= n ? WB_T_PAD : (g0 == 0 ? WB_T_FIRST : (g0 + 3 <= n - 1 ? WB_T_STD : WB_T_GOAL)); }

// ---- device / emulator ---------------------------------------------------------------------------------
// Every wavefront copies the table from the kernel arguments into its LDS block (144 16-byte cells; every version is shifted
// by 16 more bytes so that the addresses a 16-lane group reads from -- up to four versions -- hit different banks).  Two steps: the loads are
// ISSUED behind the start / goal rows and ahead of the SDF taps, and COMMITTED to LDS inside the first solve once the taps have
// arrived -- vector loads return in order, so by then the table cells are there too and nothing waits for them.  (A first
// version loaded and stored on the spot at kernel entry: two exposed kernel-argument round trips in front of everything else
// ate the whole gain; issued at entry and committed behind the th rows they still cost 0.16 us of the load phase.)  Lane l carries
// cells l, 64 + l and 128 + (l & 15) (duplicates write the same value to the same address: no branch).
struct WbStaged { double c0 __attribute__((vector_size(16))); double c1 __attribute__((vector_size(16))); double c2 __attribute__((vector_size(16))); };
// RAGGED = false (n = 4 LPT): only versions 0 and 1 are needed -- 72 cells, lane l carries cells l and 64 + (l & 7).
// RAGGED = true: all four versions -- 144 cells, lane l carries cells l, 64 + l and 128 + (l & 15).
template <bool RAGGED, typename Ctx>
DGP_HD void wb_stage_issue(const GnParams& p, Ctx& cx, WbStaged& w) {
  typedef double V2 __attribute__((vector_size(16)));
  static_assert(WB_TYPES * WB_TYPE_DOUBLES / 2 == 144 && WB_T_STD == 0 && WB_T_FIRST == 1, "cell-to-lane assignment");
  const int lane = cx.lane();
#if defined(__HIP_DEVICE_COMPILE__)
  const V2* src = (const V2*)cx.wb_source(p);      // (the kernel-argument segment is 64-byte aligned and wb_tab starts on a multiple of 16)
  w.c0 = src[lane];
  if constexpr (RAGGED) { w.c1 = src[64 + lane]; w.c2 = src[128 + (lane & 15)]; }
  else w.c1 = src[64 + (lane & 7)];
#else
  // host (tests/emul): GnParams lives on a stack that is only 8-byte aligned -- a 16-byte ALIGNED vector load of its table is undefined there (found by the
  // -ftrivial-auto-var-init=pattern build of the emulator, round 6: a general-protection fault once the stack layout shifted)
  const double* src = cx.wb_source(p);
  auto cell = [&](int c) { V2 v; __builtin_memcpy(&v, src + 2 * c, 16); return v; };
  w.c0 = cell(lane);
  if constexpr (RAGGED) { w.c1 = cell(64 + lane); w.c2 = cell(128 + (lane & 15)); }
  else w.c1 = cell(64 + (lane & 7));
#endif
}
template <bool RAGGED, typename Ctx>
DGP_HD void wb_stage_commit(Ctx& cx, const WbStaged& w) {
  typedef double V2 __attribute__((vector_size(16)));
  constexpr int kCellsPerType = WB_TYPE_DOUBLES / 2;
  char* dst = cx.wb_lds();
  const int lane = cx.lane();
  const int c1 = RAGGED ? 64 + lane : 64 + (lane & 7);
  *(V2*)(dst + (lane / kCellsPerType) * kWbTypeStrideBytes + (lane % kCellsPerType) * 16) = w.c0;
  *(V2*)(dst + (c1 / kCellsPerType) * kWbTypeStrideBytes + (c1 % kCellsPerType) * 16) = w.c1;
  if constexpr (RAGGED) {
    const int c2 = 128 + (lane & 15);
    *(V2*)(dst + (c2 / kCellsPerType) * kWbTypeStrideBytes + (c2 % kCellsPerType) * 16) = w.c2;
  }
  cx.lds_sync();
}

// Single-state factors that make up the low-rank part: type 0 = obstacle (slots x, y of the position half), type 1 =
// non-holonomic (slots theta, v_x, v_y; dof == 3 only).  Factor f = k * NF + type belongs to interior state k.
template <int DOF>
struct WbF {
  static constexpr int NF = (DOF == 3) ? 2 : 1;
  static constexpr int R = 3 * NF;
  static constexpr DGP_HD bool nz(int ft, int a) { return ft == 0 ? (a == 0 || a == 1) : (a == 2 || a == DOF || a == DOF + 1); }
  // does a factor of type ft touch the degree of freedom of slot c at all
  static constexpr DGP_HD bool touches(int ft, int c) { return ft == 0 ? ((c % DOF) == 0 || (c % DOF) == 1) : true; }
};

// 16-byte LDS cells per lane the PARK option of gn_linear_solve_wb needs: t (3 d) + non-zeros of the R factor rows [+ PARK == 1: the strict lower
// triangle of L; PARK == 2 leaves L in registers -- the fused loop, whose LDS block also holds the fp64 trajectory, has no room for it:
// four workgroups must fit the CU's 160 KB, or a 4096-trajectory batch no longer runs as ONE wave of workgroups]
template <int DOF, int PARK> struct WbParkCells {
  static constexpr int kNzRows = (DOF == 3) ? 3 * (2 + 3) : 3 * 2;
  static constexpr int value = PARK == 0 ? 0 : (3 * 2 * DOF + kNzRows + (PARK == 1 ? WbF<DOF>::R * (WbF<DOF>::R - 1) / 2 : 0) + 1) / 2;
};

// which kernels park (bit MODE of DGP_WB_PARK_MODES, d = 6 only; bit 3: the backward kernel's adjoint solve)
#ifndef DGP_WB_PARK_MODES
#define DGP_WB_PARK_MODES 11      // STEP, SOLVE, backward (profiles/r04_kernel_variants.txt: 26.7 -> 22.7 us, 21.2 -> 19.6 us per iteration, 42.1 -> 34.5 us)
#endif
template <int DOF, int MODE> struct WbParks { static constexpr int value = ((DOF == 3) && (((DGP_WB_PARK_MODES) >> MODE) & 1) != 0) ? (MODE == MODE_SOLVE ? 2 : 1) : 0; };

// S = I + H^T K H (R x R, SPD) in the form the elimination uses it: products B^T S^-1 B' as Y^T Z' with
//   R = 6 (primary template): S = L Dg L^T (unit lower L);  Y = L^-1 B,  Z = Dg^-1 Y;  S^-1 c = L^-T Dg^-1 L^-1 c
//          (forward substitutions instead of products with an explicit inverse: R (R - 1) / 2 instead of R^2 operations per column,
//          and a factorisation of ~R^3 / 6 operations instead of the ~200 of the 6 x 6 block inverse);
//   R = 3: S^-1 explicitly (one reciprocal; three dependent pivots would cost more than they save):  Y = B,  Z = S^-1 B.
// Z is produced one column at a time (zcol) where it is consumed: for R = 6 it is a row scaling of Y, and keeping Z_p, Z_s next to
// Y_p, Y_s (4 x 36 doubles) is what pushed the d = 6 kernel's Schur assembly into scratch.
template <int R>
struct WbSolver {
  double L[R][R], dinv[R];
  template <typename OK>
  DGP_HD void factor(const Sym<R>& S, OK& ok) {
    double dd[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      double v = S(j, j);
      double w[R];                       // w_k = L[j][k] d_k
#pragma unroll
      for (int k = 0; k < j; ++k) { w[k] = L[j][k] * dd[k]; v -= L[j][k] * w[k]; }
      dd[j] = v;
      ok.require(v > 0.0);
      dinv[j] = pivot_rcp(v);
#pragma unroll
      for (int i = j + 1; i < R; ++i) {
        double u = S(i, j);
#pragma unroll
        for (int k = 0; k < j; ++k) u -= L[i][k] * w[k];
        L[i][j] = u * dinv[j];
      }
    }
  }
  template <int N>
  DGP_HD void y(const double (&B)[R][N], double (&Y)[R][N]) const {
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
      for (int f = 0; f < R; ++f) {
        double v = B[f][c];
#pragma unroll
        for (int g = 0; g < f; ++g) v -= L[f][g] * Y[g][c];
        Y[f][c] = v;
      }
  }
  template <int N>
  DGP_HD void zcol(const double (&Y)[R][N], int c, double (&z)[R]) const {
#pragma unroll
    for (int f = 0; f < R; ++f) z[f] = dinv[f] * Y[f][c];
  }
  DGP_HD void solve(const double (&c)[R], double (&m)[R]) const {
    double yv[R];
#pragma unroll
    for (int f = 0; f < R; ++f) {
      double v = c[f];
#pragma unroll
      for (int g = 0; g < f; ++g) v -= L[f][g] * yv[g];
      yv[f] = v;
    }
#pragma unroll
    for (int f = R - 1; f >= 0; --f) {
      double v = dinv[f] * yv[f];
#pragma unroll
      for (int g = f + 1; g < R; ++g) v -= L[g][f] * m[g];
      m[f] = v;
    }
  }
};
template <>
struct WbSolver<3> {
  Sym<3> M;
  template <typename OK>
  DGP_HD void factor(const Sym<3>& S, OK& ok) { sym_inverse<3>(S, M, ok); }
  template <int N>
  DGP_HD void y(const double (&B)[3][N], double (&Y)[3][N]) const {
#pragma unroll
    for (int c = 0; c < N; ++c)
#pragma unroll
      for (int f = 0; f < 3; ++f) Y[f][c] = B[f][c];
  }
  template <int N>
  DGP_HD void zcol(const double (&Y)[3][N], int c, double (&z)[3]) const {
#pragma unroll
    for (int f = 0; f < 3; ++f) z[f] = M(f, 0) * Y[0][c] + M(f, 1) * Y[1][c] + M(f, 2) * Y[2][c];
  }
  DGP_HD void solve(const double (&c)[3], double (&m)[3]) const {
#pragma unroll
    for (int f = 0; f < 3; ++f) m[f] = M(f, 0) * c[0] + M(f, 1) * c[1] + M(f, 2) * c[2];
  }
};

// `staged`: the table cells this lane still has to commit to LDS (first solve of a launch), or null.
// COLWISE: order of the Schur assembly.  true: one column of S^-1 B at a time (no Z arrays); false: Z_p, Z_s formed up front, entries
// row by row.  Same arithmetic; which one the compiler turns into the better d = 6 kernel depends on the surrounding program
// (measured, B = 4096: d = 6 step 25.6 us row-wise / 26.8 us column-wise, d = 6 fused loop 34.9 / 22.2 us per iteration,
// d = 4 step 10.08 / 9.92 us) -- the callers choose.
// RAGGED: the trajectory does not fill the shape (n < 4 LPT): the lane picks one of four table versions from g0 and n; otherwise
// (every row of every lane exists) the first lane of a trajectory reads version 1, all others version 0.
// PF: scheduling fences at the phase boundaries (phase_fence in gn_lane.h: the STEP kernels only).
// PARK (d = 6, round 4): what only the interior recovery needs again -- t = K r (18 doubles), the scaled factor rows (15), L of S (15) -- is
// written to lane-private LDS slots (LdsPark in gn_lane.h: cx.stash(), 26 conflict-free 128-bit cells per lane) the moment the forward
// substitutions Y = L^-1 B are done, and read back after the PCR rounds.  The register allocator cannot do that itself (its only spill targets
// are the accumulation registers, two VALU moves per double each way, and scratch): left alone, the Schur assembly -- Y_p, Y_s, Z_p, Z_s, D_s,
// U_red: 207 doubles -- and the PCR rounds shuffled this state in and out of AGPRs and kept 22 doubles of it in scratch (244 B per lane, 17.6 MB
// of HBM traffic per launch).  configs[3] step: 26.4 -> 22.6 us, no scratch (profiles/r04_kernel_variants.txt: parking in AGPRs by hand, across
// the PCR rounds only or from the same point, gains nothing -- 26.3-26.5 us; LDS reads and writes take no VALU issue slot).
template <int DOF, int LPT, typename IO, bool RHS_OVERRIDE, bool COLWISE, bool RAGGED, bool PF = false, int PARK = 0, typename Ctx, typename Hook>
DGP_HD void gn_linear_solve_wb(const GnParams& p, Ctx& cx, int64_t b, int j, bool traj_ok, const double (&x)[4][2 * DOF],
                               const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF], const double (&rhs)[4][2 * DOF],
                               double (&dx)[4][2 * DOF], ErrAcc& acc, SpdCheck<Ctx>& ok, const WbStaged* staged, Hook&& before_pcr) {
  constexpr int D = 2 * DOF, C = 4;
  typedef WbF<DOF> F;
  constexpr int NF = F::NF, R = F::R;
  const int n = p.n;
  const Nbr<LPT, 1, Ctx> nb(cx, j);
  double x_prev[D], x_next[D];
#pragma unroll
  for (int a = 0; a < D; ++a) { x_prev[a] = nb.lo(x[C - 1][a]); x_next[a] = nb.hi(x[0][a]); }
  const int g0 = j * C;
  LaneFactors<C> lf;
  double rgp[C][D];              // prior + GP part of eta, computed while the SDF taps are in flight (as in gn_linear_solve)
  {
    LaneTaps<C, IO> taps;
    lane_obstacle_loads<DOF, C, IO>(p, b, g0, traj_ok, x, taps);
    DGP_STAMP_NOWAIT(p, cx, 8);
    phase_fence<PF>();
    double mu_ga[D];
#pragma unroll
    for (int a = 0; a < D; ++a) mu_ga[a] = mu_g[a];
    lane_after_addresses<C, IO, D>(taps, mu_ga);
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const double (&xm)[D] = (k == 0) ? x_prev : x[k > 0 ? k - 1 : 0];
      const double (&xp)[D] = (k == C - 1) ? x_next : x[k < C - 1 ? k + 1 : 0];
      static_rhs<DOF>(p, g0 + k, traj_ok && g0 + k < n, x[k], xm, xp, mu_s, mu_ga, rgp[k], acc);
    }
    double anchor[2 * C];
#pragma unroll
    for (int k = 0; k < C; ++k) { anchor[2 * k] = rgp[k][0]; anchor[2 * k + 1] = rgp[k][D - 1]; }
    DGP_STAMP_NOWAIT(p, cx, 9);
    phase_fence<PF>();
    lane_taps_use_after<C, IO, 2 * C>(taps, anchor);
    DGP_STAMP_NOWAIT(p, cx, 10);
    phase_fence<PF>();
    // the table: its loads were issued before the tap loads and vector loads return in order, so the cells are here
    if (staged) wb_stage_commit<RAGGED>(cx, *staged);
    lane_obstacle_finish<C, IO>(p, g0, traj_ok, taps, lf);
  }
  DGP_STAMP_NOWAIT(p, cx, 2);
  phase_fence<PF>();
  constexpr bool kFence = (D == 6);      // d = 6: keep the scheduler from interleaving the phases (see sched_fence)
  if constexpr (kFence) sched_fence();
  // this lane's version of the constants (ordinary rows / first lane / goal row inside / padding only)
  const char* tb = cx.wb_lds() + (RAGGED ? wb_lane_type(g0, n) : (j == 0 ? (int)WB_T_FIRST : (int)WB_T_STD)) * kWbTypeStrideBytes;
  auto T = [&](int i) -> double { return *(const double*)(tb + 8 * i); };
  const double m_prev0 = (traj_ok && g0 > 0 && (!RAGGED || g0 < n)) ? 1.0 : 0.0;        // L_0 = m_prev0 * u_fix^T
  const double m_last2 = (!RAGGED || (traj_ok && g0 + 2 < n - 1)) ? 1.0 : 0.0;           // U_2 = m_last2 * u_fix  (row g0+2 couples to the separator)

  // ---- the single-state factors of the three interior states: hh_f = sqrt(w_f) h_f, ch_f = sqrt(w_f) cost_f
  double hh[R][D], ch[R];
  // sqrt(w) of the obstacle factors: a per-state tensor only in the learned modes.  A real (wave-uniform) branch: written as a select the
  // compiler computes the three fp64 square roots speculatively and selects afterwards -- 65 VALU instructions of the static step for nothing.
  double sws[3];
  if (p.obs_w) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");                           // (keeps the branch a branch)
#endif
#pragma unroll
    for (int k = 0; k < 3; ++k) sws[k] = sqrt(lf.ow[k]);
  } else {
#pragma unroll
    for (int k = 0; k < 3; ++k) sws[k] = p.obs_w_sqrt;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double oc = lf.oc[k], ow = lf.ow[k];
    acc.e += 0.5 * ow * oc * oc;                           // obstacle factor (as eval_state_local)
    acc.eext += 0.5 * p.obs_w_fix * oc * oc;               // plan_layer.py:329-332
    acc.uobs += 0.5 * oc * oc;
    const double sw = sws[k];                              // (invalid rows have oc = h = 0)
#pragma unroll
    for (int a = 0; a < D; ++a) hh[k * NF][a] = 0.0;
    hh[k * NF][0] = sw * lf.ohx[k]; hh[k * NF][1] = sw * lf.ohy[k];
    ch[k * NF] = sw * oc;
    if constexpr (DOF == 3) {
#pragma unroll
      for (int a = 0; a < D; ++a) hh[k * NF + 1][a] = 0.0;
      ch[k * NF + 1] = 0.0;
      if (p.flags & FLAG_NONHOLONOMIC) {                   // nonholonomic_factor.py:16-30, H as the reference writes it (as eval_state_local)
        const double th = x[k][2], vx = x[k][DOF], vy = x[k][DOF + 1];
        const double sn = sin(th), cs = cos(th);
        const double e = vy * cs - vx * sn;
        acc.e += 0.5 * p.w_d * e * e; acc.eext += 0.5 * p.w_d * e * e;
        hh[k * NF + 1][2] = p.w_d_sqrt * (-vy * sn + vx * cs);
        hh[k * NF + 1][DOF] = p.w_d_sqrt * (-sn);
        hh[k * NF + 1][DOF + 1] = p.w_d_sqrt * cs;
        ch[k * NF + 1] = p.w_d_sqrt * e;
      }
    }
  }
  // ---- separator row (block elimination form: D_s, r_s with every factor of the state)
  Sym<D> Ds; Mat<D> Us; double rs[D];
  double m_s;
  static_diag<DOF>(p, g0 + C - 1, traj_ok && g0 + C - 1 < n, Ds, m_s);
#pragma unroll
  for (int a = 0; a < D; ++a) rs[a] = rgp[C - 1][a];
  eval_state_local<DOF, true>(p, x[C - 1], lf.ow[C - 1], lf.oc[C - 1], lf.ohx[C - 1], lf.ohy[C - 1], Ds, rs, acc);
  if (RHS_OVERRIDE) {
#pragma unroll
    for (int a = 0; a < D; ++a) rs[a] = (traj_ok && g0 + C - 1 < n) ? rhs[C - 1][a] : 0.0;
  }
  // ---- interior right-hand sides and t = K r
  double t[3][D];
  {
    double ri[3][D];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double v = rgp[k][a];
#pragma unroll
        for (int ft = 0; ft < NF; ++ft) if (F::nz(ft, a)) v += hh[k * NF + ft][a] * ch[k * NF + ft];
        ri[k][a] = RHS_OVERRIDE ? ((traj_ok && g0 + k < n) ? rhs[k][a] : 0.0) : v;
      }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int pv = 0; pv < 2; ++pv) {
        double kr[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) kr[q] = T(WB_K6 + (k * 2 + pv) * 6 + q);
#pragma unroll
        for (int dd = 0; dd < DOF; ++dd) {
          double s = 0.0;
#pragma unroll
          for (int q = 0; q < 6; ++q) s += kr[q] * ri[q >> 1][(q & 1) * DOF + dd];
          t[k][pv * DOF + dd] = s;
        }
      }
  }
  if constexpr (kFence) sched_fence();
  // ---- S = I + H^T K H, factorised
  WbSolver<R> sv;
  {
    Sym<R> Ah;
#pragma unroll
    for (int f = 0; f < R; ++f)
#pragma unroll
      for (int g = f; g < R; ++g) {
        const int kf = f / NF, ff = f % NF, kg = g / NF, fg = g % NF;
        double s = (f == g) ? 1.0 : 0.0;
#pragma unroll
        for (int pv = 0; pv < 2; ++pv)
#pragma unroll
          for (int pw = 0; pw < 2; ++pw) {
            double hs = 0.0;
            bool any = false;
#pragma unroll
            for (int dd = 0; dd < DOF; ++dd)
              if (F::nz(ff, pv * DOF + dd) && F::nz(fg, pw * DOF + dd)) { hs += hh[f][pv * DOF + dd] * hh[g][pw * DOF + dd]; any = true; }
            if (any) s += T(WB_K6 + (kf * 2 + pv) * 6 + kg * 2 + pw) * hs;
          }
        Ah(f, g) = s;
      }
    sv.factor(Ah, ok);
  }
  if constexpr (kFence) sched_fence();
  // ---- b_r = H^T t,  B_p = H^T K Cp,  B_s = H^T K Cs;  Schur pieces as Y^T Z (WbSolver), every entry folded into the reduced
  //      separator row the moment it exists: own block (Ess, es) subtracted here, the NEXT lane's (Epp', Eps', ep') fetched across
  //      lanes entry by entry -- no E block is ever live as a whole
  constexpr bool kPark = (PARK != 0) && (D == 6);
  constexpr bool kParkL = (PARK == 1);
  constexpr int kParkCells = kPark ? WbParkCells<DOF, PARK>::value : 1;
  {
    double Yp[R][D], Ys[R][D], yr[R][1];
    {
      double br[R][1], Bp[R][D], Bs[R][D];
#pragma unroll
      for (int f = 0; f < R; ++f) {
        const int kf = f / NF, ff = f % NF;
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < D; ++a) if (F::nz(ff, a)) s += hh[f][a] * t[kf][a];
        br[f][0] = s;
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double sp = 0.0, ss = 0.0;
#pragma unroll
          for (int a = 0; a < D; ++a)
            if (F::nz(ff, a) && (a % DOF) == (c % DOF)) {
              sp += hh[f][a] * T(WB_KCP + (kf * 2 + a / DOF) * 2 + c / DOF);
              ss += hh[f][a] * T(WB_KCS + (kf * 2 + a / DOF) * 2 + c / DOF);
            }
          Bp[f][c] = sp; Bs[f][c] = ss;
        }
      }
      sv.template y<1>(br, yr);
      sv.template y<D>(Bs, Ys);
      sv.template y<D>(Bp, Yp);
    }
    const bool has_next = (j + 1 < LPT);
    constexpr bool kZeroFill = Nbr<LPT, 1, Ctx>::kDpp;       // a DPP row shift already yields 0 where there is no next lane
    // right-hand sides:  Cp^T t = L_0^T t_0 = m_prev0 u_fix t_0 ;   Cs^T t = U_2^T t_2 = u_fix^T t_2
    {
      double zr[R];
      sv.template zcol<1>(yr, 0, zr);
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double sp = 0.0, ss = 0.0;
#pragma unroll
        for (int q = 0; q < D; ++q) if (gp_nz<D>(a, q)) { sp += p.u_fix[a * D + q] * t[0][q]; ss += p.u_fix[q * D + a] * t[2][q]; }
        sp *= m_prev0;
        if constexpr (RAGGED) ss *= m_last2;
#pragma unroll
        for (int f = 0; f < R; ++f) { sp -= Yp[f][a] * zr[f]; ss -= Ys[f][a] * zr[f]; }
        const double epn = nb.hi(sp);
        rs[a] -= ss + ((kZeroFill || has_next) ? epn : 0.0);
      }
    }
    if constexpr (kPark) {
      sched_fence();
      double flat[2 * kParkCells];
      int q = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int a = 0; a < D; ++a) flat[q++] = t[k][a];
#pragma unroll
      for (int f = 0; f < R; ++f) {
#pragma unroll
        for (int a = 0; a < D; ++a) if (F::nz(f % NF, a)) flat[q++] = hh[f][a];
        if constexpr (kParkL) {
#pragma unroll
          for (int g = 0; g < f; ++g) flat[q++] = sv.L[f][g];
        }
      }
#pragma unroll
      for (; q < 2 * kParkCells; ++q) flat[q] = 0.0;
      lds_park_put<kParkCells>(cx, flat);
      sched_fence();
    }
    if constexpr (COLWISE) {
      // blocks, one COLUMN c at a time: z_p = (S^-1 B_p)_.c, z_s = (S^-1 B_s)_.c, then every entry of that column
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double zp[R], zs[R];
        sv.template zcol<D>(Yp, c, zp);
        sv.template zcol<D>(Ys, c, zs);
#pragma unroll
        for (int a = 0; a < D; ++a) {
          const bool same = (a % DOF) == (c % DOF);
          if (a <= c) {
            double spp = same ? T(WB_GPP + (a / DOF) + (c / DOF)) : 0.0;      // packed 2 x 2 symmetric: (0,0) (0,1) (1,1)
            double sss = same ? T(WB_GSS + (a / DOF) + (c / DOF)) : 0.0;
#pragma unroll
            for (int f = 0; f < R; ++f) { spp -= Yp[f][a] * zp[f]; sss -= Ys[f][a] * zs[f]; }
            const double en = nb.hi(spp);
            Ds(a, c) -= sss + ((kZeroFill || has_next) ? en : 0.0);
          }
          double sps = same ? T(WB_GPS + (a / DOF) * 2 + (c / DOF)) : 0.0;
#pragma unroll
          for (int f = 0; f < R; ++f) sps -= Yp[f][a] * zs[f];
          const double un = nb.hi(sps);
          Us.v[a][c] = (kZeroFill || has_next) ? -un : 0.0;                    // block (s_j, s_{j+1}) = -Eps of lane j+1
        }
        if constexpr (kFence) { if (c % 2 == 1) sched_fence(); }
      }
    } else {
      double Zp[R][D], Zs[R][D];
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double zp[R], zs[R];
        sv.template zcol<D>(Yp, c, zp);
        sv.template zcol<D>(Ys, c, zs);
#pragma unroll
        for (int f = 0; f < R; ++f) { Zp[f][c] = zp[f]; Zs[f][c] = zs[f]; }
      }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          const bool same = (a % DOF) == (c % DOF);
          if (c >= a) {
            double spp = same ? T(WB_GPP + (a / DOF) + (c / DOF)) : 0.0;
            double sss = same ? T(WB_GSS + (a / DOF) + (c / DOF)) : 0.0;
#pragma unroll
            for (int f = 0; f < R; ++f) { spp -= Yp[f][a] * Zp[f][c]; sss -= Ys[f][a] * Zs[f][c]; }
            const double en = nb.hi(spp);
            Ds(a, c) -= sss + ((kZeroFill || has_next) ? en : 0.0);
          }
          double sps = same ? T(WB_GPS + (a / DOF) * 2 + (c / DOF)) : 0.0;
#pragma unroll
          for (int f = 0; f < R; ++f) sps -= Yp[f][a] * Zs[f][c];
          const double un = nb.hi(sps);
          Us.v[a][c] = (kZeroFill || has_next) ? -un : 0.0;
        }
    }
  }
  DGP_STAMP_NOWAIT(p, cx, 3);
  phase_fence<PF>();
  if constexpr (kFence) sched_fence();
  before_pcr(acc);
  double xs[D];
  // d = 6 forward kernels: the rounds on LDL^T factors, as d = 4 (measured with the recovery state parked in LDS, B = 4096: step 22.43 -> 21.91 us, fused loop
  // 20.13 -> 19.11 us per iteration; the adjoint solve of the backward kernels, RHS_OVERRIDE, loses 1 % with them and keeps the block inverse)
#ifndef DGP_WB_LDL6
#define DGP_WB_LDL6 1
#endif
  pcr_solve<D, LPT, true, (DGP_WB_LDL6 != 0) && !RHS_OVERRIDE>(cx, j, Ds, Us, rs, xs, ok);
  if constexpr (kFence) sched_fence();
  if constexpr (kPark) {
    double flat[2 * kParkCells];
    lds_park_get<kParkCells>(cx, flat);
    int q = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int a = 0; a < D; ++a) t[k][a] = flat[q++];
#pragma unroll
    for (int f = 0; f < R; ++f) {
#pragma unroll
      for (int a = 0; a < D; ++a) if (F::nz(f % NF, a)) hh[f][a] = flat[q++];
      if constexpr (kParkL) {
#pragma unroll
        for (int g = 0; g < f; ++g) sv.L[f][g] = flat[q++];
      }
    }
    sched_fence();
  }
#pragma unroll
  for (int a = 0; a < D; ++a) dx[C - 1][a] = xs[a];
  // ---- interior rows: y = K q - K H M H^T K q,  K q = t - (K Cp) x_ps - (K Cs) x_s
  {
    double xps[D];
#pragma unroll
    for (int a = 0; a < D; ++a) xps[a] = nb.lo(xs[a]);     // (the first lane's K Cp is zero)
    double Kq[3][D];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int pv = 0; pv < 2; ++pv) {
        const double cp0 = T(WB_KCP + (k * 2 + pv) * 2), cp1 = T(WB_KCP + (k * 2 + pv) * 2 + 1);
        const double cs0 = T(WB_KCS + (k * 2 + pv) * 2), cs1 = T(WB_KCS + (k * 2 + pv) * 2 + 1);
#pragma unroll
        for (int dd = 0; dd < DOF; ++dd)
          Kq[k][pv * DOF + dd] = t[k][pv * DOF + dd] - (cp0 * xps[dd] + cp1 * xps[DOF + dd]) - (cs0 * xs[dd] + cs1 * xs[DOF + dd]);
      }
    double cf[R], hm[R][D];
#pragma unroll
    for (int f = 0; f < R; ++f) {
      double s = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) if (F::nz(f % NF, a)) s += hh[f][a] * Kq[f / NF][a];
      cf[f] = s;
    }
    double mf[R];
    sv.solve(cf, mf);
#pragma unroll
    for (int f = 0; f < R; ++f)
#pragma unroll
      for (int a = 0; a < D; ++a) hm[f][a] = F::nz(f % NF, a) ? hh[f][a] * mf[f] : 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double y = Kq[k][a];
#pragma unroll
        for (int f = 0; f < R; ++f)
#pragma unroll
          for (int e = 0; e < D; ++e)
            if (F::nz(f % NF, e) && (e % DOF) == (a % DOF)) y -= T(WB_K6 + (k * 2 + a / DOF) * 6 + (f / NF) * 2 + e / DOF) * hm[f][e];
        dx[k][a] = y;
      }
  }
}

}  // namespace dgp
