// gn_lane.h -- the per-lane program of the fused Gauss-Newton kernel.
//
// Mapping (MI355X, wave64): one trajectory occupies LPT consecutive lanes of a wavefront (LPT = 16, 32 or 64) and every
// lane OWNS C consecutive support states (n <= LPT*C).  Every lane
//   1. loads its C states, gets the two states across its lane boundaries with cross-lane moves (DPP row shifts), computes the
//      bilinear tap addresses and issues the SDF loads (two column-pair loads per state); everything that does not need the
//      taps (priors, GP factors) is evaluated while they are in flight,
//   2. evaluates the factors touching its states and writes their block rows of the block-tridiagonal normal
//      equations (D sym dxd, U dxd, eta) straight into registers (static covariances: the constant GP blocks are scalar
//      kernel arguments, see eval_state / static_rhs / Coupling),
//   3. eliminates its C-1 interior rows locally (streamed block forward sweep that accumulates the Schur-complement pieces
//      the separator rows need), which leaves ONE row per lane,
//   4. takes part in a block parallel-cyclic-reduction (PCR) solve over the LPT lanes: log2(LPT) rounds, in each
//      round a lane inverts its own D, fetches (D^-1, U, eta) of lanes j-s and j+s, and eliminates them (the last round has
//      a single partner per row),
//   5. recovers its interior unknowns by a vector forward / back substitution from the two neighbouring separator unknowns.
// Lambda never exists in memory.  All arithmetic is fp64 (the reference is fp64-only).
//
// The program is written against a tiny "lane context" (cross-lane fetch + ids) so that the very same
// source also compiles for the host, where tests/emul runs 64 lock-stepped threads as a wavefront
// emulator to check the kernel logic without a GPU.  The product path only ever uses the device context.
//
// NOTE on the comments that speak of kernels hipcc "miscompiled" (rounds 2-5): one mechanism explains them all -- register-allocator spill stores placed in front of
// the exec restore of a join block -- and the build repairs it in the device assembly since round 6 (profiles/r06_compiler_fault.md, __graft_entry__.compile_hip_unit).
// The design decisions those comments record were taken before that was known; where they also rest on a measurement they still stand.
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
#include <stddef.h>
#include <type_traits>

#ifndef DGP_HD
#define DGP_HD __host__ __device__ __forceinline__
#endif

// profiles/tools/phase_probe.hip -DDGP_PHASE_STAMPS: lane 0 of every wavefront writes s_memrealtime (100 MHz) at a few
// points of the program into the buffer smuggled in through GnParams::err_hist (unused by MODE_STEP).  Never in the product.
#if defined(DGP_PHASE_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
#define DGP_STAMP(p, cx, slot)                                                                                          \
  do {                                                                                                                 \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                        \
    const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                                   \
    if ((cx).lane() == 0 && (p).err_hist) ((unsigned long long*)(p).err_hist)[(size_t)(cx).wave() * 16 + (slot)] = t_;  \
  } while (0)
#define DGP_STAMP_NOWAIT(p, cx, slot)                                                                                   \
  do {                                                                                                                 \
    const unsigned long long t_ = __builtin_amdgcn_s_memrealtime();                                                   \
    if ((cx).lane() == 0 && (p).err_hist) ((unsigned long long*)(p).err_hist)[(size_t)(cx).wave() * 16 + (slot)] = t_;  \
  } while (0)
#elif defined(DGP_ISA_MARKS) && defined(__HIP_DEVICE_COMPILE__)      // "; DGPMARK n" comments in the ISA at the same points
#define DGP_STAMP(p, cx, slot) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; DGPMARK " #slot ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define DGP_STAMP_NOWAIT(p, cx, slot) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; DGPMARK " #slot ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define DGP_STAMP(p, cx, slot) ((void)0)
#define DGP_STAMP_NOWAIT(p, cx, slot) ((void)0)
#endif

namespace dgp {

enum { MODE_STEP = 0, MODE_SOLVE = 1, MODE_EVAL = 2 };
enum { QC_STATIC = 0, QC_PERSTATE = 1, QC_QFULL = 2, QC_SCALAR = 3 };
enum { FLAG_NONHOLONOMIC = 1u, FLAG_VEL_LIMITS = 2u };
// Kernel variant by covariance representation (template parameter QK of the kernels; see Coupling):
//   QK_STATIC : static covariances with a diagonal Q_c_inv (the reference's default planner) -- constant GP blocks as scalar operands
//   QK_KRON   : per-state Q_c^-1 tensors (qc_mode PERSTATE: the learned modes diag_identity / qc_full) -- Q^-1 = T (x) C_k is never
//               formed, a row keeps the symmetric dof x dof C_k only
//   QK_GENERAL: per-state full Q^-1 (q_full) or a non-diagonal static Q_c_inv -- a row keeps its symmetric d x d Q_k^-1
//   QK_WB     : QK_STATIC with Q_c_inv = c I, no velocity-limit factors, C = 4 and n >= 4 -- the three interior rows of a lane are
//               eliminated through the Woodbury identity on a constant interior block (gn_woodbury.h)
//   QK_WBR    : the same for trajectory lengths that do not fill the shape (n < 4 LPT: the goal row inside a lane, lanes of padding rows);
//               a separate instantiation because the lane-type arithmetic and the two extra table versions cost the n = 4 LPT kernel
//               (the benchmark's) 0.25 us when they are compiled into it
//   QK_SCALED : one scalar per GP factor, Q_c^-1 = s_k Q_c_inv with a diagonal Q_c_inv (qc_mode SCALAR: the learned mode diag_identity, q_k^2 I) -- every GP block of
//               factor k is s_k times the constant block of QK_STATIC, so the lane masks that multiply those blocks carry s_k and the static block elimination
//               applies (STEP kernels only: the learned planner chains step() calls)
enum { QK_GENERAL = 0, QK_STATIC = 1, QK_KRON = 2, QK_WB = 3, QK_WBR = 4, QK_SCALED = 5 };
constexpr DGP_HD bool is_wb(int qk) { return qk == QK_WB || qk == QK_WBR; }

// Layout of the QK_WB constant table (GnParams::wb_tab; doubles, per version -- by what the lane's three interior rows are:
// 0 (WB_T_STD) three ordinary rows, 1 (WB_T_FIRST) rows 0..2 of a trajectory, 2 (WB_T_GOAL) the goal row n-1 among them and padding
// rows behind it (position (n-1) mod 4: one per trajectory length), 3 (WB_T_PAD) padding rows only):
//   K6 (6 x 6, row-major; index (k * 2 + pv) of interior row k, pv = 0 position / 1 velocity), K6 Cp (6 x 2), K6 Cs (6 x 2),
//   Cp^T K6 Cp (packed symmetric 2 x 2 + pad), Cp^T K6 Cs (2 x 2), Cs^T K6 Cs (packed + pad)
enum { WB_K6 = 0, WB_KCP = 36, WB_KCS = 48, WB_GPP = 60, WB_GPS = 64, WB_GSS = 68, WB_TYPE_DOUBLES = 72, WB_TYPES = 4 };
enum { WB_T_STD = 0, WB_T_FIRST = 1, WB_T_GOAL = 2, WB_T_PAD = 3 };
constexpr int kWbTypeStrideBytes = WB_TYPE_DOUBLES * 8 + 16;      // LDS copy: every version starts 16 more bytes (4 banks) off a multiple of 64 banks
constexpr int kWbLdsBytes = WB_TYPES * kWbTypeStrideBytes;

// Kernel arguments (plain data, passed by value).
struct GnParams {
  int32_t B, n;
  int32_t sdf_rows, sdf_cols;
  int64_t sdf_bstride;
  int32_t qc_mode;
  uint32_t flags;
  int32_t max_iters;
  int32_t vec_io;            // 1: th / dtheta / th_out / gradient rows are 16-byte aligned -> vector row accesses
  int32_t vec_mu;            // 1: start / goal rows too
  int32_t vec_qc;            // 1: the qc tensor is 16-byte aligned -> vector accesses of a factor's block
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
  double inv_res;                 // 1 / res, correctly rounded (host division)
  double w_d, w_v, vmax[2];  // 1/K_d^2, 1/K_v^2, (v_x, v_y)
  double M, inv_M;           // plan_layer.py:43-45; 1 / M correctly rounded
  double tol_delta;
  double e10, e11, f11;      // with Phi2 = [[1, dt],[0, 1]], T = [[qa, qb],[qb, qc]]:  E = Phi2^T T = [[qa, qb],[e10, e11]] (U = -E (x) C),
                             // F = E Phi2 = [[qa, e10],[e10, f11]] (Phi^T Q Phi = F (x) C)      (QK_KRON kernels)
  // Static covariances (qc_mode == QC_STATIC): the three constant blocks every GP factor contributes, precomputed on the
  // host from (dt, Q_c_inv) so that the kernels read them as scalar (SGPR) operands instead of holding them in vector
  // registers.  d = 2 dof; symmetric blocks packed like Sym<d>, u_fix row-major d x d.
  int32_t qc_diag;           // 1: Q_c_inv is diagonal -> the static (QSTAT) kernels apply; else static covariances run the generic ones
  int32_t sdf_layout;        // DgpSdf::layout: 0 row-major, 1 4 x 4 tiles -- read by the host (which translation unit to launch) and by the emulator; see DGP_TL
  double q_fix[21];          // Q^-1                                   (gp_factor.py:65-73)
  double a_fix[21];          // Phi^T Q^-1 Phi                         (block (i,i) share of factor i -> i+1)
  double u_fix[36];          // U = -Phi^T Q^-1                        (block (i,i+1))
  // QK_WB kernels (gn_woodbury.h): sqrt of the fixed obstacle / non-holonomic weights, validity of the table, the table (LAST: the
  // kernels read it with vector loads, not through the scalar cache, and warm_kernarg stops in front of it)
  double obs_w_sqrt, w_d_sqrt;
  int32_t wb_ok, pad3_;
  double wb_tab[WB_TYPES * WB_TYPE_DOUBLES];
};
// The table is read as 16-byte cells.  NO alignment attribute on the member: it raises the alignment of the whole by-value kernel
// argument to 16, hipcc then lowers the argument loads of EVERY kernel differently, and the <3,64,2,double,STEP,per-state>
// kernel built that way writes through a wild address (same source without the attribute: identical code to the previous round's,
// exact results).  The offset is a multiple of 16 as it stands; the kernel-argument segment itself is 64-byte aligned.
static_assert(offsetof(GnParams, wb_tab) % 16 == 0, "wb_tab must start on a 16-byte boundary of the kernel-argument segment");

// The QK_STATIC kernel variants (see Coupling below) apply to static covariances with a diagonal Q_c_inv.
DGP_HD bool use_static_kernels(const GnParams& p) { return p.qc_mode == QC_STATIC && p.qc_diag != 0; }
DGP_HD int kernel_variant(const GnParams& p) {
  if (p.qc_mode == QC_SCALAR) return QK_SCALED;      // (host-checked: diagonal Q_c_inv)
  return use_static_kernels(p) ? QK_STATIC : (p.qc_mode == QC_PERSTATE ? QK_KRON : QK_GENERAL);
}

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

// Static covariances with a DIAGONAL Q_c_inv (the reference's default, Q_c_inv = I / sigma^2): Q^-1 = [[qa C, qb C],[qb C, qc C]]
// with C diagonal, so Q^-1, Phi^T Q^-1 Phi and U = -Phi^T Q^-1 only couple position and velocity of the SAME degree of
// freedom: entry (a, c) is structurally zero unless a = c (mod dof).  The static kernels skip those terms.
template <int D> constexpr DGP_HD bool gp_nz(int a, int c) { return (a % (D / 2)) == (c % (D / 2)); }

// 1/v for a pivot: on the device v_rcp_f64 refined by two Newton steps (full double accuracy for normal, finite v;
// the IEEE division expansion's scaling / fix-up of denormals and overflow is not needed for SPD pivots).
DGP_HD double pivot_rcp(double v) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r = __builtin_amdgcn_rcp(v);
  double e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-v, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
#else
  return 1.0 / v;
#endif
}

// Positive-definiteness tracker of one wavefront: `require(cond)` is evaluated where it stands -- one v_cmp into a scalar
// lane mask (a ballot) and one scalar OR.  A plain `bool ok = ok && pivot > 0` is legal for the compiler to SINK into the
// `if (p.info)` block at the very end of the kernel, and it did: every pivot and determinant of the whole solve then stayed
// alive (in AGPRs and, for d = 6, in scratch) until the last instruction -- 6 us of serialised scratch reloads on the d = 6
// kernel when the flags were requested, and register pressure in every kernel even when they were not.  A ballot is a
// convergent operation and cannot be moved under a new control dependency.
template <typename Ctx>
struct SpdCheck {
  Ctx* cx;
  uint64_t bad;          // bit l: lane l met a non-positive (or NaN) pivot
  DGP_HD void require(bool cond) { bad |= cx->ballot(!cond); }
};

// A^-1 of an SPD matrix through LDL^T; ok flags a pivot that is <= 0 or NaN.  (Generic fallback; the kernels use the
// block forms below, whose dependency chains are much shorter.)
template <int D, typename OK>
DGP_HD void sym_inverse_ldlt(const Sym<D>& A, Sym<D>& Ai, OK& ok) {
  double L[D][D];      // unit lower (strict part used)
  double dinv[D];
  double dd[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double v = A(j, j);
#pragma unroll
    for (int k = 0; k < j; ++k) v -= L[j][k] * L[j][k] * dd[k];
    dd[j] = v;
    ok.require(v > 0.0);
    dinv[j] = pivot_rcp(v);
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

// Inverse of an SPD 2x2 / 3x3 block given as scalars (adjugate / determinant); ok tracks the leading minors.
template <typename OK>
DGP_HD void inv2(double a, double b, double c, double& ia, double& ib, double& ic, OK& ok) {      // [[a,b],[b,c]]
  const double det = a * c - b * b;
  ok.require((a > 0.0) && (det > 0.0));
  const double r = pivot_rcp(det);
  ia = c * r; ib = -b * r; ic = a * r;
}
template <typename OK>
DGP_HD void inv3(const double (&m)[6], double (&o)[6], OK& ok) {      // packed upper: m00 m01 m02 m11 m12 m22
  const double c00 = m[3] * m[5] - m[4] * m[4];
  const double c01 = m[2] * m[4] - m[1] * m[5];
  const double c02 = m[1] * m[4] - m[2] * m[3];
  const double c11 = m[0] * m[5] - m[2] * m[2];
  const double c12 = m[1] * m[2] - m[0] * m[4];
  const double c22 = m[0] * m[3] - m[1] * m[1];
  const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  ok.require((m[0] > 0.0) && (c22 > 0.0) && (det > 0.0));
  const double r = pivot_rcp(det);
  o[0] = c00 * r; o[1] = c01 * r; o[2] = c02 * r; o[3] = c11 * r; o[4] = c12 * r; o[5] = c22 * r;
}

// A^-1 of an SPD dxd matrix by 2x2 block elimination with (d/2)x(d/2) blocks:
//   A = [[P, Q],[Q^T, R]],  S = R - Q^T P^-1 Q,  A^-1 = [[P^-1 + Y S^-1 Y^T, -Y S^-1],[., S^-1]],  Y = P^-1 Q.
// Two reciprocals and a dependency depth of ~12 operations instead of d sequential pivots; ok=false if not SPD.
template <int D, typename OK>
DGP_HD void sym_inverse(const Sym<D>& A, Sym<D>& Ai, OK& ok) {
  if constexpr (D == 3) {
    inv3(A.v, Ai.v, ok);
  } else if constexpr (D == 4) {
    double p0, p1, p2;
    inv2(A(0, 0), A(0, 1), A(1, 1), p0, p1, p2, ok);
    // Y = P^-1 Q (2x2), Q = A[0:2, 2:4]
    const double y00 = p0 * A(0, 2) + p1 * A(1, 2), y01 = p0 * A(0, 3) + p1 * A(1, 3);
    const double y10 = p1 * A(0, 2) + p2 * A(1, 2), y11 = p1 * A(0, 3) + p2 * A(1, 3);
    // S = R - Q^T Y (symmetric)
    const double s00 = A(2, 2) - (A(0, 2) * y00 + A(1, 2) * y10);
    const double s01 = A(2, 3) - (A(0, 2) * y01 + A(1, 2) * y11);
    const double s11 = A(3, 3) - (A(0, 3) * y01 + A(1, 3) * y11);
    double t0, t1, t2;
    inv2(s00, s01, s11, t0, t1, t2, ok);
    // Z = Y S^-1
    const double z00 = y00 * t0 + y01 * t1, z01 = y00 * t1 + y01 * t2;
    const double z10 = y10 * t0 + y11 * t1, z11 = y10 * t1 + y11 * t2;
    Ai(2, 2) = t0; Ai(2, 3) = t1; Ai(3, 3) = t2;
    Ai(0, 2) = -z00; Ai(0, 3) = -z01; Ai(1, 2) = -z10; Ai(1, 3) = -z11;
    Ai(0, 0) = p0 + (z00 * y00 + z01 * y01);
    Ai(0, 1) = p1 + (z00 * y10 + z01 * y11);
    Ai(1, 1) = p2 + (z10 * y10 + z11 * y11);
  } else if constexpr (D == 6) {
    double Pm[6] = {A(0, 0), A(0, 1), A(0, 2), A(1, 1), A(1, 2), A(2, 2)}, Pi[6];
    inv3(Pm, Pi, ok);
    auto P = [&](int i, int j) { return Pi[Sym<3>::idx(i, j)]; };
    double Y[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Y[i][j] = P(i, 0) * A(0, 3 + j) + P(i, 1) * A(1, 3 + j) + P(i, 2) * A(2, 3 + j);
    double Sm[6], Si[6];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = i; j < 3; ++j)
        Sm[Sym<3>::idx(i, j)] = A(3 + i, 3 + j) - (A(0, 3 + i) * Y[0][j] + A(1, 3 + i) * Y[1][j] + A(2, 3 + i) * Y[2][j]);
    inv3(Sm, Si, ok);
    auto S = [&](int i, int j) { return Si[Sym<3>::idx(i, j)]; };
    double Z[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) Z[i][j] = Y[i][0] * S(0, j) + Y[i][1] * S(1, j) + Y[i][2] * S(2, j);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        Ai(i, 3 + j) = -Z[i][j];
        if (j >= i) {
          Ai(3 + i, 3 + j) = S(i, j);
          Ai(i, j) = P(i, j) + (Z[i][0] * Y[j][0] + Z[i][1] * Y[j][1] + Z[i][2] * Y[j][2]);
        }
      }
  } else {
    sym_inverse_ldlt<D>(A, Ai, ok);
  }
}

// x = A^-1 b through LDL^T
template <int D, typename OK>
DGP_HD void sym_solve(const Sym<D>& A, const double (&b)[D], double (&x)[D], OK& ok) {
  double L[D][D];
  double dd[D], dinv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    double v = A(j, j);
#pragma unroll
    for (int k = 0; k < j; ++k) v -= L[j][k] * L[j][k] * dd[k];
    dd[j] = v;
    ok.require(v > 0.0);
    dinv[j] = pivot_rcp(v);
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

// One state row (D consecutive elements) as 16-byte (8-byte for f32, D = 6) vector accesses when the host found every
// row pointer suitably aligned (GnParams::vec_io), scalar accesses otherwise.
template <typename IO, int D> struct RowVec;
template <> struct RowVec<float, 4> { typedef float T __attribute__((vector_size(16))); enum { N = 1, W = 4 }; };
template <> struct RowVec<float, 6> { typedef float T __attribute__((vector_size(8))); enum { N = 3, W = 2 }; };
template <> struct RowVec<double, 4> { typedef double T __attribute__((vector_size(16))); enum { N = 2, W = 2 }; };
template <> struct RowVec<double, 6> { typedef double T __attribute__((vector_size(16))); enum { N = 3, W = 2 }; };

template <typename IO, int D>
DGP_HD void ld_row(const void* p, int64_t row, bool vec, double (&x)[D]) {
  if (vec) {
    typedef typename RowVec<IO, D>::T V;
    const V* q = (const V*)((const IO*)p + row * D);
#pragma unroll
    for (int k = 0; k < RowVec<IO, D>::N; ++k) {
      const V v = q[k];
#pragma unroll
      for (int a = 0; a < RowVec<IO, D>::W; ++a) x[k * RowVec<IO, D>::W + a] = (double)v[a];
    }
  } else {
#pragma unroll
    for (int a = 0; a < D; ++a) x[a] = ld<IO>(p, row * D + a);
  }
}
template <typename IO, int D>
DGP_HD void st_row(void* p, int64_t row, bool vec, const double (&x)[D]) {
  if (vec) {
    typedef typename RowVec<IO, D>::T V;
    V* q = (V*)((IO*)p + row * D);
#pragma unroll
    for (int k = 0; k < RowVec<IO, D>::N; ++k) {
      V v;
#pragma unroll
      for (int a = 0; a < RowVec<IO, D>::W; ++a) v[a] = (IO)x[k * RowVec<IO, D>::W + a];
      q[k] = v;
    }
  } else {
#pragma unroll
    for (int a = 0; a < D; ++a) st<IO>(p, row * D + a, x[a]);
  }
}

// The C state rows of a lane, loaded without branches: rows that do not exist read row 0 of the tensor and are zeroed.
template <int DOF, int C, typename IO>
DGP_HD void load_lane_rows(const GnParams& p, const void* src, int64_t b, int g0, bool traj_ok, bool vec, double (&x)[C][2 * DOF]) {
  constexpr int D = 2 * DOF;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const bool valid = traj_ok && (g0 + k) < p.n;
    ld_row<IO, D>(src, valid ? b * p.n + g0 + k : 0, vec, x[k]);
#pragma unroll
    for (int a = 0; a < D; ++a) x[k][a] = valid ? x[k][a] : 0.0;
  }
}

// ---------------------------------------------------------------------------------------------------
// bilinear SDF lookup + hinge, bit-for-bit the reference's fp64 op order (no FMA contraction here, so
// the `dist <= eps + r` decision equals the CPU oracle's on identical inputs).
//   utils/sdf_utils.py:57-94, gpmp2/obstacle/obstacle_cost.py:30,36-37
// ---------------------------------------------------------------------------------------------------
// Everything the backward needs from one bilinear lookup (same arithmetic as obstacle_finish below).
struct ObsTaps {
  int64_t i11, i21, i12, i22;      // element offsets of the four taps inside the grid
  double wja, wjb, wjc, wjd;       // (fy2-py), (py-fy1), (fx2-px), (px-fx1)
  double cross;                    // d22 - d12 - d21 + d11
  bool act;
};

// Bilinear lookup, phase 1 (utils/sdf_utils.py:57-72): pixel coordinates and the four (clamped) tap offsets.
struct ObsAddr {
  double px, py;
  int32_t x1, x2, y1, y2;          // clamped to [0, W-1] / [0, H-1]: 32 bits are plenty (and int32 -> fp64 is one instruction)
};

// a / res, correctly rounded, without the ~11-instruction IEEE division expansion and its quarter-rate v_rcp_f64:
// q0 = a * (1/res), one exact-residual correction q1 = q0 + (a - q0 res) (1/res) with FMAs (Markstein's division step; with
// 1/res correctly rounded the result equals the IEEE quotient -- checked against a / res on 2.6e9 operand pairs, including
// near-integer quotients, in profiles/tools/div_check.c).  Non-finite a gives NaN instead of +-inf; both end as NaN costs.
DGP_HD double div_res(const GnParams& p, double a) {
  const double q0 = a * p.inv_res;
  const double r = __builtin_fma(-q0, p.res, a);
  return __builtin_fma(r, p.inv_res, q0);
}
// a / M the same way (err, err_ext normalisation, plan_layer.py:308,345)
DGP_HD double div_M(const GnParams& p, double a) {
  const double q0 = a * p.inv_M;
  const double r = __builtin_fma(-q0, p.M, a);
  return __builtin_fma(r, p.inv_M, q0);
}

DGP_HD int32_t imin32(int32_t a, int32_t b) { return a < b ? a : b; }
DGP_HD int32_t imax32(int32_t a, int32_t b) { return a > b ? a : b; }

// Grid layout the kernels of a translation unit are compiled for (DgpSdf::layout):
//   DGP_TL 0  row-major grids only -- the standard kernels (every round-4 instantiation keeps its code: no layout test anywhere in the device code);
//   DGP_TL 1  every grid is stored as 4 x 4 tiles -- the TILED translation units (gn_inst.hip with -DDGP_TL=1: the same kernels for the launch shapes with four
//             states per lane, distinct symbols through the kernels' trailing template argument);
//   DGP_TL 2  decided at run time from GnParams::sdf_layout -- the CPU wavefront emulator (tests/emul), one build for both layouts.
// A run-time branch inside the standard kernels was measured first (profiles/r05_tiled_ab.txt): the untaken branch splits the block that schedules the tap loads
// under the factor arithmetic and costs the ROW-MAJOR path 0.45 us per kernel (headline step 9.22 -> 9.68 us; 3 us on a training iteration) -- and with it hipcc
// produced wrong 64-lane d = 6 step kernels.  Compiled apart, the standard kernels are byte-identical to the verified ones.
#ifndef DGP_TL
#define DGP_TL 0
#endif
// DGP_STEP_ERRS 1: the MODE_STEP kernels of this translation unit carry an EPILOGUE -- the unweighted errors at th + dtheta (dgp_gn_step_errors, one iteration of
// learning/train_planner.py:311-327 in ONE launch instead of two) -- behind a test of GnParams::unw_*.  Compiled into twin units only (gn_inst.hip with
// -DDGP_STEP_ERRS=1: the STEP kernels of the two four-states-per-lane shapes), like the tiled kernels: the standard kernels stay byte-identical.  The emulator: 1.
#ifndef DGP_STEP_ERRS
#define DGP_STEP_ERRS 0
#endif
// Reproducer builds only (profiles/tools/r06_twin_repro.sh): DGP_TWIN_REPRO 1 = the run-time switch DGP_TWIN_NO_ERRS of dgp_host::gn_step_errors;
// DGP_EXCLUDE_REPAIRED_TWINS 1 = the round-5 host-side exclusion of the two step-errors twins that came out wrong before the build repaired its assembly
#ifndef DGP_TWIN_REPRO
#define DGP_TWIN_REPRO 0
#endif
#ifndef DGP_EXCLUDE_REPAIRED_TWINS
#define DGP_EXCLUDE_REPAIRED_TWINS 0
#endif
DGP_HD bool grid_is_tiled(const GnParams& p) { return DGP_TL == 1 ? true : (DGP_TL == 2 ? p.sdf_layout != 0 : false); }
// Elements of one grid: row-major H x W, or ceil(H / 4) x ceil(W / 4) tiles of 16 -- tile (y / 4, x / 4), row (y % 4), column (x % 4) inside it: the 2 x 2 footprint
// of a bilinear lookup then lies in one 64-byte (fp32) tile in 9 cases of 16, where the row-major grid spreads it over two lines a row pitch apart (per-sample
// grids: 70 distinct 128-byte lines per 64-state trajectory against 29, profiles/r05_tile_probe.txt)
DGP_HD int64_t grid_elems(const GnParams& p) {
  if (!grid_is_tiled(p)) return (int64_t)p.sdf_rows * p.sdf_cols;
  return (int64_t)((p.sdf_rows + 3) >> 2) * ((p.sdf_cols + 3) >> 2) * 16;
}
// the four tap offsets (x1,y1), (x2,y1), (x1,y2), (x2,y2) of a lookup in a tiled grid
DGP_HD void tiled_tap_offsets(const GnParams& p, const ObsAddr& o, int32_t& i11, int32_t& i21, int32_t& i12, int32_t& i22) {
  const int32_t Wt = (p.sdf_cols + 3) >> 2;
  const int32_t t1 = ((o.y1 >> 2) * Wt) << 4, t2 = ((o.y2 >> 2) * Wt) << 4, r1 = (o.y1 & 3) << 2, r2 = (o.y2 & 3) << 2;
  const int32_t c1 = ((o.x1 >> 2) << 4) + (o.x1 & 3), c2 = ((o.x2 >> 2) << 4) + (o.x2 & 3);
  i11 = t1 + r1 + c1; i21 = t1 + r1 + c2; i12 = t2 + r2 + c1; i22 = t2 + r2 + c2;
}

DGP_HD void obstacle_addr(const GnParams& p, double x, double y, ObsAddr& o) {
#pragma clang fp contract(off)
  o.px = p.orig_px + div_res(p, x);                       // :61   orig + x / res
  o.py = p.orig_py - div_res(p, y);                       // :62
  double fpx = floor(o.px), fpy = floor(o.py);
  // floor -> int64 -> clamp (:64-72); saturate first so that huge |px| cannot overflow the conversion.  fmax / fmin return
  // the non-NaN operand, so NaN coordinates land on an in-range index too (the interpolated result is NaN anyway).
  const double big = 1.0e9;                               // |floor| <= 1e9 fits int32 and int32 + 1 does not overflow
  const double cx = fmin(fmax(fpx, -big), big), cy = fmin(fmax(fpy, -big), big);
  const int32_t x1 = (int32_t)cx, y1 = (int32_t)cy;
  const int32_t x2 = x1 + 1, y2 = y1 + 1;                 // :65,67 (before clamping)
  const int32_t W1 = p.sdf_cols - 1, H1 = p.sdf_rows - 1;
  o.x1 = imax32(0, imin32(x1, W1));
  o.x2 = imax32(0, imin32(x2, W1));
  o.y1 = imax32(0, imin32(y1, H1));
  o.y2 = imax32(0, imin32(y2, H1));
}

// Phase 2 (utils/sdf_utils.py:81-94, obstacle_cost.py:30,36-37): weights, distance, gradient, hinge -- bit-for-bit the
// reference's fp64 op order (no FMA contraction here, so the `dist <= eps + r` decision equals the CPU oracle's).
DGP_HD void obstacle_finish(const GnParams& p, const ObsAddr& o, double d11, double d21, double d12, double d22, double eps,
                            double& cost, double& hx, double& hy, ObsTaps* taps = nullptr) {
#pragma clang fp contract(off)
  const double px = o.px, py = o.py;
  double fx1 = (double)o.x1, fx2 = (double)o.x2, fy1 = (double)o.y1, fy2 = (double)o.y2;
  double wa = (fx2 - px) * (fy2 - py);                    // :81-84
  double wb = (px - fx1) * (fy2 - py);
  double wc = (fx2 - px) * (py - fy1);
  double wd = (px - fx1) * (py - fy1);
  double dist = wa * d11 + wb * d21 + wc * d12 + wd * d22;   // :90
  double Jx = div_res(p, -1.0 * ((fy2 - py) * (d21 - d11) + (py - fy1) * (d22 - d12)));   // :93   (...) / res
  double Jy = div_res(p, (fx2 - px) * (d12 - d11) + (px - fx1) * (d22 - d21));            // :94
  double eps_tot = eps + p.radius;                        // obstacle_cost.py:30
  bool act = dist <= eps_tot;                             // :36
  cost = act ? (eps_tot - dist) : 0.0;
  hx = act ? (-1.0 * Jx) : 0.0;                           // :37
  hy = act ? (-1.0 * Jy) : 0.0;
  if (taps) {
    if (grid_is_tiled(p)) {
      int32_t i11, i21, i12, i22;
      tiled_tap_offsets(p, o, i11, i21, i12, i22);
      taps->i11 = i11; taps->i21 = i21; taps->i12 = i12; taps->i22 = i22;
    } else {
      const int64_t W = p.sdf_cols;
      taps->i11 = (int64_t)o.y1 * W + o.x1; taps->i21 = (int64_t)o.y1 * W + o.x2;
      taps->i12 = (int64_t)o.y2 * W + o.x1; taps->i22 = (int64_t)o.y2 * W + o.x2;
    }
    taps->wja = fy2 - py; taps->wjb = py - fy1; taps->wjc = fx2 - px; taps->wjd = px - fx1;
    taps->cross = d22 - d12 - d21 + d11;
    taps->act = act;
  }
}

template <typename IO>
DGP_HD void obstacle_eval(const GnParams& p, const IO* grid, double x, double y, double eps, double& cost, double& hx,
                          double& hy, ObsTaps* taps = nullptr) {
  ObsAddr o;
  obstacle_addr(p, x, y, o);
  int64_t i11, i21, i12, i22;
  if (grid_is_tiled(p)) {
    int32_t a11, a21, a12, a22;
    tiled_tap_offsets(p, o, a11, a21, a12, a22);
    i11 = a11; i21 = a21; i12 = a12; i22 = a22;
  } else {
    const int64_t W = p.sdf_cols;
    i11 = (int64_t)o.y1 * W + o.x1; i21 = (int64_t)o.y1 * W + o.x2; i12 = (int64_t)o.y2 * W + o.x1; i22 = (int64_t)o.y2 * W + o.x2;
  }
  const double d11 = (double)grid[i11], d21 = (double)grid[i21];      // dx1y1, dx2y1 (:76-77)
  const double d12 = (double)grid[i12], d22 = (double)grid[i22];      // dx1y2, dx2y2 (:78-79)
  obstacle_finish(p, o, d11, d21, d12, d22, eps, cost, hx, hy, taps);
}

// Q^-1 of one GP factor (gp_factor.py:65-73 / plan_layer.py:90)
// Per-state Q^-1 of GP factor f (only for the non-static modes: p.qc != null).  All loads are unconditional and issued
// together (a wave-uniform address select instead of per-element branches, each of which would cost its own s_waitcnt);
// every entry of Q is assigned exactly once, in straight-line code (an aggregate written on two control-flow paths ends
// up partly in scratch memory).
template <int DOF, typename IO>
DGP_HD void load_Qinv(const GnParams& p, int64_t b, int f, Sym<2 * DOF>& Q) {
  constexpr int D = 2 * DOF;
  const bool full = (p.qc_mode == QC_QFULL);
  const IO* src = (const IO*)p.qc + (b * (p.n - 1) + f) * (full ? D * D : DOF * DOF);
  double c[DOF][DOF];
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = 0; j < DOF; ++j) c[i][j] = (double)src[i * DOF + j];          // per-state mode: C = Q_c^-1 (dof x dof)
  double fq[D * (D + 1) / 2];
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) fq[Sym<D>::idx(i, j)] = (double)src[full ? i * D + j : 0];   // q_full mode: Q^-1 itself
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      const double coef = (j < DOF) ? p.qa : ((i >= DOF) ? p.qc_ : p.qb);      // blocks [[qa C, qb C],[qb C, qc C]]
      Q(i, j) = full ? fq[Sym<D>::idx(i, j)] : coef * c[i % DOF][j % DOF];
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

template <int D> DGP_HD void sym_times_vec_fwd(const Sym<D>& S, const double (&v)[D], double (&o)[D]) {   // o = S v
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) t += S(a, k) * v[k];
    o[a] = t;
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
// factor evaluation for ONE support state (row g of the block-tridiagonal system)
//   -> diagonal block Dm, coupling U to row g+1, eta r, and the partial error sums.
// Qown is Q^-1 of the GP factor (g -> g+1), Qm is Q^-1 of the factor (g-1 -> g) (ignored for g == 0);
// (oc, ohx, ohy) is the obstacle factor of this state and mu_s / mu_g the start / goal means of the trajectory, all loaded
// beforehand (every load of a lane is issued up front, branch-free, so that the memory latencies overlap).
// ---------------------------------------------------------------------------------------------------
struct ErrAcc {
  double e, eext;          // partial sums of err / err_ext (un-normalised)
  double usg, ugp, uobs;   // unweighted partials (plan_layer.py:374-388)
};

// The factors that touch ONE state only (obstacle, velocity limits, non-holonomic), for a VALID row.
template <int DOF, bool ASSEMBLE>
DGP_HD void eval_state_local(const GnParams& p, const double (&x)[2 * DOF], double ow, double oc, double ohx, double ohy,
                             Sym<2 * DOF>& Dm, double (&r)[2 * DOF], ErrAcc& acc) {
  // ---- obstacle factor (obstacle_factor.py:35-40): sphere centre = x[0:2], H = H_e H_fk, H_fk = I_d[0:2,:]
  {
    acc.e += 0.5 * ow * oc * oc;
    acc.eext += 0.5 * p.obs_w_fix * oc * oc;               // plan_layer.py:329-332 (fixed weight, current eps)
    acc.uobs += 0.5 * oc * oc;
    if (ASSEMBLE) {
      Dm(0, 0) += ow * ohx * ohx; Dm(0, 1) += ow * ohx * ohy; Dm(1, 1) += ow * ohy * ohy;
      r[0] += ow * ohx * oc; r[1] += ow * ohy * oc;
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
      acc.e += 0.5 * p.w_v * c * c; acc.eext += 0.5 * p.w_v * c * c;
      if (ASSEMBLE) { Dm(DOF + a, DOF + a) += p.w_v * h * h; r[DOF + a] += p.w_v * h * c; }
    }
  }
  // ---- non-holonomic factor (nonholonomic_factor.py:16-30), state [x,y,th,vx,vy,w]; H as the reference writes it
  if constexpr (DOF == 3) if (p.flags & FLAG_NONHOLONOMIC) {
    const double th = x[2], vx = x[DOF], vy = x[DOF + 1];
    const double sn = sin(th), cs = cos(th);
    const double e = vy * cs - vx * sn;
    const double h[3] = {-vy * sn + vx * cs, -sn, cs};    // columns 2,3,4
    acc.e += 0.5 * p.w_d * e * e; acc.eext += 0.5 * p.w_d * e * e;
    if (ASSEMBLE) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int c = a; c < 3; ++c) Dm(2 + a, 2 + c) += p.w_d * h[a] * h[c];
        r[2 + a] += p.w_d * h[a] * e;
      }
    }
  }
}

template <int DOF, typename IO, bool ASSEMBLE>
DGP_HD void eval_state(const GnParams& p, int64_t b, int g, bool valid, const double (&x)[2 * DOF], const double (&xm)[2 * DOF],
                       const double (&xp)[2 * DOF], const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF],
                       const Sym<2 * DOF>& Qown, const Sym<2 * DOF>& Qm, double ow, double oc, double ohx, double ohy,
                       Sym<2 * DOF>& Dm, Mat<2 * DOF>& U, double (&r)[2 * DOF], ErrAcc& acc) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  if (ASSEMBLE) {
#pragma unroll
    for (int a = 0; a < D; ++a) {
      r[a] = 0.0;
#pragma unroll
      for (int c = 0; c < D; ++c) {
        U.v[a][c] = 0.0;
        if (c >= a) Dm(a, c) = (a == c) ? 1.0 : 0.0;       // padding rows: identity row, x = 0
      }
    }
  }
  if (!valid) return;
  if (ASSEMBLE) {
#pragma unroll
    for (int a = 0; a < D; ++a) Dm(a, a) = p.reg;          // delta I (plan_layer.py:219)
  }
  // ---- start / goal priors: e = mu - x, H = +I, weight I/K^2 (prior_factor.py:15-18; plan_layer.py:64-68)
  if (g == 0 || g == n - 1) {
    const bool is_start = (g == 0);                        // n >= 2 (host-enforced): rows 0 and n-1 are distinct
    const double w = is_start ? p.w_s : p.w_g;
    double s2 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) {
      const double ea = (is_start ? mu_s[a] : mu_g[a]) - x[a];
      s2 += ea * ea;
      if (ASSEMBLE) { Dm(a, a) += w; r[a] += w * ea; }
    }
    acc.e += 0.5 * w * s2; acc.eext += 0.5 * w * s2; acc.usg += 0.5 * s2;
  }
  // ---- GP factor (g -> g+1), owned by row g: e = x_{g+1} - Phi x_g (gp_factor.py:105)
  const double dt = p.dt;
  if (g < n - 1) {
    double e[D];
#pragma unroll
    for (int a = 0; a < DOF; ++a) {
      e[a] = xp[a] - (x[a] + dt * x[DOF + a]);
      e[DOF + a] = xp[DOF + a] - x[DOF + a];
    }
    const double q = quad<D>(Qown, e);
    acc.e += 0.5 * q;
    double s2 = 0.0;
#pragma unroll
    for (int a = 0; a < D; ++a) s2 += e[a] * e[a];
    acc.ugp += 0.5 * s2;
    if (p.qc_mode == QC_STATIC) {
      acc.eext += 0.5 * q;
    } else {
      Sym<D> Qf;
      fixed_Qinv<DOF>(p, Qf);
      acc.eext += 0.5 * quad<D>(Qf, e);                    // plan_layer.py:318-321 (fixed GP weight)
    }
    if (ASSEMBLE) {
      // U = -Phi^T Q : rows pos = Q[pos,:], rows vel = dt*Q[pos,:] + Q[vel,:]     (block (g,g+1) = H1^T Q^-1 H2)
#pragma unroll
      for (int a = 0; a < DOF; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          U.v[a][c] = -Qown(a, c);
          U.v[DOF + a][c] = -(dt * Qown(a, c) + Qown(DOF + a, c));
        }
      // Dm += Phi^T Q Phi = -U Phi : cols pos = -U[:,pos], cols vel = -(dt*U[:,pos] + U[:,vel]);   eta += Phi^T Q e = -U e
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < DOF; ++c) {
          if (c >= a) Dm(a, c) -= U.v[a][c];
          if (DOF + c >= a) Dm(a, DOF + c) -= dt * U.v[a][c] + U.v[a][DOF + c];
        }
#pragma unroll
        for (int c = 0; c < D; ++c) t += U.v[a][c] * e[c];
        r[a] -= t;
      }
    }
  }
  // ---- GP factor (g-1 -> g): contributes Q_{g-1}^-1 to D_g and -Q_{g-1}^-1 e_{g-1} to eta_g
  if (ASSEMBLE && g > 0) {
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
        if (c >= a) Dm(a, c) += Qm(a, c);
        t += Qm(a, c) * e[c];
      }
      r[a] -= t;
    }
  }
  eval_state_local<DOF, ASSEMBLE>(p, x, ow, oc, ohx, ohy, Dm, r, acc);
}

// Static covariances (qc_mode == QC_STATIC, the reference's default planner): the same row, assembled branch-free from
// the host-precomputed blocks p.q_fix / p.a_fix / p.u_fix.  Every GP term is weighted by a 0/1 lane mask instead of being
// guarded by a branch (each wavefront holds first and last rows, so both sides of such a branch run anyway), and the
// constant blocks are scalar operands of the FMAs.  The coupling block itself is NOT written: U_g = m_next * p.u_fix.
//   m_next = 1 iff row g is valid and g < n-1;   m_prev = 1 iff row g is valid and g > 0
template <int DOF>
DGP_HD void static_rhs(const GnParams& p, int g, bool valid, const double (&x)[2 * DOF], const double (&xm)[2 * DOF],
                       const double (&xp)[2 * DOF], const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF],
                       double (&r)[2 * DOF], ErrAcc& acc, double s_next = 1.0, double s_prev = 1.0) {
  // s_next / s_prev (QK_SCALED): the scalars of the factors (g -> g+1) and (g-1 -> g), Q^-1 = s * q_fix; they weigh the system and err,
  // not err_ext / the unweighted errors (fixed weights, plan_layer.py:318-321, 374-377)
  // priors + GP factors: everything of row g that does NOT depend on the SDF lookup (runs while the taps are in flight)
  constexpr int D = 2 * DOF;
  const int n = p.n;
  const double dt = p.dt;
  const bool is_start = valid && g == 0, is_goal = valid && g == n - 1;
  const double mN = (valid && g < n - 1) ? 1.0 : 0.0, mP = (valid && g > 0) ? 1.0 : 0.0;
  // ---- start / goal priors (prior_factor.py:15-18; plan_layer.py:64-68)
  const double w = is_start ? p.w_s : (is_goal ? p.w_g : 0.0);
  double s2 = 0.0;
  double ep[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    ep[a] = (is_start ? mu_s[a] : mu_g[a]) - x[a];
    s2 += ep[a] * ep[a];
  }
  acc.e += 0.5 * w * s2; acc.eext += 0.5 * w * s2; acc.usg += (is_start || is_goal) ? 0.5 * s2 : 0.0;
  // ---- GP factors (g -> g+1) and (g-1 -> g): e = x_{g+1} - Phi x_g (gp_factor.py:105)
  double eo[D], em[D];
#pragma unroll
  for (int a = 0; a < DOF; ++a) {
    eo[a] = xp[a] - (x[a] + dt * x[DOF + a]);
    eo[DOF + a] = xp[DOF + a] - x[DOF + a];
    em[a] = x[a] - (xm[a] + dt * xm[DOF + a]);
    em[DOF + a] = x[DOF + a] - xm[DOF + a];
  }
  double q = 0.0, so = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) if (gp_nz<D>(a, c)) t += p.q_fix[Sym<D>::idx(a, c)] * eo[c];
    q += eo[a] * t;
    so += eo[a] * eo[a];
  }
  const double wN = mN * s_next, wP = mP * s_prev;
  acc.e += wN * (0.5 * q); acc.eext += mN * (0.5 * q); acc.ugp += mN * (0.5 * so);
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double tu = 0.0, tq = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) if (gp_nz<D>(a, c)) {
      tu += p.u_fix[a * D + c] * eo[c];                    // (U e_own)_a :  eta += Phi^T Q e = -U e
      tq += p.q_fix[Sym<D>::idx(a, c)] * em[c];            // (Q e_prev)_a:  eta -= Q e_prev
    }
    r[a] = w * ep[a] - wN * tu - wP * tq;
  }
}

// diagonal block of row g without the single-state factors; m_next = 1 iff the row couples to row g+1
template <int DOF>
DGP_HD void static_diag(const GnParams& p, int g, bool valid, Sym<2 * DOF>& Dm, double& m_next, double s_next = 1.0, double s_prev = 1.0) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  const bool is_start = valid && g == 0, is_goal = valid && g == n - 1;
  const double mN = ((valid && g < n - 1) ? 1.0 : 0.0) * s_next, mP = ((valid && g > 0) ? 1.0 : 0.0) * s_prev;      // (QK_SCALED: mask x the factor's scalar)
  m_next = mN;
  const double w = is_start ? p.w_s : (is_goal ? p.w_g : 0.0);
  const double dbase = valid ? p.reg : 1.0;                // delta I (plan_layer.py:219); padding rows: identity row, x = 0
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = a; c < D; ++c)
      Dm(a, c) = gp_nz<D>(a, c) ? ((a == c) ? dbase + w : 0.0) + mN * p.a_fix[Sym<D>::idx(a, c)] + mP * p.q_fix[Sym<D>::idx(a, c)] : 0.0;
}

// ---------------------------------------------------------------------------------------------------
// Generic covariances (per-state / q_full tensors, or a non-diagonal static Q_c_inv): Q^-1 of the C + 1 GP factors that touch
// a lane's rows, fetched ONCE per launch -- all loads issued together, ahead of the SDF taps, as 16-byte vector accesses
// where the factor's block is a multiple of 16 bytes (GnParams::vec_qc) -- and shared by the assembly, the elimination and
// (backward kernel) the chain rule.  In the fused loop they are loop-invariant: loaded before the first GN iteration.
//   q[k]: factor (g0+k -> g0+k+1), k = 0..C-1;   qm0: factor (g0-1 -> g0).
// Rows / factors that do not exist read a clamped factor index of the SAME trajectory (finite data, so that the 0/1 masks
// that multiply them cannot turn it into NaN; a NaN inside one trajectory's tensors stays inside that trajectory).
// ---------------------------------------------------------------------------------------------------
template <int D, int C, int QK> struct LaneQ;
template <int D, int C> struct LaneQ<D, C, QK_GENERAL> { Sym<D> q[C]; Sym<D> qm0; };
template <int D, int C> struct LaneQ<D, C, QK_STATIC> {};
template <int D, int C> struct LaneQ<D, C, QK_WB> {};
template <int D, int C> struct LaneQ<D, C, QK_WBR> {};
template <int D, int C> struct LaneQ<D, C, QK_SCALED> { double s[C]; double sm0; };      // the scalars of the lane's C + 1 GP factors
template <int D, int C> struct LaneQ<D, C, QK_KRON> { Sym<D / 2> c[C]; Sym<D / 2> cm0; };      // C_k = Q_c^-1 of the factor, read as a symmetric matrix

// N consecutive elements starting at src: 16-byte vector loads when `vec` (host-checked alignment) and N is a whole number of
// 16-byte cells, scalar loads otherwise.  Values stay in the I/O type (converted at first use).
template <typename IO, int N>
DGP_HD void ld_block(const IO* src, bool vec, IO (&v)[N]) {
  constexpr int EPV = 16 / (int)sizeof(IO);
  if constexpr ((N % EPV) == 0) {
    if (vec) {
      typedef IO V16 __attribute__((vector_size(16)));
#pragma unroll
      for (int k = 0; k < N / EPV; ++k) {
        const V16 t = ((const V16*)src)[k];
#pragma unroll
        for (int e = 0; e < EPV; ++e) v[k * EPV + e] = t[e];
      }
      return;
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = src[k];
}

template <int DOF, typename IO>
DGP_HD void load_Qinv_block(const GnParams& p, int64_t b, int f, Sym<2 * DOF>& Q) {
  constexpr int D = 2 * DOF;
  const bool stat = (p.qc_mode == QC_STATIC), full = (p.qc_mode == QC_QFULL);
  const bool vec = p.vec_qc != 0;
  IO raw[D * D];
#pragma unroll
  for (int i = 0; i < D * D; ++i) raw[i] = (IO)0;
  if (full) {                                      // wave-uniform: q_full, Q^-1 itself (d x d per factor)
    ld_block<IO, D * D>((const IO*)p.qc + (b * (p.n - 1) + f) * (D * D), vec, raw);
  } else if (!stat) {                              // per-state Q_c^-1 (dof x dof per factor)
    IO c[DOF * DOF];
    ld_block<IO, DOF * DOF>((const IO*)p.qc + (b * (p.n - 1) + f) * (DOF * DOF), vec, c);
#pragma unroll
    for (int i = 0; i < DOF * DOF; ++i) raw[i] = c[i];
  }
#pragma unroll
  for (int i = 0; i < D; ++i)
#pragma unroll
    for (int j = i; j < D; ++j) {
      const double coef = (j < DOF) ? p.qa : ((i >= DOF) ? p.qc_ : p.qb);      // blocks [[qa C, qb C],[qb C, qc C]] (gp_factor.py:65-73)
      const double cij = stat ? p.qc_fix[(i % DOF) * DOF + (j % DOF)] : (double)raw[(i % DOF) * DOF + (j % DOF)];
      Q(i, j) = full ? (double)raw[i * D + j] : coef * cij;
    }
}

template <int DOF, int C, typename IO>
DGP_HD void load_lane_Q(const GnParams& p, int64_t b, int g0, bool traj_ok, LaneQ<2 * DOF, C, QK_GENERAL>& L) {
  const int64_t bb = traj_ok ? b : 0;
  const int fmax = p.n - 2;                        // n >= 2 (host-enforced)
#pragma unroll
  for (int k = 0; k < C; ++k) load_Qinv_block<DOF, IO>(p, bb, imin32(g0 + k, fmax), L.q[k]);
  load_Qinv_block<DOF, IO>(p, bb, imin32(imax32(g0 - 1, 0), fmax), L.qm0);
}
template <int DOF, int C, typename IO>
DGP_HD void load_lane_Q(const GnParams&, int64_t, int, bool, LaneQ<2 * DOF, C, QK_STATIC>&) {}
template <int DOF, int C, typename IO>
DGP_HD void load_lane_Q(const GnParams&, int64_t, int, bool, LaneQ<2 * DOF, C, QK_WB>&) {}
template <int DOF, int C, typename IO>
DGP_HD void load_lane_Q(const GnParams&, int64_t, int, bool, LaneQ<2 * DOF, C, QK_WBR>&) {}

template <int DOF, int C, typename IO>
DGP_HD void load_lane_Q(const GnParams& p, int64_t b, int g0, bool traj_ok, LaneQ<2 * DOF, C, QK_SCALED>& L) {
  const int64_t bb = traj_ok ? b : 0;
  const int fmax = p.n - 2;
  const IO* sc = (const IO*)p.qc + bb * (p.n - 1);
#pragma unroll
  for (int k = 0; k < C; ++k) L.s[k] = (double)sc[imin32(g0 + k, fmax)];
  L.sm0 = (double)sc[imin32(imax32(g0 - 1, 0), fmax)];
}
// per-state C = Q_c^-1 (dof x dof) of factor f, upper triangle (qc_mode PERSTATE only: p.qc is the (B, n-1, dof, dof) tensor)
template <int DOF, typename IO>
DGP_HD void load_Qc_block(const GnParams& p, int64_t b, int f, Sym<DOF>& Cm) {
  IO raw[DOF * DOF];
  ld_block<IO, DOF * DOF>((const IO*)p.qc + (b * (p.n - 1) + f) * (DOF * DOF), p.vec_qc != 0, raw);
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = i; j < DOF; ++j) Cm(i, j) = (double)raw[i * DOF + j];
}
template <int DOF, int C, typename IO>
DGP_HD void load_lane_Q(const GnParams& p, int64_t b, int g0, bool traj_ok, LaneQ<2 * DOF, C, QK_KRON>& L) {
  const int64_t bb = traj_ok ? b : 0;
  const int fmax = p.n - 2;
#pragma unroll
  for (int k = 0; k < C; ++k) load_Qc_block<DOF, IO>(p, bb, imin32(g0 + k, fmax), L.c[k]);
  load_Qc_block<DOF, IO>(p, bb, imin32(imax32(g0 - 1, 0), fmax), L.cm0);
}

// (M (x) C) v for a 2 x 2 M = [[m00, m01],[m10, m11]] of scalars and a symmetric dof x dof C:  [C v_p, C v_v] mixed by M
template <int DOF>
DGP_HD void kron_apply(double m00, double m01, double m10, double m11, const Sym<DOF>& Cm, const double (&v)[2 * DOF], double (&o)[2 * DOF]) {
  double wp[DOF], wv[DOF];
#pragma unroll
  for (int a = 0; a < DOF; ++a) {
    double tp = 0.0, tv = 0.0;
#pragma unroll
    for (int c = 0; c < DOF; ++c) { tp += Cm(a, c) * v[c]; tv += Cm(a, c) * v[DOF + c]; }
    wp[a] = tp; wv[a] = tv;
  }
#pragma unroll
  for (int a = 0; a < DOF; ++a) {
    o[a] = m00 * wp[a] + m01 * wv[a];
    o[DOF + a] = m10 * wp[a] + m11 * wv[a];
  }
}
// Q^-1 = T (x) C as a symmetric d x d matrix (gp_factor.py:65-73)
template <int DOF>
DGP_HD void kron_to_sym(const GnParams& p, const Sym<DOF>& Cm, Sym<2 * DOF>& Q) {
#pragma unroll
  for (int i = 0; i < DOF; ++i)
#pragma unroll
    for (int j = 0; j < DOF; ++j) {
      if (j >= i) { Q(i, j) = p.qa * Cm(i, j); Q(DOF + i, DOF + j) = p.qc_ * Cm(i, j); }
      Q(i, DOF + j) = p.qb * Cm(i, j);
    }
}

// generic_rhs / generic_diag for QK_KRON: Cown = C of factor (g -> g+1), Cprev of (g-1 -> g)
template <int DOF>
DGP_HD void kron_rhs(const GnParams& p, int g, bool valid, const double (&x)[2 * DOF], const double (&xm)[2 * DOF],
                     const double (&xp)[2 * DOF], const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF],
                     const Sym<DOF>& Cown, const Sym<DOF>& Cprev, double (&r)[2 * DOF], ErrAcc& acc) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  const double dt = p.dt;
  const bool is_start = valid && g == 0, is_goal = valid && g == n - 1;
  const double mN = (valid && g < n - 1) ? 1.0 : 0.0, mP = (valid && g > 0) ? 1.0 : 0.0;
  const double w = is_start ? p.w_s : (is_goal ? p.w_g : 0.0);
  double s2 = 0.0, ep[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    ep[a] = (is_start ? mu_s[a] : mu_g[a]) - x[a];
    s2 += ep[a] * ep[a];
  }
  acc.e += 0.5 * w * s2; acc.eext += 0.5 * w * s2; acc.usg += (is_start || is_goal) ? 0.5 * s2 : 0.0;
  double eo[D], em[D];
#pragma unroll
  for (int a = 0; a < DOF; ++a) {
    eo[a] = xp[a] - (x[a] + dt * x[DOF + a]);              // e = x_{g+1} - Phi x_g (gp_factor.py:105)
    eo[DOF + a] = xp[DOF + a] - x[DOF + a];
    em[a] = x[a] - (xm[a] + dt * xm[DOF + a]);
    em[DOF + a] = x[DOF + a] - xm[DOF + a];
  }
  double wo[D], wm[D];                                     // Q_own e_own, Q_prev e_prev
  kron_apply<DOF>(p.qa, p.qb, p.qb, p.qc_, Cown, eo, wo);
  kron_apply<DOF>(p.qa, p.qb, p.qb, p.qc_, Cprev, em, wm);
  double q = 0.0, so = 0.0, qf = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    q += eo[a] * wo[a];
    so += eo[a] * eo[a];
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) t += p.q_fix[Sym<D>::idx(a, c)] * eo[c];     // fixed GP weight of err_ext (plan_layer.py:318-321), scalar operands
    qf += eo[a] * t;
  }
  acc.e += mN * (0.5 * q); acc.eext += mN * (0.5 * qf); acc.ugp += mN * (0.5 * so);
#pragma unroll
  for (int a = 0; a < DOF; ++a) {                          // eta_g = w (mu - x) + Phi^T Q_own e_own - Q_prev e_prev
    r[a] = w * ep[a] + mN * wo[a] - mP * wm[a];
    r[DOF + a] = w * ep[DOF + a] + mN * (dt * wo[a] + wo[DOF + a]) - mP * wm[DOF + a];
  }
}
template <int DOF>
DGP_HD void kron_diag(const GnParams& p, int g, bool valid, const Sym<DOF>& Cown, const Sym<DOF>& Cprev, Sym<2 * DOF>& Dm, double& m_next) {
  const int n = p.n;
  const bool is_start = valid && g == 0, is_goal = valid && g == n - 1;
  const double mN = (valid && g < n - 1) ? 1.0 : 0.0, mP = (valid && g > 0) ? 1.0 : 0.0;
  m_next = mN;
  const double w = is_start ? p.w_s : (is_goal ? p.w_g : 0.0);
  const double dbase = valid ? p.reg : 1.0;                // delta I (plan_layer.py:219); padding rows: identity row, x = 0
  // delta I + prior + mN (F (x) C_own) + mP (T (x) C_prev)
#pragma unroll
  for (int a = 0; a < DOF; ++a)
#pragma unroll
    for (int c = 0; c < DOF; ++c) {
      const double co = mN * Cown(a, c), cp_ = mP * Cprev(a, c);
      if (c >= a) {
        Dm(a, c) = ((a == c) ? dbase + w : 0.0) + p.qa * co + p.qa * cp_;
        Dm(DOF + a, DOF + c) = ((a == c) ? dbase + w : 0.0) + p.f11 * co + p.qc_ * cp_;
      }
      Dm(a, DOF + c) = p.e10 * co + p.qb * cp_;
    }
}

// Row g of the generic path, the part that does NOT depend on the SDF lookup (runs while the taps are in flight): priors + both GP
// factors -> eta (without the single-state factors) and the error partials.  Qown = Q^-1 of factor (g -> g+1), Qm of (g-1 -> g);
// every GP term carries a 0/1 lane mask instead of a branch, exactly like static_rhs.
template <int DOF>
DGP_HD void generic_rhs(const GnParams& p, int g, bool valid, const double (&x)[2 * DOF], const double (&xm)[2 * DOF],
                        const double (&xp)[2 * DOF], const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF],
                        const Sym<2 * DOF>& Qown, const Sym<2 * DOF>& Qm, double (&r)[2 * DOF], ErrAcc& acc) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  const double dt = p.dt;
  const bool is_start = valid && g == 0, is_goal = valid && g == n - 1;
  const double mN = (valid && g < n - 1) ? 1.0 : 0.0, mP = (valid && g > 0) ? 1.0 : 0.0;
  const double w = is_start ? p.w_s : (is_goal ? p.w_g : 0.0);
  double s2 = 0.0, ep[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    ep[a] = (is_start ? mu_s[a] : mu_g[a]) - x[a];
    s2 += ep[a] * ep[a];
  }
  acc.e += 0.5 * w * s2; acc.eext += 0.5 * w * s2; acc.usg += (is_start || is_goal) ? 0.5 * s2 : 0.0;
  double eo[D], em[D];
#pragma unroll
  for (int a = 0; a < DOF; ++a) {
    eo[a] = xp[a] - (x[a] + dt * x[DOF + a]);              // e = x_{g+1} - Phi x_g (gp_factor.py:105)
    eo[DOF + a] = xp[DOF + a] - x[DOF + a];
    em[a] = x[a] - (xm[a] + dt * xm[DOF + a]);
    em[DOF + a] = x[DOF + a] - xm[DOF + a];
  }
  double wo[D], wm[D];                                     // Qown e_own, Qm e_prev
  sym_times_vec_fwd<D>(Qown, eo, wo);
  sym_times_vec_fwd<D>(Qm, em, wm);
  double q = 0.0, so = 0.0, qf = 0.0;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    q += eo[a] * wo[a];
    so += eo[a] * eo[a];
    double t = 0.0;
#pragma unroll
    for (int c = 0; c < D; ++c) t += p.q_fix[Sym<D>::idx(a, c)] * eo[c];     // fixed GP weight of err_ext (plan_layer.py:318-321), scalar operands
    qf += eo[a] * t;
  }
  acc.e += mN * (0.5 * q);
  acc.eext += mN * (0.5 * ((p.qc_mode == QC_STATIC) ? q : qf));
  acc.ugp += mN * (0.5 * so);
  // eta_g = w (mu - x) + Phi^T Qown e_own - Qm e_prev ;   Phi^T v = [v_p ; dt v_p + v_v]
#pragma unroll
  for (int a = 0; a < DOF; ++a) {
    r[a] = w * ep[a] + mN * wo[a] - mP * wm[a];
    r[DOF + a] = w * ep[DOF + a] + mN * (dt * wo[a] + wo[DOF + a]) - mP * wm[DOF + a];
  }
}

// diagonal block of row g without the single-state factors: delta I + prior + mN Phi^T Qown Phi + mP Qm
template <int DOF>
DGP_HD void generic_diag(const GnParams& p, int g, bool valid, const Sym<2 * DOF>& Qown, const Sym<2 * DOF>& Qm, Sym<2 * DOF>& Dm,
                         double& m_next) {
  constexpr int D = 2 * DOF;
  const int n = p.n;
  const double dt = p.dt;
  const bool is_start = valid && g == 0, is_goal = valid && g == n - 1;
  const double mN = (valid && g < n - 1) ? 1.0 : 0.0, mP = (valid && g > 0) ? 1.0 : 0.0;
  m_next = mN;
  const double w = is_start ? p.w_s : (is_goal ? p.w_g : 0.0);
  const double dbase = valid ? p.reg : 1.0;                // delta I (plan_layer.py:219); padding rows: identity row, x = 0
  // A = Phi^T Q Phi:  A_pp = Q_pp,  A_pv = dt Q_pp + Q_pv,  A_vv = dt (dt Q_pp + Q_pv) + dt Q_vp + Q_vv
#pragma unroll
  for (int a = 0; a < DOF; ++a)
#pragma unroll
    for (int c = 0; c < DOF; ++c) {
      const double apv = dt * Qown(a, c) + Qown(a, DOF + c);
      if (c >= a) {
        Dm(a, c) = ((a == c) ? dbase + w : 0.0) + mN * Qown(a, c) + mP * Qm(a, c);
        Dm(DOF + a, DOF + c) = ((a == c) ? dbase + w : 0.0) + mN * (dt * apv + dt * Qown(DOF + a, c) + Qown(DOF + a, DOF + c)) + mP * Qm(DOF + a, DOF + c);
      }
      Dm(a, DOF + c) = mN * apv + mP * Qm(a, DOF + c);
    }
}

// Per-lane inputs that do not depend on the elimination: the obstacle factors of the lane's C states (all SDF loads
// issued together).
template <int C>
struct LaneFactors {          // obstacle factor of each of the lane's C states: weight, hinge cost, H = [ohx, ohy, 0..]
  double ow[C], oc[C], ohx[C], ohy[C];
};

// All loads are UNCONDITIONAL (rows / trajectories that do not exist read element 0 of the same tensor and are masked
// afterwards): a load inside a divergent `if (valid)` costs its own s_waitcnt, i.e. one exposed memory round trip each.
// The tap values stay in the I/O element type until they are used (lane_obstacle_finish): a conversion here would be
// the first USE of the load and pin the memory wait in front of whatever is scheduled under the loads.
// The four taps of a state are fetched as TWO pair loads (dwordx2 for f32 grids): elements xb, xb+1 of rows y1 and y2 with
// xb = min(x1, W-2), so that both clamped columns x1, x2 of sdf_utils.py:64-72 are among the two elements whatever the
// clamping did (x1 = x2 = 0 left of the grid, x1 = x2 = W-1 right of it).  Half the load instructions and half the L1
// tag look-ups of four scalar gathers -- the gather of 1024 lane addresses per wavefront is what the texture path of a CU
// spends ~0.5 us on per wavefront.  (Needs W >= 2: host-checked.)
template <typename IO> struct __attribute__((packed, aligned(sizeof(IO)))) TapPair { IO a, b; };
template <int C, typename IO>
struct LaneTaps {
  ObsAddr oa[C];
  IO d11[C], d21[C], d12[C], d22[C];    // RAW pair elements: row y1 (a, b), row y2 (a, b) -- see tap_values()
  bool first1[C], first2[C];            // column x1 / x2 is the FIRST element of the pair
  double eps[C], ow[C];
};
// the taps in the reference's naming: d11 = (x1,y1), d21 = (x2,y1), d12 = (x1,y2), d22 = (x2,y2)   (sdf_utils.py:76-79)
template <int C, typename IO>
DGP_HD void tap_values(const LaneTaps<C, IO>& t, int k, double& d11, double& d21, double& d12, double& d22) {
  d11 = (double)(t.first1[k] ? t.d11[k] : t.d21[k]); d21 = (double)(t.first2[k] ? t.d11[k] : t.d21[k]);
  d12 = (double)(t.first1[k] ? t.d12[k] : t.d22[k]); d22 = (double)(t.first2[k] ? t.d12[k] : t.d22[k]);
}

template <int DOF, int C, typename IO>
DGP_HD void lane_obstacle_loads(const GnParams& p, int64_t b, int g0, bool traj_ok, const double (&x)[C][2 * DOF], LaneTaps<C, IO>& t) {
  const int n = p.n;
  const int64_t W = p.sdf_cols;
  const IO* grid = (const IO*)p.sdf + (traj_ok ? b : 0) * p.sdf_bstride;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    t.eps[k] = p.eps_static; t.ow[k] = p.obs_w_fix;
    obstacle_addr(p, x[k][0], x[k][1], t.oa[k]);
  }
  if (p.eps) {
#pragma unroll
    for (int k = 0; k < C; ++k) t.eps[k] = ld<IO>(p.eps, (traj_ok && (g0 + k) < n) ? b * n + g0 + k : 0);
  }
  if (p.obs_w) {
#pragma unroll
    for (int k = 0; k < C; ++k) t.ow[k] = ld<IO>(p.obs_w, (traj_ok && (g0 + k) < n) ? b * n + g0 + k : 0);
  }
  if (grid_is_tiled(p)) {            // (compile-time in the product's kernels, see DGP_TL) 4 x 4 tiles: the four taps as scalar loads -- a column pair may straddle two tiles
#pragma unroll
    for (int k = 0; k < C; ++k) {
      int32_t o11, o21, o12, o22;
      tiled_tap_offsets(p, t.oa[k], o11, o21, o12, o22);
      t.d11[k] = grid[o11]; t.d21[k] = grid[o21]; t.d12[k] = grid[o12]; t.d22[k] = grid[o22];
      t.first1[k] = true; t.first2[k] = false;          // (tap_values(): d11 / d12 are the x1 taps, d21 / d22 the x2 taps)
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < C; ++k) {
    // grids have fewer than 2^31 elements (host-checked): 32-bit element offsets
    const int32_t xb = imin32(t.oa[k].x1, (int32_t)W - 2);
    const int32_t r1 = t.oa[k].y1 * (int32_t)W + xb, r2 = t.oa[k].y2 * (int32_t)W + xb;
#if defined(DGP_TAPS_NT) && defined(__HIP_DEVICE_COMPILE__)
    // experiment (profiles/tools/devbuild.py -DDGP_TAPS_NT): non-temporal tap loads -- per-sample grids are read once per launch
    TapPair<IO> p1, p2;
    if constexpr (sizeof(IO) == 4) {
      typedef uint64_t u64a4 __attribute__((aligned(4)));
      const uint64_t w1 = __builtin_nontemporal_load((const u64a4*)(grid + r1)), w2 = __builtin_nontemporal_load((const u64a4*)(grid + r2));
      __builtin_memcpy(&p1, &w1, 8); __builtin_memcpy(&p2, &w2, 8);
    } else {
      p1 = *(const TapPair<IO>*)(grid + r1); p2 = *(const TapPair<IO>*)(grid + r2);
    }
#else
    const TapPair<IO> p1 = *(const TapPair<IO>*)(grid + r1), p2 = *(const TapPair<IO>*)(grid + r2);
#endif
    t.d11[k] = p1.a; t.d21[k] = p1.b; t.d12[k] = p2.a; t.d22[k] = p2.b;
    t.first1[k] = (t.oa[k].x1 == xb); t.first2[k] = (t.oa[k].x2 == xb);
  }
}

// Orders the first USE of the lane's tap loads after the values in `after` have been computed: an empty asm statement that
// takes the loaded registers in/out and `after` as inputs.  Without it the compiler schedules the consumers of the loads
// (and with them the s_waitcnt) directly behind the loads, and the arithmetic that does not need them behind that, so a
// wavefront that is alone on its SIMD sits out the whole L2 / HBM round trip.  Emits no instruction; no-op on the host.
template <int C, typename IO, int N>
DGP_HD void lane_taps_use_after(LaneTaps<C, IO>& t, const double (&after)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(N <= 16, "too many anchor operands");
#pragma unroll
  for (int k = 0; k < C; ++k) {
    if constexpr (N == 16)
      asm volatile("" : "+v"(t.d11[k]), "+v"(t.d21[k]), "+v"(t.d12[k]), "+v"(t.d22[k])
                   : "v"(after[0]), "v"(after[1]), "v"(after[2]), "v"(after[3]), "v"(after[4]), "v"(after[5]), "v"(after[6]),
                     "v"(after[7]), "v"(after[8]), "v"(after[9]), "v"(after[10]), "v"(after[11]), "v"(after[12]),
                     "v"(after[13]), "v"(after[14]), "v"(after[15]));
    else if constexpr (N == 8)
      asm volatile("" : "+v"(t.d11[k]), "+v"(t.d21[k]), "+v"(t.d12[k]), "+v"(t.d22[k])
                   : "v"(after[0]), "v"(after[1]), "v"(after[2]), "v"(after[3]), "v"(after[4]), "v"(after[5]), "v"(after[6]),
                     "v"(after[7]));
    else
      asm volatile("" : "+v"(t.d11[k]), "+v"(t.d21[k]), "+v"(t.d12[k]), "+v"(t.d22[k]) : "v"(after[0]), "v"(after[N - 1]));
  }
#else
  (void)t; (void)after;
#endif
}

// Orders arithmetic on `v` after the tap addresses of all C states are known (so the loads go out first).
template <int C, typename IO, int D>
DGP_HD void lane_after_addresses(const LaneTaps<C, IO>& t, double (&v)[D]) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int a = 0; a < D; ++a)
    asm volatile("" : "+v"(v[a]) : "v"(t.oa[0].y2), "v"(t.oa[C / 2].y2), "v"(t.oa[C - 1].x2), "v"(t.oa[C - 1].y2));
#else
  (void)t; (void)v;
#endif
}

template <int C, typename IO>
DGP_HD void lane_obstacle_finish(const GnParams& p, int g0, bool traj_ok, const LaneTaps<C, IO>& t, LaneFactors<C>& f) {
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const bool valid = traj_ok && (g0 + k) < p.n;
    double d11, d21, d12, d22;
    tap_values<C, IO>(t, k, d11, d21, d12, d22);
    obstacle_finish(p, t.oa[k], d11, d21, d12, d22, t.eps[k], f.oc[k], f.ohx[k], f.ohy[k]);
    f.ow[k] = t.ow[k];
    if (!valid) { f.ow[k] = 0.0; f.oc[k] = 0.0; f.ohx[k] = 0.0; f.ohy[k] = 0.0; }
  }
}

template <int DOF, int C, typename IO>
DGP_HD void lane_prefetch(const GnParams& p, int64_t b, int g0, bool traj_ok, const double (&x)[C][2 * DOF], LaneFactors<C>& f) {
  LaneTaps<C, IO> t;
  lane_obstacle_loads<DOF, C, IO>(p, b, g0, traj_ok, x, t);
  lane_obstacle_finish<C, IO>(p, g0, traj_ok, t, f);
}

// small dense helpers on dxd blocks -----------------------------------------------------------------
template <int D> DGP_HD void sym_times_mat(const Sym<D>& S, const Mat<D>& A, Mat<D>& O) {       // O = S A
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t += S(a, k) * A.v[k][c];
      O.v[a][c] = t;
    }
}
template <int D> DGP_HD void sym_times_vec(const Sym<D>& S, const double (&v)[D], double (&o)[D]) {   // o = S v
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) t += S(a, k) * v[k];
    o[a] = t;
  }
}
template <int D> DGP_HD void sub_At_B_sym(Sym<D>& S, const Mat<D>& A, const Mat<D>& B) {          // S -= A^T B (symmetric result)
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = a; c < D; ++c) {
      double t = S(a, c);
#pragma unroll
      for (int k = 0; k < D; ++k) t -= A.v[k][a] * B.v[k][c];
      S(a, c) = t;
    }
}
template <int D> DGP_HD void sub_A_B_sym(Sym<D>& S, const Mat<D>& A, const Mat<D>& B) {           // S -= A B (symmetric result)
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = a; c < D; ++c) {
      double t = S(a, c);
#pragma unroll
      for (int k = 0; k < D; ++k) t -= A.v[a][k] * B.v[k][c];
      S(a, c) = t;
    }
}
template <int D> DGP_HD void sub_At_v(double (&o)[D], const Mat<D>& A, const double (&v)[D]) {    // o -= A^T v
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = o[a];
#pragma unroll
    for (int k = 0; k < D; ++k) t -= A.v[k][a] * v[k];
    o[a] = t;
  }
}
template <int D> DGP_HD void sub_A_v(double (&o)[D], const Mat<D>& A, const double (&v)[D]) {     // o -= A v
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = o[a];
#pragma unroll
    for (int k = 0; k < D; ++k) t -= A.v[a][k] * v[k];
    o[a] = t;
  }
}
template <int D> DGP_HD void neg_At_B(const Mat<D>& A, const Mat<D>& B, Mat<D>& O) {              // O = -A^T B
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t -= A.v[k][a] * B.v[k][c];
      O.v[a][c] = t;
    }
}
template <int D> DGP_HD void neg_A_B(const Mat<D>& A, const Mat<D>& B, Mat<D>& O) {               // O = -A B
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t -= A.v[a][k] * B.v[k][c];
      O.v[a][c] = t;
    }
}
template <int D> DGP_HD void sub_A_B(Mat<D>& O, const Mat<D>& A, const Mat<D>& B) {               // O -= A B
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = O.v[a][c];
#pragma unroll
      for (int k = 0; k < D; ++k) t -= A.v[a][k] * B.v[k][c];
      O.v[a][c] = t;
    }
}

template <int D> DGP_HD void add_A_B(Mat<D>& O, const Mat<D>& A, const Mat<D>& B) {               // O += A B
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = O.v[a][c];
#pragma unroll
      for (int k = 0; k < D; ++k) t += A.v[a][k] * B.v[k][c];
      O.v[a][c] = t;
    }
}
template <int D> DGP_HD void add_A_v(double (&o)[D], const Mat<D>& A, const double (&v)[D]) {     // o += A v
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = o[a];
#pragma unroll
    for (int k = 0; k < D; ++k) t += A.v[a][k] * v[k];
    o[a] = t;
  }
}
template <int D> DGP_HD void A_B(const Mat<D>& A, const Mat<D>& B, Mat<D>& O) {                   // O = A B
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t += A.v[a][k] * B.v[k][c];
      O.v[a][c] = t;
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
// Position of a lane inside its trajectory's lane group <-> index j of the block row it owns in the LPT-row system.
// LPT = 16 / 64: identity.  LPT = 32 (two 16-lane DPP rows per trajectory): rows are interleaved by parity,
//   lane = (j & 1) * 16 + (j >> 1),
// so that rows j and j +- s sit in the SAME DPP row, |s|/2 lanes apart, for every PCR stride s >= 2; only the stride-1
// exchanges cross the DPP rows (and go through ds_bpermute).
template <int LPT> DGP_HD int lane_to_row(int within) {
  if constexpr (LPT == 32) return ((within & 15) << 1) | (within >> 4);
  else return within;
}
template <int LPT> DGP_HD int row_to_lane(int j) {
  if constexpr (LPT == 32) return ((j & 1) << 4) | (j >> 1);
  else return j;
}

// Cross-lane access helper: value held by the lane that owns row j - S (lo) / j + S (hi) of the same trajectory.
// Whenever both rows live in one 16-lane DPP row the neighbour is read with a DPP row shift (a plain VALU move: no LDS
// round trip, no s_waitcnt); otherwise through ds_bpermute.  Rows without such a neighbour receive 0 (DPP bound_ctrl) or
// their own value (bpermute); both are harmless because the coupling block that multiplies the fetched data is zero there.
template <int LPT, int S, typename Ctx>
struct Nbr {
  static constexpr bool kDpp = (LPT == 16) || (LPT == 32 && S >= 2);     // DPP path: missing neighbours read as 0
  static constexpr int kShift = (LPT == 32) ? S / 2 : S;                  // lane distance inside the DPP row
  Ctx& cx;
  int src_lo, src_hi, src_par;
  DGP_HD Nbr(Ctx& c, int j) : cx(c) {
    const int lane = c.lane();
    const int base = lane & ~(LPT - 1);
    src_lo = (j >= S) ? base + row_to_lane<LPT>(j - S) : lane;
    src_hi = (j + S < LPT) ? base + row_to_lane<LPT>(j + S) : lane;
    src_par = base + row_to_lane<LPT>((j ^ S) & (LPT - 1));
  }
  DGP_HD double lo(double v) const {
    if constexpr (kDpp) return cx.template row_from_lower<(kShift > 0 ? kShift : 1)>(v);
    else return cx.fetch(v, src_lo);
  }
  DGP_HD double hi(double v) const {
    if constexpr (kDpp) return cx.template row_from_upper<(kShift > 0 ? kShift : 1)>(v);
    else return cx.fetch(v, src_hi);
  }
  // value held by the lane that owns row j ^ S, for S = LPT / 2 (the only partner of row j in the last PCR round):
  // with the DPP mappings that lane is 8 positions away inside the 16-lane row in either direction -> one row rotate
  DGP_HD double partner(double v) const {
    static_assert(2 * S == LPT || S == 1, "partner() is for the last round");
    if constexpr (kDpp) return cx.template row_rotate<8>(v);
    else return cx.fetch(v, src_par);
  }
};

// The LAST round (S = LPT / 2): row i has exactly one partner, row i ^ S -- rows i < S couple to it through their own U
// (block (i, i+S)), rows i >= S through the partner's U transposed (block (i, i-S) = U_{i-S}^T) and their own U is zero.
// So K = U_i + U_partner^T (one of the two terms is zero) and one elimination  D_i -= K D_p^-1 K^T,  r_i -= K D_p^-1 r_p
// replaces the two half-empty ones of the generic round; no coupling survives.
template <int D, int LPT, int S, typename Ctx>
DGP_HD void pcr_last_round(Ctx& cx, int i, Sym<D>& Dm, const Mat<D>& U, double (&r)[D], SpdCheck<Ctx>& ok) {
  const Nbr<LPT, S, Ctx> nb(cx, i);
  Sym<D> Di, DiP;
  sym_inverse<D>(Dm, Di, ok);
  double rP[D];
  Mat<D> K;
#pragma unroll
  for (int k = 0; k < D * (D + 1) / 2; ++k) DiP.v[k] = nb.partner(Di.v[k]);
#pragma unroll
  for (int a = 0; a < D; ++a) rP[a] = nb.partner(r[a]);
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) K.v[c][a] = nb.partner(U.v[a][c]);            // U_partner^T
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) K.v[a][c] += U.v[a][c];
  double T[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t += K.v[a][k] * DiP(k, c);
      T[a][c] = t;
    }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = r[a];
#pragma unroll
    for (int k = 0; k < D; ++k) t -= T[a][k] * rP[k];
    r[a] = t;
#pragma unroll
    for (int c = a; c < D; ++c) {
      double w = Dm(a, c);
#pragma unroll
      for (int k = 0; k < D; ++k) w -= T[a][k] * K.v[c][k];
      Dm(a, c) = w;
    }
  }
}

// one PCR round at stride S.  Owner-computes form: what row i contributes to its RIGHT neighbour's update,
//     W_i = U_i^T D_i^-1 U_i (symmetric)   and   v_i = U_i^T D_i^-1 r_i,
// only involves row i's own data, so lane i forms them (through G_i = D_i^-1 U_i and y_i = D_i^-1 r_i, which its LEFT
// neighbour needs anyway) and ships d(d+1)/2 + d values to the right instead of D^-1, U, r (d(d+1)/2 + d^2 + d) for the
// receiver to multiply out.  From the right a lane fetches D_R^-1, y_R, G_R:
//     D_i' = D_i - W_L - (U_i D_R^-1) U_i^T ;   r_i' = r_i - v_L - U_i y_R ;   U_i' = -U_i G_R.
// Same flop count as eliminating with the fetched rows, 29 % fewer cross-lane moves (d = 4: 44 instead of 60 values per round),
// and no fetched left row (U_L, D_L^-1) or U_L^T D_L^-1 product alive next to the lane's own state.
template <int D, int LPT, int S, typename Ctx>
DGP_HD void pcr_round(Ctx& cx, int i, Sym<D>& Dm, Mat<D>& U, double (&r)[D], SpdCheck<Ctx>& ok) {
  constexpr bool last = (2 * S >= LPT);
  if constexpr (last && LPT >= 2) { pcr_last_round<D, LPT, S>(cx, i, Dm, U, r, ok); return; }
  typedef Nbr<LPT, S, Ctx> NB;
  const NB nb(cx, i);
  const bool has_l = (i >= S);
  Sym<D> Di;
  sym_inverse<D>(Dm, Di, ok);
  Mat<D> G;                       // G = D^-1 U
  double y[D];                    // y = D^-1 r
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) t += Di(a, k) * r[k];
    y[a] = t;
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double g = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) g += Di(a, k) * U.v[k][c];
      G.v[a][c] = g;
    }
  }
  // ---- left neighbour: its W and v (a DPP row shift already yields 0 where there is no left neighbour)
  double rn[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) v += U.v[k][a] * y[k];
    const double vL = nb.lo(v);
    rn[a] = r[a] - ((NB::kDpp || has_l) ? vL : 0.0);
#pragma unroll
    for (int c = a; c < D; ++c) {
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) w += U.v[k][a] * G.v[k][c];
      const double wL = nb.lo(w);
      Dm(a, c) -= (NB::kDpp || has_l) ? wL : 0.0;
    }
  }
  // ---- right neighbour (U_i == 0 where there is none, so whatever was fetched there is multiplied away)
  {
    Sym<D> DiR;
    double yR[D];
#pragma unroll
    for (int k = 0; k < D * (D + 1) / 2; ++k) DiR.v[k] = nb.hi(Di.v[k]);
#pragma unroll
    for (int a = 0; a < D; ++a) yR[a] = nb.hi(y[a]);
    double T[D][D];               // T = U D_R^-1
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
      for (int k = 0; k < D; ++k) t -= U.v[a][k] * yR[k];
      rn[a] = t;
#pragma unroll
      for (int c = a; c < D; ++c) {
        double w = Dm(a, c);
#pragma unroll
        for (int k = 0; k < D; ++k) w -= T[a][k] * U.v[c][k];
        Dm(a, c) = w;
      }
    }
    Mat<D> GR, Un;
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = 0; c < D; ++c) GR.v[a][c] = nb.hi(G.v[a][c]);
#pragma unroll
    for (int a = 0; a < D; ++a)
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t -= U.v[a][k] * GR.v[k][c];
        Un.v[a][c] = t;
      }
    U = Un;
  }
#pragma unroll
  for (int a = 0; a < D; ++a) r[a] = rn[a];
}

// Scheduling fence: nothing is moved across it by the instruction scheduler (no instruction is emitted).  The d = 6 kernels
// use it between the stages of a PCR round and between the phases of the Woodbury elimination: left alone, the scheduler
// interleaves the stages of these ~2000-instruction straight-line blocks, every block of every stage is live at once, and the
// working set lands in AGPRs (1000 parking moves) and scratch.
DGP_HD void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// Scheduling fences at the PHASE boundaries of the program (where profiles/tools/phase_probe.hip puts its time stamps): with them the Woodbury STEP kernels
// come out with fewer AGPR parking moves (d = 4: 134 -> 55) and run 0.7 % (d = 4: 10.14 -> 10.07 us) / 1.8 % (d = 6: 27.04 -> 26.54 us) faster
// (profiles/r03_kernel_variants.txt, "marks" builds); the fused loop gets SLOWER with them (76.2 -> 77.3 us per 10 iterations), so only MODE_STEP of the
// QK_WB / QK_WBR kernels has them (-DDGP_PHASE_FENCES=0: none).
#ifndef DGP_PHASE_FENCES
#define DGP_PHASE_FENCES 1
#endif
template <bool ON> DGP_HD void phase_fence() {
  if constexpr (ON && DGP_PHASE_FENCES != 0) sched_fence();
}

// pcr_round in a register-lean order (same arithmetic, same results as pcr_round): stage by stage, each block dies before the
// next one is born -- peak 117 doubles for d = 6 (D, U, r, D^-1, y, D_R^-1, y_R) instead of ~200:
//   1. D^-1, y = D^-1 r                       2. fetch D_R^-1, y_R;  row by row: T_a = U_a D_R^-1,  r_a -= U_a y_R,  D_a. -= T_a U^T
//   3. G = D^-1 U (D^-1 dies)                 4. v = U^T y -> right neighbour's r (y dies)
//   5. column by column: W_.c = U^T G_.c -> right neighbour's D;  fetch G_R,.c (G_.c dies);  U'_.c = -U G_R,.c
template <int D, int LPT, int S, typename Ctx>
DGP_HD void pcr_round_lean(Ctx& cx, int i, Sym<D>& Dm, Mat<D>& U, double (&r)[D], SpdCheck<Ctx>& ok) {
  typedef Nbr<LPT, S, Ctx> NB;
  const NB nb(cx, i);
  const bool has_l = (i >= S);
  Sym<D> Di;
  double y[D];
  sched_fence();
  sym_inverse<D>(Dm, Di, ok);
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) t += Di(a, k) * r[k];
    y[a] = t;
  }
  sched_fence();
  {
    Sym<D> DiR;
    double yR[D];
#pragma unroll
    for (int k = 0; k < D * (D + 1) / 2; ++k) DiR.v[k] = nb.hi(Di.v[k]);
#pragma unroll
    for (int a = 0; a < D; ++a) yR[a] = nb.hi(y[a]);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double Ta[D];
#pragma unroll
      for (int c = 0; c < D; ++c) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < D; ++k) t += U.v[a][k] * DiR(k, c);
        Ta[c] = t;
      }
      double t = r[a];
#pragma unroll
      for (int k = 0; k < D; ++k) t -= U.v[a][k] * yR[k];
      r[a] = t;
#pragma unroll
      for (int c = a; c < D; ++c) {
        double w = Dm(a, c);
#pragma unroll
        for (int k = 0; k < D; ++k) w -= Ta[k] * U.v[c][k];
        Dm(a, c) = w;
      }
    }
  }
  sched_fence();
  Mat<D> G;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double g = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) g += Di(a, k) * U.v[k][c];
      G.v[a][c] = g;
    }
  sched_fence();
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) v += U.v[k][a] * y[k];
    const double vL = nb.lo(v);
    r[a] -= (NB::kDpp || has_l) ? vL : 0.0;
  }
  sched_fence();
  Mat<D> Un;
#pragma unroll
  for (int c = 0; c < D; ++c) {
#pragma unroll
    for (int a = 0; a <= c; ++a) {
      double w = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) w += U.v[k][a] * G.v[k][c];
      const double wL = nb.lo(w);
      Dm(a, c) -= (NB::kDpp || has_l) ? wL : 0.0;
    }
    double GRc[D];
#pragma unroll
    for (int k = 0; k < D; ++k) GRc[k] = nb.hi(G.v[k][c]);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) t -= U.v[a][k] * GRc[k];
      Un.v[a][c] = t;
    }
    if (c % 2 == 1) sched_fence();
  }
  U = Un;
  sched_fence();
}

// ---------------------------------------------------------------------------------------------------
// The PCR round on the LDL^T factors of D instead of on D^-1 (-DDGP_PCR_LDL=<d>: for state dimension d; 1: every dimension; 0: never).
//   D = L Dg L^T (unit lower L);  Z = L^-1 U,  Zs = Dg^-1 Z,  q = L^-1 r,  qs = Dg^-1 q
//   to the right neighbour (owner computes):  W = U^T D^-1 U = Z^T Zs,   v = U^T D^-1 r = Z^T qs
//   from the right neighbour (L_R, Dg_R^-1, qs_R, Zs_R -- as many values as D_R^-1, y_R, G_R):
//       Zt = U L_R^-T (D forward substitutions),  D -= (Zt Dg_R^-1) Zt^T,  r -= Zt qs_R,  U' = -Zt Zs_R
// Triangular solves instead of products with explicit inverses: per round 310 instead of 365 multiply-adds for d = 4, ~960 instead of
// ~1 160 for d = 6 -- at the price of d dependent pivots where the block inverse has two.  Stage order as in pcr_round_lean.
// Measured (round 3, profiles/r03_kernel_variants.txt, B = 4096): d = 4 step 3 718 -> 3 436 instructions, 10.23 -> 10.10 us, fused loop 8.02 -> 7.58 us
// per iteration; d = 6 8 793 instead of 9 511 instructions but 26.5-27.2 against 26.9-27.1 us (the six dependent pivots and 260 more AGPR moves eat the
// saving).  A variant with two BLOCK pivots (the chain of the block inverse, the flops of the LDL^T; profiles/tools/r03_experiments.patch) is slower
// than the scalar one for d = 4 (10.14 us) and for the d = 6 step (27.5 us).  Hence: d = 4 only, scalar pivots.
// ---------------------------------------------------------------------------------------------------
#ifndef DGP_PCR_LDL
#define DGP_PCR_LDL 4
#endif
template <int D> struct Ldl {
  double L[D][D], dinv[D];            // strict lower part of L used
  template <typename OK>
  DGP_HD void factor(const Sym<D>& S, OK& ok) {
    double dd[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      double v = S(j, j);
      double w[D];
#pragma unroll
      for (int k = 0; k < j; ++k) { w[k] = L[j][k] * dd[k]; v -= L[j][k] * w[k]; }
      dd[j] = v;
      ok.require(v > 0.0);
      dinv[j] = pivot_rcp(v);
#pragma unroll
      for (int i = j + 1; i < D; ++i) {
        double u = S(i, j);
#pragma unroll
        for (int k = 0; k < j; ++k) u -= L[i][k] * w[k];
        L[i][j] = u * dinv[j];
      }
    }
  }
  // x <- L^-1 x
  DGP_HD void fwd(double (&x)[D]) const {
#pragma unroll
    for (int f = 1; f < D; ++f) {
      double v = x[f];
#pragma unroll
      for (int g = 0; g < f; ++g) v -= L[f][g] * x[g];
      x[f] = v;
    }
  }
  DGP_HD void scale(const double (&x)[D], double (&xs)[D]) const {      // xs = Dg^-1 x
#pragma unroll
    for (int f = 0; f < D; ++f) xs[f] = dinv[f] * x[f];
  }
  DGP_HD void solve(const double (&c)[D], double (&m)[D]) const {      // m = D^-1 c
    double y[D];
#pragma unroll
    for (int f = 0; f < D; ++f) y[f] = c[f];
    fwd(y);
#pragma unroll
    for (int f = D - 1; f >= 0; --f) {
      double v = dinv[f] * y[f];
#pragma unroll
      for (int g = f + 1; g < D; ++g) v -= L[g][f] * m[g];
      m[f] = v;
    }
  }
  // the factor of another lane, fetched value by value with `get` (nb.hi / nb.partner)
  template <typename Get> DGP_HD void fetch(const Ldl& own, Get&& get) {
#pragma unroll
    for (int f = 0; f < D; ++f) {
      dinv[f] = get(own.dinv[f]);
#pragma unroll
      for (int g = 0; g < f; ++g) L[f][g] = get(own.L[f][g]);
    }
  }
};

template <int D, int LPT, int S, typename Ctx>
DGP_HD void pcr_round_ldl(Ctx& cx, int i, Sym<D>& Dm, Mat<D>& U, double (&r)[D], SpdCheck<Ctx>& ok) {
  typedef Nbr<LPT, S, Ctx> NB;
  typedef Ldl<D> FT;
  const NB nb(cx, i);
  const bool has_l = (i >= S);
  constexpr bool kFence = (D == 6);
  if constexpr (kFence) sched_fence();
  FT F;
  F.factor(Dm, ok);
  double q[D], qs[D];
#pragma unroll
  for (int a = 0; a < D; ++a) q[a] = r[a];
  F.fwd(q);
  F.scale(q, qs);
  if constexpr (kFence) sched_fence();
  // ---- right neighbour: its factor and qs
  Mat<D> Zt;                        // Zt = U L_R^-T  (row a: L_R^-1 applied to row a of U)
  {
    FT FR;
    double qR[D];
    FR.fetch(F, [&](double v) { return nb.hi(v); });
#pragma unroll
    for (int f = 0; f < D; ++f) qR[f] = nb.hi(qs[f]);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double row[D];
#pragma unroll
      for (int c = 0; c < D; ++c) row[c] = U.v[a][c];
      FR.fwd(row);
      double t = r[a];
#pragma unroll
      for (int f = 0; f < D; ++f) { Zt.v[a][f] = row[f]; t -= row[f] * qR[f]; }
      r[a] = t;
    }
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double zs[D];
      FR.scale(Zt.v[a], zs);
#pragma unroll
      for (int c = a; c < D; ++c) {
        double w = Dm(a, c);
#pragma unroll
        for (int f = 0; f < D; ++f) w -= zs[f] * Zt.v[c][f];
        Dm(a, c) = w;
      }
    }
  }
  if constexpr (kFence) sched_fence();
  // ---- own Z = L^-1 U (in place of U, which is dead now), column by column; v and W to the right neighbour
#pragma unroll
  for (int c = 0; c < D; ++c) {
    double col[D];
#pragma unroll
    for (int a = 0; a < D; ++a) col[a] = U.v[a][c];
    F.fwd(col);
#pragma unroll
    for (int a = 0; a < D; ++a) U.v[a][c] = col[a];
  }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double v = 0.0;
#pragma unroll
    for (int f = 0; f < D; ++f) v += U.v[f][a] * qs[f];
    const double vL = nb.lo(v);
    r[a] -= (NB::kDpp || has_l) ? vL : 0.0;
  }
  if constexpr (kFence) sched_fence();
  Mat<D> Un;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    double zc[D], zsc[D];           // column c of Z and of Zs = B^-1 Z
#pragma unroll
    for (int f = 0; f < D; ++f) zc[f] = U.v[f][c];
    F.scale(zc, zsc);
#pragma unroll
    for (int a = 0; a <= c; ++a) {
      double w = 0.0;
#pragma unroll
      for (int f = 0; f < D; ++f) w += U.v[f][a] * zsc[f];
      const double wL = nb.lo(w);
      Dm(a, c) -= (NB::kDpp || has_l) ? wL : 0.0;
    }
    double zR[D];
#pragma unroll
    for (int f = 0; f < D; ++f) zR[f] = nb.hi(zsc[f]);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      double t = 0.0;
#pragma unroll
      for (int f = 0; f < D; ++f) t -= Zt.v[a][f] * zR[f];
      Un.v[a][c] = t;
    }
    if constexpr (kFence) { if (c % 2 == 1) sched_fence(); }
  }
  U = Un;
  if constexpr (kFence) sched_fence();
}

// the last round (one partner, see pcr_last_round) and the final solve on the factors
template <int D, int LPT, int S, typename Ctx>
DGP_HD void pcr_last_round_ldl(Ctx& cx, int i, Sym<D>& Dm, const Mat<D>& U, double (&r)[D], SpdCheck<Ctx>& ok) {
  typedef Ldl<D> FT;
  const Nbr<LPT, S, Ctx> nb(cx, i);
  FT F, FP;
  F.factor(Dm, ok);
  double q[D], qs[D], qP[D];
#pragma unroll
  for (int a = 0; a < D; ++a) q[a] = r[a];
  F.fwd(q);
  F.scale(q, qs);
  FP.fetch(F, [&](double v) { return nb.partner(v); });
#pragma unroll
  for (int f = 0; f < D; ++f) qP[f] = nb.partner(qs[f]);
  Mat<D> K;
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) K.v[c][a] = nb.partner(U.v[a][c]);            // U_partner^T
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) K.v[a][c] += U.v[a][c];
  Mat<D> Kt;
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double row[D];
#pragma unroll
    for (int c = 0; c < D; ++c) row[c] = K.v[a][c];
    FP.fwd(row);
    double t = r[a];
#pragma unroll
    for (int f = 0; f < D; ++f) { Kt.v[a][f] = row[f]; t -= row[f] * qP[f]; }
    r[a] = t;
  }
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double zs[D];
    FP.scale(Kt.v[a], zs);
#pragma unroll
    for (int c = a; c < D; ++c) {
      double w = Dm(a, c);
#pragma unroll
      for (int f = 0; f < D; ++f) w -= zs[f] * Kt.v[c][f];
      Dm(a, c) = w;
    }
  }
}

// which dimension gets the register-lean round (tuning aid: -DDGP_PCR_LEAN_ALL=1 / -DDGP_PCR_LEAN_NONE=1)
#if defined(DGP_PCR_LEAN_ALL)
#define DGP_PCR_LEAN_D(D) true
#elif defined(DGP_PCR_LEAN_NONE)
#define DGP_PCR_LEAN_D(D) false
#else
#define DGP_PCR_LEAN_D(D) ((D) == 6)
#endif
// LDL6: the caller asks for the LDL^T rounds at d = 6 as well (round 4: the forward Woodbury kernels, whose recovery state is parked in LDS during the rounds)
template <int D, int LPT, int S, bool LEAN, bool LDL6 = false, typename Ctx>
DGP_HD void pcr_round_any(Ctx& cx, int i, Sym<D>& Dm, Mat<D>& U, double (&r)[D], SpdCheck<Ctx>& ok) {
  constexpr bool last = (2 * S >= LPT);
  // Only the rounds whose exchanges are DPP row shifts (LPT = 16; LPT = 32 from stride 2 on).  With the lean order on the ds_bpermute
  // rounds of LPT = 64, hipcc built a <3,64,2,double,STEP,general> kernel that returns wrong results (4e-2 off; the same source
  // is exact on the CPU wavefront emulator, every other shape is exact on the GPU) -- found by tests/stress_random_configs.py.
  // And not in the general-covariance kernels (LEAN = false there): <3,16,4,double,STEP,general> -- 355 spilled VGPRs, 1.4 KB of scratch
  // per lane -- came out wrong (O(1) errors) with the lean rounds, again only on the GPU (tests/test_hip_every_kernel.py pins every
  // instantiation against the C oracle since).
  constexpr bool kLdl = LEAN && (DGP_PCR_LDL == 1 || DGP_PCR_LDL == D || (LDL6 && D == 6)) && Nbr<LPT, S, Ctx>::kDpp && LPT != 64;
  if constexpr (kLdl && !last) pcr_round_ldl<D, LPT, S>(cx, i, Dm, U, r, ok);
  else if constexpr (kLdl && last) pcr_last_round_ldl<D, LPT, S>(cx, i, Dm, U, r, ok);
  else if constexpr (LEAN && DGP_PCR_LEAN_D(D) && !last && Nbr<LPT, S, Ctx>::kDpp && LPT != 64) pcr_round_lean<D, LPT, S>(cx, i, Dm, U, r, ok);
  else pcr_round<D, LPT, S>(cx, i, Dm, U, r, ok);
}

template <int D, int LPT, bool LEAN, bool LDL6 = false, typename Ctx>
DGP_HD void pcr_solve(Ctx& cx, int i, Sym<D>& Dm, Mat<D>& U, double (&r)[D], double (&x)[D], SpdCheck<Ctx>& ok) {
  if constexpr (LPT > 1) pcr_round_any<D, LPT, 1, LEAN, LDL6>(cx, i, Dm, U, r, ok);
  if constexpr (LPT > 2) pcr_round_any<D, LPT, 2, LEAN, LDL6>(cx, i, Dm, U, r, ok);
  if constexpr (LPT > 4) pcr_round_any<D, LPT, 4, LEAN, LDL6>(cx, i, Dm, U, r, ok);
  if constexpr (LPT > 8) pcr_round_any<D, LPT, 8, LEAN, LDL6>(cx, i, Dm, U, r, ok);
  if constexpr (LPT > 16) pcr_round_any<D, LPT, 16, LEAN, LDL6>(cx, i, Dm, U, r, ok);
  if constexpr (LPT > 32) pcr_round_any<D, LPT, 32, LEAN, LDL6>(cx, i, Dm, U, r, ok);
  if constexpr (LEAN && DGP_PCR_LEAN_D(D) && LPT != 64) sched_fence();
  if constexpr (LEAN && (DGP_PCR_LDL == 1 || DGP_PCR_LDL == D || (LDL6 && D == 6)) && LPT == 16) {
    Ldl<D> F;
    F.factor(Dm, ok);
    F.solve(r, x);
    return;
  }
  Sym<D> Di;
  sym_inverse<D>(Dm, Di, ok);                   // (block inverse: two reciprocals deep, against d sequential pivots of a solve)
  sym_times_vec<D>(Di, r, x);
}

// sum over the LPT lanes of one trajectory (butterfly); every lane gets the total
template <int LPT, typename Ctx>
DGP_HD double group_sum(Ctx& cx, double v) {
  if constexpr (LPT == 16) {       // the group is one DPP row: four row rotations, no LDS round trips
    v += cx.template row_rotate<8>(v);
    v += cx.template row_rotate<4>(v);
    v += cx.template row_rotate<2>(v);
    v += cx.template row_rotate<1>(v);
    return v;
  } else {
    const int lane = cx.lane();
#pragma unroll
    for (int m = LPT / 2; m >= 1; m >>= 1) v += cx.fetch(v, lane ^ m);
    return v;
  }
}
// sum over the LPT lanes of one trajectory, valid in the FIRST lane of the group only (that is where err / err_ext are
// written from).  With 16 lanes per trajectory it is four DPP row shifts + adds, no LDS round trips.
template <int LPT, typename Ctx>
DGP_HD double group_sum_to_first(Ctx& cx, double v) {
  if constexpr (LPT == 16) {
    v += cx.template row_from_upper<8>(v);
    v += cx.template row_from_upper<4>(v);
    v += cx.template row_from_upper<2>(v);
    v += cx.template row_from_upper<1>(v);
    return v;
  } else {
    return group_sum<LPT>(cx, v);
  }
}

// ---------------------------------------------------------------------------------------------------
// The coupling blocks U_k (block (k, k+1)) of the rows a lane owns, in two representations:
//   generic (QSTAT = false): U_k = -m_k Phi^T Q_k with the row's symmetric Q_k^-1 (d(d+1)/2 values instead of d^2) and a
//                            0/1 mask; products with U_k are formed from Q_k and the two-band Phi on the fly;
//   static  (QSTAT = true) : static covariances with a diagonal Q_c_inv.  U_k = m_k * p.u_fix with a 0/1 mask per row --
//                            nothing but the masks is kept in vector registers, the block itself is a scalar operand and
//                            its structural zeros (gp_nz) are skipped.
// ---------------------------------------------------------------------------------------------------
//   Kronecker (QK_KRON)    : per-state Q_c^-1 tensors.  Q_k^-1 = T (x) C_k, hence U_k = -m_k (E (x) C_k) and Phi^T Q_k^-1 Phi = F (x) C_k
//                            with the 2 x 2 constants T, E, F (GnParams::qa.., e10, e11, f11; scalar operands): a row keeps the
//                            symmetric dof x dof C_k (3 values for d = 4 instead of the 10 of Q_k^-1) and every product with U_k
//                            is "C_k applied to the position and velocity halves, mixed by a constant 2 x 2" (kron_apply).
template <int D, int N, int QK> struct Coupling;
template <int D, int N> struct Coupling<D, N, QK_GENERAL> { Sym<D> q[N]; double m[N]; double dt; };
template <int D, int N> struct Coupling<D, N, QK_STATIC> { double m[N]; };
template <int D, int N> struct Coupling<D, N, QK_KRON> { Sym<D / 2> c[N]; double m[N]; };
template <int D, int N> struct Coupling<D, N, QK_SCALED> : Coupling<D, N, QK_STATIC> {};      // m[k] = mask x s_k: every product below is linear in it, except ...

// G = S^-1 U_k
template <int D, int N>
DGP_HD void coup_SinvU(const GnParams&, const Coupling<D, N, QK_GENERAL>& cp, int k, const Sym<D>& Si, Mat<D>& G) {
  // S^-1 U = -m (S^-1 Phi^T) Q,   (S^-1 Phi^T)[a][c] = S^-1[a][c] + (c < dof ? dt S^-1[a][dof + c] : 0)
  constexpr int DOF = D / 2;
  double A[D][D];
  const double nm = -cp.m[k];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) A[a][c] = nm * ((c < DOF) ? Si(a, c) + cp.dt * Si(a, DOF + c) : Si(a, c));
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) t += A[a][q] * cp.q[k](q, c);
      G.v[a][c] = t;
    }
}
template <int D, int N>
DGP_HD void coup_SinvU(const GnParams& p, const Coupling<D, N, QK_STATIC>& cp, int k, const Sym<D>& Si, Mat<D>& G) {
  Sym<D> Sm;
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) Sm.v[i] = cp.m[k] * Si.v[i];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) if (gp_nz<D>(q, c)) t += Sm(a, q) * p.u_fix[q * D + c];
      G.v[a][c] = t;
    }
}
// S -= U_k^T B (symmetric result).  Static form: B is G_k = S_k^-1 U_k, which already carries the mask.
template <int D, int N>
DGP_HD void coup_sub_UtB_sym(const GnParams&, const Coupling<D, N, QK_GENERAL>& cp, int k, Sym<D>& S, const Mat<D>& B) {
  // U^T B = -Q Phi B (B carries the mask):  S += Q (Phi B),  (Phi B)[a] = B[a] + dt B[dof + a] for a < dof
  constexpr int DOF = D / 2;
  double PB[D][D];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) PB[a][c] = (a < DOF) ? B.v[a][c] + cp.dt * B.v[DOF + a][c] : B.v[a][c];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = a; c < D; ++c) {
      double t = S(a, c);
#pragma unroll
      for (int q = 0; q < D; ++q) t += cp.q[k](a, q) * PB[q][c];
      S(a, c) = t;
    }
}
template <int D, int N>
DGP_HD void coup_sub_UtB_sym(const GnParams& p, const Coupling<D, N, QK_STATIC>&, int, Sym<D>& S, const Mat<D>& B) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = a; c < D; ++c) {
      double t = S(a, c);
#pragma unroll
      for (int q = 0; q < D; ++q) if (gp_nz<D>(q, a)) t -= p.u_fix[q * D + a] * B.v[q][c];
      S(a, c) = t;
    }
}
// (... U^T B with B = S^-1 U: the static form above leans on m^2 = m; with a scalar in the mask the second factor is applied here)
template <int D, int N>
DGP_HD void coup_sub_UtB_sym(const GnParams& p, const Coupling<D, N, QK_SCALED>& cp, int k, Sym<D>& S, const Mat<D>& B) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = a; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) if (gp_nz<D>(q, a)) t += p.u_fix[q * D + a] * B.v[q][c];
      S(a, c) -= cp.m[k] * t;
    }
}
// o -= U_k^T v
template <int D, int N>
DGP_HD void coup_sub_Ut_v(const GnParams&, const Coupling<D, N, QK_GENERAL>& cp, int k, double (&o)[D], const double (&v)[D]) {
  // o -= U^T v = o + m Q (Phi v)
  constexpr int DOF = D / 2;
  double pv[D];
#pragma unroll
  for (int a = 0; a < D; ++a) pv[a] = (a < DOF) ? v[a] + cp.dt * v[DOF + a] : v[a];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q) t += cp.q[k](a, q) * pv[q];
    o[a] += cp.m[k] * t;
  }
}
template <int D, int N>
DGP_HD void coup_sub_Ut_v(const GnParams& p, const Coupling<D, N, QK_STATIC>& cp, int k, double (&o)[D], const double (&v)[D]) {
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q) if (gp_nz<D>(q, a)) t += p.u_fix[q * D + a] * v[q];
    o[a] -= cp.m[k] * t;
  }
}
// o -= U_k v
template <int D, int N>
DGP_HD void coup_sub_U_v(const GnParams&, const Coupling<D, N, QK_GENERAL>& cp, int k, double (&o)[D], const double (&v)[D]) {
  // o -= U v = o + m Phi^T (Q v)
  constexpr int DOF = D / 2;
  double w[D];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q) t += cp.q[k](a, q) * v[q];
    w[a] = t;
  }
#pragma unroll
  for (int a = 0; a < D; ++a) o[a] += cp.m[k] * ((a < DOF) ? w[a] : cp.dt * w[a - DOF] + w[a]);
}
template <int D, int N>
DGP_HD void coup_sub_U_v(const GnParams& p, const Coupling<D, N, QK_STATIC>& cp, int k, double (&o)[D], const double (&v)[D]) {
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < D; ++q) if (gp_nz<D>(a, q)) t += p.u_fix[a * D + q] * v[q];
    o[a] -= cp.m[k] * t;
  }
}
// O = sgn * A U_k   (sgn = +1 / -1)
template <int D, int N>
DGP_HD void coup_A_U(const GnParams&, const Coupling<D, N, QK_GENERAL>& cp, int k, const Mat<D>& A, double sgn, Mat<D>& O) {
  // A U = -m (A Phi^T) Q,   (A Phi^T)[a][c] = A[a][c] + (c < dof ? dt A[a][dof + c] : 0)
  constexpr int DOF = D / 2;
  double AP[D][D];
  const double f = -sgn * cp.m[k];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) AP[a][c] = f * ((c < DOF) ? A.v[a][c] + cp.dt * A.v[a][DOF + c] : A.v[a][c]);
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) t += AP[a][q] * cp.q[k](q, c);
      O.v[a][c] = t;
    }
}
template <int D, int N>
DGP_HD void coup_A_U(const GnParams& p, const Coupling<D, N, QK_STATIC>& cp, int k, const Mat<D>& A, double sgn, Mat<D>& O) {
  const double f = sgn * cp.m[k];
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double t = 0.0;
#pragma unroll
      for (int q = 0; q < D; ++q) if (gp_nz<D>(q, c)) t += A.v[a][q] * p.u_fix[q * D + c];
      O.v[a][c] = f * t;
    }
}
// the block itself, in vector registers (the separator row's coupling is PCR state)
template <int D, int N>
DGP_HD void coup_get(const GnParams&, const Coupling<D, N, QK_GENERAL>& cp, int k, Mat<D>& U) {
  constexpr int DOF = D / 2;
  const double nm = -cp.m[k];
#pragma unroll
  for (int a = 0; a < DOF; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) {
      U.v[a][c] = nm * cp.q[k](a, c);
      U.v[DOF + a][c] = nm * (cp.dt * cp.q[k](a, c) + cp.q[k](DOF + a, c));
    }
}
template <int D, int N>
DGP_HD void coup_get(const GnParams& p, const Coupling<D, N, QK_STATIC>& cp, int k, Mat<D>& U) {
#pragma unroll
  for (int a = 0; a < D; ++a)
#pragma unroll
    for (int c = 0; c < D; ++c) U.v[a][c] = gp_nz<D>(a, c) ? cp.m[k] * p.u_fix[a * D + c] : 0.0;
}

// ---- QK_KRON forms of the six products above (E = [[qa, qb],[e10, e11]]; U_k = -m_k (E (x) C_k), U_k^T = -m_k (E^T (x) C_k)) ----
template <int D, int N>
DGP_HD void coup_SinvU(const GnParams& p, const Coupling<D, N, QK_KRON>& cp, int k, const Sym<D>& Si, Mat<D>& G) {
  // row a of S^-1 U = -m ( (E^T (x) C) S^-1[a,:]^T )^T
  const double nm = -cp.m[k];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double row[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) row[c] = Si(a, c);
    kron_apply<D / 2>(p.qa, p.e10, p.qb, p.e11, cp.c[k], row, o);
#pragma unroll
    for (int c = 0; c < D; ++c) G.v[a][c] = nm * o[c];
  }
}
template <int D, int N>
DGP_HD void coup_sub_UtB_sym(const GnParams& p, const Coupling<D, N, QK_KRON>& cp, int k, Sym<D>& S, const Mat<D>& B) {
  // S -= U^T B = S + (E^T (x) C) B   (B carries the mask), upper triangle only
#pragma unroll
  for (int c = 0; c < D; ++c) {
    double col[D], o[D];
#pragma unroll
    for (int a = 0; a < D; ++a) col[a] = B.v[a][c];
    kron_apply<D / 2>(p.qa, p.e10, p.qb, p.e11, cp.c[k], col, o);
#pragma unroll
    for (int a = 0; a <= c; ++a) S(a, c) += o[a];
  }
}
template <int D, int N>
DGP_HD void coup_sub_Ut_v(const GnParams& p, const Coupling<D, N, QK_KRON>& cp, int k, double (&o)[D], const double (&v)[D]) {
  double t[D];
  kron_apply<D / 2>(p.qa, p.e10, p.qb, p.e11, cp.c[k], v, t);            // (E^T (x) C) v
#pragma unroll
  for (int a = 0; a < D; ++a) o[a] += cp.m[k] * t[a];
}
template <int D, int N>
DGP_HD void coup_sub_U_v(const GnParams& p, const Coupling<D, N, QK_KRON>& cp, int k, double (&o)[D], const double (&v)[D]) {
  double t[D];
  kron_apply<D / 2>(p.qa, p.qb, p.e10, p.e11, cp.c[k], v, t);            // (E (x) C) v
#pragma unroll
  for (int a = 0; a < D; ++a) o[a] += cp.m[k] * t[a];
}
template <int D, int N>
DGP_HD void coup_A_U(const GnParams& p, const Coupling<D, N, QK_KRON>& cp, int k, const Mat<D>& A, double sgn, Mat<D>& O) {
  const double f = -sgn * cp.m[k];
#pragma unroll
  for (int a = 0; a < D; ++a) {
    double o[D];
    kron_apply<D / 2>(p.qa, p.e10, p.qb, p.e11, cp.c[k], A.v[a], o);     // row a of A (E (x) C)
#pragma unroll
    for (int c = 0; c < D; ++c) O.v[a][c] = f * o[c];
  }
}
template <int D, int N>
DGP_HD void coup_get(const GnParams& p, const Coupling<D, N, QK_KRON>& cp, int k, Mat<D>& U) {
  constexpr int DOF = D / 2;
  const double nm = -cp.m[k];
#pragma unroll
  for (int a = 0; a < DOF; ++a)
#pragma unroll
    for (int c = 0; c < DOF; ++c) {
      const double v = nm * cp.c[k](a, c);
      U.v[a][c] = p.qa * v; U.v[a][DOF + c] = p.qb * v;
      U.v[DOF + a][c] = p.e10 * v; U.v[DOF + a][DOF + c] = p.e11 * v;
    }
}

// ---------------------------------------------------------------------------------------------------
// LDS stash of the interior rows' S_k^-1 (d = 6 backward kernel).  Between the forward sweep and the interior recovery the PCR
// rounds need every register; for d = 6 the 3 x 21 doubles of S_k^-1 do not fit next to them.  They can be parked in LDS: lane-private
// slots of 11 x 16 bytes per block, lane stride a multiple of 16 bytes whose dword count / 4 is odd -> conflict-free 128-bit accesses.
// NS = number of blocks the kernel's LDS budget allows (0: no stash; the rest stays in registers).
// ---------------------------------------------------------------------------------------------------
// how many S_k^-1 blocks a kernel parks in LDS.  Measured on d = 6, (16,4), B = 4096, round 2: the backward kernel (whose chain rule
// follows the adjoint solve and needs every register again) 91.3 -> 71.2 us with all three blocks stashed; the static step kernel of that
// round was indifferent to three blocks (30.2 -> 30.1 us) and the fused loop LOST (34.2 -> 38.7 us per iteration with two blocks).
enum { MODE_BACKWARD_SOLVE = 3 };
template <int D, int C, int MODE> struct SinvStashBlocks {
  static constexpr int kInterior = (C > 1) ? C - 1 : 0;
  // Round 3, STEP kernels with four states per lane (block elimination: static with velocity limits, per-state, general): TWO blocks parked --
  // per-state 48.1 -> 44.7 us (one block 45.5, three 48.3), static + velocity limits 30.6 -> 29.3 us, q_full (16,4) 85.4 -> 83.7 us; the two-states-per-lane
  // shapes do not react, and the fused loop loses with one block or two (profiles/r03_kernel_variants.txt).
#ifndef DGP_STASH_STEP_D4
#define DGP_STASH_STEP_D4 0
#endif
#ifndef DGP_STASH_STEP_D6
#define DGP_STASH_STEP_D6 2
#endif
  static constexpr int kWant = (D == 6 && MODE == MODE_BACKWARD_SOLVE) ? 3 : ((D == 6 && MODE == MODE_STEP && C == 4) ? DGP_STASH_STEP_D6 : ((D == 4 && MODE == MODE_STEP && C == 4) ? DGP_STASH_STEP_D4 : 0));
  static constexpr int value = kWant < kInterior ? kWant : kInterior;
};
template <int D, int NS> struct SinvStash {
  static constexpr int kBlock = ((D * (D + 1) / 2 * 8 + 15) / 16) * 16;                 // bytes per block, padded to 16
  static constexpr int kQuads = (NS * kBlock) / 16;
  static constexpr int kStride = NS == 0 ? 0 : ((kQuads % 2) ? kQuads : kQuads + 1) * 16;   // lane stride in bytes
  static constexpr int kBytes = 64 * kStride;
};
template <int D, int NS, typename Ctx>
DGP_HD void stash_put(Ctx& cx, int slot, const Sym<D>& S) {
  typedef double V2 __attribute__((vector_size(16)));
  char* l = cx.stash() + cx.lane() * SinvStash<D, NS>::kStride + slot * SinvStash<D, NS>::kBlock;
  constexpr int N = D * (D + 1) / 2;
#pragma unroll
  for (int i = 0; i < (N + 1) / 2; ++i) {
    V2 t; t[0] = S.v[2 * i]; t[1] = (2 * i + 1 < N) ? S.v[2 * i + 1 < N ? 2 * i + 1 : 0] : 0.0;
    *(V2*)(l + i * 16) = t;
  }
}
template <int D, int NS, typename Ctx>
DGP_HD void stash_get(Ctx& cx, int slot, Sym<D>& S) {
  typedef double V2 __attribute__((vector_size(16)));
  const char* l = cx.stash() + cx.lane() * SinvStash<D, NS>::kStride + slot * SinvStash<D, NS>::kBlock;
  constexpr int N = D * (D + 1) / 2;
#pragma unroll
  for (int i = 0; i < (N + 1) / 2; ++i) {
    const V2 t = *(const V2*)(l + i * 16);
    S.v[2 * i] = t[0];
    if (2 * i + 1 < N) S.v[2 * i + 1 < N ? 2 * i + 1 : 0] = t[1];
  }
}

// Lane-private LDS slots of NQ 16-byte cells (lane stride an odd number of cells: conflict-free 128-bit accesses) in the block cx.stash()
// points at: where the d = 6 Woodbury kernels park their recovery state across the Schur assembly and the PCR rounds (gn_woodbury.h, PARK).
template <int NQ> struct LdsPark {
  static constexpr int kStride = ((NQ % 2) ? NQ : NQ + 1) * 16;
  static constexpr int kBytes = 64 * kStride;
};
template <int NQ, typename Ctx>
DGP_HD void lds_park_put(Ctx& cx, const double (&v)[2 * NQ]) {
  typedef double V2 __attribute__((vector_size(16)));
  char* l = cx.stash() + cx.lane() * LdsPark<NQ>::kStride;
#pragma unroll
  for (int i = 0; i < NQ; ++i) { V2 t; t[0] = v[2 * i]; t[1] = v[2 * i + 1]; *(V2*)(l + i * 16) = t; }
}
template <int NQ, typename Ctx>
DGP_HD void lds_park_get(Ctx& cx, double (&v)[2 * NQ]) {
  typedef double V2 __attribute__((vector_size(16)));
  const char* l = cx.stash() + cx.lane() * LdsPark<NQ>::kStride;
#pragma unroll
  for (int i = 0; i < NQ; ++i) { const V2 t = *(const V2*)(l + i * 16); v[2 * i] = t[0]; v[2 * i + 1] = t[1]; }
}

// ---------------------------------------------------------------------------------------------------
// One Gauss-Newton linear solve for the C rows owned by this lane (rows j*C .. j*C+C-1 of trajectory b).
//
//  a. forward block elimination of the lane's C-1 INTERIOR rows (all but its last), carrying the right-hand side
//     and the "left spike" (the coupling of row 0 to the previous lane's last row):
//         S_0 = D_0,  S_k = D_k - U_{k-1}^T G_{k-1},   G_k = S_k^-1 U_k
//         z_k = r_k - G_{k-1}^T z_{k-1},   Zl_0 = L_0 = U_{-1}^T,   Zl_k = -G_{k-1}^T Zl_{k-1}
//     i.e.  x_k = S_k^-1 (z_k - Zl_k x_ps) - G_k x_{k+1}     (x_ps = last row of the previous lane).
//  b. STREAMED with a.: the first interior unknown as an affine function of the two neighbouring SEPARATOR unknowns
//         x_0 = P_0 - N_0 L_0 x_ps - W_0 x_s,     (x_s = last row of this lane)
//         P_0 = sum_k Pi_k S_k^-1 z_k,  N_0 = sum_k Pi_k S_k^-1 Pi_k^T,  W_0 = Pi_{C-2} G_{C-2},  Pi_0 = I, Pi_{k+1} = -Pi_k G_k
//     (Zl_k = Pi_k^T L_0, so the spike never has to be formed: N_0 is the SYMMETRIC (0,0) block of the inverse of the
//     interior system, and only running products stay live, not G_k / Zl_k of every row);
//  c. the lane's separator row, with x_{C-2} (own interior; its relation is the last line of a.) and x'_0 (first
//     interior row of the NEXT lane, whose P'_0, V'_0, W'_0 are fetched across lanes) substituted, is one row of a
//     block-tridiagonal system over the LPT lanes -> block PCR (log2 LPT rounds) gives x_s;
//  d. x_ps is fetched from the previous lane and the interior rows follow from a VECTOR forward / back substitution
//     with the kept S_k^-1, z_k:   w_0 = L_0 x_ps,  w_k = -U_{k-1}^T S_{k-1}^-1 w_{k-1},
//                                  x_k = S_k^-1 (z_k - w_k - U_k x_{k+1}),  x_{C-1} = x_s
//     (no d x d blocks survive the PCR rounds -- that is what keeps the kernel inside the vector register file).
// With C == 1 there are no interior rows and this is plain block PCR on the original system.
// QSTAT: static covariances, rows assembled by eval_state_static, U_k = m_k * p.u_fix (see Coupling).
// ---------------------------------------------------------------------------------------------------
// `before_pcr(acc)` is called once every factor of the lane has been evaluated (the error partials are complete) and before
// the PCR rounds: MODE_STEP reduces and stores err / err_ext there, off the tail of the kernel.
template <int DOF, int LPT, int C, typename IO, bool RHS_OVERRIDE, int QK, int NS, typename Ctx, typename Hook>
DGP_HD void gn_linear_solve(const GnParams& p, Ctx& cx, int64_t b, int j, bool traj_ok, const double (&x)[C][2 * DOF],
                            const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF], const LaneQ<2 * DOF, C, QK>& lq,
                            const double (&rhs)[C][2 * DOF], double (&dx)[C][2 * DOF], ErrAcc& acc, SpdCheck<Ctx>& ok, Hook&& before_pcr) {
  constexpr int D = 2 * DOF;
  constexpr bool QSTAT = (QK == QK_STATIC || QK == QK_SCALED);
  constexpr bool QSCAL = (QK == QK_SCALED);
  constexpr int CI = (C > 1) ? C - 1 : 1;       // interior rows (array extent; unused when C == 1)
  constexpr int KL = (C > 1) ? C - 2 : 0;       // last interior row
  const int n = p.n;
  const Nbr<LPT, 1, Ctx> nb(cx, j);
  // neighbouring states across the lane boundary
  double x_prev[D], x_next[D];
#pragma unroll
  for (int a = 0; a < D; ++a) { x_prev[a] = nb.lo(x[C - 1][a]); x_next[a] = nb.hi(x[0][a]);  }

  const int g0 = j * C;
  LaneFactors<C> lf;
  double rgp[C][D];              // prior + GP part of eta, computed while the SDF taps are in flight
  {
    LaneTaps<C, IO> taps;
    lane_obstacle_loads<DOF, C, IO>(p, b, g0, traj_ok, x, taps);
    DGP_STAMP_NOWAIT(p, cx, 8);
    // (the goal mean, which every row's arithmetic reads, is tied to the tap ADDRESSES, so that the loads are issued first)
    double mu_ga[D];
#pragma unroll
    for (int a = 0; a < D; ++a) mu_ga[a] = mu_g[a];
    lane_after_addresses<C, IO, D>(taps, mu_ga);
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const double (&xm)[D] = (k == 0) ? x_prev : x[k > 0 ? k - 1 : 0];
      const double (&xp)[D] = (k == C - 1) ? x_next : x[k < C - 1 ? k + 1 : 0];
      if constexpr (QK == QK_STATIC) static_rhs<DOF>(p, g0 + k, traj_ok && g0 + k < n, x[k], xm, xp, mu_s, mu_ga, rgp[k], acc);
      else if constexpr (QSCAL) static_rhs<DOF>(p, g0 + k, traj_ok && g0 + k < n, x[k], xm, xp, mu_s, mu_ga, rgp[k], acc, lq.s[k], (k == 0) ? lq.sm0 : lq.s[k > 0 ? k - 1 : 0]);
      else if constexpr (QK == QK_KRON) kron_rhs<DOF>(p, g0 + k, traj_ok && g0 + k < n, x[k], xm, xp, mu_s, mu_ga, lq.c[k],
                                                      (k == 0) ? lq.cm0 : lq.c[k > 0 ? k - 1 : 0], rgp[k], acc);
      else generic_rhs<DOF>(p, g0 + k, traj_ok && g0 + k < n, x[k], xm, xp, mu_s, mu_ga, lq.q[k], (k == 0) ? lq.qm0 : lq.q[k > 0 ? k - 1 : 0],
                            rgp[k], acc);
    }
    // first components of every row's eta (each is the end of that row's dependency chain)
    double anchor[2 * C];
#pragma unroll
    for (int k = 0; k < C; ++k) { anchor[2 * k] = rgp[k][0]; anchor[2 * k + 1] = rgp[k][D - 1]; }
    DGP_STAMP_NOWAIT(p, cx, 9);
    lane_taps_use_after<C, IO, 2 * C>(taps, anchor);
    DGP_STAMP_NOWAIT(p, cx, 10);
    lane_obstacle_finish<C, IO>(p, g0, traj_ok, taps, lf);
  }
  DGP_STAMP_NOWAIT(p, cx, 2);
#if defined(DGP_PHASE_STOP)     // profiles/tools/phase_probe.hip: cut the program short after a phase (timing aid, never in the product build)
  if (DGP_PHASE_STOP == 1 || DGP_PHASE_STOP == 2) {
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
      for (int a = 0; a < D; ++a)
        dx[k][a] = x[k][a] + x_prev[a] + x_next[a] + (DGP_PHASE_STOP == 2 ? lf.oc[k] + lf.ohx[k] + lf.ohy[k] + rgp[k][a] : 0.0);
    return;
  }
#endif
  // kept for the interior recovery (d.)
  Sym<D> Sinv[CI];
  double z[CI][D];
  Coupling<D, C, QK> cp;         // U_k of all C rows
  if constexpr (QK == QK_GENERAL) cp.dt = p.dt;
  double m_prev0 = 0.0;          // L_0 = -m_prev0 Qm0 Phi  (static path: m_prev0 * p.u_fix^T)
  // running quantities of the streamed elimination
  Mat<D> G, Pi, W0;
  Sym<D> N0;
  double P0[D], Pl[D];

  // assemble one row (D_k, r_k; U_k goes into cp)
  auto assemble = [&](int k, Sym<D>& Dk, double (&rk)[D]) {
    const int g = g0 + k;
    const bool valid = traj_ok && g < n;
    if constexpr (QSCAL) {
      static_diag<DOF>(p, g, valid, Dk, cp.m[k], lq.s[k], (k == 0) ? lq.sm0 : lq.s[k > 0 ? k - 1 : 0]);
    } else if constexpr (QK == QK_STATIC) {
      static_diag<DOF>(p, g, valid, Dk, cp.m[k]);
    } else if constexpr (QK == QK_KRON) {
      kron_diag<DOF>(p, g, valid, lq.c[k], (k == 0) ? lq.cm0 : lq.c[k > 0 ? k - 1 : 0], Dk, cp.m[k]);
      cp.c[k] = lq.c[k];
    } else {
      generic_diag<DOF>(p, g, valid, lq.q[k], (k == 0) ? lq.qm0 : lq.q[k > 0 ? k - 1 : 0], Dk, cp.m[k]);
      cp.q[k] = lq.q[k];
    }
#pragma unroll
    for (int a = 0; a < D; ++a) rk[a] = rgp[k][a];
    eval_state_local<DOF, true>(p, x[k], lf.ow[k], lf.oc[k], lf.ohx[k], lf.ohy[k], Dk, rk, acc);
    if (RHS_OVERRIDE) {
#pragma unroll
      for (int a = 0; a < D; ++a) rk[a] = valid ? rhs[k][a] : 0.0;
    }
  };

  // ---- a. + b. forward sweep over the interior rows
#pragma unroll
  for (int k = 0; k < C - 1; ++k) {
    Sym<D> Dk; double rk[D];
    assemble(k, Dk, rk);
    if (k == 0) {
      // left spike L_0 = (block (g, g-1)) = U_{g-1}^T = -(Phi^T Qm)^T = -Qm Phi   (zero for the first row of a trajectory)
      const bool has_prev = traj_ok && g0 > 0 && g0 < n;
      m_prev0 = has_prev ? 1.0 : 0.0;
      if constexpr (QSCAL) m_prev0 *= lq.sm0;
#pragma unroll
      for (int a = 0; a < D; ++a) z[0][a] = rk[a];
    } else {
      coup_sub_UtB_sym<D, C>(p, cp, k - 1, Dk, G);          // S_k = D_k - U_{k-1}^T G_{k-1}
#pragma unroll
      for (int a = 0; a < D; ++a) z[k][a] = rk[a];
      sub_At_v<D>(z[k], G, z[k > 0 ? k - 1 : 0]);          // z_k = r_k - G_{k-1}^T z_{k-1}
    }
    sym_inverse<D>(Dk, Sinv[k], ok);
    coup_SinvU<D, C>(p, cp, k, Sinv[k], G);                // G_k = S_k^-1 U_k
    if (k == 0) {
      N0 = Sinv[0];                                        // N_0 = Pi_0 S_0^-1 Pi_0^T, Pi_0 = I
      sym_times_vec<D>(Sinv[0], z[0], P0);                 // P_0 = S_0^-1 z_0
      if (k == KL) {
        W0 = G;                                            // C == 2: W_0 = G_0
      } else {
#pragma unroll
        for (int a = 0; a < D; ++a)
#pragma unroll
          for (int c = 0; c < D; ++c) Pi.v[a][c] = -G.v[a][c];       // Pi_1 = -G_0
      }
    } else {
      Mat<D> Mk;                                           // M_k = Pi_k S_k^-1
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double t = 0.0;
#pragma unroll
          for (int q = 0; q < D; ++q) t += Pi.v[a][q] * Sinv[k](q, c);
          Mk.v[a][c] = t;
        }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = a; c < D; ++c) {                      // N += M_k Pi_k^T  (symmetric)
          double t = N0(a, c);
#pragma unroll
          for (int q = 0; q < D; ++q) t += Mk.v[a][q] * Pi.v[c][q];
          N0(a, c) = t;
        }
      add_A_v<D>(P0, Mk, z[k]);                            // P_0 += Pi_k S_k^-1 z_k
      Mat<D> T;                                            // Pi_k G_k = (Pi_k S_k^-1) U_k = M_k U_k: U's structure applies
      if (k == KL) { coup_A_U<D, C>(p, cp, k, Mk, 1.0, T); W0 = T; }      // W_0 = Pi_{C-2} G_{C-2}
      else { coup_A_U<D, C>(p, cp, k, Mk, -1.0, T); Pi = T; }            // Pi_{k+1} = -Pi_k G_k
    }
    if (k == KL) sym_times_vec<D>(Sinv[k], z[k], Pl);      // P_{C-2} = S^-1 z_{C-2}   (x_{C-2} = P - (..) x_ps - G x_s)
    if constexpr (NS > 0) { if (k < NS) stash_put<D, NS>(cx, k, Sinv[k]); }      // parked in LDS until the recovery (d.)
  }
  // ---- c. separator row -> reduced system row
  Sym<D> Ds; Mat<D> Us; double rs[D];
  assemble(C - 1, Ds, rs);
  if (C == 1 || !QSTAT) coup_get<D, C>(p, cp, C - 1, Us);
  if (C > 1) {
    coup_sub_UtB_sym<D, C>(p, cp, KL, Ds, G);             // D_s -= U_{C-2}^T W_{C-2},  W_{C-2} = G_{C-2}
    coup_sub_Ut_v<D, C>(p, cp, KL, rs, Pl);               // r_s -= U_{C-2}^T P_{C-2}
    // first interior row of the next lane: x'_0 = P'_0 - N'_0 L'_0 x_s - W'_0 x'_s, and L'_0 = U_s^T
    Sym<D> Nn; Mat<D> Wn; double Pn[D];
#pragma unroll
    for (int i = 0; i < D * (D + 1) / 2; ++i) Nn.v[i] = nb.hi(N0.v[i]);
#pragma unroll
    for (int a = 0; a < D; ++a) {
      Pn[a] = nb.hi(P0[a]);
#pragma unroll
      for (int c = 0; c < D; ++c) Wn.v[a][c] = nb.hi(W0.v[a][c]);
    }
    // (U_s == 0 whenever there is no next lane / next row, so fetched-own values are harmless)
    Mat<D> T, Ur;
    if constexpr (QSTAT) {
      // U_s = m_s * p.u_fix: scalar operands, structural zeros skipped, the mask applied to the results
      const double ms = cp.m[C - 1];
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {
          double t = 0.0, u = 0.0;
#pragma unroll
          for (int q = 0; q < D; ++q) if (gp_nz<D>(a, q)) {
            t += p.u_fix[a * D + q] * Nn(q, c);            // U N'_0
            u += p.u_fix[a * D + q] * Wn.v[q][c];          // U W'_0
          }
          T.v[a][c] = ms * t;
          Ur.v[a][c] = -ms * u;                            // U_red = -U_s W'_0
        }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = a; c < D; ++c) {                      // D_s -= U_s N'_0 U_s^T   (m_s^2 = m_s)
          double t = Ds(a, c);
#pragma unroll
          for (int q = 0; q < D; ++q) if (gp_nz<D>(c, q)) t -= (QSCAL ? ms * T.v[a][q] : T.v[a][q]) * p.u_fix[c * D + q];      // (QK_SCALED: m_s^2 is not m_s)
          Ds(a, c) = t;
        }
      coup_sub_U_v<D, C>(p, cp, C - 1, rs, Pn);           // r_s -= U_s P'_0
    } else {
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = 0; c < D; ++c) {                      // T = U_s N'_0
          double t = 0.0;
#pragma unroll
          for (int q = 0; q < D; ++q) t += Us.v[a][q] * Nn(q, c);
          T.v[a][c] = t;
        }
#pragma unroll
      for (int a = 0; a < D; ++a)
#pragma unroll
        for (int c = a; c < D; ++c) {                      // D_s -= U_s N'_0 U_s^T
          double t = Ds(a, c);
#pragma unroll
          for (int q = 0; q < D; ++q) t -= T.v[a][q] * Us.v[c][q];
          Ds(a, c) = t;
        }
      sub_A_v<D>(rs, Us, Pn);                             // r_s -= U_s P'_0
      neg_A_B<D>(Us, Wn, Ur);                             // U_red = -U_s W'_0
    }
    Us = Ur;
  }
#if defined(DGP_PHASE_STOP)
  if (DGP_PHASE_STOP == 3) {
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
      for (int a = 0; a < D; ++a) dx[k][a] = Ds(a, a) + Ds(0, a) + rs[a] + Us.v[a][k % D] + Us.v[k % D][a] + (k < CI ? z[k < CI ? k : 0][a] + Sinv[k < CI ? k : 0](a, a) : 0.0);
    return;
  }
#endif
  DGP_STAMP_NOWAIT(p, cx, 3);
  before_pcr(acc);
  double xs[D];
  pcr_solve<D, LPT, (QK != QK_GENERAL)>(cx, j, Ds, Us, rs, xs, ok);
#pragma unroll
  for (int a = 0; a < D; ++a) dx[C - 1][a] = xs[a];
#if defined(DGP_PHASE_STOP)
  if (DGP_PHASE_STOP == 4) {
#pragma unroll
    for (int k = 0; k < C - 1; ++k)
#pragma unroll
      for (int a = 0; a < D; ++a) dx[k][a] = xs[a] + z[k][a] + Sinv[k](a, a);
    return;
  }
#endif
  // ---- d. interior rows
  if (C > 1) {
    double xps[D];
#pragma unroll
    for (int a = 0; a < D; ++a) xps[a] = nb.lo(xs[a]);                // L_0 == 0 where there is no previous separator
    double w[D], q[CI][D];
    // w_0 = L_0 x_ps
    if constexpr (QSTAT) {
#pragma unroll
      for (int a = 0; a < D; ++a) {
        double t = 0.0;
#pragma unroll
        for (int c = 0; c < D; ++c) if (gp_nz<D>(c, a)) t += p.u_fix[c * D + a] * xps[c];
        w[a] = m_prev0 * t;
      }
    } else {
      // L_0 x_ps = -m_prev0 Q_{g0-1} (Phi x_ps)
      double pv[D];
#pragma unroll
      for (int a = 0; a < D; ++a) pv[a] = (a < DOF) ? xps[a] + p.dt * xps[DOF + a] : xps[a];
      if constexpr (QK == QK_KRON) {
        double t[D];
        kron_apply<DOF>(p.qa, p.qb, p.qb, p.qc_, lq.cm0, pv, t);
#pragma unroll
        for (int a = 0; a < D; ++a) w[a] = -m_prev0 * t[a];
      } else {
#pragma unroll
        for (int a = 0; a < D; ++a) {
          double t = 0.0;
#pragma unroll
          for (int c = 0; c < D; ++c) t += lq.qm0(a, c) * pv[c];
          w[a] = -m_prev0 * t;
        }
      }
    }
#pragma unroll
    for (int a = 0; a < D; ++a) q[0][a] = z[0][a] - w[a];
#pragma unroll
    for (int k = 1; k < C - 1; ++k) {
      double t[D];
      if constexpr (NS > 0) {
        if (k - 1 < NS) { Sym<D> Sk; stash_get<D, NS>(cx, k - 1, Sk); sym_times_vec<D>(Sk, w, t); }
        else sym_times_vec<D>(Sinv[k - 1], w, t);
      } else {
        sym_times_vec<D>(Sinv[k - 1], w, t);
      }
#pragma unroll
      for (int a = 0; a < D; ++a) w[a] = 0.0;
      coup_sub_Ut_v<D, C>(p, cp, k - 1, w, t);                         // w_k = -U_{k-1}^T (S_{k-1}^-1 w_{k-1})
#pragma unroll
      for (int a = 0; a < D; ++a) q[k][a] = z[k][a] - w[a];
    }
    double xn[D];
#pragma unroll
    for (int a = 0; a < D; ++a) xn[a] = xs[a];
#pragma unroll
    for (int k = C - 2; k >= 0; --k) {
      coup_sub_U_v<D, C>(p, cp, k, q[k], xn);                          // q_k - U_k x_{k+1}
      if constexpr (NS > 0) {
        if (k < NS) { Sym<D> Sk; stash_get<D, NS>(cx, k, Sk); sym_times_vec<D>(Sk, q[k], xn); }
        else sym_times_vec<D>(Sinv[k], q[k], xn);
      } else {
        sym_times_vec<D>(Sinv[k], q[k], xn);
      }
#pragma unroll
      for (int a = 0; a < D; ++a) dx[k][a] = xn[a];
    }
  }
}

template <int DOF, int LPT, int C, typename IO, bool RHS_OVERRIDE, int QK, int NS, typename Ctx>
DGP_HD void gn_linear_solve(const GnParams& p, Ctx& cx, int64_t b, int j, bool traj_ok, const double (&x)[C][2 * DOF],
                            const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF], const LaneQ<2 * DOF, C, QK>& lq,
                            const double (&rhs)[C][2 * DOF], double (&dx)[C][2 * DOF], ErrAcc& acc, SpdCheck<Ctx>& ok) {
  gn_linear_solve<DOF, LPT, C, IO, RHS_OVERRIDE, QK, NS>(p, cx, b, j, traj_ok, x, mu_s, mu_g, lq, rhs, dx, acc, ok, [](const ErrAcc&) {});
}

// errors only (no assembly): sums over the lane's C rows
template <int DOF, int LPT, int C, typename IO, typename Ctx>
DGP_HD void gn_eval_only(const GnParams& p, Ctx& cx, int64_t b, int j, bool traj_ok, const double (&x)[C][2 * DOF],
                         const double (&mu_s)[2 * DOF], const double (&mu_g)[2 * DOF], ErrAcc& acc) {
  constexpr int D = 2 * DOF;
  const Nbr<LPT, 1, Ctx> nb(cx, j);
  double x_next[D];
#pragma unroll
  for (int a = 0; a < D; ++a) x_next[a] = nb.hi(x[0][a]);
  LaneFactors<C> lf;
  if (p.sdf) {               // wave-uniform.  No grid (host-checked: no output that depends on it was requested): no obstacle factors
    lane_prefetch<DOF, C, IO>(p, b, j * C, traj_ok, x, lf);
  } else {
#pragma unroll
    for (int k = 0; k < C; ++k) { lf.ow[k] = 0.0; lf.oc[k] = 0.0; lf.ohx[k] = 0.0; lf.ohy[k] = 0.0; }
  }
  const bool stat = (p.qc_mode == QC_STATIC);
  Sym<D> Qown;
  fixed_Qinv<DOF>(p, Qown);
  Sym<D> Dk; Mat<D> Uk; double rk[D];
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int g = j * C + k;
    const bool valid = traj_ok && g < p.n;
    if (!stat && valid && g < p.n - 1) load_Qinv<DOF, IO>(p, b, g, Qown);
    eval_state<DOF, IO, false>(p, b, g, valid, x[k], x[k], (k < C - 1) ? x[k < C - 1 ? k + 1 : 0] : x_next, mu_s, mu_g, Qown, Qown,
                               lf.ow[k], lf.oc[k], lf.ohx[k], lf.ohy[k], Dk, Uk, rk, acc);
  }
}

// The three UNWEIGHTED errors of DiffGPMP2Planner.unweighted_errors_batch at a trajectory (plan_layer.py:374-388: start_goal_error, gp_error's sum over the
// factors, obs_error's sum over the states), per lane, summed over its C rows in the order eval_state accumulates them -- no covariance enters (only the current
// epsilons): the step kernels' epilogue for the errors at th + dtheta (DGP_STEP_ERRS), which must not re-read per-state covariance blocks.
template <int DOF, int LPT, int C, typename IO, typename Ctx>
DGP_HD void unweighted_errors_at(const GnParams& p, Ctx& cx, int64_t b, int j, bool traj_ok, const double (&x)[C][2 * DOF], const double (&mu_s)[2 * DOF],
                                 const double (&mu_g)[2 * DOF], double& usg, double& ugp, double& uobs) {
  constexpr int D = 2 * DOF;
  const Nbr<LPT, 1, Ctx> nb(cx, j);
  double x_next[D];
#pragma unroll
  for (int a = 0; a < D; ++a) x_next[a] = nb.hi(x[0][a]);
  LaneFactors<C> lf;
  if (p.sdf) lane_prefetch<DOF, C, IO>(p, b, j * C, traj_ok, x, lf);
  usg = 0.0; ugp = 0.0; uobs = 0.0;
  const int n = p.n;
  const double dt = p.dt;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    const int g = j * C + k;
    if (!(traj_ok && g < n)) continue;
    const double* xk = x[k];
    const double* xp = (k < C - 1) ? x[k < C - 1 ? k + 1 : 0] : x_next;
    if (g == 0 || g == n - 1) {
      double s2 = 0.0;
#pragma unroll
      for (int a = 0; a < D; ++a) { const double ea = ((g == 0) ? mu_s[a] : mu_g[a]) - xk[a]; s2 += ea * ea; }
      usg += 0.5 * s2;
    }
    if (g < n - 1) {
      double e[D], s2 = 0.0;
#pragma unroll
      for (int a = 0; a < DOF; ++a) { e[a] = xp[a] - (xk[a] + dt * xk[DOF + a]); e[DOF + a] = xp[DOF + a] - xk[DOF + a]; }
#pragma unroll
      for (int a = 0; a < D; ++a) s2 += e[a] * e[a];
      ugp += 0.5 * s2;
    }
    if (p.sdf) uobs += 0.5 * lf.oc[k] * lf.oc[k];
  }
}

// ---------------------------------------------------------------------------------------------------
// Row stores of a whole wavefront through LDS.  A lane owns C consecutive rows = one contiguous chunk of C*D elements, so
// a direct store instruction writes 16 bytes at a C*D*sizeof(IO)-byte lane stride: 64 partially written cache lines per
// instruction, which the memory pipeline acknowledges ~0.4 us later than the same bytes written as full lines
// (profiles/tools/launch_probe.hip).  When the wavefront's rows are one contiguous block of memory (n == LPT*C, identity
// lane mapping, every trajectory of the wavefront exists, 16-byte aligned rows) the chunks are written to LDS (lane stride
// padded by 16 bytes: conflict-free 128-bit writes) and read back so that lane l stores the l-th 16 bytes of each 1 KB.
// ---------------------------------------------------------------------------------------------------
template <typename IO, int C, int D>
struct WaveStore {
  static constexpr int kChunk = C * D * (int)sizeof(IO);          // bytes per lane
  static constexpr bool kUsable = (kChunk % 16) == 0;
  static constexpr int kCells = kUsable ? kChunk / 16 : 1;        // 16-byte cells per lane
  static constexpr int kStride = kChunk + 16;
  static constexpr int kLdsBytes = kUsable ? 64 * kStride : 16;
};

template <typename IO, int C, int D, typename Ctx>
DGP_HD void store_rows_through_lds(Ctx& cx, void* out, int64_t wave_first_elem, const double (&v)[C][D]) {
  typedef WaveStore<IO, C, D> WS;
  typedef IO V16 __attribute__((vector_size(16)));
  constexpr int EPV = 16 / (int)sizeof(IO);                       // elements per 16-byte cell
  const int lane = cx.lane();
  char* l = cx.lds();
#pragma unroll
  for (int i = 0; i < WS::kCells; ++i) {
    V16 t;
#pragma unroll
    for (int e = 0; e < EPV; ++e) t[e] = (IO)v[(i * EPV + e) / D][(i * EPV + e) % D];
    *(V16*)(l + lane * WS::kStride + i * 16) = t;
  }
  cx.lds_sync();
  char* o = (char*)out + wave_first_elem * (int64_t)sizeof(IO);
#pragma unroll
  for (int i = 0; i < WS::kCells; ++i) {
    const int c = i * 64 + lane;                                  // cell index inside the wavefront's block
    const V16 t = *(const V16*)(l + (c / WS::kCells) * WS::kStride + (c % WS::kCells) * 16);
    *(V16*)(o + (int64_t)c * 16) = t;
  }
}

// The same block stored WRITE-THROUGH (`buffer_store_dwordx4 ... sc0 sc1`, system scope): the lines do not stay dirty in the XCD's L2,
// so the release at the end of the kernel has nothing to write back and the next launch of an in-order stream starts ~0.2 us earlier
// (DESIGN.md section 5 "(j)").  A raw buffer store because that is the 16-byte store whose cache-scope bits HIP source can set without
// inline asm (which would hide the store from the compiler's hazard and wait-count bookkeeping); the wavefront's block is the buffer.
template <typename IO, int C, int D, typename Ctx>
DGP_HD void store_rows_through_lds_wt(Ctx& cx, void* out, int64_t wave_first_elem, const double (&v)[C][D]) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef WaveStore<IO, C, D> WS;
  typedef IO V16 __attribute__((vector_size(16)));
  typedef unsigned int U4 __attribute__((ext_vector_type(4)));
  constexpr int EPV = 16 / (int)sizeof(IO);
  const int lane = cx.lane();
  char* l = cx.lds();
#pragma unroll
  for (int i = 0; i < WS::kCells; ++i) {
    V16 t;
#pragma unroll
    for (int e = 0; e < EPV; ++e) t[e] = (IO)v[(i * EPV + e) / D][(i * EPV + e) % D];
    *(V16*)(l + lane * WS::kStride + i * 16) = t;
  }
  cx.lds_sync();
  char* o = (char*)out + wave_first_elem * (int64_t)sizeof(IO);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(o, 0, WS::kCells * 64 * 16, 0x00020000);
#pragma unroll
  for (int i = 0; i < WS::kCells; ++i) {
    const int c = i * 64 + lane;                                  // cell index inside the wavefront's block
    const U4 t = *(const U4*)(l + (c / WS::kCells) * WS::kStride + (c % WS::kCells) * 16);
    __builtin_amdgcn_raw_buffer_store_b128(t, rs, c * 16, 0, /* sc0 | sc1 */ 1 | 16);
  }
#else
  store_rows_through_lds<IO, C, D>(cx, out, wave_first_elem, v);
#endif
}

// The mirror image for loads: the wavefront's block is read as full lines (lane l reads the l-th 16 bytes of each 1 KB),
// staged in LDS and picked up by the owning lanes (a load instruction that reads 16 bytes at a 64-byte lane stride makes 64
// partial-line requests; four of them per lane cost ~0.3 us more than the same bytes as 16 full lines per instruction).
template <typename IO, int C, int D, typename Ctx>
DGP_HD void load_rows_through_lds(Ctx& cx, const void* in, int64_t wave_first_elem, double (&v)[C][D]) {
  typedef WaveStore<IO, C, D> WS;
  typedef IO V16 __attribute__((vector_size(16)));
  constexpr int EPV = 16 / (int)sizeof(IO);
  const int lane = cx.lane();
  char* l = cx.lds();
  const char* src = (const char*)in + wave_first_elem * (int64_t)sizeof(IO);
  V16 t[WS::kCells];
#pragma unroll
  for (int i = 0; i < WS::kCells; ++i) t[i] = *(const V16*)(src + (int64_t)(i * 64 + lane) * 16);
#pragma unroll
  for (int i = 0; i < WS::kCells; ++i) {
    const int c = i * 64 + lane;
    *(V16*)(l + (c / WS::kCells) * WS::kStride + (c % WS::kCells) * 16) = t[i];
  }
  cx.lds_sync();
#pragma unroll
  for (int i = 0; i < WS::kCells; ++i) {
    const V16 u = *(const V16*)(l + lane * WS::kStride + i * 16);
#pragma unroll
    for (int e = 0; e < EPV; ++e) v[(i * EPV + e) / D][(i * EPV + e) % D] = (double)u[e];
  }
  cx.lds_sync();      // the staging block is reused by the output stores
}

// MODE_SOLVE keeps the trajectory (the lane's C fp64 state rows) in the wavefront's LDS block between the uses at the two ends
// of an iteration -- factor evaluation at the start, th += dtheta at the end -- instead of in 2 C d vector registers across the
// whole elimination + PCR, where every register is contended (lane-private slots, padded lane stride: conflict-free 128-bit accesses).
template <int C, int D, typename Ctx>
DGP_HD void lds_put_rows(Ctx& cx, const double (&v)[C][D]) {
  typedef double V2 __attribute__((vector_size(16)));
  char* l = cx.lds() + cx.lane() * WaveStore<double, C, D>::kStride;
#pragma unroll
  for (int i = 0; i < C * D / 2; ++i) {
    V2 t; t[0] = v[(2 * i) / D][(2 * i) % D]; t[1] = v[(2 * i + 1) / D][(2 * i + 1) % D];
    *(V2*)(l + i * 16) = t;
  }
}
template <int C, int D, typename Ctx>
DGP_HD void lds_get_rows(Ctx& cx, double (&v)[C][D]) {
  typedef double V2 __attribute__((vector_size(16)));
  const char* l = cx.lds() + cx.lane() * WaveStore<double, C, D>::kStride;
#pragma unroll
  for (int i = 0; i < C * D / 2; ++i) {
    const V2 t = *(const V2*)(l + i * 16);
    v[(2 * i) / D][(2 * i) % D] = t[0]; v[(2 * i + 1) / D][(2 * i + 1) % D] = t[1];
  }
}

// Lane-private LDS slots of the chain backward kernels (gn_backward.h, cx.chain_lds()): the running cotangent's C rows and the two accumulated
// mean gradients (start, goal), d doubles each; lane stride an odd number of 16-byte cells (conflict-free 128-bit accesses).  Vector index v:
// 0 .. C-1 the rows, C the start gradient, C + 1 the goal gradient.
template <int C, int D> struct ChainSlots {
  static constexpr int kCells = (C + 2) * D / 2;
  static constexpr int kStride = ((kCells % 2) ? kCells : kCells + 1) * 16;
  static constexpr int kBytes = 64 * kStride;
};
template <int C, int D>
DGP_HD void chain_put(char* base, int lane, int v, const double (&x)[D]) {
  typedef double V2 __attribute__((vector_size(16)));
  char* l = base + lane * ChainSlots<C, D>::kStride + v * D * 8;
#pragma unroll
  for (int i = 0; i < D / 2; ++i) { V2 t; t[0] = x[2 * i]; t[1] = x[2 * i + 1]; *(V2*)(l + i * 16) = t; }
}
template <int C, int D>
DGP_HD void chain_get(const char* base, int lane, int v, double (&x)[D]) {
  typedef double V2 __attribute__((vector_size(16)));
  const char* l = base + lane * ChainSlots<C, D>::kStride + v * D * 8;
#pragma unroll
  for (int i = 0; i < D / 2; ++i) { const V2 t = *(const V2*)(l + i * 16); x[2 * i] = t[0]; x[2 * i + 1] = t[1]; }
}

// Lane-private LDS slots of the d = 4 single-step backward kernels (round 5, dgp_gn_step_errors_backward in one launch): what the errors' prologue hands to
// the main program of the SAME lane -- the C trajectory-gradient rows (vectors 0 .. C-1), its shares of the start / goal gradients (C, C + 1) and of the C
// epsilon gradients (C + 2, first C doubles).  Through memory these cost the main program one exposed round trip per row: every re-read sat behind the
// previous row's stores.  (d = 6 keeps the memory hand-over: its backward kernels hold 34 KB of LDS already, and four wavefronts per CU must fit in 160 KB.)
template <int C, int D> struct FoldSlots {
  static constexpr int kCells = (C + 3) * D / 2;
  static constexpr int kStride = ((kCells % 2) ? kCells : kCells + 1) * 16;
  static constexpr int kBytes = 64 * kStride;
};
template <int C, int D>
DGP_HD void fold_put(char* base, int lane, int v, const double (&x)[D]) {
  typedef double V2 __attribute__((vector_size(16)));
  char* l = base + lane * FoldSlots<C, D>::kStride + v * D * 8;
#pragma unroll
  for (int i = 0; i < D / 2; ++i) { V2 t; t[0] = x[2 * i]; t[1] = x[2 * i + 1]; *(V2*)(l + i * 16) = t; }
}
template <int C, int D>
DGP_HD void fold_get(const char* base, int lane, int v, double (&x)[D]) {
  typedef double V2 __attribute__((vector_size(16)));
  const char* l = base + lane * FoldSlots<C, D>::kStride + v * D * 8;
#pragma unroll
  for (int i = 0; i < D / 2; ++i) { const V2 t = *(const V2*)(l + i * 16); x[2 * i] = t[0]; x[2 * i + 1] = t[1]; }
}

}  // namespace dgp
#include "gn_woodbury.h"
namespace dgp {

// Makes the per-state covariance blocks of a lane opaque to the optimiser at the top of a GN iteration of the fused loop (emits no
// instruction).  They ARE loop-invariant, and so is everything assembled from them alone -- the GP part of every diagonal block, the
// coupling blocks, their products with the constant 2 x 2 factors: left alone, loop-invariant code motion hoists all of it out of the
// loop and keeps it alive across the whole body (d = 6: ~100 doubles on top of a body that already needs every register), i.e. in
// scratch (d = 6 per-state fused loop: 2 052 -> 1 428 B per lane with the blocks opaque), reloaded with exposed latency.  Recomputing those few
// hundred FMAs per iteration is cheaper than the reloads.
// Measured (profiles/r03_kernel_variants.txt, B = 4096, 10 iterations): d = 6 per-state 761.9 -> 717.5 us, d = 6 q_full 1 410 -> 1 365 us, d = 4 per-state
// 103.7 -> 103.3 us (no spill to begin with) -- used for d = 6 only.
#ifndef DGP_SOLVE_OPAQUE_Q
#define DGP_SOLVE_OPAQUE_Q 6      // the state dimension it applies to (0: never, 1: every dimension)
#endif
DGP_HD void opaque(double& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#else
  (void)v;
#endif
}
template <int D, int C> DGP_HD void lane_q_opaque(LaneQ<D, C, QK_GENERAL>& L) {
#pragma unroll
  for (int i = 0; i < D * (D + 1) / 2; ++i) {
    opaque(L.qm0.v[i]);
#pragma unroll
    for (int k = 0; k < C; ++k) opaque(L.q[k].v[i]);
  }
}
template <int D, int C> DGP_HD void lane_q_opaque(LaneQ<D, C, QK_KRON>& L) {
#pragma unroll
  for (int i = 0; i < (D / 2) * (D / 2 + 1) / 2; ++i) {
    opaque(L.cm0.v[i]);
#pragma unroll
    for (int k = 0; k < C; ++k) opaque(L.c[k].v[i]);
  }
}
template <int D, int C> DGP_HD void lane_q_opaque(LaneQ<D, C, QK_STATIC>&) {}
template <int D, int C> DGP_HD void lane_q_opaque(LaneQ<D, C, QK_SCALED>&) {}
template <int D, int C> DGP_HD void lane_q_opaque(LaneQ<D, C, QK_WB>&) {}
template <int D, int C> DGP_HD void lane_q_opaque(LaneQ<D, C, QK_WBR>&) {}

// ---------------------------------------------------------------------------------------------------
// the lane program: LPT lanes per trajectory, C consecutive states per lane (n <= LPT * C)
// ---------------------------------------------------------------------------------------------------
template <int DOF, int LPT, int C, typename IO, int MODE, int QK, typename Ctx>
DGP_HD void gn_lane_program(const GnParams& p, Ctx& cx) {
  constexpr int D = 2 * DOF;
  constexpr int TPW = 64 / LPT;                 // trajectories per wavefront
  const int lane = cx.lane();
  const int j = lane_to_row<LPT>(lane & (LPT - 1));      // block row of the LPT-row system owned by this lane
  const int64_t b = (int64_t)cx.wave() * TPW + (lane / LPT);
  const int n = p.n;
  const bool traj_ok = b < p.B;
  DGP_STAMP_NOWAIT(p, cx, 0);
  constexpr bool kPF = is_wb(QK) && MODE == MODE_STEP;      // phase fences (see phase_fence)
  phase_fence<kPF>();
#if defined(__HIP_DEVICE_COMPILE__)
  // the scalars of the pixel-coordinate / tap-address arithmetic are fetched now, under the th load, instead of at their
  // first use right after it (the compiler places scalar loads in the block that first needs them)
  asm volatile("" :: "s"(p.res), "s"(p.inv_res), "s"(p.orig_px), "s"(p.orig_py), "s"(p.sdf), "s"(p.sdf_cols), "s"(p.sdf_rows),
               "s"(p.sdf_bstride), "s"(p.eps), "s"(p.obs_w));
#endif

  // constants of the Woodbury elimination (host-checked: C == 4, n == LPT * C): loads issued ahead of the th rows, committed to LDS behind them
  struct NoStage {};
  typename std::conditional<is_wb(QK), WbStaged, NoStage>::type wbv;
  const bool vec = p.vec_io != 0;
  double x[C][D], mu_s[D], mu_g[D];
  LaneQ<D, C, QK> lq;                           // generic covariances: Q^-1 of the lane's C + 1 GP factors, loaded once (loop-invariant in MODE_SOLVE)
  if constexpr (MODE != MODE_EVAL) load_lane_Q<DOF, C, IO>(p, b, j * C, traj_ok, lq);
  // wave-uniform: the wavefront's th rows are one contiguous, fully populated block -> full-line loads via LDS (-0.3 us)
  bool block_load = false;
  if constexpr (WaveStore<IO, C, D>::kUsable && LPT != 32 && MODE == MODE_STEP)
    block_load = vec && n == LPT * C && ((int64_t)cx.wave() + 1) * TPW <= (int64_t)p.B;
  if (block_load) {
    if constexpr (WaveStore<IO, C, D>::kUsable && LPT != 32 && MODE == MODE_STEP)
      load_rows_through_lds<IO, C, D>(cx, p.th, (int64_t)cx.wave() * TPW * n * D, x);
  } else {
    load_lane_rows<DOF, C, IO>(p, p.th, b, j * C, traj_ok, vec, x);
  }
  ld_row<IO, D>(p.start, traj_ok ? b : 0, vec && p.vec_mu, mu_s);
  ld_row<IO, D>(p.goal, traj_ok ? b : 0, vec && p.vec_mu, mu_g);
  if constexpr (is_wb(QK)) wb_stage_issue<(QK == QK_WBR)>(p, cx, wbv);
  DGP_STAMP(p, cx, 1);
  phase_fence<kPF>();

  if (MODE == MODE_EVAL) {
    ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
    if (p.dtheta) {               // wave-uniform.  dgp_gn_step_errors: the errors at th + dtheta (learning/train_planner.py:313,327), the sum formed in the
      double dq[C][D];            // I/O type exactly as torch forms th_curr_b + dthetab before it hands the result to unweighted_errors_batch
      load_lane_rows<DOF, C, IO>(p, p.dtheta, b, j * C, traj_ok, vec, dq);
#pragma unroll
      for (int k = 0; k < C; ++k)
#pragma unroll
        for (int a = 0; a < D; ++a) x[k][a] = (double)(IO)((IO)x[k][a] + (IO)dq[k][a]);
    }
    gn_eval_only<DOF, LPT, C, IO>(p, cx, b, j, traj_ok, x, mu_s, mu_g, acc);
    const double e = group_sum_to_first<LPT>(cx, acc.e), ee = group_sum_to_first<LPT>(cx, acc.eext);
    const double usg = group_sum_to_first<LPT>(cx, acc.usg), ugp = group_sum_to_first<LPT>(cx, acc.ugp);
    const double uobs = group_sum_to_first<LPT>(cx, acc.uobs);
    if (traj_ok && j == 0) {
      if (p.err) st<IO>(p.err, b, div_M(p, e));
      if (p.err_ext) st<IO>(p.err_ext, b, div_M(p, ee));
      if (p.unw_sg) st<IO>(p.unw_sg, b, usg);
      if (p.unw_gp) st<IO>(p.unw_gp, b, ugp / (double)(n - 1));     // torch.mean over the n-1 factors
      if (p.unw_obs) st<IO>(p.unw_obs, b, uobs / (double)n);
    }
    return;
  }

  const int iters_max = (MODE == MODE_SOLVE) ? p.max_iters : 1;
  bool active = traj_ok;
  int my_iters = 0;
  SpdCheck<Ctx> ok = {&cx, 0};            // accumulates over the GN iterations of MODE_SOLVE
  // MODE_SOLVE, d = 4: the trajectory is parked in LDS between the two ends of an iteration (lds_put_rows).  NOT for d = 6: with the
  // state parked, hipcc produced d = 6 fused-loop kernels -- <3,64,2,double,SOLVE,per-state>, <3,64,4,double,SOLVE,general>; the
  // ones that spill hundreds of SGPR lane masks into VGPR lanes -- that return garbage, non-deterministically, although the LDS
  // contents are intact (checked in-kernel against a register copy).  Found by tests/stress_random_configs.py, which now drives
  // the fused loop; the d = 6 fused kernels therefore keep their state in registers, as in round 1.
  constexpr bool kPark = (MODE == MODE_SOLVE) && (D == 4);
  if constexpr (kPark) lds_put_rows<C, D>(cx, x);
#pragma unroll 1
  for (int it = 0; it < iters_max; ++it) {
    ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
    double dx[C][D];
    if constexpr (MODE == MODE_SOLVE && (DGP_SOLVE_OPAQUE_Q == 1 || DGP_SOLVE_OPAQUE_Q == D)) lane_q_opaque<D, C>(lq);
    if constexpr (kPark) lds_get_rows<C, D>(cx, x);       // the state comes back from LDS
    if constexpr (MODE == MODE_SOLVE) {
      // dgp_gn_solve's optional trajectory history (what dgp_gn_solve_backward re-assembles the passes from): th_it in fp64 whatever the I/O type,
      // row (it, b, g).  The pointer travels in GnParams::dtheta, which the fused loop does not otherwise use.  Wave-uniform test.
      if (p.dtheta) {
        double* hist = (double*)p.dtheta;
        // d = 4, a fully populated wavefront block of a length that fills the shape, every trajectory of it still iterating: the parked state in LDS IS
        // the staging layout of store_rows_through_lds<double> (lane-private chunks, padded stride), so the block goes out as full cache lines
        // straight from LDS -- lane l stores the l-th 16 bytes of each 1 KB -- instead of 32 bytes per lane at a 128-byte stride
        bool block_hist = false;
        if constexpr (kPark && LPT != 32) block_hist = n == LPT * C && ((int64_t)cx.wave() + 1) * TPW <= (int64_t)p.B && !cx.any(!active);
        if (block_hist) {
          if constexpr (kPark && LPT != 32) {
            typedef WaveStore<double, C, D> WS;
            typedef double V2 __attribute__((vector_size(16)));
            cx.lds_sync();
            const char* l = cx.lds();
            char* o = (char*)(hist + (((int64_t)it * p.B + (int64_t)cx.wave() * TPW) * n) * D);
#pragma unroll
            for (int i = 0; i < WS::kCells; ++i) {
              const int c = i * 64 + lane;
              *(V2*)(o + (int64_t)c * 16) = *(const V2*)(l + (c / WS::kCells) * WS::kStride + (c % WS::kCells) * 16);
            }
            cx.lds_sync();
          }
        } else {
#pragma unroll
          for (int k = 0; k < C; ++k) {
            const int g = j * C + k;
            if (active && g < n) {
              double* row = hist + (((int64_t)it * p.B + b) * n + g) * D;
#pragma unroll
              for (int a = 0; a < D; ++a) row[a] = x[k][a];
            }
          }
        }
      }
    }
    double e = 0.0, ee = 0.0;
    auto before_pcr = [&](const ErrAcc& a) {
      e = group_sum_to_first<LPT>(cx, a.e); ee = group_sum_to_first<LPT>(cx, a.eext);
      if (MODE == MODE_STEP && traj_ok && j == 0) {
        if (p.err) st<IO>(p.err, b, div_M(p, e));
        if (p.err_ext) st<IO>(p.err_ext, b, div_M(p, ee));
      }
    };
    if constexpr (is_wb(QK)) {
      static_assert(C == 4, "the Woodbury kernels are built for four states per lane");
      gn_linear_solve_wb<DOF, LPT, IO, false, (D == 4 || MODE == MODE_SOLVE), (QK == QK_WBR), (MODE == MODE_STEP), WbParks<DOF, MODE>::value>(p, cx, b, j, traj_ok, x, mu_s, mu_g, x, dx, acc, ok, it == 0 ? &wbv : nullptr, before_pcr);
    } else {
#if defined(DGP_BISECT_LAMBDA)
      gn_linear_solve<DOF, LPT, C, IO, false, QK, SinvStashBlocks<2 * DOF, C, MODE>::value>(p, cx, b, j, traj_ok, x, mu_s, mu_g, lq, x, dx, acc, ok, [&](const ErrAcc& a) {
        e = group_sum_to_first<LPT>(cx, a.e); ee = group_sum_to_first<LPT>(cx, a.eext);
        if (MODE == MODE_STEP && traj_ok && j == 0) {
          if (p.err) st<IO>(p.err, b, div_M(p, e));
          if (p.err_ext) st<IO>(p.err_ext, b, div_M(p, ee));
        }
      });
#else
      gn_linear_solve<DOF, LPT, C, IO, false, QK, SinvStashBlocks<2 * DOF, C, MODE>::value>(p, cx, b, j, traj_ok, x, mu_s, mu_g, lq, x, dx, acc, ok, before_pcr);
#endif
    }
    DGP_STAMP_NOWAIT(p, cx, 4);
    phase_fence<kPF>();
    if (MODE == MODE_STEP) {
      // wave-uniform: the wavefront's dtheta rows are one contiguous, fully populated block -> full-line stores via LDS
      bool block_store = false;
      if constexpr (WaveStore<IO, C, D>::kUsable && LPT != 32)
        block_store = vec && n == LPT * C && ((int64_t)cx.wave() + 1) * TPW <= (int64_t)p.B;
      if (block_store) {
        if constexpr (WaveStore<IO, C, D>::kUsable && LPT != 32)
        {
          // write-through for d = 4 only: on d = 6 it is worth 0.2 of 26 us, and with it hipcc miscompiled <3,64,2,float,STEP,per-state>
          // (wrong by O(1) for every n, found by tests/test_hip_every_kernel.py) -- the d = 6 kernels keep the code of
          // the build they were verified on
          if constexpr (D == 4) store_rows_through_lds_wt<IO, C, D>(cx, p.dtheta, (int64_t)cx.wave() * TPW * n * D, dx);
          else store_rows_through_lds<IO, C, D>(cx, p.dtheta, (int64_t)cx.wave() * TPW * n * D, dx);
        }
      } else {
#pragma unroll
        for (int k = 0; k < C; ++k) {
          const int g = j * C + k;
          if (traj_ok && g < n) st_row<IO, D>(p.dtheta, b * n + g, vec, dx[k]);
        }
      }
#if DGP_STEP_ERRS
      if (p.unw_sg || p.unw_gp || p.unw_obs) {      // wave-uniform: the training iteration's errors at th + dtheta, same launch
        // th is read again (L2 hits: carried through the solve it would cost 2 C d registers), the sum formed in the I/O type from the STORED dtheta values,
        // exactly as the separate error launch forms it from memory (torch: th_curr_b + dthetab)
        double xe[C][D];
        if (block_load) {
          if constexpr (WaveStore<IO, C, D>::kUsable && LPT != 32 && MODE == MODE_STEP)
            load_rows_through_lds<IO, C, D>(cx, p.th, (int64_t)cx.wave() * TPW * n * D, xe);
        } else {
          load_lane_rows<DOF, C, IO>(p, p.th, b, j * C, traj_ok, vec, xe);
        }
#pragma unroll
        for (int k = 0; k < C; ++k)
#pragma unroll
          for (int a = 0; a < D; ++a) xe[k][a] = (traj_ok && j * C + k < n) ? (double)(IO)((IO)xe[k][a] + (IO)dx[k][a]) : 0.0;
        double usg, ugp, uobs, ms[D], mg[D];      // (the means re-read too: two rows less to carry through the solve)
        ld_row<IO, D>(p.start, traj_ok ? b : 0, vec && p.vec_mu, ms);
        ld_row<IO, D>(p.goal, traj_ok ? b : 0, vec && p.vec_mu, mg);
        unweighted_errors_at<DOF, LPT, C, IO>(p, cx, b, j, traj_ok, xe, ms, mg, usg, ugp, uobs);
        usg = group_sum_to_first<LPT>(cx, usg); ugp = group_sum_to_first<LPT>(cx, ugp); uobs = group_sum_to_first<LPT>(cx, uobs);
        if (traj_ok && j == 0) {
          if (p.unw_sg) st<IO>(p.unw_sg, b, usg);
          if (p.unw_gp) st<IO>(p.unw_gp, b, ugp / (double)(n - 1));     // torch.mean over the n-1 factors
          if (p.unw_obs) st<IO>(p.unw_obs, b, uobs / (double)n);
        }
      }
#endif
    } else {
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < C; ++k)
#pragma unroll
        for (int a = 0; a < D; ++a) s2 += (traj_ok && j * C + k < n) ? dx[k][a] * dx[k][a] : 0.0;
      s2 = group_sum<LPT>(cx, s2);
      if (active) {
        // (the history stores stay HERE, behind the solve: issued from the before_pcr hook, hipcc built a
        //  <3,16,1,double,SOLVE,general> kernel that stores through a wild address -- tests/stress_random_configs.py)
        if (j == 0) {
          if (p.err_hist) st<IO>(p.err_hist, b * (int64_t)p.max_iters + it, div_M(p, e));
          if (p.errext_hist) st<IO>(p.errext_hist, b * (int64_t)p.max_iters + it, div_M(p, ee));
        }
        if constexpr (kPark) {
          double xc[C][D];
          lds_get_rows<C, D>(cx, xc);
#pragma unroll
          for (int k = 0; k < C; ++k)
#pragma unroll
            for (int a = 0; a < D; ++a) xc[k][a] += (j * C + k < n) ? dx[k][a] : 0.0;   // th_new = th_curr + dtheta (:144)
          lds_put_rows<C, D>(cx, xc);
        } else {
#pragma unroll
          for (int k = 0; k < C; ++k)
#pragma unroll
            for (int a = 0; a < D; ++a) x[k][a] += (j * C + k < n) ? dx[k][a] : 0.0;
        }
        my_iters = it + 1;
        if (sqrt(s2) < p.tol_delta) active = false;                                    // planner_utils.py:4
      }
      if (!cx.any(active)) break;
    }
  }
  DGP_STAMP_NOWAIT(p, cx, 5);
  phase_fence<kPF>();
  if (p.info) {                                        // (wave-uniform)
    const int base = lane & ~(LPT - 1);                // first lane of this trajectory's group
    const uint64_t grp = (LPT == 64) ? ok.bad : ((ok.bad >> base) & ((uint64_t(1) << (LPT & 63)) - 1));
    if (traj_ok && j == 0) p.info[b] = grp != 0 ? 1 : 0;
  }
  DGP_STAMP(p, cx, 6);
  phase_fence<kPF>();
  if (MODE == MODE_SOLVE) {
    if constexpr (kPark) lds_get_rows<C, D>(cx, x);
#pragma unroll
    for (int k = 0; k < C; ++k) {
      const int g = j * C + k;
      if (traj_ok && g < n) st_row<IO, D>(p.th_out, b * n + g, vec, x[k]);
    }
    if (traj_ok && j == 0 && p.iters) p.iters[b] = my_iters;
    if (p.err_final) {
      ErrAcc acc = {0.0, 0.0, 0.0, 0.0, 0.0};
      gn_eval_only<DOF, LPT, C, IO>(p, cx, b, j, traj_ok, x, mu_s, mu_g, acc);
      const double e = group_sum_to_first<LPT>(cx, acc.e);
      if (traj_ok && j == 0) st<IO>(p.err_final, b, div_M(p, e));
    }
  }
}

}  // namespace dgp
