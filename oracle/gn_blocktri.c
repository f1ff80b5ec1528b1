/* gn_blocktri.c -- CPU oracle #2 (TEST / BENCH INFRASTRUCTURE, not product code).
 *
 * The same Gauss-Newton step as oracle/gpmp2_oracle.py (the literal dense restatement of the reference), but
 * assembled directly as the block-tridiagonal normal equations and solved by a scalar block-Cholesky (Thomas)
 * sweep in fp64 -- O(n d^3) per trajectory instead of O((n d)^3), so the FULL benchmark batch (4096 x 64 states)
 * can be checked on the CPU in seconds, and bench.py can quote a "best CPU" line next to the dense baseline.
 * It is validated against the numpy oracle and the reference's golden fixtures in tests/test_oracle_golden.py.
 *
 * Reference math (paths relative to /root/reference/diff_gpmp2/):
 *   GP factor        gpmp2/gp/gp_factor.py:31-37 (Phi), :65-73 (Q^-1), :100-110 (e = x_{i+1} - Phi x_i, H1=Phi, H2=-I)
 *   prior factor     gpmp2/gp/prior_factor.py:15-18, weights gpmp2/plan_layer.py:64-68
 *   obstacle factor  gpmp2/obstacle/obstacle_factor.py:35-40, obstacle_cost.py:29-38, utils/sdf_utils.py:38-107
 *   velocity limit   gpmp2/custom_factors/velocity_limit_factor.py:17-29
 *   non-holonomic    gpmp2/custom_factors/nonholonomic_factor.py:16-30
 *   system / solve   gpmp2/plan_layer.py:152-234 (LAM = A^T K A + delta I, dtheta = LAM^-1 A^T K b); err :273-308
 *
 * Only tests/, __graft_entry__ and bench.py's cpu_baseline leg may load the resulting library.
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp)
 *
 * Second build, libgn_blocktri_ld.so (-DORC_REAL="long double" -DORC_SQRT=sqrtl): the factor errors, the assembly and the block solve
 * in x87 80-bit extended precision (64-bit mantissa) -- the ARBITER of tests/stress_random_configs.py when a weakly regularised system
 * (cond(Lambda) up to 1e7) puts two fp64 solvers a few 1e-9 apart: which of them is further from the extended-precision solution.
 * The bilinear lookup and the hinge decision stay in fp64 in the reference's operation order in both builds (same active set).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define MAXD 6
#ifndef ORC_REAL
#define ORC_REAL double
#define ORC_SQRT sqrt
#endif
typedef ORC_REAL real_t;

typedef struct {
  int32_t n, dof, sdf_rows, sdf_cols;
  int64_t sdf_bstride;       /* 0 = shared grid */
  int32_t flags;             /* 1 non-holonomic, 2 velocity limits */
  int32_t qc_mode;           /* 0 static, 1 per-state dof x dof, 2 per-state full d x d */
  double dt, w_s, w_g, reg, radius, eps_static, obs_w_fix, qc_fix[9];
  double x_lims[2], y_lims[2];
  double w_d, w_v, vmax[2], M;
} OrcParams;

static void bilinear(const OrcParams* p, const double* grid, double x, double y, double eps, double* cost, double* hx, double* hy) {
  /* utils/sdf_utils.py:57-94 in the reference's operation order; obstacle_cost.py:30,34,36-37 */
  const double res = (p->x_lims[1] - p->x_lims[0]) / (double)p->sdf_cols;
  const double opx = (0. - p->x_lims[0] / res), opy = (0. - p->y_lims[0] / res);
  double px = opx + x / res, py = opy - y / res;
  double fx = floor(px), fy = floor(py);
  if (fx < -1e9) fx = -1e9; if (fx > 1e9) fx = 1e9; if (fy < -1e9) fy = -1e9; if (fy > 1e9) fy = 1e9;
  int64_t x1 = (int64_t)fx, y1 = (int64_t)fy, x2 = x1 + 1, y2 = y1 + 1;
  const int64_t W = p->sdf_cols, H = p->sdf_rows;
  x1 = x1 < 0 ? 0 : (x1 > W - 1 ? W - 1 : x1); x2 = x2 < 0 ? 0 : (x2 > W - 1 ? W - 1 : x2);
  y1 = y1 < 0 ? 0 : (y1 > H - 1 ? H - 1 : y1); y2 = y2 < 0 ? 0 : (y2 > H - 1 ? H - 1 : y2);
  double d11 = grid[y1 * W + x1], d21 = grid[y1 * W + x2], d12 = grid[y2 * W + x1], d22 = grid[y2 * W + x2];
  double fx1 = (double)x1, fx2 = (double)x2, fy1 = (double)y1, fy2 = (double)y2;
  double wa = (fx2 - px) * (fy2 - py), wb = (px - fx1) * (fy2 - py), wc = (fx2 - px) * (py - fy1), wd = (px - fx1) * (py - fy1);
  double dist = wa * d11 + wb * d21 + wc * d12 + wd * d22;
  double Jx = (-1.0 * ((fy2 - py) * (d21 - d11) + (py - fy1) * (d22 - d12))) / res;
  double Jy = ((fx2 - px) * (d12 - d11) + (px - fx1) * (d22 - d21)) / res;
  double et = eps + p->radius;
  int act = dist <= et;
  *cost = act ? (et - dist) : 0.0; *hx = act ? (-1.0 * Jx) : 0.0; *hy = act ? (-1.0 * Jy) : 0.0;
}

static void q_inv(const OrcParams* p, const double* qc, int64_t b, int f, real_t Q[MAXD][MAXD], const real_t* abc) {
  const int dof = p->dof, d = 2 * dof, n = p->n;
  if (p->qc_mode == 2) {
    const double* s = qc + (b * (n - 1) + f) * d * d;
    for (int i = 0; i < d; ++i) for (int j = 0; j < d; ++j) Q[i][j] = s[i * d + j];
    return;
  }
  const double* C = p->qc_mode == 1 ? qc + (b * (n - 1) + f) * dof * dof : p->qc_fix;
  const real_t a = abc[0], bb = abc[1], c = abc[2];                                                  /* gp_factor.py:66-68 */
  for (int i = 0; i < dof; ++i) for (int j = 0; j < dof; ++j) {
    Q[i][j] = a * C[i * dof + j]; Q[i][dof + j] = bb * C[i * dof + j];
    Q[dof + i][j] = bb * C[i * dof + j]; Q[dof + i][dof + j] = c * C[i * dof + j];
  }
}

/* Cholesky of a d x d SPD block in place (lower); returns 0 if not SPD */
static int chol(int d, real_t A[MAXD][MAXD]) {
  for (int j = 0; j < d; ++j) {
    real_t v = A[j][j];
    for (int k = 0; k < j; ++k) v -= A[j][k] * A[j][k];
    if (!(v > 0.0)) return 0;
    A[j][j] = ORC_SQRT(v);
    for (int i = j + 1; i < d; ++i) {
      real_t w = A[i][j];
      for (int k = 0; k < j; ++k) w -= A[i][k] * A[j][k];
      A[i][j] = w / A[j][j];
    }
  }
  return 1;
}
static void fsub(int d, real_t L[MAXD][MAXD], real_t* v) {   /* v <- L^-1 v */
  for (int i = 0; i < d; ++i) { real_t w = v[i]; for (int k = 0; k < i; ++k) w -= L[i][k] * v[k]; v[i] = w / L[i][i]; }
}
static void bsub(int d, real_t L[MAXD][MAXD], real_t* v) {   /* v <- L^-T v */
  for (int i = d - 1; i >= 0; --i) { real_t w = v[i]; for (int k = i + 1; k < d; ++k) w -= L[k][i] * v[k]; v[i] = w / L[i][i]; }
}

/* One trajectory.  Returns 0 ok / 1 not SPD.  work: n*(d*d + d*d + d) real_t. */
static int step_one(const OrcParams* p, int64_t b, const double* th, const double* start, const double* goal, const double* sdf,
                    const double* qc, const double* obs_w, const double* epsv, double* dtheta, double* err, double* err_ext,
                    real_t* work) {
  const int n = p->n, dof = p->dof, d = 2 * dof;
  const real_t dt = p->dt;
  const double* X = th + b * n * d;
  const double* grid = sdf + b * p->sdf_bstride;
  real_t (*Lc)[MAXD][MAXD] = (real_t (*)[MAXD][MAXD])work;                     /* chol(S_i) */
  real_t (*Wm)[MAXD][MAXD] = (real_t (*)[MAXD][MAXD])(work + (size_t)n * MAXD * MAXD);   /* W_i = L_i^-1 U_i */
  real_t (*y)[MAXD] = (real_t (*)[MAXD])(work + (size_t)2 * n * MAXD * MAXD);
  real_t e_tot = 0.0, eext_tot = 0.0;
  real_t Qprev[MAXD][MAXD], eprev[MAXD];
  real_t Qfix[MAXD][MAXD];
  const real_t abc[3] = {12.0 * pow(p->dt, -3.0), -6.0 * pow(p->dt, -2.0), 4.0 * pow(p->dt, -1.0)};      /* (fp64 in both builds: what Python computes) */
  { OrcParams pf = *p; pf.qc_mode = 0; q_inv(&pf, 0, b, 0, Qfix, abc); }
  int bad = 0;
  for (int i = 0; i < n; ++i) {
    const double* x = X + i * d;
    real_t Dm[MAXD][MAXD], U[MAXD][MAXD], r[MAXD];
    memset(Dm, 0, sizeof(Dm)); memset(U, 0, sizeof(U)); memset(r, 0, sizeof(r));
    for (int a = 0; a < d; ++a) Dm[a][a] = p->reg;
    if (i == 0 || i == n - 1) {
      const double* mu = (i == 0 ? start : goal) + b * d;
      const real_t w = i == 0 ? p->w_s : p->w_g;
      real_t s2 = 0.0;
      for (int a = 0; a < d; ++a) { real_t ea = (real_t)mu[a] - x[a]; s2 += ea * ea; Dm[a][a] += w; r[a] += w * ea; }
      e_tot += 0.5 * w * s2; eext_tot += 0.5 * w * s2;
    }
    if (i > 0) {
      for (int a = 0; a < d; ++a) { real_t t = 0.0; for (int c = 0; c < d; ++c) { Dm[a][c] += Qprev[a][c]; t += Qprev[a][c] * eprev[c]; } r[a] -= t; }
    }
    if (i < n - 1) {
      real_t Q[MAXD][MAXD], e[MAXD], PQ[MAXD][MAXD];
      q_inv(p, qc, b, i, Q, abc);
      const double* xp = X + (i + 1) * d;
      for (int a = 0; a < dof; ++a) { e[a] = (real_t)xp[a] - ((real_t)x[a] + dt * x[dof + a]); e[dof + a] = (real_t)xp[dof + a] - x[dof + a]; }
      real_t q = 0.0, qf = 0.0;
      for (int a = 0; a < d; ++a) for (int c = 0; c < d; ++c) { q += e[a] * Q[a][c] * e[c]; qf += e[a] * Qfix[a][c] * e[c]; }
      e_tot += 0.5 * q; eext_tot += 0.5 * qf;
      for (int a = 0; a < dof; ++a) for (int c = 0; c < d; ++c) { PQ[a][c] = Q[a][c]; PQ[dof + a][c] = dt * Q[a][c] + Q[dof + a][c]; }
      for (int a = 0; a < d; ++a) {
        real_t t = 0.0;
        for (int c = 0; c < dof; ++c) { Dm[a][c] += PQ[a][c]; Dm[a][dof + c] += dt * PQ[a][c] + PQ[a][dof + c]; }
        for (int c = 0; c < d; ++c) { U[a][c] = -PQ[a][c]; t += PQ[a][c] * e[c]; }
        r[a] += t;
      }
      memcpy(Qprev, Q, sizeof(Q)); memcpy(eprev, e, sizeof(e));
    }
    {
      const double eps = epsv ? epsv[b * n + i] : p->eps_static;
      const real_t w = obs_w ? obs_w[b * n + i] : p->obs_w_fix;
      double c_, hx_, hy_;
      bilinear(p, grid, x[0], x[1], eps, &c_, &hx_, &hy_);
      const real_t c = c_, hx = hx_, hy = hy_;
      e_tot += 0.5 * w * c * c; eext_tot += 0.5 * (real_t)p->obs_w_fix * c * c;
      Dm[0][0] += w * hx * hx; Dm[0][1] += w * hx * hy; Dm[1][0] += w * hx * hy; Dm[1][1] += w * hy * hy;
      r[0] += w * hx * c; r[1] += w * hy * c;
    }
    if (p->flags & 2) {
      for (int a = 0; a < 2; ++a) {
        const double v = x[dof + a], av = fabs(v);
        const int act = av >= p->vmax[a];
        const real_t c = act ? (real_t)av - p->vmax[a] : 0.0, sg = v > 0 ? 1.0 : (v < 0 ? -1.0 : 0.0), h = act ? -sg : 0.0;
        e_tot += 0.5 * (real_t)p->w_v * c * c; eext_tot += 0.5 * (real_t)p->w_v * c * c;
        Dm[dof + a][dof + a] += p->w_v * h * h; r[dof + a] += p->w_v * h * c;
      }
    }
    if (dof == 3 && (p->flags & 1)) {
      const real_t t = x[2], vx = x[3], vy = x[4], sn = sin(x[2]), cs = cos(x[2]);      /* (sin / cos in fp64 in both builds) */
      (void)t;
      const real_t e = vy * cs - vx * sn, h[3] = {-vy * sn + vx * cs, -sn, cs};
      e_tot += 0.5 * (real_t)p->w_d * e * e; eext_tot += 0.5 * (real_t)p->w_d * e * e;
      for (int a = 0; a < 3; ++a) { for (int c = 0; c < 3; ++c) Dm[2 + a][2 + c] += p->w_d * h[a] * h[c]; r[2 + a] += p->w_d * h[a] * e; }
    }
    /* block Cholesky forward sweep: S_i = D_i - W_{i-1}^T W_{i-1}; y_i = L_i^-1 (r_i - W_{i-1}^T y_{i-1}) */
    if (i > 0) {
      for (int a = 0; a < d; ++a) {
        real_t t = 0.0;
        for (int c = 0; c < d; ++c) { real_t s = 0.0; for (int k = 0; k < d; ++k) s += Wm[i - 1][k][a] * Wm[i - 1][k][c]; Dm[a][c] -= s; }
        for (int k = 0; k < d; ++k) t += Wm[i - 1][k][a] * y[i - 1][k];
        r[a] -= t;
      }
    }
    memcpy(Lc[i], Dm, sizeof(Dm));
    if (!chol(d, Lc[i])) { bad = 1; break; }
    for (int c = 0; c < d; ++c) { real_t col[MAXD]; for (int a = 0; a < d; ++a) col[a] = U[a][c]; fsub(d, Lc[i], col); for (int a = 0; a < d; ++a) Wm[i][a][c] = col[a]; }
    memcpy(y[i], r, sizeof(r)); fsub(d, Lc[i], y[i]);
  }
  if (bad) { for (int k = 0; k < n * d; ++k) dtheta[b * n * d + k] = NAN; }
  else {
    real_t xn[MAXD] = {0};
    for (int i = n - 1; i >= 0; --i) {
      real_t v[MAXD];
      for (int a = 0; a < d; ++a) { real_t t = y[i][a]; if (i < n - 1) for (int c = 0; c < d; ++c) t -= Wm[i][a][c] * xn[c]; v[a] = t; }
      bsub(d, Lc[i], v);
      for (int a = 0; a < d; ++a) { dtheta[(b * n + i) * d + a] = (double)v[a]; xn[a] = v[a]; }
    }
  }
  if (err) err[b] = (double)(e_tot / p->M);
  if (err_ext) err_ext[b] = (double)(eext_tot / p->M);
  return bad;
}

/* Whole batch; all arrays fp64, C-contiguous with the reference's shapes.  info (B) int32 may be NULL.
 * nthreads <= 0: use OpenMP's default. */
int orc_gn_step(const OrcParams* p, int64_t B, const double* th, const double* start, const double* goal, const double* sdf,
                const double* qc, const double* obs_w, const double* eps, double* dtheta, double* err, double* err_ext,
                int32_t* info, void* work_per_thread, int nthreads) {
  const size_t wsz = (size_t)p->n * (2 * MAXD * MAXD + MAXD);
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
  for (int64_t b = 0; b < B; ++b) {
    int tid = 0;
#ifdef _OPENMP
    extern int omp_get_thread_num(void);
    tid = omp_get_thread_num();
#endif
    int bad = step_one(p, b, th, start, goal, sdf, qc, obs_w, eps, dtheta, err, err_ext, (real_t*)work_per_thread + (size_t)tid * wsz);
    if (info) info[b] = bad;
  }
  return 0;
}

/* per-thread work space in units of 8 bytes */
int64_t orc_work_doubles(int32_t n) { return (int64_t)n * (2 * MAXD * MAXD + MAXD) * (int64_t)(sizeof(real_t) / 8 + (sizeof(real_t) % 8 != 0)); }
int32_t orc_real_bytes(void) { return (int32_t)sizeof(real_t); }
int32_t orc_sizeof_params(void) { return (int32_t)sizeof(OrcParams); }
