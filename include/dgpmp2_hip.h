/* dgpmp2_hip.h -- C-ABI of the MI355X-native inner Gauss-Newton solver for dGPMP2.
 *
 * The reference (mhmukadam/dgpmp2) has NO FFI: its boundary for this path is a Python API,
 *   PlanLayer.forward(thb,startb,goalb,imb,sdfb,qc_inv_trajb,obscov_inv_trajb,eps_trajb)
 *       -> (dthetab, err, err_ext)                         diff_gpmp2/gpmp2/plan_layer.py:87-99
 *   DiffGPMP2Planner.step(...) / .forward(...)              diff_gpmp2/gpmp2/diff_gpmp2_planner.py:176-211 / :92-174
 *   PlanLayer.error_batch / error_ext_batch / gp_error / obs_error / start_goal_error
 *                                                           plan_layer.py:273-345, :374-388
 * Each entry point below names the reference function it replaces.  The host-side mirror of the
 * Python classes (same names, arguments, return tuples) lives in dgpmp2_amd/gpmp2/ and calls these
 * through ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - every data pointer is a BORROWED DEVICE pointer (torch tensor.data_ptr()), contiguous row-major
 *    with exactly the torch shapes quoted; element type is the handle's io_dtype (all tensors alike).
 *  - all arithmetic inside the kernels is IEEE binary64 (the reference is fp64-only).
 *  - calls are asynchronous on the HIP stream passed as `stream` (a hipStream_t cast to void*;
 *    NULL = the null stream).  Nothing is allocated, freed or synchronised inside a call.
 *  - return value: DGP_OK or a negative DGP_E* code; never throws.  dgp_last_error() gives the text
 *    of the calling thread's last failure.
 *  - a handle is immutable after dgp_create() => the same handle may be used from several threads
 *    and streams concurrently (unlike the reference PlanLayer, which mutates factor state per call).
 */
#ifndef DGPMP2_HIP_H
#define DGPMP2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGP_ABI_VERSION 6

/* status codes */
#define DGP_OK              0
#define DGP_EINVAL         -1   /* bad argument (NULL pointer, size out of range, bad enum)            */
#define DGP_EUNSUPPORTED   -2   /* valid request outside what the kernels implement (e.g. nlinks != 1) */
#define DGP_EHIP           -3   /* HIP runtime error (launch failure, no device)                       */

/* io_dtype */
#define DGP_F32 0
#define DGP_F64 1
#define DGP_U8  2   /* dgp_sdf_2d images only */

/* flags */
#define DGP_FLAG_NONHOLONOMIC 1u   /* planner_params['non_holonomic'] (plan_layer.py:32), needs dof == 3 */
#define DGP_FLAG_VEL_LIMITS   2u   /* planner_params['use_vel_limits'] (plan_layer.py:34)                 */

/* GP covariance input modes (plan_layer.py:90-91; diff_gpmp2_planner.py:247-290) */
#define DGP_QC_STATIC   0   /* qc_inv == NULL: gp_params['Q_c_inv'] for every factor (diff_gpmp2_planner.py:41-46)  */
#define DGP_QC_PERSTATE 1   /* qc_inv (B, n-1, dof, dof): Q^-1 built as in gp_factor.py:65-73                       */
#define DGP_QC_QFULL    2   /* qc_inv (B, n-1, d, d) is Q^-1 itself ('q_full', plan_layer.py:90)                    */
#define DGP_QC_SCALAR   3   /* qc_inv (B, n-1): one scalar s_k per GP factor, Q_c^-1 = s_k * DgpConfig.Q_c_inv (diagonal) --      */
                            /* 'diag_identity' (diff_gpmp2_planner.py:255-258: q_k^2 I with Q_c_inv = I); dgp_gn_step[_errors] and their backward (g_qc_inv: the (B,n-1,dof,dof) gradient of the blocks s_k I), n <= 256 */

/* Constructor arguments of PlanLayer / DiffGPMP2Planner that the math depends on
 * (plan_layer.py:14-81).  All lengths in the reference's units. */
typedef struct DgpConfig {
  uint32_t struct_size;      /* = sizeof(DgpConfig), ABI guard                                           */
  int32_t  num_states;       /* n = planner_params['total_time_step'] + 1          (plan_layer.py:30);   */
                             /* 2 <= n <= 1024 (dof 2) / 640 (dof 3): up to 256 states the unrolled       */
                             /* kernels, beyond that the loop kernels of csrc/gn_long.h (LDS-bounded)     */
  int32_t  dof;              /* planner_params['dof']: 2 (point robot) or 3 (x,y,theta); d = 2*dof       */
  int32_t  nlinks;           /* robot_model.nlinks; only 1 is implemented                                */
  int32_t  io_dtype;         /* DGP_F32 or DGP_F64                                                       */
  uint32_t flags;            /* DGP_FLAG_*                                                               */
  double   total_time_sec;   /* planner_params['total_time_sec']; dt = total_time_sec/(n-1) (:31)        */
  double   x_lims[2];        /* env_params['x_lims']                                                     */
  double   y_lims[2];        /* env_params['y_lims']                                                     */
  double   K_s, K_g;         /* gp_params['K_s'], ['K_g']: prior sigma, weight 1/K^2 (:64-65)            */
  double   reg;              /* optim_params['reg']: delta added to the diagonal (:96,:219)              */
  double   sphere_radius;    /* robot_model.get_sphere_radii()  (obstacle_cost.py:30)                    */
  double   Q_c_inv[9];       /* gp_params['Q_c_inv'] (dof x dof, row-major): static GP weight and the    */
                             /* FIXED weight of err_ext (plan_layer.py:70-73,80)                         */
  double   cost_sigma;       /* obs_params['cost_sigma']: static / fixed obstacle weight 1/sigma^2 (:74) */
  double   epsilon_dist;     /* obs_params['epsilon_dist']: static epsilon                               */
  double   K_d;              /* gp_params['K_d'] (non-holonomic factor sigma)                            */
  double   K_v, v_x, v_y;    /* gp_params['K_v'], ['v_x'], ['v_y'] (velocity-limit factor)               */
} DgpConfig;

typedef struct DgpHandle DgpHandle;

/* Signed-distance field argument: sdfb (B,1,H',W') of PlanLayer.forward (only sdfb[:,0] is read,
 * obstacle_cost.py:35).  batch_stride is in ELEMENTS between consecutive samples' grids; 0 means one
 * grid shared by the whole batch (an expand()ed tensor). */
/* Memory layout of a grid (DgpSdf::layout).  ROWMAJOR is the reference's tensor.  TILED4: the same H' x W' values stored as 4 x 4 tiles,
 * element (y, x) at offset ((y / 4) * ceil(W' / 4) + x / 4) * 16 + (y % 4) * 4 + (x % 4) of a grid of ceil(H'/4) * ceil(W'/4) * 16 elements (cells past
 * the last row / column are padding, never read): the 2 x 2 footprint of a bilinear lookup then lies in ONE 64-byte (fp32) tile in 9 cases of 16 instead of
 * in two rows 4 W' bytes apart -- what dgp_sdf_2d(..., out_layout) writes directly for per-sample grids (DESIGN.md section 3 "SDF"). */
#define DGP_SDF_ROWMAJOR 0
#define DGP_SDF_TILED4   1

/* How the backward entry points deliver dL/d(sdf) (DgpSdf::grad_mode; the `g_sdf` argument is the destination):
 *   DGP_GSDF_DENSE     g_sdf = grid(s) of the handle's io_dtype in the layout given by g_sdf_batch_stride / g_sdf_copies, accumulated with atomics
 *                      (the caller zeroes them) -- the reference's dense sdf.grad;
 *   DGP_GSDF_DENSE_F64 the same with grids of DOUBLES whatever io_dtype: the partial copies of a shared grid, summed (and cast) by the caller -- the
 *                      accumulation over thousands of trajectories no longer depends on the order of fp32 atomics;
 *   DGP_GSDF_SPARSE    no grid at all: g_sdf = (passes, B, n, 4) tap VALUES (io_dtype) and grad_indices = the (4, passes*B*n*4) int64 COO indices
 *                      (b, 0, y, x) of a sparse tensor of sdfb's shape (torch.sparse_coo_tensor; explicit zeros and duplicates included).  A TILED grid
 *                      (layout DGP_SDF_TILED4, tensor (B,1,H'/4,W'/4,4,4)): SIX index rows (b, 0, y / 4, x / 4, y % 4, x % 4), a (6, nnz) array.  passes = 1 (dgp_gn_step_backward, dgp_eval_errors_backward), 2 (dgp_gn_step_errors_backward with an
 *                      unweighted-error cotangent: the taps at th + dtheta, then those at th) or max_iters (dgp_gn_solve_backward, whose caller
 *                      ZERO-FILLS both arrays: passes a trajectory did not run stay untouched).  No atomics, no O(B H' W') zero fill: the dense
 *                      gradient of B per-sample grids is 1 GiB of zeros at B = 4096, 256 x 256 around 4 MB of taps.  num_states <= 256.  */
#define DGP_GSDF_DENSE     0
#define DGP_GSDF_DENSE_F64 1
#define DGP_GSDF_SPARSE    2

typedef struct DgpSdf {
  const void* data;
  int32_t     rows;          /* H' */
  int32_t     cols;          /* W' >= 2; res = (x_lims[1]-x_lims[0])/W'  (obstacle_cost.py:34); a single-column grid
                                is rejected with DGP_EUNSUPPORTED: the taps are fetched as column pairs */
  int64_t     batch_stride;
  int32_t     layout;        /* DGP_SDF_*: layout of `data` (and of a dense g_sdf)                                                  */
  int32_t     grad_mode;     /* DGP_GSDF_*: backward entry points only                                                              */
  int64_t*    grad_indices;  /* DGP_GSDF_SPARSE: the (4, nnz) -- tiled grids: (6, nnz) -- int64 index array, else ignored              */
} DgpSdf;

/* Per-call covariance inputs = the three trailing arguments of PlanLayer.forward. */
typedef struct DgpCovs {
  int32_t     qc_mode;       /* DGP_QC_*                                                                 */
  const void* qc_inv;        /* see DGP_QC_*; NULL iff DGP_QC_STATIC                                     */
  const void* obs_w;         /* obscov_inv_trajb (B,n,1,1) or NULL = static 1/cost_sigma^2               */
  const void* eps;           /* eps_trajb (B,n,1,1) or NULL = static epsilon_dist                        */
} DgpCovs;

int         dgp_abi_version(void);
const char* dgp_last_error(void);

/* PlanLayer.__init__ (plan_layer.py:14-81).  Host-only; validates cfg, precomputes constants. */
int  dgp_create(const DgpConfig* cfg, DgpHandle** out);
void dgp_destroy(DgpHandle* h);

/* M of plan_layer.py:43-45 (rows of the dense system; the normaliser of err) for this handle. */
int  dgp_num_factor_rows(const DgpHandle* h);

/* Launch shape the library will use for a batch of `batch` trajectories: lanes per trajectory and consecutive states per
 * lane (reporting / tuning aid; the environment variable DGP_FORCE_SHAPE="LPT,C" read by dgp_create pins it).  num_states > 256:
 * (64, ceil(num_states / 64)) -- one trajectory per wavefront, the rows of a lane walked in a loop. */
int  dgp_launch_shape(const DgpHandle* h, int32_t batch, int32_t* lanes_per_trajectory, int32_t* states_per_lane);

/* Kernel variant dgp_gn_step / dgp_gn_solve launch for a batch of `batch` trajectories WITH STATIC covariances (covs == NULL):
 * 1 = block elimination with the constant GP blocks as scalar operands, 3 / 4 = interior rows eliminated through the Woodbury identity
 * on the constant GP block (Q_c_inv = c I, no velocity-limit factors, a launch shape with four states per lane; gn_woodbury.h --
 * 3 when num_states fills the shape exactly, 4 otherwise),
 * 0 = the general kernels (non-diagonal Q_c_inv).  Per-state covariance tensors select their own kernels per call.  Reporting
 * only (bench.py names the kernel whose instruction counts it quotes); DGP_NO_WOODBURY=1 at dgp_create keeps variant 1. */
int  dgp_step_kernel_variant(const DgpHandle* h, int32_t batch);

/* One batched Gauss-Newton step == PlanLayer.forward (plan_layer.py:87-99):
 * factor evaluation (gp_factor.py:100-110, prior_factor.py:15-18, obstacle_factor.py:35-40 ->
 * obstacle_cost.py:29-38 -> sdf_utils.py:38-107, custom_factors/), assembly of the block-tridiagonal
 * A^T K A + delta I and A^T K b (plan_layer.py:152-220) and the solve (plan_layer.py:226-228), plus
 * err (error_batch, :273-308) and err_ext (error_ext_batch, :310-345) at the INPUT trajectory.
 *   th (B,n,d)  start,goal (B,1,d)  ->  dtheta (B,n,d)  err (B,1,1)  err_ext (B,1,1)
 *   info (B) int32, optional: 0 ok, 1 = a non-positive pivot was met (system not SPD; the reference
 *   raises from torch.cholesky there).  err / err_ext may be NULL. */
int dgp_gn_step(const DgpHandle* h, int32_t batch,
                const void* th, const void* start, const void* goal,
                const DgpSdf* sdf, const DgpCovs* covs,
                void* dtheta, void* err, void* err_ext, int32_t* info, void* stream);

/* Fused Gauss-Newton loop == DiffGPMP2Planner.forward's inner `while True` (diff_gpmp2_planner.py:122-156)
 * for every trajectory of the batch at once, the trajectory staying in registers across iterations.
 * Per trajectory: repeat {dtheta,err,err_ext = step(th); th += dtheta; j++} until
 * ||dtheta||_F < tol_delta or j >= max_iters (utils/planner_utils.py:3-16; the last dtheta IS applied).
 *   th_out (B,n,d); iters (B) int32; err_hist, errext_hist (B,max_iters), entries at or past iters[b] untouched;
 *   err_final (B): error_batch at th_out (diff_gpmp2_planner.py:145,162).  Optional outputs may be NULL. */
int dgp_gn_solve(const DgpHandle* h, int32_t batch,
                 const void* th_init, const void* start, const void* goal,
                 const DgpSdf* sdf, const DgpCovs* covs,
                 int32_t max_iters, double tol_delta,
                 void* th_out, int32_t* iters, void* err_hist, void* errext_hist, void* err_final,
                 int32_t* info, void* stream);

/* Factor evaluation only == PlanLayer.error_batch / error_ext_batch (plan_layer.py:273-345) and the
 * unweighted errors of DiffGPMP2Planner.unweighted_errors_batch (diff_gpmp2_planner.py:229-237 ->
 * plan_layer.py:374-388).  Any output may be NULL.  err, err_ext, unw_sg, unw_gp, unw_obs: (B).
 * `sdf` (or sdf->data) may be NULL when err, err_ext and unw_obs are all NULL: PlanLayer.gp_error(thb) and
 * start_goal_error(thb) (plan_layer.py:374-377, :384-388) take no grid, and then none is read. */
int dgp_eval_errors(const DgpHandle* h, int32_t batch,
                    const void* th, const void* start, const void* goal,
                    const DgpSdf* sdf, const DgpCovs* covs,
                    void* err, void* err_ext, void* unw_sg, void* unw_gp, void* unw_obs, void* stream);

/* Backward of dgp_gn_step (the reference gets it from torch autograd over plan_layer.py:152-234;
 * consumers: learning/train_planner.py:366-374, examples/diff_gpmp2_2d_example.py:77).
 * Given the forward inputs, the forward output dtheta (B,n,d), g_dtheta = dL/d(dtheta) (B,n,d) and
 * g_err_ext = dL/d(err_ext) (B) (either cotangent may be NULL = 0; dtheta may be NULL iff g_dtheta is),
 * re-assembles the system, solves the adjoint system Lambda lambda = g_dtheta and writes dL/d{th,start,goal}
 * (same shapes), dL/d(qc_inv) (shape of the qc_mode), dL/d(obs_w), dL/d(eps) (B,n), and ACCUMULATES dL/d(sdf)
 * into g_sdf with atomics (layout given by g_sdf_batch_stride; the caller zeroes it).  NULL outputs are skipped.
 * g_sdf_copies: 1, or (shared grid, g_sdf_batch_stride == 0, only) the number of PARTIAL grids laid out back to back
 * in g_sdf: a wavefront accumulates into copy (xcc_id % g_sdf_copies) and the caller sums the copies afterwards.  With
 * g_sdf_copies >= 8 (gfx950 has at most 8 XCDs, so no two XCDs share a copy) the accumulation uses XCD-local L2 atomics
 * instead of device-scope ones (the per-XCD L2s are not coherent with each other; every copy is then only ever touched
 * by ONE L2 during the kernel); with 2..7 copies the atomics stay device-scope and the copies only spread contention. */
int dgp_gn_step_backward(const DgpHandle* h, int32_t batch,
                         const void* th, const void* start, const void* goal,
                         const DgpSdf* sdf, const DgpCovs* covs,
                         const void* dtheta, const void* g_dtheta, const void* g_err_ext,
                         void* g_th, void* g_start, void* g_goal,
                         void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies,
                         void* g_qc_inv, void* g_obs_w, void* g_eps, void* stream);

/* The partial copies of a shared grid's gradient (g_sdf_copies above) -> the gradient: out[e] = scale * sum_c partial[c * elems + e], e < elems = H' W',
 * one launch at memory speed; partial_dtype DGP_F32 / DGP_F64 (DGP_GSDF_DENSE_F64), out_dtype DGP_F32 / DGP_F64.  `scale`: 1, or 1 / B for a caller whose
 * autograd will sum B equal shares of an expand()ed grid.  (torch.sum over the copies costs 45 us for 16 x 256 x 256 doubles on MI355X; this 3.) */
int dgp_sum_partial_grids(const void* partial, int32_t partial_dtype, int32_t copies, int64_t elems, double scale,
                          void* out, int32_t out_dtype, void* stream);

/* Backward of dgp_eval_errors == torch autograd through PlanLayer.error_ext_batch (plan_layer.py:310-345) and the unweighted errors
 * gp_error / obs_error / start_goal_error (:374-388), which the reference's training loss differentiates
 * (learning/train_planner.py:327,342 -> one_step_loss :75-120 -> final_loss.backward() :366-374).
 * Cotangents g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs: (B), any may be NULL (= 0).  Writes dL/d{th,start,goal} (same shapes as the
 * inputs), dL/d(eps) (B,n; only with covs->eps, the current epsilons of plan_layer.py:94,329) and ACCUMULATES dL/d(sdf) into g_sdf
 * (layout and partial copies exactly as in dgp_gn_step_backward; the caller zeroes it).  NULL outputs are skipped.  err (error_batch)
 * runs under no_grad in the reference (:275) and has no cotangent; none of these errors depends on qc_inv / obs_w (fixed or unit
 * weights), so covs->qc_inv and covs->obs_w are ignored.  `sdf` (or sdf->data) may be NULL when g_err_ext, g_unw_obs and g_sdf are. */
int dgp_eval_errors_backward(const DgpHandle* h, int32_t batch,
                             const void* th, const void* start, const void* goal,
                             const DgpSdf* sdf, const DgpCovs* covs,
                             const void* g_err_ext, const void* g_unw_sg, const void* g_unw_gp, const void* g_unw_obs,
                             void* g_th, void* g_start, void* g_goal,
                             void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies,
                             void* g_eps, void* stream);

/* ---- round 4: what the reference's callers do around the step, as single calls ------------------------------------------------ */

/* dgp_gn_solve with the trajectory history the backward pass needs: th_hist (max_iters, B, n, d) in fp64 WHATEVER the handle's io_dtype
 * (dtheta_k = th_{k+1} - th_k is formed from it), row (k, b) = th_k of trajectory b for k < iters[b]; rows at or past iters[b] are not
 * written.  iters must be given with th_hist.  th_hist == NULL: exactly dgp_gn_solve.  num_states <= 256. */
int dgp_gn_solve_traced(const DgpHandle* h, int32_t batch,
                        const void* th_init, const void* start, const void* goal,
                        const DgpSdf* sdf, const DgpCovs* covs,
                        int32_t max_iters, double tol_delta,
                        void* th_out, int32_t* iters, void* err_hist, void* errext_hist, void* err_final,
                        int32_t* info, double* th_hist, void* stream);

/* Backward of the whole Gauss-Newton loop == torch autograd through DiffGPMP2Planner.forward, which keeps the graph across its
 * iterations (diff_gpmp2_planner.py:122-156; consumer examples/diff_gpmp2_2d_example.py:77): given dgp_gn_solve_traced's th_hist,
 * th_out, iters and the cotangent g_th_out (B,n,d) of the final trajectory, ONE launch walks th_{k+1} = th_k + dtheta(th_k) backwards
 * per trajectory (adjoint solve + per-factor chain rule per iteration, the running cotangent in registers) and writes dL/d th_init
 * (B,n,d), dL/d start, dL/d goal (B,1,d) and ACCUMULATES dL/d sdf (layout / partial copies as in dgp_gn_step_backward; the caller
 * zeroes it).  Static covariances (what forward() runs without learn modules; any Q_c_inv since round 6 -- a non-diagonal one runs the general chain kernels), num_states <= 256;
 * DGP_EUNSUPPORTED otherwise -- chain dgp_gn_step_backward then.  The per-iteration errors have no cotangent: forward() returns them
 * as python floats (:138-141). */
int dgp_gn_solve_backward(const DgpHandle* h, int32_t batch,
                          const void* start, const void* goal, const DgpSdf* sdf,
                          int32_t max_iters, const double* th_hist, const void* th_out, const int32_t* iters,
                          const void* g_th_out,
                          void* g_th_init, void* g_start, void* g_goal,
                          void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies, void* stream);

/* One iteration of the reference's training loop (learning/train_planner.py:311-327): dgp_gn_step, then the unweighted errors of
 * DiffGPMP2Planner.unweighted_errors_batch at th + dtheta (the sum formed in io_dtype, as torch forms th_curr_b + dthetab), no th + dtheta
 * tensor in between.  unw_* (B), any may be NULL (all NULL: dgp_gn_step).  ONE launch (the step kernels with an errors epilogue) for row-major
 * grids and num_states <= 128 -- dof = 2: every covariance representation; dof = 3: everything but the general family (q_full tensors, a non-diagonal
 * Q_c_inv: measured slower in one launch than in two, profiles/r06_d6_general_twin.txt); otherwise two stream-ordered launches. */
int dgp_gn_step_errors(const DgpHandle* h, int32_t batch,
                       const void* th, const void* start, const void* goal,
                       const DgpSdf* sdf, const DgpCovs* covs,
                       void* dtheta, void* err, void* err_ext, int32_t* info,
                       void* unw_sg, void* unw_gp, void* unw_obs, void* stream);

/* Backward of dgp_gn_step_errors: cotangents of dtheta, err_ext and of the three unweighted errors at th + dtheta in, every gradient
 * of dgp_gn_step_backward out (same conventions; g_sdf accumulated).  dof = 2, num_states <= 256: ONE launch -- the errors' backward at th + dtheta
 * runs as a prologue of the step's backward kernel and hands dL/d(th + dtheta) over inside the launch (lane-private LDS), `workspace` may be NULL.
 * dof = 3 and longer trajectories: two stream-ordered launches, the first leaves dL/d(th + dtheta) in `workspace` ((B,n,d) elements of io_dtype,
 * caller-provided: nothing is allocated inside a call).  No unweighted-error cotangent: exactly dgp_gn_step_backward (workspace may be NULL). */
int dgp_gn_step_errors_backward(const DgpHandle* h, int32_t batch,
                                const void* th, const void* start, const void* goal,
                                const DgpSdf* sdf, const DgpCovs* covs,
                                const void* dtheta, const void* g_dtheta, const void* g_err_ext,
                                const void* g_unw_sg, const void* g_unw_gp, const void* g_unw_obs,
                                void* g_th, void* g_start, void* g_goal,
                                void* g_sdf, int64_t g_sdf_batch_stride, int32_t g_sdf_copies,
                                void* g_qc_inv, void* g_obs_w, void* g_eps, void* workspace, void* stream);

/* DiffGPMP2Planner.get_covariances for a single-link robot in the modes whose tensors are plain squares of the learn module's output (diff_gpmp2_planner.py:247-290,
 * 'diag_identity' and 'fix_dynamics', with or without learned epsilons) as ONE small launch each way -- the reference spends some twenty tiny torch kernels on the
 * slices, outer products, broadcasts and their backward, which cost as much as the solver once a training iteration is replayed from a HIP graph.
 * raw (B, width): the module's output vector out[:, 0, :], column blocks [0, n_gp) one scalar q_k per GP factor (n_gp = n - 1: 'diag_identity'; 0: 'fix_dynamics'),
 * [n_gp, n_gp + n) the raw obstacle weights o_i, [n_gp + n, n_gp + 2 n) the raw epsilons e_i (learn_eps != 0).  Outputs, every one optional, dense, io dtype of `dtype`:
 *   sq_scalars (B, n_gp) = q_k^2 (what DGP_QC_SCALAR takes), sq_qc_inv (B, n_gp, dof, dof) = q_k^2 I (the tensor the reference returns), sq_obs_w (B, n) = o_i^2,
 *   sq_eps (B, n) = e_i^2 -- the squares formed in the I/O type, as torch's v * v. */
int dgp_square_covariances(const void* raw, int32_t dtype, int32_t batch, int32_t width, int32_t n_gp, int32_t num_states, int32_t learn_eps, int32_t dof,
                           void* sq_scalars, void* sq_qc_inv, void* sq_obs_w, void* sq_eps, void* stream);
/* Its backward: g_raw (B, width) = 2 raw * [trace of g_qc_inv's dof x dof blocks (the gradient dgp_gn_step_backward writes under DGP_QC_SCALAR) | g_obs_w | g_eps],
 * zero in columns the mode does not consume; any gradient input may be NULL (= 0). */
int dgp_square_covariances_backward(const void* raw, int32_t dtype, int32_t batch, int32_t width, int32_t n_gp, int32_t num_states, int32_t learn_eps, int32_t dof,
                                    const void* g_qc_inv, const void* g_obs_w, const void* g_eps, void* g_raw, void* stream);

/* Signed distance fields of a batch of occupancy images on the GPU: utils/sdf_utils.py:6-21 (sdf_2d) for every image of the batch --
 *   im = image > 0.75 (free space), padded by `padlen` pixels of free space on every side (:13-15),
 *   sdf = (distance_transform_edt(im) - distance_transform_edt(1 - im)) * res (:16-20; scipy.ndimage underneath),
 * i.e. positive distances to the nearest obstacle pixel in free space, negative distances to the nearest free pixel inside obstacles.
 * image (batch, rows, cols) contiguous, image_dtype DGP_F32 / DGP_F64 / DGP_U8; sdf_out (batch, rows + 2 padlen, cols + 2 padlen),
 * out_dtype DGP_F32 / DGP_F64 (the reference returns float64); out_layout DGP_SDF_ROWMAJOR, or DGP_SDF_TILED4: batch grids of ceil(H'/4) * ceil(W'/4) * 16
 * elements in the tiled layout of DgpSdf::layout (what the GN kernels read with half the cache lines when every trajectory has its own grid; padding cells
 * of the last tile row / column are not written).  Bit-identical to scipy's result in float64 (squared distances are
 * integers), including its convention for an image with no pixel of the other kind (distances from the pixel at row -1, column 0).
 * workspace: device memory of at least dgp_sdf_2d_workspace_bytes(...) bytes, 4-byte aligned, owned by the call until the stream has
 * passed it.  Two stream-ordered launches (+ one memset of batch words); padded sides up to 8192, batch up to 65535. */
size_t dgp_sdf_2d_workspace_bytes(int32_t batch, int32_t rows, int32_t cols, int32_t padlen);
int dgp_sdf_2d(const void* image, int32_t image_dtype, int32_t batch, int32_t rows, int32_t cols, int32_t padlen, double res,
               void* sdf_out, int32_t out_dtype, int32_t out_layout, void* workspace, size_t workspace_bytes, void* stream);

/* Measurement aid (no counterpart in the reference): the NEXT kernel launched by the calling thread through any entry point
 * above records its own begin and end on the two HIP events (hipEvent_t, created with timing enabled, cast to void*), the way
 * hipExtLaunchKernelGGL does -- hipEventElapsedTime(start, stop) is then that kernel's execution time, the quantity
 * rocprofv3 --kernel-trace reports, with no marker packets between back-to-back launches.  One-shot: consumed by that launch.
 * Passing NULL for either event cancels a pending request. */
int dgp_time_next_launch(void* start_event, void* stop_event);
/* Events for dgp_time_next_launch, created and read through the HIP runtime this library is linked against (hipEvent_t as void*);
 * dgp_event_elapsed_ms needs both events completed (synchronise the stream first). */
int  dgp_event_create(void** out_event);
void dgp_event_destroy(void* event);
int  dgp_event_elapsed_ms(void* start_event, void* stop_event, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* DGPMP2_HIP_H */
