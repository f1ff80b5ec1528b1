/* dgp_pycall.c -- CPython trampoline onto the C-ABI of include/dgpmp2_hip.h (module dgpmp2_amd.lib._dgp_pycall).
 *
 * ctypes spends 3-5 us converting the dozen-odd arguments of one dgp_gn_step call -- a third of the 10.9 us the kernel runs
 * (planner_step_api).  This module does the same job with METH_FASTCALL: integers in, the two by-reference
 * structs (DgpSdf, DgpCovs) built on the C stack, one indirect call.  It links against NOTHING of the product: bind() receives
 * the addresses of the C-ABI entry points that dgpmp2_amd/_capi.py resolved in libdgpmp2_hip.so, so the boundary stays the
 * C-ABI and this file contains no solver logic.  Pointers are passed as Python ints (tensor.data_ptr()), None or 0 = NULL.
 * Every function returns the C-ABI status code; the caller turns a non-zero code into DgpError (dgp_last_error()).
 *
 * Built by __graft_entry__.build() with the host compiler (no HIP).
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include "../../include/dgpmp2_hip.h"

typedef int (*gn_step_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*, void*, void*, void*,
                          int32_t*, void*);
typedef int (*gn_solve_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*, int32_t, double, void*,
                           int32_t*, void*, void*, void*, int32_t*, void*);
typedef int (*eval_errors_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*, void*, void*, void*,
                              void*, void*, void*);
typedef int (*gn_step_backward_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*, const void*,
                                   const void*, const void*, void*, void*, void*, void*, int64_t, int32_t, void*, void*, void*, void*);
typedef int (*eval_errors_backward_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*,
                                       const void*, const void*, const void*, const void*, void*, void*, void*, void*, int64_t, int32_t, void*,
                                       void*);

typedef int (*gn_solve_traced_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*, int32_t, double, void*,
                                  int32_t*, void*, void*, void*, int32_t*, double*, void*);
typedef int (*gn_solve_backward_fn)(const DgpHandle*, int32_t, const void*, const void*, const DgpSdf*, int32_t, const double*, const void*, const int32_t*,
                                    const void*, void*, void*, void*, void*, int64_t, int32_t, void*);
typedef int (*gn_step_errors_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*, void*, void*, void*,
                                 int32_t*, void*, void*, void*, void*);
typedef int (*gn_step_errors_backward_fn)(const DgpHandle*, int32_t, const void*, const void*, const void*, const DgpSdf*, const DgpCovs*, const void*,
                                          const void*, const void*, const void*, const void*, const void*, void*, void*, void*, void*, int64_t, int32_t,
                                          void*, void*, void*, void*, void*);

typedef int (*square_covs_fn)(const void*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, void*, void*, void*, void*, void*);
typedef int (*square_covs_bwd_fn)(const void*, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, const void*, const void*, const void*, void*, void*);
static square_covs_fn f_square_covs;
static square_covs_bwd_fn f_square_covs_bwd;
typedef int (*sum_partial_grids_fn)(const void*, int32_t, int32_t, int64_t, double, void*, int32_t, void*);
static sum_partial_grids_fn f_sum_partial_grids;
static gn_step_fn f_gn_step;
static gn_solve_traced_fn f_gn_solve_traced;
static gn_solve_backward_fn f_gn_solve_backward;
static gn_step_errors_fn f_gn_step_errors;
static gn_step_errors_backward_fn f_gn_step_errors_backward;
static gn_solve_fn f_gn_solve;
static eval_errors_fn f_eval_errors;
static gn_step_backward_fn f_gn_step_backward;
static eval_errors_backward_fn f_eval_errors_backward;

/* int / None -> address; sets *bad on a conversion error */
static inline void* as_ptr(PyObject* o, int* bad) {
  if (o == Py_None) return NULL;
  void* p = PyLong_AsVoidPtr(o);
  if (p == NULL && PyErr_Occurred()) *bad = 1;
  return p;
}
static inline int64_t as_i64(PyObject* o, int* bad) {
  const long long v = PyLong_AsLongLong(o);
  if (v == -1 && PyErr_Occurred()) *bad = 1;
  return (int64_t)v;
}

#define NEED(n, name)                                                                                       \
  if (nargs != (n)) {                                                                                       \
    PyErr_Format(PyExc_TypeError, name " takes exactly %d positional arguments (%zd given)", (n), nargs);   \
    return NULL;                                                                                            \
  }                                                                                                         \
  int bad = 0
#define P(i) as_ptr(a[i], &bad)
#define I(i) as_i64(a[i], &bad)

/* common prefix of every entry point: handle, batch, th, start, goal, the seven fields of DgpSdf (data, rows, cols, batch_stride, layout, grad_mode,
 * grad_indices), the four of DgpCovs (qc_mode, qc_inv, obs_w, eps) -- 16 arguments.
 * sdf_data None/0 -> a NULL DgpSdf* (only dgp_eval_errors[_backward] accept that). */
#define PREFIX 16
#define BUILD_PREFIX                                                                                        \
  const DgpHandle* h = (const DgpHandle*)P(0);                                                              \
  const int32_t batch = (int32_t)I(1);                                                                      \
  const void *th = P(2), *start = P(3), *goal = P(4);                                                       \
  DgpSdf sdf = {P(5), (int32_t)I(6), (int32_t)I(7), I(8), (int32_t)I(9), (int32_t)I(10), (int64_t*)P(11)};  \
  DgpCovs covs = {(int32_t)I(12), P(13), P(14), P(15)};                                                     \
  const DgpSdf* sdfp = sdf.data ? &sdf : NULL

static PyObject* py_bind(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(12, "bind");
  f_gn_step = (gn_step_fn)P(0);
  f_gn_solve = (gn_solve_fn)P(1);
  f_eval_errors = (eval_errors_fn)P(2);
  f_gn_step_backward = (gn_step_backward_fn)P(3);
  f_eval_errors_backward = (eval_errors_backward_fn)P(4);
  f_gn_solve_traced = (gn_solve_traced_fn)P(5);
  f_gn_solve_backward = (gn_solve_backward_fn)P(6);
  f_gn_step_errors = (gn_step_errors_fn)P(7);
  f_gn_step_errors_backward = (gn_step_errors_backward_fn)P(8);
  f_sum_partial_grids = (sum_partial_grids_fn)P(9);
  f_square_covs = (square_covs_fn)P(10);
  f_square_covs_bwd = (square_covs_bwd_fn)P(11);
  if (bad) return NULL;
  Py_RETURN_NONE;
}

#define BOUND(f)                                                                                            \
  if (!(f)) { PyErr_SetString(PyExc_RuntimeError, "_dgp_pycall.bind() has not been called"); return NULL; }

/* gn_step(PREFIX..., dtheta, err, err_ext, info, stream) */
static PyObject* py_gn_step(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 5, "gn_step");
  BOUND(f_gn_step);
  BUILD_PREFIX;
  void *dth = P(16), *err = P(17), *eex = P(18), *info = P(19), *stream = P(20);
  if (bad) return NULL;
  return PyLong_FromLong(f_gn_step(h, batch, th, start, goal, sdfp, &covs, dth, err, eex, (int32_t*)info, stream));
}

/* gn_solve(PREFIX..., max_iters, tol_delta, th_out, iters, err_hist, errext_hist, err_final, info, stream) */
static PyObject* py_gn_solve(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 9, "gn_solve");
  BOUND(f_gn_solve);
  BUILD_PREFIX;
  const int32_t max_iters = (int32_t)I(16);
  const double tol = PyFloat_AsDouble(a[17]);
  if (tol == -1.0 && PyErr_Occurred()) return NULL;
  void *th_out = P(18), *iters = P(19), *eh = P(20), *eeh = P(21), *ef = P(22), *info = P(23), *stream = P(24);
  if (bad) return NULL;
  return PyLong_FromLong(f_gn_solve(h, batch, th, start, goal, sdfp, &covs, max_iters, tol, th_out, (int32_t*)iters, eh, eeh, ef, (int32_t*)info, stream));
}

/* eval_errors(PREFIX..., err, err_ext, unw_sg, unw_gp, unw_obs, stream) */
static PyObject* py_eval_errors(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 6, "eval_errors");
  BOUND(f_eval_errors);
  BUILD_PREFIX;
  void *err = P(16), *eex = P(17), *usg = P(18), *ugp = P(19), *uobs = P(20), *stream = P(21);
  if (bad) return NULL;
  return PyLong_FromLong(f_eval_errors(h, batch, th, start, goal, sdfp, &covs, err, eex, usg, ugp, uobs, stream));
}

/* gn_step_backward(PREFIX..., dtheta, g_dtheta, g_err_ext, g_th, g_start, g_goal, g_sdf, g_sdf_batch_stride, g_sdf_copies, g_qc_inv, g_obs_w, g_eps, stream) */
static PyObject* py_gn_step_backward(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 13, "gn_step_backward");
  BOUND(f_gn_step_backward);
  BUILD_PREFIX;
  const void *dth = P(16), *g_dth = P(17), *g_eex = P(18);
  void *g_th = P(19), *g_st = P(20), *g_go = P(21), *g_sdf = P(22);
  const int64_t g_stride = I(23);
  const int32_t copies = (int32_t)I(24);
  void *g_qc = P(25), *g_ow = P(26), *g_eps = P(27), *stream = P(28);
  if (bad) return NULL;
  return PyLong_FromLong(f_gn_step_backward(h, batch, th, start, goal, sdfp, &covs, dth, g_dth, g_eex, g_th, g_st, g_go, g_sdf, g_stride, copies,
                                            g_qc, g_ow, g_eps, stream));
}

/* eval_errors_backward(PREFIX..., g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th, g_start, g_goal, g_sdf, g_sdf_batch_stride, g_sdf_copies, g_eps, stream) */
static PyObject* py_eval_errors_backward(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 12, "eval_errors_backward");
  BOUND(f_eval_errors_backward);
  BUILD_PREFIX;
  const void *g_eex = P(16), *g_usg = P(17), *g_ugp = P(18), *g_uobs = P(19);
  void *g_th = P(20), *g_st = P(21), *g_go = P(22), *g_sdf = P(23);
  const int64_t g_stride = I(24);
  const int32_t copies = (int32_t)I(25);
  void *g_eps = P(26), *stream = P(27);
  if (bad) return NULL;
  return PyLong_FromLong(f_eval_errors_backward(h, batch, th, start, goal, sdfp, &covs, g_eex, g_usg, g_ugp, g_uobs, g_th, g_st, g_go, g_sdf, g_stride,
                                                copies, g_eps, stream));
}

/* gn_solve_traced(PREFIX..., max_iters, tol_delta, th_out, iters, err_hist, errext_hist, err_final, info, th_hist, stream) */
static PyObject* py_gn_solve_traced(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 10, "gn_solve_traced");
  BOUND(f_gn_solve_traced);
  BUILD_PREFIX;
  const int32_t max_iters = (int32_t)I(16);
  const double tol = PyFloat_AsDouble(a[17]);
  if (tol == -1.0 && PyErr_Occurred()) return NULL;
  void *th_out = P(18), *iters = P(19), *eh = P(20), *eeh = P(21), *ef = P(22), *info = P(23), *hist = P(24), *stream = P(25);
  if (bad) return NULL;
  return PyLong_FromLong(f_gn_solve_traced(h, batch, th, start, goal, sdfp, &covs, max_iters, tol, th_out, (int32_t*)iters, eh, eeh, ef, (int32_t*)info,
                                           (double*)hist, stream));
}

/* gn_solve_backward(handle, batch, start, goal, sdf_data, sdf_rows, sdf_cols, sdf_batch_stride, sdf_layout, sdf_grad_mode, sdf_grad_indices, max_iters, th_hist, th_out, iters, g_th_out,
 *                   g_th_init, g_start, g_goal, g_sdf, g_sdf_batch_stride, g_sdf_copies, stream) */
static PyObject* py_gn_solve_backward(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(23, "gn_solve_backward");
  BOUND(f_gn_solve_backward);
  const DgpHandle* h = (const DgpHandle*)P(0);
  const int32_t batch = (int32_t)I(1);
  const void *start = P(2), *goal = P(3);
  DgpSdf sdf = {P(4), (int32_t)I(5), (int32_t)I(6), I(7), (int32_t)I(8), (int32_t)I(9), (int64_t*)P(10)};
  const int32_t max_iters = (int32_t)I(11);
  const void *hist = P(12), *th_out = P(13), *iters = P(14), *g_out = P(15);
  void *g_th = P(16), *g_st = P(17), *g_go = P(18), *g_sdf = P(19);
  const int64_t g_stride = I(20);
  const int32_t copies = (int32_t)I(21);
  void* stream = P(22);
  if (bad) return NULL;
  return PyLong_FromLong(f_gn_solve_backward(h, batch, start, goal, sdf.data ? &sdf : NULL, max_iters, (const double*)hist, th_out, (const int32_t*)iters,
                                             g_out, g_th, g_st, g_go, g_sdf, g_stride, copies, stream));
}

/* gn_step_errors(PREFIX..., dtheta, err, err_ext, info, unw_sg, unw_gp, unw_obs, stream) */
static PyObject* py_gn_step_errors(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 8, "gn_step_errors");
  BOUND(f_gn_step_errors);
  BUILD_PREFIX;
  void *dth = P(16), *err = P(17), *eex = P(18), *info = P(19), *usg = P(20), *ugp = P(21), *uobs = P(22), *stream = P(23);
  if (bad) return NULL;
  return PyLong_FromLong(f_gn_step_errors(h, batch, th, start, goal, sdfp, &covs, dth, err, eex, (int32_t*)info, usg, ugp, uobs, stream));
}

/* gn_step_errors_backward(PREFIX..., dtheta, g_dtheta, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th, g_start, g_goal, g_sdf, g_sdf_batch_stride,
 *                         g_sdf_copies, g_qc_inv, g_obs_w, g_eps, workspace, stream) */
static PyObject* py_gn_step_errors_backward(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(PREFIX + 17, "gn_step_errors_backward");
  BOUND(f_gn_step_errors_backward);
  BUILD_PREFIX;
  const void *dth = P(16), *g_dth = P(17), *g_eex = P(18), *g_usg = P(19), *g_ugp = P(20), *g_uobs = P(21);
  void *g_th = P(22), *g_st = P(23), *g_go = P(24), *g_sdf = P(25);
  const int64_t g_stride = I(26);
  const int32_t copies = (int32_t)I(27);
  void *g_qc = P(28), *g_ow = P(29), *g_eps = P(30), *ws = P(31), *stream = P(32);
  if (bad) return NULL;
  return PyLong_FromLong(f_gn_step_errors_backward(h, batch, th, start, goal, sdfp, &covs, dth, g_dth, g_eex, g_usg, g_ugp, g_uobs, g_th, g_st, g_go, g_sdf,
                                                   g_stride, copies, g_qc, g_ow, g_eps, ws, stream));
}

/* sum_partial_grids(partial, partial_dtype, copies, elems, scale, out, out_dtype, stream) */
static PyObject* py_sum_partial_grids(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(8, "sum_partial_grids");
  BOUND(f_sum_partial_grids);
  const void* part = P(0);
  const int32_t pdt = (int32_t)I(1), copies = (int32_t)I(2);
  const int64_t elems = I(3);
  const double scale = PyFloat_AsDouble(a[4]);
  if (scale == -1.0 && PyErr_Occurred()) return NULL;
  void* out = P(5);
  const int32_t odt = (int32_t)I(6);
  void* stream = P(7);
  if (bad) return NULL;
  return PyLong_FromLong(f_sum_partial_grids(part, pdt, copies, elems, scale, out, odt, stream));
}

/* square_covariances(raw, dtype, batch, width, n_gp, num_states, learn_eps, dof, sq_scalars, sq_qc_inv, sq_obs_w, sq_eps, stream) */
static PyObject* py_square_covs(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(13, "square_covariances");
  BOUND(f_square_covs);
  const void* raw = P(0);
  const int32_t dt = (int32_t)I(1), B = (int32_t)I(2), W = (int32_t)I(3), ngp = (int32_t)I(4), n = (int32_t)I(5), le = (int32_t)I(6), dof = (int32_t)I(7);
  void *s = P(8), *blk = P(9), *ow = P(10), *ep = P(11), *stream = P(12);
  if (bad) return NULL;
  return PyLong_FromLong(f_square_covs(raw, dt, B, W, ngp, n, le, dof, s, blk, ow, ep, stream));
}

/* square_covariances_backward(raw, dtype, batch, width, n_gp, num_states, learn_eps, dof, g_qc_inv, g_obs_w, g_eps, g_raw, stream) */
static PyObject* py_square_covs_bwd(PyObject* self, PyObject* const* a, Py_ssize_t nargs) {
  NEED(13, "square_covariances_backward");
  BOUND(f_square_covs_bwd);
  const void* raw = P(0);
  const int32_t dt = (int32_t)I(1), B = (int32_t)I(2), W = (int32_t)I(3), ngp = (int32_t)I(4), n = (int32_t)I(5), le = (int32_t)I(6), dof = (int32_t)I(7);
  const void *gq = P(8), *gw = P(9), *ge = P(10);
  void *graw = P(11), *stream = P(12);
  if (bad) return NULL;
  return PyLong_FromLong(f_square_covs_bwd(raw, dt, B, W, ngp, n, le, dof, gq, gw, ge, graw, stream));
}

static PyMethodDef methods[] = {
    {"bind", (PyCFunction)(void (*)(void))py_bind, METH_FASTCALL,
     "bind(gn_step, gn_solve, eval_errors, gn_step_backward, eval_errors_backward, gn_solve_traced, gn_solve_backward, gn_step_errors, gn_step_errors_backward, sum_partial_grids, square_covariances, square_covariances_backward): "
     "addresses of the C-ABI entry points"},
    {"gn_solve_traced", (PyCFunction)(void (*)(void))py_gn_solve_traced, METH_FASTCALL, "dgp_gn_solve_traced"},
    {"gn_solve_backward", (PyCFunction)(void (*)(void))py_gn_solve_backward, METH_FASTCALL, "dgp_gn_solve_backward"},
    {"gn_step_errors", (PyCFunction)(void (*)(void))py_gn_step_errors, METH_FASTCALL, "dgp_gn_step_errors"},
    {"gn_step_errors_backward", (PyCFunction)(void (*)(void))py_gn_step_errors_backward, METH_FASTCALL, "dgp_gn_step_errors_backward"},
    {"sum_partial_grids", (PyCFunction)(void (*)(void))py_sum_partial_grids, METH_FASTCALL, "dgp_sum_partial_grids"},
    {"square_covariances", (PyCFunction)(void (*)(void))py_square_covs, METH_FASTCALL, "dgp_square_covariances"},
    {"square_covariances_backward", (PyCFunction)(void (*)(void))py_square_covs_bwd, METH_FASTCALL, "dgp_square_covariances_backward"},
    {"gn_step", (PyCFunction)(void (*)(void))py_gn_step, METH_FASTCALL, "dgp_gn_step"},
    {"gn_solve", (PyCFunction)(void (*)(void))py_gn_solve, METH_FASTCALL, "dgp_gn_solve"},
    {"eval_errors", (PyCFunction)(void (*)(void))py_eval_errors, METH_FASTCALL, "dgp_eval_errors"},
    {"gn_step_backward", (PyCFunction)(void (*)(void))py_gn_step_backward, METH_FASTCALL, "dgp_gn_step_backward"},
    {"eval_errors_backward", (PyCFunction)(void (*)(void))py_eval_errors_backward, METH_FASTCALL, "dgp_eval_errors_backward"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_dgp_pycall", "METH_FASTCALL trampoline onto the dgpmp2_hip C-ABI", -1, methods};

PyMODINIT_FUNC PyInit__dgp_pycall(void) { return PyModule_Create(&moddef); }
