"""ctypes driver of oracle/gn_blocktri.c (TEST / BENCH INFRASTRUCTURE): block-tridiagonal fp64 CPU restatement of the
Gauss-Newton step, fast enough for the full 4096 x 64 benchmark batch.  See gn_blocktri.c for the reference citations."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, 'libgn_blocktri.so')


class OrcParams(C.Structure):
  _fields_ = [('n', C.c_int32), ('dof', C.c_int32), ('sdf_rows', C.c_int32), ('sdf_cols', C.c_int32), ('sdf_bstride', C.c_int64),
              ('flags', C.c_int32), ('qc_mode', C.c_int32), ('dt', C.c_double), ('w_s', C.c_double), ('w_g', C.c_double),
              ('reg', C.c_double), ('radius', C.c_double), ('eps_static', C.c_double), ('obs_w_fix', C.c_double),
              ('qc_fix', C.c_double * 9), ('x_lims', C.c_double * 2), ('y_lims', C.c_double * 2), ('w_d', C.c_double),
              ('w_v', C.c_double), ('vmax', C.c_double * 2), ('M', C.c_double)]


_LIB_LD = os.path.join(_HERE, 'libgn_blocktri_ld.so')
_libs = {}


def lib(extended=False):
  """fp64 build, or (extended=True) the build whose assembly and block solve run in 80-bit extended precision."""
  if extended not in _libs:
    path = _LIB_LD if extended else _LIB
    src = os.path.join(_HERE, 'gn_blocktri.c')
    if not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path):
      subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])
    L = C.CDLL(path)
    assert L.orc_sizeof_params() == C.sizeof(OrcParams)
    L.orc_work_doubles.restype = C.c_int64
    assert L.orc_real_bytes() == (16 if extended else 8)
    _libs[extended] = L
  return _libs[extended]


def gn_step(p, th, start, goal, sdf, qc=None, ow=None, eps=None, q_full=False, nthreads=1, extended=False):
  """p: oracle.gpmp2_oracle.OracleParams; arrays as in gpmp2_oracle.plan_layer_forward (sdf (B|1,1,H,W)).
  -> dtheta (B,n,d), err (B,), err_ext (B,), info (B,).  extended: solve in extended precision (results rounded to fp64)."""
  L = lib(extended)
  B, n, d = th.shape
  f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
  th, start, goal, sdf, qc, ow, eps = f(th), f(start), f(goal), f(sdf), f(qc), f(ow), f(eps)
  P = OrcParams()
  P.n, P.dof, P.sdf_rows, P.sdf_cols = n, p.dof, sdf.shape[-2], sdf.shape[-1]
  P.sdf_bstride = 0 if sdf.shape[0] == 1 else sdf.shape[-1] * sdf.shape[-2]
  P.flags = (1 if p.non_holonomic else 0) | (2 if p.use_vel_limits else 0)
  P.qc_mode = 0 if qc is None else (2 if q_full else 1)
  P.dt, P.w_s, P.w_g, P.reg, P.radius = p.dt, 1.0 / p.K_s ** 2.0, 1.0 / p.K_g ** 2.0, p.reg, p.radius
  P.eps_static, P.obs_w_fix = p.epsilon_dist, 1.0 / p.cost_sigma ** 2.0
  q = np.asarray(p.Q_c_inv, dtype=np.float64).reshape(-1)
  for k in range(9): P.qc_fix[k] = q[k] if k < q.size else 0.0
  P.x_lims[0], P.x_lims[1], P.y_lims[0], P.y_lims[1] = p.x_lims[0], p.x_lims[1], p.y_lims[0], p.y_lims[1]
  P.w_d = 1.0 / p.K_d ** 2.0 if p.non_holonomic else 0.0
  P.w_v = 1.0 / p.K_v ** 2.0 if p.use_vel_limits else 0.0
  P.vmax[0], P.vmax[1] = p.v_x, p.v_y
  P.M = float(p.M)
  dth = np.empty((B, n, d)); err = np.empty(B); eex = np.empty(B); info = np.empty(B, dtype=np.int32)
  work = np.empty(int(L.orc_work_doubles(n)) * max(1, nthreads) + 2)
  work = work[(-work.ctypes.data // 8) % 2:]      # 16-byte aligned (long double)
  ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
  L.orc_gn_step(C.byref(P), C.c_int64(B), ptr(th), ptr(start), ptr(goal), ptr(sdf), ptr(qc), ptr(ow), ptr(eps), ptr(dth), ptr(err),
                ptr(eex), ptr(info), ptr(work), C.c_int(nthreads))
  return dth, err, eex, info
