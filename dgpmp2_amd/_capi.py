"""ctypes binding of the C-ABI declared in include/dgpmp2_hip.h.

The product path loads dgpmp2_amd/lib/libdgpmp2_hip.so (built in-tree by __graft_entry__.build()) and
fails loudly when it is missing -- there is no CPU fallback.  The binding class is parameterised by
(library, symbol prefix) only so that the test-suite can drive tests/emul's CPU wavefront emulator,
which exports the same entry points with the prefix `emul_`, through the very same marshalling code.
"""
import ctypes as C
import importlib.util
import os

import torch  # noqa: F401  -- must be imported BEFORE the HIP library is dlopen'ed: torch ships its own libamdhip64.so.7 /
# libhsa-runtime64; loading ours first would bind /opt/rocm's copies and leave the process with two runtimes ("no ROCm-capable
# device" on the first launch).  With torch first, our library's libamdhip64.so.7 dependency resolves to the one already loaded.

DGP_OK, DGP_EINVAL, DGP_EUNSUPPORTED, DGP_EHIP = 0, -1, -2, -3
DGP_F32, DGP_F64, DGP_U8 = 0, 1, 2
DGP_FLAG_NONHOLONOMIC, DGP_FLAG_VEL_LIMITS = 1, 2
DGP_QC_STATIC, DGP_QC_PERSTATE, DGP_QC_QFULL, DGP_QC_SCALAR = 0, 1, 2, 3
DGP_SDF_ROWMAJOR, DGP_SDF_TILED4 = 0, 1
DGP_GSDF_DENSE, DGP_GSDF_DENSE_F64, DGP_GSDF_SPARSE = 0, 1, 2
DGP_ABI_VERSION = 6

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DGP_LIB_PATH') or os.path.join(_HERE, 'lib', 'libdgpmp2_hip.so')   # override: tuning builds only


class DgpConfig(C.Structure):
  _fields_ = [('struct_size', C.c_uint32), ('num_states', C.c_int32), ('dof', C.c_int32), ('nlinks', C.c_int32),
              ('io_dtype', C.c_int32), ('flags', C.c_uint32), ('total_time_sec', C.c_double),
              ('x_lims', C.c_double * 2), ('y_lims', C.c_double * 2), ('K_s', C.c_double), ('K_g', C.c_double),
              ('reg', C.c_double), ('sphere_radius', C.c_double), ('Q_c_inv', C.c_double * 9),
              ('cost_sigma', C.c_double), ('epsilon_dist', C.c_double), ('K_d', C.c_double), ('K_v', C.c_double),
              ('v_x', C.c_double), ('v_y', C.c_double)]


class DgpSdf(C.Structure):
  _fields_ = [('data', C.c_void_p), ('rows', C.c_int32), ('cols', C.c_int32), ('batch_stride', C.c_int64), ('layout', C.c_int32),
              ('grad_mode', C.c_int32), ('grad_indices', C.c_void_p)]


class DgpCovs(C.Structure):
  _fields_ = [('qc_mode', C.c_int32), ('qc_inv', C.c_void_p), ('obs_w', C.c_void_p), ('eps', C.c_void_p)]


class DgpError(RuntimeError):
  def __init__(self, code, msg):
    super(DgpError, self).__init__('dgpmp2_hip error %d: %s' % (code, msg))
    self.code = code


class CApi(object):
  """Thin typed wrapper over one shared library exporting <prefix>create, <prefix>gn_step, ..."""

  SYMBOLS = ('abi_version', 'last_error', 'create', 'destroy', 'num_factor_rows', 'launch_shape', 'step_kernel_variant', 'gn_step', 'gn_solve',
             'eval_errors', 'gn_step_backward', 'eval_errors_backward', 'gn_solve_traced', 'gn_solve_backward', 'gn_step_errors',
             'gn_step_errors_backward', 'sum_partial_grids', 'square_covariances', 'square_covariances_backward', 'sdf_2d_workspace_bytes', 'sdf_2d', 'time_next_launch', 'event_create', 'event_destroy', 'event_elapsed_ms')

  def __init__(self, path, prefix='dgp_'):
    if not os.path.exists(path):
      raise ImportError('%s is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                        '(hipcc --offload-arch=gfx950).  dgpmp2_amd has no CPU fallback.' % path)
    self.path, self.prefix = path, prefix
    self.lib = C.CDLL(path)
    f = lambda name: getattr(self.lib, prefix + name)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    self.abi_version = f('abi_version'); self.abi_version.restype = C.c_int; self.abi_version.argtypes = []
    self.last_error = f('last_error'); self.last_error.restype = C.c_char_p; self.last_error.argtypes = []
    self.create = f('create'); self.create.restype = C.c_int; self.create.argtypes = [C.POINTER(DgpConfig), C.POINTER(vp)]
    self.destroy = f('destroy'); self.destroy.restype = None; self.destroy.argtypes = [vp]
    self.num_factor_rows = f('num_factor_rows'); self.num_factor_rows.restype = C.c_int; self.num_factor_rows.argtypes = [vp]
    self.launch_shape = f('launch_shape'); self.launch_shape.restype = C.c_int
    self.launch_shape.argtypes = [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    self.step_kernel_variant = f('step_kernel_variant'); self.step_kernel_variant.restype = C.c_int; self.step_kernel_variant.argtypes = [vp, i32]
    self.gn_step = f('gn_step'); self.gn_step.restype = C.c_int
    self.gn_step.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), vp, vp, vp, vp, vp]
    self.gn_solve = f('gn_solve'); self.gn_solve.restype = C.c_int
    self.gn_solve.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), i32, dbl, vp, vp, vp, vp, vp, vp, vp]
    self.eval_errors = f('eval_errors'); self.eval_errors.restype = C.c_int
    self.eval_errors.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), vp, vp, vp, vp, vp, vp]
    self.gn_step_backward = f('gn_step_backward'); self.gn_step_backward.restype = C.c_int
    self.gn_step_backward.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), vp, vp, vp, vp, vp, vp, vp, i64,
                                      i32, vp, vp, vp, vp]
    self.eval_errors_backward = f('eval_errors_backward'); self.eval_errors_backward.restype = C.c_int
    self.eval_errors_backward.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp]
    self.gn_solve_traced = f('gn_solve_traced'); self.gn_solve_traced.restype = C.c_int
    self.gn_solve_traced.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), i32, dbl, vp, vp, vp, vp, vp, vp, vp, vp]
    self.gn_solve_backward = f('gn_solve_backward'); self.gn_solve_backward.restype = C.c_int
    self.gn_solve_backward.argtypes = [vp, i32, vp, vp, C.POINTER(DgpSdf), i32, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]
    self.gn_step_errors = f('gn_step_errors'); self.gn_step_errors.restype = C.c_int
    self.gn_step_errors.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), vp, vp, vp, vp, vp, vp, vp, vp]
    self.gn_step_errors_backward = f('gn_step_errors_backward'); self.gn_step_errors_backward.restype = C.c_int
    self.gn_step_errors_backward.argtypes = [vp, i32, vp, vp, vp, C.POINTER(DgpSdf), C.POINTER(DgpCovs), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64,
                                             i32, vp, vp, vp, vp, vp]
    self.sum_partial_grids = f('sum_partial_grids'); self.sum_partial_grids.restype = C.c_int
    self.sum_partial_grids.argtypes = [vp, i32, i32, i64, dbl, vp, i32, vp]
    self.square_covariances = f('square_covariances'); self.square_covariances.restype = C.c_int
    self.square_covariances.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    self.square_covariances_backward = f('square_covariances_backward'); self.square_covariances_backward.restype = C.c_int
    self.square_covariances_backward.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    self.sdf_2d_workspace_bytes = f('sdf_2d_workspace_bytes'); self.sdf_2d_workspace_bytes.restype = C.c_size_t
    self.sdf_2d_workspace_bytes.argtypes = [i32, i32, i32, i32]
    self.sdf_2d = f('sdf_2d'); self.sdf_2d.restype = C.c_int
    self.sdf_2d.argtypes = [vp, i32, i32, i32, i32, i32, dbl, vp, i32, i32, vp, C.c_size_t, vp]
    self.time_next_launch = f('time_next_launch'); self.time_next_launch.restype = C.c_int; self.time_next_launch.argtypes = [vp, vp]
    self.event_create = f('event_create'); self.event_create.restype = C.c_int; self.event_create.argtypes = [C.POINTER(vp)]
    self.event_destroy = f('event_destroy'); self.event_destroy.restype = None; self.event_destroy.argtypes = [vp]
    self.event_elapsed_ms = f('event_elapsed_ms'); self.event_elapsed_ms.restype = C.c_int
    self.event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
    v = self.abi_version()
    if v != DGP_ABI_VERSION:
      raise ImportError('%s: ABI version %d, binding expects %d' % (path, v, DGP_ABI_VERSION))

  def check(self, rc):
    if rc != DGP_OK:
      raise DgpError(rc, (self.last_error() or b'').decode('utf-8', 'replace'))


class KernelTimer(object):
  """Per-kernel execution times through dgp_time_next_launch: a pool of HIP event pairs, created and read through the library's own
  dgp_event_* helpers (i.e. the HIP runtime the kernels are launched with); `arm()` before a launch makes that launch record its own
  begin / end, `durations_ms()` reads them back after a synchronisation.  Measurement aid for bench.py and the profiling tools --
  not used by the planner."""

  def __init__(self, n, api=None):
    self.api = api if api is not None else get_api()
    self.pairs = []
    for _ in range(n):
      a, b = C.c_void_p(), C.c_void_p()
      self.api.check(self.api.event_create(C.byref(a)))
      self.api.check(self.api.event_create(C.byref(b)))
      self.pairs.append((a, b))
    self.used = 0

  def arm(self):
    a, b = self.pairs[self.used]; self.used += 1
    self.api.check(self.api.time_next_launch(a, b))

  def reset(self):
    self.used = 0

  def durations_ms(self):
    out = []
    for a, b in self.pairs[:self.used]:
      ms = C.c_float()
      self.api.check(self.api.event_elapsed_ms(a, b, C.byref(ms)))
      out.append(ms.value)
    return out

  def __del__(self):
    for a, b in getattr(self, 'pairs', []):
      self.api.event_destroy(a); self.api.event_destroy(b)
    self.pairs = []


_api = None
_pycall = None
import sysconfig
# the trampoline is a CPython extension: its file name carries the interpreter's ABI tag, so that a library built for another Python is
# never loaded by mistake
PYCALL_PATH = os.path.join(_HERE, 'lib', '_dgp_pycall' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))
_PYCALL_ENTRIES = ('gn_step', 'gn_solve', 'eval_errors', 'gn_step_backward', 'eval_errors_backward', 'gn_solve_traced', 'gn_solve_backward',
                   'gn_step_errors', 'gn_step_errors_backward', 'sum_partial_grids', 'square_covariances', 'square_covariances_backward')


def get_api():
  """The product library (HIP).  Raises ImportError if it has not been built."""
  global _api
  if _api is None:
    _api = CApi(LIB_PATH, 'dgp_')
  return _api


class CtypesPycall(object):
  """Drop-in for the trampoline module, on the plain ctypes binding: the same positional signatures (addresses as ints / None, the two
  structs flattened into their seven + four fields), 3-5 us slower per call.  Used when the trampoline has not been built or does not load
  (no Python.h on the build host, another interpreter): the product path is then still the C-ABI, only the marshalling is slower."""

  def __init__(self, api):
    self.api = api

  @staticmethod
  def _sdf(a, i):
    return C.byref(DgpSdf(a[i], int(a[i + 1]), int(a[i + 2]), int(a[i + 3]), int(a[i + 4]), int(a[i + 5]), a[i + 6])) if a[i] else None

  @staticmethod
  def _covs(a, i):
    return C.byref(DgpCovs(int(a[i]), a[i + 1], a[i + 2], a[i + 3]))

  def _prefixed(self, fn, a):
    return fn(a[0], a[1], a[2], a[3], a[4], self._sdf(a, 5), self._covs(a, 12), *a[16:])

  def gn_step(self, *a): return self._prefixed(self.api.gn_step, a)
  def gn_solve(self, *a): return self._prefixed(self.api.gn_solve, a)
  def eval_errors(self, *a): return self._prefixed(self.api.eval_errors, a)
  def gn_step_backward(self, *a): return self._prefixed(self.api.gn_step_backward, a)
  def eval_errors_backward(self, *a): return self._prefixed(self.api.eval_errors_backward, a)
  def gn_solve_traced(self, *a): return self._prefixed(self.api.gn_solve_traced, a)
  def gn_step_errors(self, *a): return self._prefixed(self.api.gn_step_errors, a)
  def gn_step_errors_backward(self, *a): return self._prefixed(self.api.gn_step_errors_backward, a)

  def gn_solve_backward(self, *a):
    return self.api.gn_solve_backward(a[0], a[1], a[2], a[3], self._sdf(a, 4), *a[11:])

  def sum_partial_grids(self, *a): return self.api.sum_partial_grids(*a)
  def square_covariances(self, *a): return self.api.square_covariances(*a)
  def square_covariances_backward(self, *a): return self.api.square_covariances_backward(*a)


def get_pycall():
  """The METH_FASTCALL trampoline onto the product library's entry points (csrc/dgp_pycall.c): the same C-ABI calls as the ctypes
  binding above at a tenth of the per-call marshalling cost.  What the torch-facing layer (dgpmp2_amd.gpmp2) launches through;
  the ctypes methods of `Solver` stay the reference binding (tests, tools).  The product library itself must exist
  (ImportError otherwise: there is no CPU path); a missing or unloadable trampoline falls back to CtypesPycall with a warning."""
  global _pycall
  if _pycall is None:
    api = get_api()
    try:
      if not os.path.exists(PYCALL_PATH):
        raise ImportError('%s is missing' % PYCALL_PATH)
      spec = importlib.util.spec_from_file_location('_dgp_pycall', PYCALL_PATH)
      mod = importlib.util.module_from_spec(spec)
      spec.loader.exec_module(mod)
      addr = lambda f: C.cast(f, C.c_void_p).value
      mod.bind(*[addr(getattr(api, name)) for name in _PYCALL_ENTRIES])
      _pycall = mod
    except (ImportError, OSError, TypeError) as e:
      import warnings
      warnings.warn('dgpmp2_amd: the call trampoline %s is not usable (%s: %s); using the ctypes binding of the same C-ABI entry points '
                    '(3-5 us more host time per launch).  Build it with `python -c "import __graft_entry__ as g; g.build()"`.'
                    % (os.path.basename(PYCALL_PATH), type(e).__name__, e))
      _pycall = CtypesPycall(api)
  return _pycall


def make_config(num_states, dof, io_dtype, total_time_sec, x_lims, y_lims, K_s, K_g, reg, sphere_radius, Q_c_inv,
                cost_sigma, epsilon_dist, non_holonomic=False, use_vel_limits=False, K_d=0.0, K_v=0.0, v_x=0.0, v_y=0.0,
                nlinks=1):
  cfg = DgpConfig()
  cfg.struct_size = C.sizeof(DgpConfig)
  cfg.num_states, cfg.dof, cfg.nlinks, cfg.io_dtype = int(num_states), int(dof), int(nlinks), int(io_dtype)
  cfg.flags = (DGP_FLAG_NONHOLONOMIC if non_holonomic else 0) | (DGP_FLAG_VEL_LIMITS if use_vel_limits else 0)
  cfg.total_time_sec = float(total_time_sec)
  cfg.x_lims[0], cfg.x_lims[1] = float(x_lims[0]), float(x_lims[1])
  cfg.y_lims[0], cfg.y_lims[1] = float(y_lims[0]), float(y_lims[1])
  cfg.K_s, cfg.K_g, cfg.reg, cfg.sphere_radius = float(K_s), float(K_g), float(reg), float(sphere_radius)
  q = [float(v) for row in Q_c_inv for v in row]
  if len(q) != int(dof) * int(dof):
    raise ValueError('Q_c_inv must be dof x dof')
  for k in range(9):
    cfg.Q_c_inv[k] = q[k] if k < len(q) else 0.0
  cfg.cost_sigma, cfg.epsilon_dist = float(cost_sigma), float(epsilon_dist)
  cfg.K_d, cfg.K_v, cfg.v_x, cfg.v_y = float(K_d), float(K_v), float(v_x), float(v_y)
  return cfg


class Solver(object):
  """One DgpHandle.  All methods take raw addresses (ints) -- see dgpmp2_amd.gpmp2.plan_layer for the
  torch-facing layer.  Immutable after construction, hence safe to share between streams/threads."""

  def __init__(self, cfg, api=None):
    self.api = api if api is not None else get_api()
    self.cfg = cfg
    h = C.c_void_p()
    self.api.check(self.api.create(C.byref(cfg), C.byref(h)))
    self.handle = h
    self.h = h.value                       # the handle as a plain int (what the trampoline of get_pycall() takes)
    self.M = self.api.num_factor_rows(h)
    self.n, self.dof, self.d = cfg.num_states, cfg.dof, 2 * cfg.dof

  def __del__(self):
    h = getattr(self, 'handle', None)
    if h is not None and h.value:
      self.api.destroy(h)
      self.handle = None

  def launch_shape(self, batch):
    """(lanes per trajectory, states per lane) the kernels will run with for this batch size."""
    l, c = C.c_int32(), C.c_int32()
    self.api.check(self.api.launch_shape(self.handle, int(batch), C.byref(l), C.byref(c)))
    return l.value, c.value

  def step_kernel_variant(self, batch):
    """Kernel variant a static-covariance step of this batch launches: 1 block elimination, 3 / 4 Woodbury interior elimination (num_states fills the shape / does not), 0 general."""
    v = self.api.step_kernel_variant(self.handle, int(batch))
    if v < 0: self.api.check(v)
    return v

  @staticmethod
  def sdf_arg(ptr, rows, cols, batch_stride, layout=DGP_SDF_ROWMAJOR, grad_mode=DGP_GSDF_DENSE, grad_indices=None):
    return DgpSdf(ptr, int(rows), int(cols), int(batch_stride), int(layout), int(grad_mode), grad_indices)

  @staticmethod
  def covs_arg(qc_mode=DGP_QC_STATIC, qc_inv=None, obs_w=None, eps=None):
    return DgpCovs(int(qc_mode), qc_inv, obs_w, eps)

  def gn_step(self, batch, th, start, goal, sdf, covs, dtheta, err=None, err_ext=None, info=None, stream=None):
    self.api.check(self.api.gn_step(self.handle, batch, th, start, goal, C.byref(sdf), C.byref(covs) if covs is not None else None,
                                    dtheta, err, err_ext, info, stream))

  def gn_solve(self, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters=None, err_hist=None,
               errext_hist=None, err_final=None, info=None, stream=None):
    self.api.check(self.api.gn_solve(self.handle, batch, th_init, start, goal, C.byref(sdf),
                                     C.byref(covs) if covs is not None else None, int(max_iters), float(tol_delta), th_out, iters,
                                     err_hist, errext_hist, err_final, info, stream))

  def eval_errors(self, batch, th, start, goal, sdf, covs, err=None, err_ext=None, unw_sg=None, unw_gp=None, unw_obs=None,
                  stream=None):
    self.api.check(self.api.eval_errors(self.handle, batch, th, start, goal, C.byref(sdf),
                                        C.byref(covs) if covs is not None else None, err, err_ext, unw_sg, unw_gp, unw_obs, stream))

  def gn_step_backward(self, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_th=None, g_start=None, g_goal=None,
                       g_sdf=None, g_sdf_batch_stride=0, g_qc_inv=None, g_obs_w=None, g_eps=None, stream=None, g_sdf_copies=1):
    self.api.check(self.api.gn_step_backward(self.handle, batch, th, start, goal, C.byref(sdf),
                                             C.byref(covs) if covs is not None else None, dtheta, g_dtheta, g_err_ext, g_th, g_start,
                                             g_goal, g_sdf, int(g_sdf_batch_stride), int(g_sdf_copies), g_qc_inv, g_obs_w, g_eps, stream))

  def gn_solve_traced(self, batch, th_init, start, goal, sdf, covs, max_iters, tol_delta, th_out, iters, err_hist=None, errext_hist=None,
                      err_final=None, info=None, th_hist=None, stream=None):
    self.api.check(self.api.gn_solve_traced(self.handle, batch, th_init, start, goal, C.byref(sdf), C.byref(covs) if covs is not None else None,
                                            int(max_iters), float(tol_delta), th_out, iters, err_hist, errext_hist, err_final, info, th_hist, stream))

  def gn_solve_backward(self, batch, start, goal, sdf, max_iters, th_hist, th_out, iters, g_th_out, g_th_init=None, g_start=None, g_goal=None,
                        g_sdf=None, g_sdf_batch_stride=0, stream=None, g_sdf_copies=1):
    self.api.check(self.api.gn_solve_backward(self.handle, batch, start, goal, C.byref(sdf), int(max_iters), th_hist, th_out, iters, g_th_out,
                                              g_th_init, g_start, g_goal, g_sdf, int(g_sdf_batch_stride), int(g_sdf_copies), stream))

  def gn_step_errors(self, batch, th, start, goal, sdf, covs, dtheta, err=None, err_ext=None, info=None, unw_sg=None, unw_gp=None, unw_obs=None,
                     stream=None):
    self.api.check(self.api.gn_step_errors(self.handle, batch, th, start, goal, C.byref(sdf), C.byref(covs) if covs is not None else None,
                                           dtheta, err, err_ext, info, unw_sg, unw_gp, unw_obs, stream))

  def gn_step_errors_backward(self, batch, th, start, goal, sdf, covs, dtheta, g_dtheta, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th=None,
                              g_start=None, g_goal=None, g_sdf=None, g_sdf_batch_stride=0, g_qc_inv=None, g_obs_w=None, g_eps=None, workspace=None,
                              stream=None, g_sdf_copies=1):
    self.api.check(self.api.gn_step_errors_backward(self.handle, batch, th, start, goal, C.byref(sdf), C.byref(covs) if covs is not None else None,
                                                    dtheta, g_dtheta, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, g_th, g_start, g_goal, g_sdf,
                                                    int(g_sdf_batch_stride), int(g_sdf_copies), g_qc_inv, g_obs_w, g_eps, workspace, stream))

  def eval_errors_backward(self, batch, th, start, goal, sdf, covs, g_err_ext=None, g_unw_sg=None, g_unw_gp=None, g_unw_obs=None,
                           g_th=None, g_start=None, g_goal=None, g_sdf=None, g_sdf_batch_stride=0, g_eps=None, stream=None, g_sdf_copies=1):
    self.api.check(self.api.eval_errors_backward(self.handle, batch, th, start, goal, C.byref(sdf),
                                                 C.byref(covs) if covs is not None else None, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs,
                                                 g_th, g_start, g_goal, g_sdf, int(g_sdf_batch_stride), int(g_sdf_copies), g_eps, stream))
