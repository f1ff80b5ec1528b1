"""Test harness: drives the C-ABI (include/dgpmp2_hip.h) from numpy inputs, through either
  - the HIP library on cuda:0  (backend 'hip', used by the -m gpu parity tests), or
  - tests/emul's CPU wavefront emulator of the same per-lane program (backend 'emul', CPU-only tests).
Both go through dgpmp2_amd._capi, i.e. the same marshalling the product uses.
"""
import ctypes as C
import os
import subprocess
import numpy as np

from dgpmp2_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMUL_DIR = os.path.join(ROOT, 'tests', 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, 'libgn_emul.so')


def build_emulator(force=False):
  src = os.path.join(EMUL_DIR, 'emul_main.cpp')
  deps = [src] + [os.path.join(ROOT, 'dgpmp2_amd', 'csrc', f) for f in ('gn_lane.h', 'gn_woodbury.h', 'gn_backward.h', 'gn_long.h', 'dgp_host.h')] + \
         [os.path.join(ROOT, 'include', 'dgpmp2_hip.h')]
  # DGP_EMUL_CXX / DGP_EMUL_LIB (profiles/tools/r06_emul_sanitize.sh): another compiler and output name -- the sanitizer / poisoned-stack builds of the lane program
  lib = os.environ.get('DGP_EMUL_LIB', EMUL_LIB)
  if force or not os.path.exists(lib) or any(os.path.getmtime(d) > os.path.getmtime(lib) for d in deps):
    extra = os.environ.get('DGP_EMUL_FLAGS', '').split()      # experiment macros of the kernel source (e.g. -DDGP_PCR_LDL=1), tuning only
    subprocess.check_call([os.environ.get('DGP_EMUL_CXX', 'g++'), '-O1', '-std=c++17', '-ffp-contract=off', '-Wno-unknown-pragmas', '-fPIC', '-shared',
                           '-pthread'] + extra + [src, '-o', lib])
  return lib


_emul_api = None


def emul_api():
  global _emul_api
  if _emul_api is None:
    _emul_api = _capi.CApi(build_emulator(), 'emul_')
  return _emul_api


def tile_np(a):
  """(B, 1, H, W) -> (B, 1, ceil(H/4), ceil(W/4), 4, 4): the 4 x 4 tiles of DgpSdf::layout = DGP_SDF_TILED4 (padding cells zero)"""
  a = np.asarray(a)
  B, C_, H, W = a.shape
  Ht, Wt = (H + 3) // 4, (W + 3) // 4
  p = np.zeros((B, C_, Ht * 4, Wt * 4), dtype=a.dtype); p[:, :, :H, :W] = a
  return np.ascontiguousarray(p.reshape(B, C_, Ht, 4, Wt, 4).transpose(0, 1, 2, 4, 3, 5))


def untile_np(t, hw):
  B, C_, Ht, Wt = t.shape[:4]
  return np.ascontiguousarray(t.transpose(0, 1, 2, 4, 3, 5).reshape(B, C_, Ht * 4, Wt * 4)[:, :, :hw[0], :hw[1]])


def config_from_oracle(p, io):
  """oracle.OracleParams -> DgpConfig."""
  return _capi.make_config(num_states=p.n, dof=p.dof, io_dtype=_capi.DGP_F64 if io == 'f64' else _capi.DGP_F32,
                           total_time_sec=p.total_time_sec, x_lims=p.x_lims, y_lims=p.y_lims, K_s=p.K_s, K_g=p.K_g, reg=p.reg,
                           sphere_radius=p.radius, Q_c_inv=p.Q_c_inv, cost_sigma=p.cost_sigma, epsilon_dist=p.epsilon_dist,
                           non_holonomic=p.non_holonomic, use_vel_limits=p.use_vel_limits, K_d=p.K_d, K_v=p.K_v, v_x=p.v_x, v_y=p.v_y,
                           nlinks=p.nlinks)


class Backend(object):
  """numpy in / numpy out driver.  kind: 'emul' (host memory) or 'hip' (cuda:0 through torch)."""

  def __init__(self, kind):
    self.kind = kind
    if kind == 'emul':
      self.api = emul_api()
    else:
      import torch
      self.torch = torch
      self.api = _capi.get_api()
    self._keep = []
    self.misalign = False      # True: hand the C-ABI buffers that start one element past a 16-byte boundary (scalar row access path)
    self.sdf_tiled = False     # True: every grid goes down as 4 x 4 tiles (DgpSdf::layout = DGP_SDF_TILED4) and dense grid gradients come back untiled -- any parity case can be re-run on the tiled layout

  # -- memory ------------------------------------------------------------------------------------
  def _np_dtype(self, io): return np.float64 if io == 'f64' else np.float32

  def to_dev(self, a, io=None, dtype=None):
    if a is None: return None, None
    dt = dtype if dtype is not None else self._np_dtype(io)
    a = np.ascontiguousarray(np.asarray(a), dtype=dt)
    if self.kind == 'emul':
      if self.misalign and a.size:
        buf = np.empty(a.size + 1, dtype=dt); v = buf[1:]; v[:] = a.ravel(); a = v.reshape(a.shape)
        assert a.ctypes.data % 16 != 0
      self._keep.append(a)
      return a, a.ctypes.data
    t = self.torch.from_numpy(a).to('cuda:0')
    if self.misalign and t.numel():
      buf = self.torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device); v = buf[1:]; v.copy_(t.reshape(-1)); t = v.view(t.shape)
      assert t.data_ptr() % 16 != 0
    self._keep.append(t)
    return t, t.data_ptr()

  def empty(self, shape, io=None, dtype=None, fill=None):
    dt = dtype if dtype is not None else self._np_dtype(io)
    a = np.full(shape, np.nan if fill is None and dt not in (np.int32, np.int64) else (fill if fill is not None else -1), dtype=dt)
    return self.to_dev(a, dtype=dt)

  def to_np(self, obj):
    if obj is None: return None
    if self.kind == 'emul': return np.array(obj, dtype=np.float64 if obj.dtype not in (np.int32, np.int64) else obj.dtype)
    self.torch.cuda.synchronize()
    a = obj.cpu().numpy()
    return a.astype(np.float64) if a.dtype not in (np.int32, np.int64) else a

  def stream(self):
    if self.kind == 'emul': return None
    return C.c_void_p(self.torch.cuda.current_stream().cuda_stream)

  # -- argument marshalling -------------------------------------------------------------------------
  def _common(self, p, io, th, start, goal, sdf, qc, ow, eps, q_full, raw=None):
    """raw = (out (B, W), n_gp, learn_eps): the covariances as a learn module's output vector, squared inside the kernels (DGP_COVS_SQUARED, row stride W);
    qc / ow / eps must then be None."""
    self._keep = []
    B = th.shape[0]
    solver = _capi.Solver(config_from_oracle(p, io), api=self.api)
    _, th_p = self.to_dev(th, io)
    _, st_p = self.to_dev(start, io)
    _, go_p = self.to_dev(goal, io)
    if sdf is None:              # dgp_eval_errors without obstacle outputs: no grid
      sdf_arg = solver.sdf_arg(None, 2, 2, 0)
    else:
      sdf = np.asarray(sdf)
      assert sdf.ndim == 4 and sdf.shape[1] == 1
      shared = sdf.shape[0] == 1 and B >= 1
      H, W = sdf.shape[-2], sdf.shape[-1]
      if self.sdf_tiled:
        til = tile_np(sdf)
        _, sdf_p = self.to_dev(til, io)
        sdf_arg = solver.sdf_arg(sdf_p, H, W, 0 if shared else til[0].size, layout=_capi.DGP_SDF_TILED4)
      else:
        _, sdf_p = self.to_dev(sdf, io)
        sdf_arg = solver.sdf_arg(sdf_p, H, W, 0 if shared else H * W)
    mode = _capi.DGP_QC_STATIC if qc is None else (_capi.DGP_QC_QFULL if q_full else _capi.DGP_QC_PERSTATE)
    if qc is not None and np.asarray(qc).ndim == 2: mode = _capi.DGP_QC_SCALAR      # (B, n-1) scalars: Q_c^-1 = s_k Q_c_inv (dgp_gn_step only)
    if raw is not None:
      # dgp_square_covariances: the squares the step takes (scalars, weights, epsilons) from the raw output vector, one launch
      assert qc is None and ow is None and eps is None
      out, n_gp, learn_eps = raw
      out = np.asarray(out)
      W, n = out.shape[1], th.shape[1]
      _, o_p = self.to_dev(out, io)
      s_t, s_p = self.empty((B, n_gp), io) if n_gp else (None, None)
      b_t, b_p = self.empty((B, n_gp, p.dof, p.dof), io) if n_gp else (None, None)
      w_t, w_p = self.empty((B, n), io)
      e_t, e_p = self.empty((B, n), io) if learn_eps else (None, None)
      self.api.check(self.api.square_covariances(o_p, _capi.DGP_F64 if io == 'f64' else _capi.DGP_F32, B, W, n_gp, n, int(learn_eps), p.dof, s_p, b_p, w_p, e_p, self.stream()))
      self._raw = (o_p, W, n_gp, n, int(learn_eps), p.dof, b_t)
      covs = solver.covs_arg(_capi.DGP_QC_SCALAR if n_gp else _capi.DGP_QC_STATIC, s_p, w_p, e_p)
      return solver, B, th_p, st_p, go_p, sdf_arg, covs
    _, qc_p = self.to_dev(qc, io)
    _, ow_p = self.to_dev(ow, io)
    _, eps_p = self.to_dev(eps, io)
    covs = solver.covs_arg(mode, qc_p, ow_p, eps_p)
    return solver, B, th_p, st_p, go_p, sdf_arg, covs

  # -- destination of the grid gradient (DgpSdf::grad_mode) -----------------------------------------------
  def _gsdf(self, sdf, sdf_arg, io, sdf_copies, sdf_grad, B, n, passes=1):
    """sdf_grad: 'dense' (grids of the I/O type), 'f64' (DGP_GSDF_DENSE_F64: double grids whatever the I/O type) or 'sparse' (DGP_GSDF_SPARSE: tap values +
    COO indices; `passes` tap blocks, zero-filled as the chain kernels' caller must).  -> (state for _gsdf_out, g_sdf address, batch stride)"""
    sdf = np.asarray(sdf)
    stride = 0 if sdf.shape[0] == 1 else sdf.shape[-1] * sdf.shape[-2]
    if self.sdf_tiled and sdf_grad == 'sparse':
      # a tiled grid tensor (B,1,Ht,Wt,4,4): six index rows (b, 0, y/4, x/4, y%4, x%4)
      assert sdf_copies == 1 and stride != 0
      H, W = sdf.shape[-2], sdf.shape[-1]
      nnz = passes * B * n * 4
      vals, vals_p = self.empty((nnz,), io, fill=0.0)
      idx, idx_p = self.empty((6, nnz), dtype=np.int64, fill=0)
      sdf_arg.grad_mode = _capi.DGP_GSDF_SPARSE; sdf_arg.grad_indices = idx_p
      return ('sparse_tiled', vals, idx, (sdf.shape[0], 1, (H + 3) // 4, (W + 3) // 4, 4, 4), (H, W)), vals_p, ((H + 3) // 4) * ((W + 3) // 4) * 16
    if self.sdf_tiled:
      # a tiled grid takes a dense gradient in its own layout
      H, W = sdf.shape[-2], sdf.shape[-1]
      tshape = ((sdf_copies if sdf_copies > 1 else sdf.shape[0]), 1, (H + 3) // 4, (W + 3) // 4, 4, 4)
      wide = sdf_grad == 'f64'
      g, g_p = self.empty(tshape, dtype=np.float64, fill=0.0) if wide else self.empty(tshape, io, fill=0.0)
      if wide: sdf_arg.grad_mode = _capi.DGP_GSDF_DENSE_F64
      return ('tiled', g, (H, W)), g_p, (0 if sdf.shape[0] == 1 else tshape[2] * tshape[3] * 16)
    if sdf_grad == 'sparse':
      assert sdf_copies == 1 and stride != 0
      nnz = passes * B * n * 4
      vals, vals_p = self.empty((nnz,), io, fill=0.0)
      idx, idx_p = self.empty((4, nnz), dtype=np.int64, fill=0)
      sdf_arg.grad_mode = _capi.DGP_GSDF_SPARSE; sdf_arg.grad_indices = idx_p
      return ('sparse', vals, idx, sdf.shape), vals_p, stride
    shape = ((sdf_copies,) + sdf.shape[1:]) if sdf_copies > 1 else sdf.shape
    if sdf_grad == 'f64':
      g, g_p = self.empty(shape, dtype=np.float64, fill=0.0)
      sdf_arg.grad_mode = _capi.DGP_GSDF_DENSE_F64
    else:
      g, g_p = self.empty(shape, io, fill=0.0)
    return ('dense', g), g_p, stride

  def _gsdf_out(self, st):
    """-> the gradient as a float64 array: the grid(s) as written (partial copies unsummed), or the sparse taps scattered into a grid of sdfb's shape"""
    if st is None: return None
    if st[0] == 'dense': return self.to_np(st[1])
    if st[0] == 'tiled': return untile_np(self.to_np(st[1]), st[2])
    if st[0] == 'sparse_tiled':
      vals = self.to_np(st[1])
      if self.kind == 'emul': idx = np.array(st[2])
      else:
        self.torch.cuda.synchronize(); idx = st[2].cpu().numpy()
      shape = st[3]
      for r in range(6): assert idx[r].min() >= 0 and idx[r].max() < shape[r], (r, idx[r].min(), idx[r].max(), shape)
      out = np.zeros(shape, dtype=np.float64)
      np.add.at(out, tuple(idx[r] for r in range(6)), vals)
      return untile_np(out, st[4])
    vals = self.to_np(st[1])
    if self.kind == 'emul': idx = np.array(st[2])
    else:
      self.torch.cuda.synchronize(); idx = st[2].cpu().numpy()
    shape = st[3]
    assert idx[0].min() >= 0 and idx[0].max() < shape[0] and (idx[1] == 0).all() and idx[2].min() >= 0 and idx[2].max() < shape[2] and idx[3].min() >= 0 and idx[3].max() < shape[3]
    out = np.zeros(shape, dtype=np.float64)
    np.add.at(out, (idx[0], idx[1], idx[2], idx[3]), vals)
    return out

  # -- entry points ------------------------------------------------------------------------------------
  def step(self, p, th, start, goal, sdf, qc=None, ow=None, eps=None, q_full=False, io='f64', raw=None):
    """-> dtheta (B,n,d), err (B,), err_ext (B,), info (B,)"""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, qc, ow, eps, q_full, raw)
    dth, dth_p = self.empty(th.shape, io)
    err, err_p = self.empty((B,), io)
    eex, eex_p = self.empty((B,), io)
    info, info_p = self.empty((B,), dtype=np.int32)
    solver.gn_step(B, th_p, st_p, go_p, sdf_arg, covs, dth_p, err_p, eex_p, info_p, self.stream())
    return self.to_np(dth), self.to_np(err), self.to_np(eex), self.to_np(info)

  def solve(self, p, th, start, goal, sdf, max_iters, tol_delta, qc=None, ow=None, eps=None, q_full=False, io='f64'):
    """-> th_out, iters, err_hist (B,max_iters; NaN where untouched), errext_hist, err_final, info"""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, qc, ow, eps, q_full)
    tho, tho_p = self.empty(th.shape, io)
    its, its_p = self.empty((B,), dtype=np.int32)
    eh, eh_p = self.empty((B, max_iters), io)
    eeh, eeh_p = self.empty((B, max_iters), io)
    ef, ef_p = self.empty((B,), io)
    info, info_p = self.empty((B,), dtype=np.int32)
    solver.gn_solve(B, th_p, st_p, go_p, sdf_arg, covs, max_iters, tol_delta, tho_p, its_p, eh_p, eeh_p, ef_p, info_p, self.stream())
    return self.to_np(tho), self.to_np(its), self.to_np(eh), self.to_np(eeh), self.to_np(ef), self.to_np(info)

  def eval_errors(self, p, th, start, goal, sdf, qc=None, ow=None, eps=None, q_full=False, io='f64'):
    """-> err, err_ext, unw_sg, unw_gp, unw_obs  (each (B,))"""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, qc, ow, eps, q_full)
    want = [sdf is not None, sdf is not None, True, True, sdf is not None]      # without a grid only unw_sg / unw_gp may be requested
    outs = [self.empty((B,), io) if w else (None, None) for w in want]
    solver.eval_errors(B, th_p, st_p, go_p, sdf_arg, covs, *[o[1] for o in outs], stream=self.stream())
    return tuple(self.to_np(o[0]) for o in outs)

  def _raw_grad_bufs(self, B, io):
    """gradient buffers of the squared tensors for a backward launch in raw mode -> (g_blocks, g_ow, g_eps) (array, address) pairs"""
    o_p, W, n_gp, n, le, dof, _ = self._raw
    return (self.empty((B, n_gp, dof, dof), io) if n_gp else (None, None), self.empty((B, n), io), self.empty((B, n), io) if le else (None, None))

  def _raw_grad_out(self, B, io, bufs):
    """dgp_square_covariances_backward: the gradients of the squared tensors -> d/d out (B, W)"""
    o_p, W, n_gp, n, le, dof, _ = self._raw
    g, g_p = self.empty((B, W), io)
    self.api.check(self.api.square_covariances_backward(o_p, _capi.DGP_F64 if io == 'f64' else _capi.DGP_F32, B, W, n_gp, n, le, dof, bufs[0][1], bufs[1][1], bufs[2][1],
                                                       g_p, self.stream()))
    return self.to_np(g)

  def backward(self, p, th, start, goal, sdf, dtheta, g_dtheta, g_err_ext, qc=None, ow=None, eps=None, q_full=False, io='f64',
               sdf_copies=1, sdf_grad='dense', raw=None):
    """-> dict of gradients: th, start, goal, sdf, qc, ow, eps (numpy fp64); raw: `out` = the gradient of the raw output vector instead of qc / ow / eps"""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, qc, ow, eps, q_full, raw)
    n = th.shape[1]
    _, dth_p = self.to_dev(dtheta, io)
    _, gd_p = self.to_dev(g_dtheta, io)
    _, ge_p = self.to_dev(g_err_ext, io)
    gth, gth_p = self.empty(th.shape, io)
    gst, gst_p = self.empty(np.asarray(start).shape, io)
    ggo, ggo_p = self.empty(np.asarray(goal).shape, io)
    gs, gsdf_p, stride = self._gsdf(sdf, sdf_arg, io, sdf_copies, sdf_grad, B, n)
    qshape = None if qc is None else (np.asarray(qc).shape if np.asarray(qc).ndim != 2 else np.asarray(qc).shape + (p.dof, p.dof))      # DGP_QC_SCALAR: the gradient of the blocks s_k I
    gqc, gqc_p = self.empty(qshape, io) if qc is not None else (None, None)
    gow, gow_p = self.empty((B, n), io) if ow is not None else (None, None)
    gep, gep_p = self.empty((B, n), io) if eps is not None else (None, None)
    bufs = None
    if raw is not None:
      bufs = self._raw_grad_bufs(B, io)
      gqc_p, gow_p, gep_p = bufs[0][1], bufs[1][1], bufs[2][1]
    solver.gn_step_backward(B, th_p, st_p, go_p, sdf_arg, covs, dth_p, gd_p, ge_p, gth_p, gst_p, ggo_p, gsdf_p, stride, gqc_p, gow_p,
                            gep_p, self.stream(), g_sdf_copies=sdf_copies)
    return dict(th=self.to_np(gth), start=self.to_np(gst), goal=self.to_np(ggo), sdf=self._gsdf_out(gs), qc=self.to_np(gqc),
                ow=self.to_np(gow), eps=self.to_np(gep), out=None if bufs is None else self._raw_grad_out(B, io, bufs))

  def eval_backward(self, p, th, start, goal, sdf, g_err_ext=None, g_unw_sg=None, g_unw_gp=None, g_unw_obs=None, eps=None, io='f64', sdf_copies=1,
                    want_sdf=True, sdf_grad='dense'):
    """dgp_eval_errors_backward -> dict of gradients: th, start, goal, sdf, eps (numpy fp64).  sdf None: no grid (sg / gp cotangents only)."""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, None, None, eps, False)
    n = th.shape[1]
    cot = [self.to_dev(None if c is None else np.reshape(c, (B,)), io)[1] for c in (g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs)]
    gth, gth_p = self.empty(th.shape, io)
    gst, gst_p = self.empty(np.asarray(start).shape, io)
    ggo, ggo_p = self.empty(np.asarray(goal).shape, io)
    gs, gsdf_p, stride = None, None, 0
    if sdf is not None and want_sdf:
      gs, gsdf_p, stride = self._gsdf(sdf, sdf_arg, io, sdf_copies, sdf_grad, B, n)
    gep, gep_p = self.empty((B, n), io) if eps is not None else (None, None)
    solver.eval_errors_backward(B, th_p, st_p, go_p, sdf_arg, covs, cot[0], cot[1], cot[2], cot[3], gth_p, gst_p, ggo_p, gsdf_p, stride, gep_p,
                                self.stream(), g_sdf_copies=sdf_copies)
    return dict(th=self.to_np(gth), start=self.to_np(gst), goal=self.to_np(ggo), sdf=self._gsdf_out(gs), eps=self.to_np(gep))

  # -- round 4: the fused-loop backward and the training iteration as single calls ---------------------------
  def solve_traced(self, p, th, start, goal, sdf, max_iters, tol_delta, io='f64'):
    """dgp_gn_solve_traced (static covariances) -> th_out, iters, th_hist (max_iters,B,n,d; NaN where untouched), info"""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, None, None, None, False)
    tho, tho_p = self.empty(th.shape, io)
    its, its_p = self.empty((B,), dtype=np.int32)
    hist, hist_p = self.empty((max_iters,) + tuple(th.shape), dtype=np.float64)
    info, info_p = self.empty((B,), dtype=np.int32)
    solver.gn_solve_traced(B, th_p, st_p, go_p, sdf_arg, None, max_iters, tol_delta, tho_p, its_p, None, None, None, info_p, hist_p, self.stream())
    return self.to_np(tho), self.to_np(its), self.to_np(hist), self.to_np(info)

  def solve_backward(self, p, start, goal, sdf, max_iters, th_hist, th_out, iters, g_th_out, io='f64', sdf_copies=1, want_sdf=True, sdf_grad='dense'):
    """dgp_gn_solve_backward -> dict of gradients: th (w.r.t. th_init), start, goal, sdf"""
    th_out = np.asarray(th_out)
    solver, B, tho_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th_out, start, goal, sdf, None, None, None, False)
    _, hist_p = self.to_dev(th_hist, dtype=np.float64)
    _, its_p = self.to_dev(iters, dtype=np.int32)
    _, g_p = self.to_dev(g_th_out, io)
    gth, gth_p = self.empty(th_out.shape, io)
    gst, gst_p = self.empty(np.asarray(start).shape, io)
    ggo, ggo_p = self.empty(np.asarray(goal).shape, io)
    sdf = np.asarray(sdf)
    gs, gsdf_p = (None, None)
    stride = 0 if sdf.shape[0] == 1 else sdf.shape[-1] * sdf.shape[-2]
    if want_sdf:
      gs, gsdf_p, stride = self._gsdf(sdf, sdf_arg, io, sdf_copies, sdf_grad, B, th_out.shape[1], passes=max_iters)
    solver.gn_solve_backward(B, st_p, go_p, sdf_arg, max_iters, hist_p, tho_p, its_p, g_p, gth_p, gst_p, ggo_p, gsdf_p, stride, self.stream(),
                             g_sdf_copies=sdf_copies)
    return dict(th=self.to_np(gth), start=self.to_np(gst), goal=self.to_np(ggo), sdf=self._gsdf_out(gs))

  def step_errors(self, p, th, start, goal, sdf, qc=None, ow=None, eps=None, q_full=False, io='f64', raw=None):
    """dgp_gn_step_errors -> dtheta, err, err_ext, info, unw_sg, unw_gp, unw_obs (the last three at th + dtheta)"""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, qc, ow, eps, q_full, raw)
    dth, dth_p = self.empty(th.shape, io)
    outs = [self.empty((B,), io) for _ in range(5)]
    info, info_p = self.empty((B,), dtype=np.int32)
    solver.gn_step_errors(B, th_p, st_p, go_p, sdf_arg, covs, dth_p, outs[0][1], outs[1][1], info_p, outs[2][1], outs[3][1], outs[4][1], self.stream())
    return (self.to_np(dth), self.to_np(outs[0][0]), self.to_np(outs[1][0]), self.to_np(info)) + tuple(self.to_np(o[0]) for o in outs[2:])

  def step_errors_backward(self, p, th, start, goal, sdf, dtheta, g_dtheta, g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs, qc=None, ow=None, eps=None,
                           q_full=False, io='f64', sdf_copies=1, sdf_grad='dense', raw=None):
    """dgp_gn_step_errors_backward -> dict of gradients: th, start, goal, sdf, qc, ow, eps (raw: `out`).  sdf_grad 'none': no grid gradient requested"""
    solver, B, th_p, st_p, go_p, sdf_arg, covs = self._common(p, io, th, start, goal, sdf, qc, ow, eps, q_full, raw)
    n = th.shape[1]
    _, dth_p = self.to_dev(dtheta, io)
    _, gd_p = self.to_dev(g_dtheta, io)
    cot = [self.to_dev(None if c is None else np.reshape(c, (B,)), io)[1] for c in (g_err_ext, g_unw_sg, g_unw_gp, g_unw_obs)]
    gth, gth_p = self.empty(th.shape, io)
    gst, gst_p = self.empty(np.asarray(start).shape, io)
    ggo, ggo_p = self.empty(np.asarray(goal).shape, io)
    errs = any(c is not None for c in (g_unw_sg, g_unw_gp, g_unw_obs))
    gs, gsdf_p, stride = (None, None, 0) if sdf_grad == 'none' else self._gsdf(sdf, sdf_arg, io, sdf_copies, sdf_grad, B, n, passes=2 if errs else 1)
    qshape = None if qc is None else (np.asarray(qc).shape if np.asarray(qc).ndim != 2 else np.asarray(qc).shape + (p.dof, p.dof))      # DGP_QC_SCALAR: the gradient of the blocks s_k I
    gqc, gqc_p = self.empty(qshape, io) if qc is not None else (None, None)
    gow, gow_p = self.empty((B, n), io) if ow is not None else (None, None)
    gep, gep_p = self.empty((B, n), io) if eps is not None else (None, None)
    ws, ws_p = self.empty(th.shape, io)
    bufs = None
    if raw is not None:
      bufs = self._raw_grad_bufs(B, io)
      gqc_p, gow_p, gep_p = bufs[0][1], bufs[1][1], bufs[2][1]
    solver.gn_step_errors_backward(B, th_p, st_p, go_p, sdf_arg, covs, dth_p, gd_p, cot[0], cot[1], cot[2], cot[3], gth_p, gst_p, ggo_p, gsdf_p, stride,
                                   gqc_p, gow_p, gep_p, ws_p, self.stream(), g_sdf_copies=sdf_copies)
    return dict(th=self.to_np(gth), start=self.to_np(gst), goal=self.to_np(ggo), sdf=self._gsdf_out(gs), qc=self.to_np(gqc), ow=self.to_np(gow),
                eps=self.to_np(gep), out=None if bufs is None else self._raw_grad_out(B, io, bufs))
