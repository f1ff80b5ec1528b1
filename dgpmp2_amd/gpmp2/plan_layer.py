"""PlanLayer -- host-side mirror of the reference's diff_gpmp2.gpmp2.plan_layer.PlanLayer (plan_layer.py:13-99).

Same constructor arguments, same forward() signature and return tuple, same public helpers; the body of forward() --
factor evaluation, linear-system assembly, solve, the two error evaluations -- is ONE launch of the fused HIP kernel
behind the C-ABI (include/dgpmp2_hip.h: dgp_gn_step), and its autograd backward is ONE launch of
dgp_gn_step_backward.  Nothing of the reference's dense A/b/K/mask machinery exists here.

Differences a caller can observe (all deliberate):
  * tensors must live on a CUDA (ROCm) device; there is no CPU path (the reference's use_cuda=False runs on CPU);
  * float32 and float64 tensors are both accepted (the reference only works in float64); arithmetic inside
    the kernel is float64 either way;
  * the batch size is NOT baked in at construction: any leading dimension works with one PlanLayer;
  * forward() itself is re-entrant (all inputs are arguments; the C-ABI handle is immutable).  Only for the reference's
    error_batch(thb, sdfb)-style helpers does the layer remember the means/covariances of the last forward();
  * a non-SPD system raises RuntimeError (as torch.cholesky does in the reference) only when `check_spd` is True,
    because the check forces a device synchronisation; the per-trajectory flags are always available in `last_info`.
    When no input requires grad (planning loops) `last_info` is ONE buffer per (batch, device, stream), overwritten by the next
    forward() on that stream -- clone() it to keep an iteration's flags; a differentiable step or check_spd gets its own tensor;
  * the autograd node of forward() holds plain references to its inputs (addresses + a version-counter check instead of
    SavedVariables, ~1.5 us per tensor): they are released with the node, not at the end of backward(), and
    torch.autograd.graph.saved_tensors_hooks (save_on_cpu, checkpointing) do not see them -- only `dtheta` is a SavedVariable;
  * a `qc_inv_trajb` tensor that DiffGPMP2Planner.get_covariances built in the 'diag_identity' mode carries its n - 1 scalars as a tag (and its version counter: an
    in-place edit voids it); forward() / forward_with_errors() and their backward then launch the scaled-mask static kernels (DGP_QC_SCALAR) instead of the per-state
    ones.  Same values to rounding, the gradient returned for the tensor is that of its dof x dof blocks either way; a copy of the tensor (no tag) takes the per-state path.
"""
import ctypes
import weakref

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from .. import _capi


def _f(x):
  """python float from a float / 0-d or 1-element tensor / 1-element list."""
  if torch.is_tensor(x):
    return float(x.reshape(-1)[0].item())
  if isinstance(x, (list, tuple)):
    return _f(x[0])
  return float(x)


def _io_code(dtype):
  if dtype == torch.float32: return _capi.DGP_F32
  if dtype == torch.float64: return _capi.DGP_F64
  raise TypeError('dgpmp2_amd supports float32 and float64 tensors, got %s' % dtype)


def solver_config(num_states, dof, io_dtype, total_time_sec=10.0, x_lims=(-5.0, 5.0), y_lims=(-5.0, 5.0), K_s=0.01, K_g=0.01,
                  reg=0.1, sphere_radius=0.4, Q_c_inv=None, cost_sigma=0.01, epsilon_dist=0.4, **kw):
  """DgpConfig with the defaults of the reference's examples/configs/{gpmp2_2d_params,robot_2d,env_2d_params}.yaml."""
  if Q_c_inv is None:
    Q_c_inv = [[1.0 if i == j else 0.0 for j in range(dof)] for i in range(dof)]
  return _capi.make_config(num_states=num_states, dof=dof, io_dtype=_io_code(io_dtype), total_time_sec=total_time_sec,
                           x_lims=x_lims, y_lims=y_lims, K_s=K_s, K_g=K_g, reg=reg, sphere_radius=sphere_radius, Q_c_inv=Q_c_inv,
                           cost_sigma=cost_sigma, epsilon_dist=epsilon_dist, **kw)


_SDF_GRAD_COPIES = 8      # MI355X has 8 XCDs, each with its own L2: one FLOAT64 partial grid per XCD (XCD-local atomics; summed afterwards by dgp_sum_partial_grids).  (Two per XCD -- round 4's fp32 choice -- run the backward kernel no faster with double grids and double the zero fill and the sum: 27.7 vs 27.5 us, profiles/tools/ubench.py --what bwd_sdf16w,bwd_sdf8w --covs perstate)
_ALL_STATIC = (True, True, True)
_NO_COVS = (_capi.DGP_QC_STATIC, None, None, None)       # the four fields of DgpCovs as the trampoline takes them

# current device / current raw stream as plain ints: torch.cuda.current_stream() builds a Stream object (1.2 us), the private getters
# are what it calls underneath (0.1 us each); fall back to the public API where a torch build lacks them
_cur_dev = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t, name):
  if not t.is_cuda:
    raise RuntimeError('dgpmp2_amd: `%s` must be a CUDA/ROCm tensor (got device %s); this build has no CPU path' % (name, t.device))


class _NoGuard(object):
  def __enter__(self): return self
  def __exit__(self, *a): return False


_NO_GUARD = _NoGuard()


def _on_device(index):
  """Context in which the CURRENT device is cuda:`index` (a launch goes to the current device's stream).  The usual case -- the tensors
  already live on the current device -- costs one integer compare instead of torch.cuda.device()'s get / set / restore."""
  return _NO_GUARD if index == _cur_dev() else torch.cuda.device(index)


def _same_device(dev, **named):
  """Every tensor argument must live on the device of `thb` (index `dev`): the kernel receives raw addresses."""
  for name, t in named.items():
    if t is not None and t.get_device() != dev:
      raise RuntimeError('dgpmp2_amd: `%s` is on %s but `thb` is on cuda:%d; all inputs of one call must share a device' % (name, t.device, dev))


def _ptr(t):
  return None if t is None else t.data_ptr()


def _launch(dev, fn, *args):
  """One C-ABI call through the trampoline with cuda:`dev` as the current device; a non-zero status raises DgpError."""
  if dev == _cur_dev():
    rc = fn(*args)
  else:
    with torch.cuda.device(dev):
      rc = fn(*args)
  if rc:
    _capi.get_api().check(rc)


class _GNStep(torch.autograd.Function):
  """dtheta, err, err_ext = GN step; backward through dgp_gn_step_backward (adjoint block-tridiagonal solve)."""

  @staticmethod
  def launch(layer, static, th, start, goal, sdf, qc, ow, eps, own_info=False):
    """The forward launch itself (no autograd bookkeeping): -> dth, err, eex, and what the backward needs (contiguous inputs,
    the marshalled SDF / covariance arguments with the tensors they keep alive)."""
    B = th.shape[0]
    dtype = th.dtype
    solver = layer._solvers.get(dtype) or layer._solver(dtype)
    dev = th.get_device()
    if start.get_device() != dev or goal.get_device() != dev:
      _same_device(dev, startb=start, goalb=goal)
    sd = layer._sdf_args(sdf, dtype, B, dev)
    if static.__class__ is _RawCovs: cv = static.cov_args(B, dtype, dev)      # (qc is the learn module's raw output vector, squared inside the kernel)
    else: cv = _NO_COVS_KEEP if static == _ALL_STATIC else layer._cov_args(qc, ow, eps, dtype, B, dev, static, True)
    thc, stc, goc = th.contiguous(), start.contiguous(), goal.contiguous()
    dth = torch.empty_like(thc)
    proto = layer._err_protos.get((B, dtype, dev))
    if proto is None: proto = layer._err_proto(B, dtype, dev, thc)
    err = torch.empty_like(proto)           # (empty_like of a cached (B,1,1) tensor: 1.3 us; new_empty / torch.empty with a shape: 2.1 us)
    eex = torch.empty_like(proto)
    stream = _raw_stream(dev)
    # SPD flags: a differentiable step (own_info) or a checking layer gets its own tensor -- `last_info` kept from iteration k stays that
    # iteration's; the planning loop without an autograd graph reuses one buffer per (batch, device, stream), see _info_buffer
    info = layer._info_buffer(B, dev, stream, thc)
    if own_info or layer.check_spd: info = torch.empty_like(info)
    _launch(dev, layer._pc.gn_step, solver.h, B, thc.data_ptr(), stc.data_ptr(), goc.data_ptr(), *sd[:7], *cv[:4], dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr(), stream)
    layer.__dict__['last_info'] = info      # (plain attribute: nn.Module.__setattr__ costs microseconds per call)
    if layer.check_spd and bool(info.any()):
      raise RuntimeError('dgpmp2_amd: A^T K A + delta I is not positive definite for %d of %d trajectories '
                         '(the reference raises from torch.cholesky here)' % (int(info.sum()), B))
    return dth, err, eex, (thc, stc, goc), (sd, cv)

  @staticmethod
  def forward(ctx, layer, static, slots, ts, box, *diff):
    """ts = (th, start, goal, sdf, qc, ow, eps), every input; diff = those of them that require grad, ts[i] for i in slots.  Only `diff` are
    tensor ARGUMENTS of the Function (the bookkeeping of Function.apply is paid per tensor argument: ~1 us each, and a TBPTT link
    differentiates one of the seven); err -- never differentiable, plan_layer.py:275 -- leaves through `box` instead of as an output."""
    th, start, goal, sdf, qc, ow, eps = ts
    dth, err, eex, (thc, stc, goc), args = _GNStep.launch(layer, static, th, start, goal, sdf, qc, ow, eps, True)
    box.append(err)
    ctx.layer = layer
    ctx.slots = slots
    ctx.args = args                               # marshalled SDF / covariance arguments (and the converted copies they point into)
    # Inputs: the backward needs their ADDRESSES (in ctx.args) and, for the gradient buffers, their shapes -- not SavedVariables, whose
    # unpacking costs ~1.5 us per tensor.  What save_for_backward would add, the "modified by an inplace operation" check, is done
    # by hand on the version counters.  (dth is an OUTPUT: it must go through save_for_backward, a plain reference would be a cycle.)
    ctx.inputs = (thc, stc, goc, sdf, qc, ow, eps, start, goal)
    ctx.versions = (thc._version, stc._version, goc._version, -1 if sdf is None else sdf._version, -1 if qc is None else qc._version,
                    -1 if ow is None else ow._version, -1 if eps is None else eps._version)
    ctx.save_for_backward(dth)
    ctx.set_materialize_grads(False)              # an unused output arrives as None: no adjoint solve for an err_ext-only loss
    return dth, eex

  @staticmethod
  def backward(ctx, g_dth, g_eex):
    # The backward is a raw kernel: a double backward must raise instead of silently returning zeros.  torch's once_differentiable
    # does that, at ~8 us of wrapper per call; grad mode is only enabled inside backward() under create_graph=True, so the wrapper
    # is only paid there.
    if torch.is_grad_enabled():
      return _GNStep._backward_once(ctx, g_dth, g_eex)
    return _GNStep._backward_impl(ctx, g_dth, g_eex)

  @staticmethod
  def _backward_impl(ctx, g_dth, g_eex):
    layer = ctx.layer
    dth, = ctx.saved_tensors
    th, stc, goc, sdf, qc, ow, eps, start, goal = ctx.inputs
    _check_versions(ctx.inputs, ctx.versions)
    sd, cv = ctx.args
    B, n, d = th.shape
    dtype = th.dtype
    solver = layer._solvers[dtype]
    dev = th.get_device()
    slots = ctx.slots
    nig = ctx.needs_input_grad                    # (layer, static, slots, ts, box, *diff)
    need = [False] * 9                            # indexed like the old fixed signature: [2 + i] <-> ts[i]
    for q, i in enumerate(slots): need[2 + i] = nig[5 + q]
    if g_dth is not None and (g_dth.dtype is not dtype or not g_dth.is_contiguous()): g_dth = g_dth.contiguous().to(dtype)
    if g_eex is not None and (g_eex.dtype is not dtype or not g_eex.is_contiguous()): g_eex = g_eex.contiguous().to(dtype)
    g_th = torch.empty_like(th) if need[2] else None
    g_st = _grad_like(start, stc) if need[3] else None
    g_go = _grad_like(goal, goc) if need[4] else None
    gs = _SdfGrad(layer, th, sdf, sd) if need[5] else _NO_SDF_GRAD
    g_qc = _grad_like(qc, th) if (need[6] and cv[1] is not None) else None
    g_ow = _grad_like(ow, th) if (need[7] and cv[2] is not None) else None
    g_eps = _grad_like(eps, th) if (need[8] and cv[3] is not None) else None
    _launch(dev, layer._pc.gn_step_backward, solver.h, B, th.data_ptr(), stc.data_ptr(), goc.data_ptr(), *gs.sd(sd),
            *cv[:4], dth.data_ptr(), _ptr(g_dth), _ptr(g_eex), _ptr(g_th), _ptr(g_st), _ptr(g_go), gs.ptr, gs.stride, gs.copies,
            _ptr(g_qc), _ptr(g_ow), _ptr(g_eps), _raw_stream(dev))
    g_sdf = gs.finish(sdf)
    grads = (g_th, _grad_out(g_st, start), _grad_out(g_go, goal), g_sdf, _grad_out(g_qc, qc), _grad_out(g_ow, ow), _grad_out(g_eps, eps))
    return (None, None, None, None, None) + tuple(grads[i] for i in slots)


def _expand_base(t):
  """sdfb as autograd should see it.  A shared grid usually arrives as `grid.expand(B, 1, H, W)` of a (1, 1, H, W) tensor: differentiating w.r.t. the
  expanded view makes autograd's ExpandBackward sum B slices of whatever the node returns -- 4096 x 65536 elements, 45 us on MI355X, for a sum the
  backward kernel has already formed.  When `t` is exactly such a view the node takes the view's BASE as its differentiable input (same storage, same
  launch) and returns the (1, 1, H, W) gradient itself.  Anything else (a view of a view, a differently strided base) is returned unchanged and gets B
  equal shares (see _SdfGrad.finish)."""
  if t is None or t.dim() not in (4, 6) or t.shape[0] <= 1 or t.stride(0) != 0 or not t._is_view(): return t
  # the view itself is wanted as a gradient target (retain_grad(), a tensor hook on it): differentiate w.r.t. the view as the reference would -- it gets B equal
  # shares and autograd's ExpandBackward sums them.  (torch.autograd.grad(..., inputs=view) cannot be seen from here: pass the base, or a non-expanded tensor.)
  if t.retains_grad or t._backward_hooks: return t
  b = t._base
  if (b is not None and b.dim() == t.dim() and b.shape[0] == 1 and b.shape[1:] == t.shape[1:] and b.stride()[1:] == t.stride()[1:] and b.data_ptr() == t.data_ptr()
      and b.dtype is t.dtype and b.requires_grad == t.requires_grad):
    if '_dgp_hw' in t.__dict__: b.__dict__.setdefault('_dgp_hw', t.__dict__['_dgp_hw'])      # (a tiled grid's logical size travels with it)
    return b
  return t


def _check_versions(tensors, versions):
  """What autograd does when a SavedVariable is unpacked: a tensor the backward reads must not have been modified in place since the forward."""
  for t, v in zip(tensors, versions):
    if t is not None and t._version != v:
      raise RuntimeError('one of the variables needed for gradient computation has been modified by an inplace operation: a %s tensor '
                         'is at version %d; expected version %d (dgpmp2_amd PlanLayer backward)' % (list(t.shape), t._version, v))


def _grad_like(ref, like):
  """Uninitialised gradient buffer for input `ref`, written by the kernel as a contiguous array of the launch dtype (that of `like`):
  in ref's own shape when ref is contiguous and of that dtype (no reshape / cast afterwards), flat otherwise (see _grad_out)."""
  if ref.dtype is like.dtype and ref.is_contiguous():
    return torch.empty_like(ref)
  return like.new_empty((ref.numel(),))


def _grad_out(g, ref):
  if g is None or g.shape == ref.shape: return g
  g = g.reshape(ref.shape)
  return g if g.dtype is ref.dtype else g.to(ref.dtype)


_GNStep._backward_once = staticmethod(once_differentiable(_GNStep._backward_impl))


class _SdfGrad(object):
  """Destination of dL/d(sdfb) for one backward launch, and its way back into autograd.

  shared grid (sdfb expand()ed / a single grid): `_SDF_GRAD_COPIES` PARTIAL grids, one set per XCD (XCD-local atomics), in FLOAT64 whatever the I/O type
      (DGP_GSDF_DENSE_F64): the sum over thousands of trajectories is formed in double and cast once, so the fp32 result no longer depends on the order in
      which the atomics landed; summed over the copies here.
  per-sample grids (the reference's API shape, sdfb (B,1,H,W) with `sdf_b.requires_grad_(True)`, learning/train_planner.py:267): the reference's gradient is a
      dense (B,1,H,W) tensor -- 1 GiB of zeros around 4 MB of taps at B = 4096, 256 x 256.  `layer.sdf_grad`:
        'dense'  (default) that tensor (zero-filled here, atomics in the kernel; the reference's layout);
        'sparse' a torch.sparse_coo_tensor of sdfb's shape holding the 4 n B taps (DGP_GSDF_SPARSE: no zero fill, no atomics; uncoalesced -- explicit zeros and
                 the duplicates of neighbouring states included; .to_dense() / .coalesce() give the reference's tensor).  AccumulateGrad takes a sparse gradient
                 for a dense leaf (sdfb.grad is then sparse, and sums with dense or sparse gradients from other nodes);
        'auto'   sparse when sdfb is a LEAF tensor, the trajectory has at most 256 states and the dense gradient would be larger than both
                 _SPARSE_MIN_DENSE_BYTES and twice the sparse one; dense otherwise (a non-leaf's producer may not accept sparse gradients).
  A tiled grid tensor (B,1,H/4,W/4,4,4) gets a gradient of its own shape either way: dense tiles, or a sparse tensor with six index rows."""

  __slots__ = ('g', 'idx', 'ptr', 'stride', 'copies', 'mode', 'shared', 'shape', 'pc', 'dev')

  def __init__(self, layer, th, sdf, sd, passes=1, zero_fill=False):
    B, n = th.shape[0], th.shape[1]
    H, W = sd[1], sd[2]
    self.shared = sd[3] == 0
    self.idx = None
    self.pc, self.dev = layer._pc, th.get_device()
    tiled = sd[4] != _capi.DGP_SDF_ROWMAJOR
    # the grid of one gradient: (1, H, W), or the (1, Ht, Wt, 4, 4) tiles of a tiled sdfb (the gradient of a tiled tensor is tiled)
    gshape = (1, (H + 3) // 4, (W + 3) // 4, 4, 4) if tiled else (1, H, W)
    gelems = gshape[1] * gshape[2] * (16 if tiled else 1)
    if self.shared:
      # partial copies: one per XCD; a single grid (device-scope atomics) for a small batch, whose few thousand taps do not contend -- zero-filling and summing
      # eight 256 x 256 double grids costs more than such a backward kernel runs
      self.copies, self.stride = (_SDF_GRAD_COPIES if B * n >= 8192 else 1), 0
      self.mode = _capi.DGP_GSDF_DENSE_F64 if n <= 256 else _capi.DGP_GSDF_DENSE
      self.g = torch.zeros((self.copies,) + gshape, dtype=torch.float64 if self.mode == _capi.DGP_GSDF_DENSE_F64 else th.dtype, device=th.device)
    else:
      self.copies, self.stride = 1, gelems
      want = layer.sdf_grad
      nnz = passes * B * n * 4
      sparse = False
      if want != 'dense' and n <= 256:
        dense_bytes, sparse_bytes = B * gelems * th.element_size(), nnz * ((48 if tiled else 32) + th.element_size())
        sparse = want == 'sparse' or (sdf.is_leaf and dense_bytes > _SPARSE_MIN_DENSE_BYTES and dense_bytes > 2 * sparse_bytes)
      if sparse:
        self.mode = _capi.DGP_GSDF_SPARSE
        mk = torch.zeros if zero_fill else torch.empty
        self.g = mk((nnz,), dtype=th.dtype, device=th.device)
        self.idx = mk((6 if tiled else 4, nnz), dtype=torch.int64, device=th.device)      # (b, 0, y, x), or the six indices of a tiled tensor (b, 0, y/4, x/4, y%4, x%4)
        self.shape = (B,) + tuple(sdf.shape[1:])
      else:
        self.mode = _capi.DGP_GSDF_DENSE
        self.g = th.new_zeros((B,) + gshape)
    self.ptr = self.g.data_ptr()

  def sd(self, sd):
    """The seven DgpSdf fields of the launch: the grid as marshalled for the forward + how its gradient is delivered."""
    return sd[:5] + (self.mode, None if self.idx is None else self.idx.data_ptr())

  def finish(self, sdf):
    """The launch's output -> the gradient in the layout autograd expects for `sdf`."""
    g = self.g
    if self.mode == _capi.DGP_GSDF_SPARSE:
      if g.dtype is not sdf.dtype: g = g.to(sdf.dtype)
      return torch.sparse_coo_tensor(self.idx, g, self.shape, check_invariants=False)      # (channel index 0: only channel 0 is read, obstacle_cost.py:35)
    if self.shared:
      # the copies summed (in double), scaled and cast in one launch; an expand()ed sdfb gets B equal shares: the kernel already accumulated all B
      # trajectories into the one grid, and autograd's expand-backward will sum the B slices of whatever is returned here
      many = sdf.shape[0] != 1
      out = torch.empty((1,) + tuple(g.shape[1:]), dtype=sdf.dtype if sdf.dtype in (torch.float32, torch.float64) else g.dtype, device=g.device)
      _launch(self.dev, self.pc.sum_partial_grids, g.data_ptr(), _io_code(g.dtype), self.copies, g[0].numel(), 1.0 / sdf.shape[0] if many else 1.0,
              out.data_ptr(), _io_code(out.dtype), _raw_stream(self.dev))
      g = out
      if sdf.shape[1] != 1:
        full = g.new_zeros((1,) + tuple(sdf.shape[1:]))
        full[:, 0:1] = g
        g = full
      if many: g = g.expand(sdf.shape)
      return g if g.dtype is sdf.dtype else g.to(sdf.dtype)
    if sdf.shape[1] != 1:       # only channel 0 is read (obstacle_cost.py:35): the other channels get a zero gradient
      full = g.new_zeros((g.shape[0],) + tuple(sdf.shape[1:]))
      full[:, 0:1] = g
      g = full
    return g if g.dtype is sdf.dtype else g.to(sdf.dtype)


class _NoSdfGrad(object):
  ptr, stride, copies = None, 0, 1

  @staticmethod
  def sd(sd): return sd[:7]

  @staticmethod
  def finish(sdf): return None


_NO_SDF_GRAD = _NoSdfGrad()
_SPARSE_MIN_DENSE_BYTES = 16 << 20      # 'auto': below this a zero-filled dense gradient costs a few microseconds and keeps the reference's layout


class _GNStepErrors(torch.autograd.Function):
  """One iteration of the reference's training loop as ONE autograd node (learning/train_planner.py:311-327): dtheta, err, err_ext of the
  step AND the three unweighted errors at th + dtheta -- forward = dgp_gn_step_errors, backward = dgp_gn_step_errors_backward (one
  launch each for the 2-D robot with a row-major grid and up to 128 / 256 states, two stream-ordered launches otherwise, behind one C-ABI call; no th + dtheta tensor,
  no second Function.apply, no second trip through the autograd engine)."""

  @staticmethod
  def launch(layer, static, th, start, goal, sdf, qc, ow, eps, own_info=False):
    B = th.shape[0]
    dtype = th.dtype
    solver = layer._solvers.get(dtype) or layer._solver(dtype)
    dev = th.get_device()
    if start.get_device() != dev or goal.get_device() != dev:
      _same_device(dev, startb=start, goalb=goal)
    sd = layer._sdf_args(sdf, dtype, B, dev)
    if static.__class__ is _RawCovs: cv = static.cov_args(B, dtype, dev)
    else: cv = _NO_COVS_KEEP if static == _ALL_STATIC else layer._cov_args(qc, ow, eps, dtype, B, dev, static, True)
    thc, stc, goc = th.contiguous(), start.contiguous(), goal.contiguous()
    dth = torch.empty_like(thc)
    proto = layer._err_protos.get((B, dtype, dev))
    if proto is None: proto = layer._err_proto(B, dtype, dev, thc)
    err, eex, usg, ugp, uobs = (torch.empty_like(proto) for _ in range(5))
    stream = _raw_stream(dev)
    info = layer._info_buffer(B, dev, stream, thc)
    if own_info or layer.check_spd: info = torch.empty_like(info)
    _launch(dev, layer._pc.gn_step_errors, solver.h, B, thc.data_ptr(), stc.data_ptr(), goc.data_ptr(), *sd[:7], *cv[:4], dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr(), usg.data_ptr(), ugp.data_ptr(),
            uobs.data_ptr(), stream)
    layer.__dict__['last_info'] = info
    if layer.check_spd and bool(info.any()):
      raise RuntimeError('dgpmp2_amd: A^T K A + delta I is not positive definite for %d of %d trajectories '
                         '(the reference raises from torch.cholesky here)' % (int(info.sum()), B))
    return dth, err, eex, usg, ugp, uobs, (thc, stc, goc), (sd, cv)

  @staticmethod
  def forward(ctx, layer, static, slots, ts, box, *diff):
    th, start, goal, sdf, qc, ow, eps = ts
    dth, err, eex, usg, ugp, uobs, (thc, stc, goc), args = _GNStepErrors.launch(layer, static, th, start, goal, sdf, qc, ow, eps, True)
    box.append(err)
    ctx.layer, ctx.slots, ctx.args = layer, slots, args
    ctx.inputs = (thc, stc, goc, sdf, qc, ow, eps, start, goal)
    ctx.versions = (thc._version, stc._version, goc._version, -1 if sdf is None else sdf._version, -1 if qc is None else qc._version,
                    -1 if ow is None else ow._version, -1 if eps is None else eps._version)
    ctx.save_for_backward(dth)
    ctx.set_materialize_grads(False)
    return dth, eex, usg, ugp, uobs

  @staticmethod
  def backward(ctx, g_dth, g_eex, g_usg, g_ugp, g_uobs):
    if torch.is_grad_enabled():
      return _GNStepErrors._backward_once(ctx, g_dth, g_eex, g_usg, g_ugp, g_uobs)
    return _GNStepErrors._backward_impl(ctx, g_dth, g_eex, g_usg, g_ugp, g_uobs)

  @staticmethod
  def _backward_impl(ctx, g_dth, g_eex, g_usg, g_ugp, g_uobs):
    layer = ctx.layer
    dth, = ctx.saved_tensors
    th, stc, goc, sdf, qc, ow, eps, start, goal = ctx.inputs
    _check_versions(ctx.inputs, ctx.versions)
    sd, cv = ctx.args
    B, n, d = th.shape
    dtype = th.dtype
    solver = layer._solvers[dtype]
    dev = th.get_device()
    slots = ctx.slots
    nig = ctx.needs_input_grad                    # (layer, static, slots, ts, box, *diff)
    need = [False] * 9
    for q, i in enumerate(slots): need[2 + i] = nig[5 + q]
    fix = lambda g: g if (g is None or (g.dtype is dtype and g.is_contiguous())) else g.contiguous().to(dtype)
    g_dth, g_eex, g_usg, g_ugp, g_uobs = fix(g_dth), fix(g_eex), fix(g_usg), fix(g_ugp), fix(g_uobs)
    g_th = torch.empty_like(th) if need[2] else None
    g_st = _grad_like(start, stc) if need[3] else None
    g_go = _grad_like(goal, goc) if need[4] else None
    errs = g_usg is not None or g_ugp is not None or g_uobs is not None
    gs = _SdfGrad(layer, th, sdf, sd, passes=2 if errs else 1) if need[5] else _NO_SDF_GRAD      # (with error cotangents the grid is read at th + dtheta and at th)
    g_qc = _grad_like(qc, th) if (need[6] and cv[1] is not None) else None
    g_ow = _grad_like(ow, th) if (need[7] and cv[2] is not None) else None
    g_eps = _grad_like(eps, th) if (need[8] and cv[3] is not None) else None
    # d = 4: the errors' backward runs as a prologue of the step's backward kernel (one launch, hand-over in LDS); the two-launch form of the (x, y, theta)
    # robot and of long trajectories hands dL/d(th + dtheta) over in a workspace
    ws = torch.empty_like(th) if (errs and (d != 4 or n > 256)) else None
    _launch(dev, layer._pc.gn_step_errors_backward, solver.h, B, th.data_ptr(), stc.data_ptr(), goc.data_ptr(), *gs.sd(sd),
            *cv[:4], dth.data_ptr(), _ptr(g_dth), _ptr(g_eex), _ptr(g_usg), _ptr(g_ugp), _ptr(g_uobs), _ptr(g_th), _ptr(g_st),
            _ptr(g_go), gs.ptr, gs.stride, gs.copies, _ptr(g_qc), _ptr(g_ow), _ptr(g_eps), _ptr(ws), _raw_stream(dev))
    g_sdf = gs.finish(sdf)
    grads = (g_th, _grad_out(g_st, start), _grad_out(g_go, goal), g_sdf, _grad_out(g_qc, qc), _grad_out(g_ow, ow), _grad_out(g_eps, eps))
    return (None, None, None, None, None) + tuple(grads[i] for i in slots)


_GNStepErrors._backward_once = staticmethod(once_differentiable(_GNStepErrors._backward_impl))


class _RawCovs(object):
  """The learn module's output vector as the covariance input of a step: `out` (B, 1, W) contiguous, column blocks [0, n_gp) one raw scalar q_k per GP factor
  ('diag_identity': Q_c^-1 = q_k^2 I; n_gp = 0: 'fix_dynamics', the static Q_c_inv), [n_gp, n_gp + n) the raw obstacle weights o_i (weight o_i^2), then --
  learn_eps -- the raw epsilons (eps = e_i^2): exactly what get_covariances slices and squares for a single-link robot (diff_gpmp2_planner.py:247-290).  ONE small
  launch (dgp_square_covariances) forms every tensor the step and its caller need -- the scalars q_k^2 the kernels take (DGP_QC_SCALAR), the blocks q_k^2 I the
  reference returns, o_i^2, e_i^2 -- and one more (dgp_square_covariances_backward) turns the gradients the backward kernel writes into d/d out."""

  __slots__ = ('out', 'n_gp', 'n', 'learn_eps', 'dof', 'scal', 'qc', 'ow', 'eps')

  def __init__(self, out, n_gp, n, learn_eps, dof):
    self.out, self.n_gp, self.n, self.learn_eps, self.dof = out, n_gp, n, learn_eps, dof
    self.scal = self.qc = self.ow = self.eps = None

  @property
  def used(self): return self.n_gp + self.n * (2 if self.learn_eps else 1)

  def square(self, pc, dev):
    """the forward launch: fills scal (B, n-1), qc (B, n-1, dof, dof), ow (B, n, 1, 1), eps (B, n, 1, 1) (those the mode has)"""
    out = self.out
    B, n = out.shape[0], self.n
    if self.n_gp:
      self.scal = out.new_empty((B, self.n_gp))
      self.qc = out.new_empty((B, self.n_gp, self.dof, self.dof))
    self.ow = out.new_empty((B, n, 1, 1))
    if self.learn_eps: self.eps = out.new_empty((B, n, 1, 1))
    _launch(dev, pc.square_covariances, out.data_ptr(), _io_code(out.dtype), B, out.shape[2], self.n_gp, n, int(self.learn_eps), self.dof, _ptr(self.scal), _ptr(self.qc),
            self.ow.data_ptr(), _ptr(self.eps), _raw_stream(dev))

  def cov_args(self, B, dtype, dev):
    return (_capi.DGP_QC_SCALAR if self.n_gp else _capi.DGP_QC_STATIC, _ptr(self.scal), self.ow.data_ptr(), _ptr(self.eps), (self.scal, self.ow, self.eps))


class _GNStepRaw(torch.autograd.Function):
  """_GNStep / _GNStepErrors with the learn module's OUTPUT VECTOR as the covariance input (planner.step() / step_with_errors() in the learned modes
  'diag_identity' and 'fix_dynamics' of a single-link robot): the squares of diff_gpmp2_planner.py:247-290 are taken inside the kernels and the backward kernel
  writes dL/d out -- none of get_covariances' slices, products and their backward (a dozen small torch kernels that cost as much as the solver under HIP-graph
  replay) is left.  Outputs: dtheta, err_ext[, the three unweighted errors at th + dtheta], and the squared tensors step() returns (differentiable: a cotangent
  on them, which the reference's loss does not produce -- its cov_loss is commented out, learning/train_planner.py:115 -- is chained by plain torch ops)."""

  @staticmethod
  def forward(ctx, layer, raw, with_errors, slots, ts, box, *diff):
    th, start, goal, sdf, out = ts
    raw.square(layer._pc, th.get_device())
    if with_errors:
      dth, err, eex, usg, ugp, uobs, (thc, stc, goc), args = _GNStepErrors.launch(layer, raw, th, start, goal, sdf, out, None, None, True)
    else:
      dth, err, eex, (thc, stc, goc), args = _GNStep.launch(layer, raw, th, start, goal, sdf, out, None, None, True)
      usg = ugp = uobs = None
    # (aliases as outputs: the node keeps raw.scal / ow / eps for its backward launch, and an OUTPUT held by its own node would be a reference cycle)
    qc, ow, eps = (None if t is None else t.view(t.shape) for t in (raw.qc, raw.ow, raw.eps))
    box.append((err,))
    ctx.layer, ctx.slots, ctx.args, ctx.raw, ctx.with_errors = layer, slots, args, raw, with_errors
    ctx.inputs = (thc, stc, goc, sdf, out, start, goal)
    ctx.versions = (thc._version, stc._version, goc._version, -1 if sdf is None else sdf._version, out._version)
    ctx.save_for_backward(dth)
    ctx.set_materialize_grads(False)
    res = (dth, eex) + ((usg, ugp, uobs) if with_errors else ()) + tuple(t for t in (qc, ow, eps) if t is not None)
    return res

  @staticmethod
  def backward(ctx, *cots):
    if torch.is_grad_enabled():
      return _GNStepRaw._backward_once(ctx, *cots)
    return _GNStepRaw._backward_impl(ctx, *cots)

  @staticmethod
  def _backward_impl(ctx, *cots):
    layer, raw = ctx.layer, ctx.raw
    dth, = ctx.saved_tensors
    th, stc, goc, sdf, out, start, goal = ctx.inputs
    _check_versions(ctx.inputs[:5], ctx.versions)
    sd, cv = ctx.args
    B, n, d = th.shape
    dtype = th.dtype
    solver = layer._solvers[dtype]
    dev = th.get_device()
    slots = ctx.slots
    nig = ctx.needs_input_grad                    # (layer, raw, with_errors, slots, ts, box, *diff)
    need = [False] * 5
    for q, i in enumerate(slots): need[i] = nig[6 + q]
    fix = lambda g: g if (g is None or (g.dtype is dtype and g.is_contiguous())) else g.contiguous().to(dtype)
    k = 5 if ctx.with_errors else 2
    g_dth, g_eex = fix(cots[0]), fix(cots[1])
    g_usg, g_ugp, g_uobs = (fix(cots[2]), fix(cots[3]), fix(cots[4])) if ctx.with_errors else (None, None, None)
    g_sq = list(cots[k:])                         # cotangents of the squared tensors (qc [if n_gp], ow, eps [if learn_eps]): normally all None
    g_th = torch.empty_like(th) if need[0] else None
    g_st = _grad_like(start, stc) if need[1] else None
    g_go = _grad_like(goal, goc) if need[2] else None
    errs = g_usg is not None or g_ugp is not None or g_uobs is not None
    gs = _SdfGrad(layer, th, sdf, sd, passes=2 if errs else 1) if need[3] else _NO_SDF_GRAD
    g_out = gq = gw = ge = None
    if need[4]:
      # the backward kernel writes the gradients of the squared tensors (the blocks' gradient under DGP_QC_SCALAR); one more small launch turns them into d/d out
      if raw.n_gp: gq = torch.empty_like(raw.qc)
      gw = torch.empty_like(raw.ow)
      if raw.learn_eps: ge = torch.empty_like(raw.eps)
    ws = torch.empty_like(th) if (errs and (d != 4 or n > 256)) else None
    if ctx.with_errors:
      _launch(dev, layer._pc.gn_step_errors_backward, solver.h, B, th.data_ptr(), stc.data_ptr(), goc.data_ptr(), *gs.sd(sd),
              *cv[:4], dth.data_ptr(), _ptr(g_dth), _ptr(g_eex), _ptr(g_usg), _ptr(g_ugp), _ptr(g_uobs), _ptr(g_th), _ptr(g_st),
              _ptr(g_go), gs.ptr, gs.stride, gs.copies, _ptr(gq), _ptr(gw), _ptr(ge), _ptr(ws), _raw_stream(dev))
    else:
      _launch(dev, layer._pc.gn_step_backward, solver.h, B, th.data_ptr(), stc.data_ptr(), goc.data_ptr(), *gs.sd(sd),
              *cv[:4], dth.data_ptr(), _ptr(g_dth), _ptr(g_eex), _ptr(g_th), _ptr(g_st), _ptr(g_go), gs.ptr, gs.stride, gs.copies,
              _ptr(gq), _ptr(gw), _ptr(ge), _raw_stream(dev))
    if need[4]:
      # someone differentiated the squared tensors themselves (the reference's loss does not): their cotangents join the kernel's gradients
      names = (['qc'] if raw.n_gp else []) + ['ow'] + (['eps'] if raw.learn_eps else [])
      for name, g in zip(names, g_sq):
        if g is None: continue
        if name == 'qc': gq = gq + g.to(dtype)
        elif name == 'ow': gw = gw + g.reshape(gw.shape).to(dtype)
        else: ge = ge + g.reshape(ge.shape).to(dtype)
      g_out = torch.empty_like(out)
      _launch(dev, layer._pc.square_covariances_backward, out.data_ptr(), _io_code(dtype), B, out.shape[2], raw.n_gp, raw.n, int(raw.learn_eps), raw.dof,
              _ptr(gq), _ptr(gw), _ptr(ge), g_out.data_ptr(), _raw_stream(dev))
    grads = (g_th, _grad_out(g_st, start), _grad_out(g_go, goal), gs.finish(sdf), g_out)
    return (None, None, None, None, None, None) + tuple(grads[i] for i in slots)


_GNStepRaw._backward_once = staticmethod(once_differentiable(_GNStepRaw._backward_impl))


class _GNSolve(torch.autograd.Function):
  """The whole Gauss-Newton loop of DiffGPMP2Planner.forward as ONE autograd node: forward = dgp_gn_solve_traced (the fused loop, keeping the
  fp64 history of the trajectory), backward = dgp_gn_solve_backward (one launch that walks the iterations backwards per trajectory).  The
  reference keeps the graph across its python loop (diff_gpmp2_planner.py:122-156); K iterations there are K x (dense solve + 2 error
  evaluations) nodes, here two launches whatever K is."""

  @staticmethod
  def forward(ctx, layer, max_iters, tol_delta, slots, ts, box, *diff):
    """ts = (th_init, start, goal, sdf); box receives (iters, err_hist, errext_hist, err_final, info) -- none of them differentiable
    (forward() returns the errors as python floats, :138-141,162-164)."""
    th, start, goal, sdf = ts
    B, n, d = th.shape
    dtype = th.dtype
    solver = layer._solvers.get(dtype) or layer._solver(dtype)
    dev = th.get_device()
    sd = layer._sdf_args(sdf, dtype, B, dev)
    thc, stc, goc = th.contiguous(), start.contiguous(), goal.contiguous()
    th_out = torch.empty_like(thc)
    m = max_iters
    buf = torch.full((B * (2 * m + 3),), float('nan'), dtype=dtype, device=th.device)      # err history | err_ext history | final error | iteration counts | SPD flags: one copy to the host
    eh, eeh, ef = buf[:B * m], buf[B * m:2 * B * m], buf[2 * B * m:2 * B * m + B]
    iters = buf[2 * B * m + B:].view(torch.int32)[:B]
    info = buf[2 * B * m + 2 * B:].view(torch.int32)[:B]
    hist = torch.empty((m, B, n, d), dtype=torch.float64, device=th.device)      # th_k, fp64 whatever the I/O type (rows past iters[b] stay unwritten and unread)
    _launch(dev, layer._pc.gn_solve_traced, solver.h, B, thc.data_ptr(), stc.data_ptr(), goc.data_ptr(), *sd[:7], *_NO_COVS,
            max_iters, tol_delta, th_out.data_ptr(), iters.data_ptr(), eh.data_ptr(), eeh.data_ptr(), ef.data_ptr(), info.data_ptr(), hist.data_ptr(),
            _raw_stream(dev))
    box.append((buf, info))
    ctx.layer, ctx.slots, ctx.max_iters = layer, slots, max_iters
    ctx.args = sd
    ctx.inputs = (stc, goc, sdf, start, goal, thc)
    ctx.versions = (stc._version, goc._version, -1 if sdf is None else sdf._version)
    ctx.hist, ctx.iters = hist, iters               # (work buffers of this node, never handed out: plain references)
    ctx.save_for_backward(th_out)
    return th_out

  @staticmethod
  @once_differentiable
  def backward(ctx, g_out):
    layer = ctx.layer
    th_out, = ctx.saved_tensors
    stc, goc, sdf, start, goal, thc = ctx.inputs
    _check_versions(ctx.inputs[:3], ctx.versions)
    sd = ctx.args
    B, n, d = th_out.shape
    dtype = th_out.dtype
    solver = layer._solvers[dtype]
    dev = th_out.get_device()
    nig = ctx.needs_input_grad                    # (layer, max_iters, tol_delta, slots, ts, box, *diff)
    need = [False] * 4
    for q, i in enumerate(ctx.slots): need[i] = nig[6 + q]
    if g_out.dtype is not dtype or not g_out.is_contiguous(): g_out = g_out.contiguous().to(dtype)
    g_th = torch.empty_like(th_out) if need[0] else None
    g_st = _grad_like(start, stc) if need[1] else None
    g_go = _grad_like(goal, goc) if need[2] else None
    gs = _SdfGrad(layer, th_out, sdf, sd, passes=ctx.max_iters, zero_fill=True) if need[3] else _NO_SDF_GRAD      # (a trajectory writes only the passes it ran)
    _launch(dev, layer._pc.gn_solve_backward, solver.h, B, stc.data_ptr(), goc.data_ptr(), *gs.sd(sd), ctx.max_iters,
            ctx.hist.data_ptr(), th_out.data_ptr(), ctx.iters.data_ptr(), g_out.data_ptr(), _ptr(g_th), _ptr(g_st), _ptr(g_go), gs.ptr, gs.stride,
            gs.copies, _raw_stream(dev))
    g_sdf = gs.finish(sdf)
    grads = (g_th, _grad_out(g_st, start), _grad_out(g_go, goal), g_sdf)
    return (None, None, None, None, None, None) + tuple(grads[i] for i in ctx.slots)


_NO_COVS_KEEP = _NO_COVS + ((),)


class _EvalErrors(torch.autograd.Function):
  """err_ext, start_goal_error, gp_error, obs_error at a trajectory = one launch of dgp_eval_errors; backward = one launch of
  dgp_eval_errors_backward.  These are the quantities the reference's training loss differentiates besides dtheta
  (learning/train_planner.py:327,342 -> one_step_loss :75-120): plain torch ops there (plan_layer.py:310-345, :374-388), hence
  differentiable w.r.t. thb, sdfb, the start / goal means and the current eps (all remembered WITH their graphs by the last
  forward(), plan_layer.py:88-94).  None of them depends on qc_inv / obscov_inv (fixed or unit weights)."""

  @staticmethod
  def forward(ctx, layer, slots, ts, *diff):
    """ts = (th, start, goal, sdf, eps); diff = ts[i] for i in slots, the inputs that require grad (see _GNStep.forward)."""
    th, start, goal, sdf, eps = ts
    eps_arg = None if (eps is None or '_dgp_static' in eps.__dict__) else eps
    o, thc, stc, goc, sd, cv = layer._eval_launch(th, sdf, start, goal, None, None, eps_arg)
    ctx.layer = layer
    ctx.slots = slots
    ctx.args = (sd, cv)
    ctx.inputs = (thc, stc, goc, sdf, eps_arg, start, goal)      # (addresses in ctx.args; version counters checked by hand, see _GNStep.forward)
    ctx.versions = (thc._version, stc._version, goc._version, -1 if sdf is None else sdf._version, -1 if eps_arg is None else eps_arg._version)
    ctx.set_materialize_grads(False)
    return o[1], o[2], o[3], o[4]            # (the three that read the grid are None without one)

  @staticmethod
  def backward(ctx, g_eex, g_usg, g_ugp, g_uobs):
    if torch.is_grad_enabled():                       # create_graph=True: see _GNStep.backward
      return _EvalErrors._backward_once(ctx, g_eex, g_usg, g_ugp, g_uobs)
    return _EvalErrors._backward_impl(ctx, g_eex, g_usg, g_ugp, g_uobs)

  @staticmethod
  def _backward_impl(ctx, g_eex, g_usg, g_ugp, g_uobs):
    layer = ctx.layer
    th, stc, goc, sdf, eps, start, goal = ctx.inputs
    _check_versions(ctx.inputs, ctx.versions)
    sd, cv = ctx.args
    B, n, d = th.shape
    dtype = th.dtype
    solver = layer._solvers[dtype]
    dev = th.get_device()
    nig = ctx.needs_input_grad                        # (layer, slots, ts, *diff)
    need = [False] * 6                                # indexed like the old fixed signature: [1 + i] <-> ts[i]
    for q, i in enumerate(ctx.slots): need[1 + i] = nig[3 + q]
    cot = [None if g is None else (g if (g.dtype is dtype and g.is_contiguous()) else g.contiguous().to(dtype)) for g in (g_eex, g_usg, g_ugp, g_uobs)]
    g_th = torch.empty_like(th) if need[1] else None
    g_st = _grad_like(start, stc) if need[2] else None
    g_go = _grad_like(goal, goc) if need[3] else None
    gs = _SdfGrad(layer, th, sdf, sd) if (sdf is not None and need[4]) else _NO_SDF_GRAD
    g_eps = _grad_like(eps, th) if (need[5] and eps is not None) else None
    _launch(dev, layer._pc.eval_errors_backward, solver.h, B, th.data_ptr(), stc.data_ptr(), goc.data_ptr(), *gs.sd(sd),
            *cv[:4], _ptr(cot[0]), _ptr(cot[1]), _ptr(cot[2]), _ptr(cot[3]), _ptr(g_th), _ptr(g_st), _ptr(g_go), gs.ptr,
            gs.stride, gs.copies, _ptr(g_eps), _raw_stream(dev))
    g_sdf = gs.finish(sdf)
    grads = (g_th, _grad_out(g_st, start), _grad_out(g_go, goal), g_sdf, _grad_out(g_eps, eps))
    return (None, None, None) + tuple(grads[i] for i in ctx.slots)


_EvalErrors._backward_once = staticmethod(once_differentiable(_EvalErrors._backward_impl))


class _AutoTile(torch.autograd.Function):
  """PlanLayer.auto_tile: a per-sample ROW-MAJOR sdfb (B,1,H,W) stands in the graph, the kernels read its 4 x 4-tiled copy `tiles` (made once per batch of grids,
  PlanLayer._auto_tiled).  forward: the cached tiles, no work; backward: the gradient of the tiled tensor -> the gradient of the row-major one -- a dense tiled gradient
  is un-tiled (views + one slice: AccumulateGrad's copy is the only pass over it), a sparse one (six index rows b,0,y/4,x/4,y%4,x%4) gets the four row-major index
  rows (b,0,y,x); taps of padding cells do not exist (the lookup clamps to H-1 / W-1)."""

  @staticmethod
  def forward(ctx, sdfb, tiles):
    ctx.hw = (int(sdfb.shape[-2]), int(sdfb.shape[-1]))
    ctx.shape = tuple(sdfb.shape)
    return tiles.detach()

  @staticmethod
  @once_differentiable
  def backward(ctx, g):
    H, W = ctx.hw
    if g.is_sparse:
      i = g._indices()
      idx = torch.stack((i[0], i[1], i[2] * 4 + i[4], i[3] * 4 + i[5]))
      return torch.sparse_coo_tensor(idx, g._values(), ctx.shape, check_invariants=False), None
    with torch._C.DisableTorchFunctionSubclass():
      B, C_, Ht, Wt = g.shape[:4]
      return g.permute(0, 1, 2, 4, 3, 5).reshape(B, C_, Ht * 4, Wt * 4)[:, :, :H, :W], None


class PlanLayer(nn.Module):
  """See module docstring.  Constructor mirrors plan_layer.py:14."""

  def __init__(self, gp_params, obs_params, planner_params, optim_params, env_params, robot_model, learn_params=None, batch_size=1,
               use_cuda=False, check_spd=False):
    super(PlanLayer, self).__init__()
    self.use_cuda = torch.cuda.is_available() if use_cuda else False
    self.device = torch.device('cuda') if torch.cuda.is_available() else torch.device('cpu')
    self.gp_params, self.obs_params, self.planner_params = gp_params, obs_params, planner_params
    self.optim_params, self.env_params, self.robot_model, self.learn_params = optim_params, env_params, robot_model, learn_params
    self.batch_size = batch_size
    self.dof = int(planner_params['dof'])
    self.state_dim = int(planner_params['state_dim'])
    if self.state_dim != 2 * self.dof:
      raise ValueError('state_dim must be 2*dof (position + velocity)')
    self.total_time_sec = planner_params['total_time_sec']
    self.total_time_step = int(planner_params['total_time_step'])
    self.num_traj_states = self.total_time_step + 1
    self.dt = self.total_time_sec * 1.0 / self.total_time_step * 1.0
    self.non_holonomic = bool(planner_params['non_holonomic']) if 'non_holonomic' in planner_params else False
    self.use_vel_limits = bool(planner_params['use_vel_limits']) if 'use_vel_limits' in planner_params else False
    self.num_gp_factors = self.num_traj_states - 1
    self.num_prior_factors = 2
    self.num_obs_factors = self.num_traj_states
    self.nlinks = robot_model.nlinks
    self.M = self.state_dim * (self.num_gp_factors + self.num_prior_factors) + self.num_obs_factors * self.nlinks
    if self.non_holonomic: self.M = self.M + self.num_traj_states
    if self.use_vel_limits: self.M = self.M + self.dof * self.num_traj_states
    self.N = self.state_dim * self.num_traj_states
    self.dynamics_mode = learn_params['dgpmp2']['dynamics_mode'] if learn_params is not None else None
    self._q_full = learn_params is not None and self.dynamics_mode == 'q_full'        # plan_layer.py:90
    self.check_spd = check_spd
    # how the gradient of PER-SAMPLE grids is returned: 'dense' (default: the reference's layout, a dense tensor of sdfb's shape) | 'sparse' | 'auto' (sparse for
    # large leaf grids) -- see _SdfGrad; sparse gradients are an opt-in because optimisers, clip_grad_norm_ and DDP buckets do not take them
    self.sdf_grad = 'dense'
    self.sdf_hw = None                 # logical (H, W) of tiled grids that arrive as PLAIN (B,1,Ht,Wt,4,4) tensors (a utils.sdf_utils.TiledSdf carries its own)
    # True: a per-sample row-major sdfb (B,1,H,W) -- the reference's API shape -- is tiled ONCE per batch of grids (per tensor object / storage / version) and every
    # launch that follows reads the tiles: half the grid traffic per step.  Pays when a batch of grids is used for more launches than the
    # tiling pass costs (one read + one write of the grids: ~0.4 ms per GiB; ~3 us saved per launch at B = 4096) -- the TBPTT window of learning/train_planner.py
    # re-uses one batch for every iteration of a window.  Off by default: a single step() per batch of grids would lose.  Grids that are already tiled, shared
    # (expand()ed) grids and trajectories of more than 128 states are left alone.
    self.auto_tile = False
    self._tile_cache = None            # (weakref to the row-major tensor, data_ptr, version, shape, dtype) -> its TiledSdf
    self.last_info = None
    self._last = None
    self._solvers = {}                 # torch dtype -> _capi.Solver
    self._sdf_cache = None             # marshalled arguments of the last SDF tensor seen (see _sdf_args)
    self._err_protos = {}              # (B, dtype, device index) -> (B,1,1) prototype for torch.empty_like
    self._info_bufs = {}               # (B, device index, raw stream) -> the int32 flag buffer launches on that stream write
    q = gp_params['Q_c_inv']
    self._qc_rows = [[float(v) for v in row] for row in (q.tolist() if torch.is_tensor(q) else q)]
    # DGP_QC_SCALAR scales the configuration's Q_c_inv: q_k^2 I (diag_identity) is that only for Q_c_inv = I
    self._scalar_qc = all(self._qc_rows[i][j] == (1.0 if i == j else 0.0) for i in range(len(self._qc_rows)) for j in range(len(self._qc_rows)))
    self._pc = _capi.get_pycall()      # fails loudly if the library / trampoline have not been built
    self._solver(torch.float64)        # validates the configuration now

  # -- C-ABI plumbing -------------------------------------------------------------------------------
  def _solver(self, dtype):
    s = self._solvers.get(dtype)
    if s is None:
      code = _io_code(dtype)
      gp, ob = self.gp_params, self.obs_params
      cfg = _capi.make_config(
          num_states=self.num_traj_states, dof=self.dof, io_dtype=code, total_time_sec=_f(self.total_time_sec),
          x_lims=[_f(v) for v in self.env_params['x_lims']], y_lims=[_f(v) for v in self.env_params['y_lims']],
          K_s=_f(gp['K_s']), K_g=_f(gp['K_g']), reg=_f(self.optim_params['reg']), sphere_radius=_f(self.robot_model.get_sphere_radii()),
          Q_c_inv=self._qc_rows, cost_sigma=_f(ob['cost_sigma']), epsilon_dist=_f(ob['epsilon_dist']),
          non_holonomic=self.non_holonomic, use_vel_limits=self.use_vel_limits,
          K_d=_f(gp['K_d']) if self.non_holonomic else 0.0, K_v=_f(gp['K_v']) if self.use_vel_limits else 0.0,
          v_x=_f(gp['v_x']) if self.use_vel_limits else 0.0, v_y=_f(gp['v_y']) if self.use_vel_limits else 0.0, nlinks=self.nlinks)
      s = _capi.Solver(cfg)
      assert s.M == self.M
      self._solvers[dtype] = s
    return s

  def _err_proto(self, B, dtype, dev, like):
    """A (B,1,1) tensor whose only purpose is to be the argument of torch.empty_like (the cheapest way to allocate err / err_ext)."""
    if len(self._err_protos) > 64: self._err_protos.clear()
    t = self._err_protos[(B, dtype, dev)] = torch.empty((B, 1, 1), dtype=dtype, device=like.device)
    return t

  def _info_buffer(self, B, dev, stream, like):
    """The (B) int32 SPD flags a launch writes (`last_info`).  One buffer per (batch, device, stream), reused by the next forward() on
    that stream (stream order makes that safe; clone() `last_info` to keep it across calls)."""
    k = (B, dev, stream)
    t = self._info_bufs.get(k)
    if t is None:
      if len(self._info_bufs) > 64: self._info_bufs.clear()
      t = self._info_bufs[k] = torch.empty(B, dtype=torch.int32, device=like.device)
    return t

  def _sdf_args(self, sdfb, dtype, B, dev):
    """sdfb (B,1,H,W) (only channel 0 is read, obstacle_cost.py:35) -> the seven DgpSdf fields (address, H, W, batch stride in elements, layout, 0, None) + a keep-alive tensor; an
    expand()ed / single grid is passed as shared (stride 0).  None: no grid (dgp_eval_errors without obstacle outputs only).
    A GN loop passes the same grid tensor every iteration, and slicing + checking it costs 5 us -- half a kernel -- so the result for
    the LAST tensor seen is cached, but ONLY in the zero-copy case: the kernel then reads sdfb's own storage, whatever was written to
    it and however (in place, through .data, through an aliasing numpy / DLPack buffer), so the entry cannot go stale; it is keyed on
    the tensor object, its storage address, shape, strides, dtype and the batch.  A grid that needs a converted copy (other dtype than
    thb, non-contiguous) is converted on EVERY call, on the current stream, like any other torch op."""
    if sdfb is None:
      return _NO_SDF
    c = self._sdf_cache
    if (c is not None and c[0]() is sdfb and c[1] == sdfb.data_ptr() and c[2] == sdfb.shape and c[3] == sdfb.stride() and c[4] is sdfb.dtype
        and c[5] == B and c[6] == dev and c[7] is dtype):
      return c[8]
    _require_cuda(sdfb, 'sdfb')
    if sdfb.get_device() != dev: _same_device(dev, sdfb=sdfb)
    if sdfb.dim() == 6 and sdfb.shape[-2:] == (4, 4): return self._tiled_sdf_args(sdfb, dtype, B, dev)      # utils.sdf_utils.tile_sdf / sdf_2d_batch(layout='tiled4')
    if sdfb.dim() != 4: raise ValueError('sdfb must be (B,1,H,W) (or the (B,1,H/4,W/4,4,4) tiles of utils.sdf_utils.tile_sdf)')
    H, W = sdfb.shape[-2], sdfb.shape[-1]
    shared = sdfb.stride(0) == 0 or sdfb.shape[0] == 1
    if not shared and sdfb.shape[0] != B:       # a per-sample grid tensor with fewer grids than trajectories would be read out of bounds
      raise ValueError('sdfb has %d grids for a batch of %d trajectories (expected %d, or 1 / an expand()ed view for a shared grid)'
                       % (sdfb.shape[0], B, B))
    t = sdfb[0:1, 0:1] if shared else sdfb[:, 0:1]
    t = t.detach()
    if t.dtype != dtype or not t.is_contiguous():
      t = t.to(dtype).contiguous()              # an owned copy: never cached (it would not see later writes to sdfb)
      return (t.data_ptr(), int(H), int(W), 0 if shared else int(t.stride(0)), _capi.DGP_SDF_ROWMAJOR, 0, None, t)
    res = (t.data_ptr(), int(H), int(W), 0 if shared else int(t.stride(0)), _capi.DGP_SDF_ROWMAJOR, 0, None, None)      # a view into sdfb: lives as long as sdfb does
    me = weakref.ref(self)

    def _drop(_, me=me):                        # the tensor died: its address may be reused by another tensor object
      s = me()
      if s is not None: s.__dict__['_sdf_cache'] = None
    try:
      self.__dict__['_sdf_cache'] = (weakref.ref(sdfb, _drop), sdfb.data_ptr(), sdfb.shape, sdfb.stride(), sdfb.dtype, B, dev, dtype, res)
    except TypeError:
      self.__dict__['_sdf_cache'] = None
    return res[:7] + (t,)                       # (the caller's copy of the result keeps the view alive for the duration of the call)

  def _auto_tiled(self, sdfb, B):
    """auto_tile: sdfb as the launches should see it -- its cached 4 x 4-tiled copy when it is a per-sample row-major batch of grids, sdfb itself otherwise.  With
    requires_grad the tiles enter the graph through _AutoTile, so gradients come back in sdfb's own (row-major) shape."""
    if (sdfb is None or sdfb.dim() != 4 or sdfb.shape[0] != B or B == 1 or sdfb.stride(0) == 0 or sdfb.shape[1] != 1 or self.num_traj_states > 128
        or sdfb.dtype not in (torch.float32, torch.float64)):
      return sdfb
    c = self._tile_cache
    # (under HIP-graph capture the tiling pass is recorded with everything else and re-run by every replay -- a cached copy would go stale when the caller writes new
    #  grids into the captured tensor: give a graphed iteration pre-tiled grids instead, PlanningDataset(sdf_layout='tiled4') / tile_sdf)
    capturing = sdfb.is_cuda and torch.cuda.is_current_stream_capturing()
    if capturing or not (c is not None and c[0]() is sdfb and c[1] == sdfb.data_ptr() and c[2] == sdfb._version and c[3] == sdfb.shape and c[4] is sdfb.dtype):
      from ..utils.sdf_utils import tile_sdf
      with torch.no_grad():
        tiles = tile_sdf(sdfb.detach())
      if capturing:
        if sdfb.requires_grad and torch.is_grad_enabled():
          out = _AutoTile.apply(sdfb, tiles)
          out.__dict__.setdefault('_dgp_hw', tiles.__dict__['_dgp_hw'])
          return out
        return tiles
      me = weakref.ref(self)

      def _drop(_, me=me):
        s_ = me()
        if s_ is not None: s_.__dict__['_tile_cache'] = None
      try: c = (weakref.ref(sdfb, _drop), sdfb.data_ptr(), sdfb._version, sdfb.shape, sdfb.dtype, tiles)
      except TypeError: return sdfb
      self.__dict__['_tile_cache'] = c
    tiles = c[5]
    if sdfb.requires_grad and torch.is_grad_enabled():
      out = _AutoTile.apply(sdfb, tiles)
      if '_dgp_hw' not in out.__dict__: out.__dict__['_dgp_hw'] = tiles.__dict__['_dgp_hw']
      return out
    return tiles

  def _tiled_sdf_args(self, sdfb, dtype, B, dev):
    """A grid stored as 4 x 4 tiles, (B | 1, 1, Ht, Wt, 4, 4) contiguous per grid (utils.sdf_utils.tile_sdf): -> the DgpSdf fields with layout DGP_SDF_TILED4.  The
    logical size (H, W) -- it sets the resolution and the clamping of the lookup, and the tile counts do not determine it (130 and 132 both give 33 tiles) -- comes
    from the tensor (a utils.sdf_utils.TiledSdf carries it through .to() / .detach() / indexing / collation) or, for a plain tensor that lost it, from
    `self.sdf_hw`; without either the call is refused rather than guessed.  Not cached: the checks are cheap next to a kernel on per-sample grids."""
    hw = sdfb.__dict__.get('_dgp_hw') or self.sdf_hw
    with torch._C.DisableTorchFunctionSubclass():      # (a TiledSdf: no __torch_function__ round trip per attribute below)
      if sdfb.shape[1] != 1: raise ValueError('tiled sdfb must be (B,1,Ht,Wt,4,4)')
      Ht, Wt = int(sdfb.shape[2]), int(sdfb.shape[3])
      if hw is None:
        raise ValueError('tiled sdfb (B,1,%d,%d,4,4) carries no logical grid size: it is not a utils.sdf_utils.TiledSdf (tile_sdf / sdf_2d_batch(layout="tiled4") return one; '
                         'as_tiled(t, (H, W)) re-declares tiles that went through foreign code) and plan_layer.sdf_hw is not set.  The size cannot be guessed: '
                         '%d x %d tiles hold any grid from %d x %d to %d x %d, and H, W set the resolution and the clamping of the lookup' %
                         (Ht, Wt, Ht, Wt, 4 * Ht - 3, 4 * Wt - 3, 4 * Ht, 4 * Wt))
      H, W = int(hw[0]), int(hw[1])
      if (H + 3) // 4 != Ht or (W + 3) // 4 != Wt: raise ValueError('tiled sdfb of %d x %d tiles does not hold a %d x %d grid' % (Ht, Wt, H, W))
      shared = sdfb.stride(0) == 0 or sdfb.shape[0] == 1
      if not shared and sdfb.shape[0] != B:
        raise ValueError('sdfb has %d grids for a batch of %d trajectories (expected %d, or 1 / an expand()ed view for a shared grid)' % (sdfb.shape[0], B, B))
      t = (sdfb[0:1] if shared else sdfb).detach()
      if t.dtype != dtype or not t.is_contiguous(): t = t.to(dtype).contiguous()
      return (t.data_ptr(), H, W, 0 if shared else Ht * Wt * 16, _capi.DGP_SDF_TILED4, 0, None, t)

  @staticmethod
  def static_flags(qc, ow, eps):
    """(qc, ow, eps) -> which of them stand for the handle's static constants: None, or a tensor that
    DiffGPMP2Planner tagged as the expand()ed view of its own static covariance."""
    return (qc is None or '_dgp_static' in qc.__dict__, ow is None or '_dgp_static' in ow.__dict__, eps is None or '_dgp_static' in eps.__dict__)

  def _cov_args(self, qc, ow, eps, dtype, B, dev, static=(False, False, False), scalar_ok=False):
    """Covariance tensors -> the four DgpCovs fields (qc_mode, the qc_inv / obs_w / eps addresses) + a keep-alive list.  A static entry selects the constants of the
    handle (no per-state tensor is streamed)."""
    if static == _ALL_STATIC:
      return _NO_COVS_KEEP
    n, dof, d = self.num_traj_states, self.dof, self.state_dim
    keep = []

    def prep(t, count, name):
      _require_cuda(t, name)
      if t.get_device() != dev: _same_device(dev, **{name: t})
      if t.shape[0] != B: raise ValueError('%s has batch %d, expected %d' % (name, t.shape[0], B))
      if t.numel() != B * count: raise ValueError('%s has %d elements, expected %d' % (name, t.numel(), B * count))
      if t.dtype is not dtype: t = t.to(dtype)
      t = t.contiguous()
      keep.append(t)
      return t.data_ptr()

    mode, qc_p, ow_p, ep_p = _capi.DGP_QC_STATIC, None, None, None
    if qc is not None and not static[0]:
      # scalar_ok (step() / step_with_errors(), forward and backward launches): a tensor the planner tagged as q_k^2 I (dynamics_mode 'diag_identity') goes down as its
      # n - 1 scalars; the gradient the backward kernel writes is that of the (B,n-1,dof,dof) blocks all the same
      tag = qc.__dict__.get('_dgp_scalar') if (scalar_ok and self._scalar_qc and not self._q_full and n <= 256) else None
      sc = tag[0] if (tag is not None and tag[1] == qc._version) else None
      if sc is not None and sc.shape == (B, n - 1):
        mode = _capi.DGP_QC_SCALAR
        qc_p = prep(sc, n - 1, 'qc_inv_trajb (one scalar per GP factor)')
      else:
        mode = _capi.DGP_QC_QFULL if self._q_full else _capi.DGP_QC_PERSTATE
        qc_p = prep(qc, (n - 1) * (d * d if self._q_full else dof * dof), 'qc_inv_trajb')
    if ow is not None and not static[1]: ow_p = prep(ow, n * self.nlinks, 'obscov_inv_trajb')
    if eps is not None and not static[2]: ep_p = prep(eps, n * self.nlinks, 'eps_trajb')
    return (mode, qc_p, ow_p, ep_p, keep)

  def _check_inputs(self, thb, startb, goalb):
    if not (thb.is_cuda and startb.is_cuda and goalb.is_cuda):
      _require_cuda(thb, 'thb'); _require_cuda(startb, 'startb'); _require_cuda(goalb, 'goalb')
    shp = thb.shape
    if len(shp) != 3 or shp[1] != self.num_traj_states or shp[2] != self.state_dim:
      raise ValueError('thb must be (B,%d,%d), got %s' % (self.num_traj_states, self.state_dim, tuple(thb.shape)))
    nd = shp[0] * self.state_dim
    if startb.numel() != nd or goalb.numel() != nd:
      raise ValueError('startb/goalb must be (B,1,%d)' % self.state_dim)
    dt = thb.dtype
    if startb.dtype is not dt or goalb.dtype is not dt:
      raise TypeError('thb, startb, goalb must share one dtype')

  # -- the reference's public surface -----------------------------------------------------------------
  def forward(self, thb, startb, goalb, imb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb):
    """plan_layer.py:87-99.  -> (dthetab (B,n,d), err (B,1,1) [no grad], err_ext (B,1,1) [grad])."""
    self._check_inputs(thb, startb, goalb)
    if self.auto_tile: sdfb = self._auto_tiled(sdfb, thb.shape[0])
    # like the reference (plan_layer.py:88-94) remember means / covariances for the error_* helpers below
    static = self.static_flags(qc_inv_trajb, obscov_inv_trajb, eps_trajb)
    # (start / goal / eps are kept WITH their graphs, as set_mean / set_eps do: error_ext_batch and the unweighted errors are
    #  differentiable w.r.t. them; qc_inv / obscov_inv only feed error_batch, which runs under no_grad, :275)
    if static == _ALL_STATIC:
      self.__dict__['_last'] = (startb, goalb, None, None, None)
    else:
      det = lambda t, st: None if (t is None or st) else t.detach()
      self.__dict__['_last'] = (startb, goalb, det(qc_inv_trajb, static[0]), det(obscov_inv_trajb, static[1]),
                                None if (eps_trajb is None or static[2]) else eps_trajb)
    if torch.is_grad_enabled():
      if sdfb is not None and sdfb.requires_grad: sdfb = _expand_base(sdfb)
      ts = (thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb)
      slots = tuple([i for i in range(7) if ts[i] is not None and ts[i].requires_grad])
      if slots:
        box = []
        dth, eex = _GNStep.apply(self, static, slots, ts, box, *[ts[i] for i in slots])
        return dth, box[0], eex
    # planning / validation loops: no autograd node, one launch
    return _GNStep.launch(self, static, thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb)[:3]

  def forward_with_errors(self, thb, startb, goalb, imb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb):
    """forward() and, in the same call, what the reference's training loop evaluates right behind it (learning/train_planner.py:313,327):
    the unweighted errors of unweighted_errors_batch at thb + dthetab.  -> (dthetab, err, err_ext, err_sg (B,1), err_gp (B,1,1), err_obs (B,1,1));
    everything but err carries the graph (ONE autograd node, ONE backward call).  Equivalent to
        dth, err, eex = layer(thb, ...); sg, gp, ob = layer.unweighted_errors(thb + dth, sdfb)"""
    self._check_inputs(thb, startb, goalb)
    if self.auto_tile: sdfb = self._auto_tiled(sdfb, thb.shape[0])
    B = thb.shape[0]
    if self.num_traj_states > 256:      # the fused entry points stop at 256 states: the two calls this method stands for (the loop kernels of gn_long.h)
      dth, err, eex = self.forward(thb, startb, goalb, imb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb)
      sg, gp, ob = self.unweighted_errors(thb + dth, sdfb)
      return dth, err, eex, sg, gp, ob
    static = self.static_flags(qc_inv_trajb, obscov_inv_trajb, eps_trajb)
    if static == _ALL_STATIC:
      self.__dict__['_last'] = (startb, goalb, None, None, None)
    else:
      det = lambda t, st: None if (t is None or st) else t.detach()
      self.__dict__['_last'] = (startb, goalb, det(qc_inv_trajb, static[0]), det(obscov_inv_trajb, static[1]),
                                None if (eps_trajb is None or static[2]) else eps_trajb)
    if torch.is_grad_enabled():
      if sdfb is not None and sdfb.requires_grad: sdfb = _expand_base(sdfb)
      ts = (thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb)
      slots = tuple([i for i in range(7) if ts[i] is not None and ts[i].requires_grad])
      if slots:
        box = []
        dth, eex, usg, ugp, uobs = _GNStepErrors.apply(self, static, slots, ts, box, *[ts[i] for i in slots])
        return dth, box[0], eex, usg.reshape(B, 1), ugp, uobs
    dth, err, eex, usg, ugp, uobs = _GNStepErrors.launch(self, static, thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb)[:6]
    return dth, err, eex, usg.reshape(B, 1), ugp, uobs

  def raw_covs(self, out, mode, learn_eps):
    """The learn module's output `out` as a covariance input the kernels square themselves (_RawCovs), or None when this call cannot take that path: single-link
    robot, dynamics_mode 'diag_identity' (with Q_c_inv = I in the configuration: q_k^2 I is then a scalar multiple of it) or 'fix_dynamics', out (B, 1, W) contiguous
    with W at least what the mode consumes, at most 256 states."""
    n = self.num_traj_states
    if self.nlinks != 1 or n > 256 or self._q_full or out.dim() != 3 or out.shape[1] != 1 or not out.is_contiguous(): return None
    if mode == 'diag_identity':
      if not self._scalar_qc: return None
      n_gp = n - 1
    elif mode == 'fix_dynamics': n_gp = 0
    else: return None
    raw = _RawCovs(out, n_gp, n, bool(learn_eps), self.dof)
    if learn_eps and raw.used != out.shape[2]: return None      # (the reference's reshape of the epsilon block would raise: keep its behaviour)
    if raw.used > out.shape[2]: return None
    return raw

  def forward_raw(self, thb, startb, goalb, imb, sdfb, raw, with_errors=False):
    """forward() / forward_with_errors() with the covariances given as the learn module's output vector (`raw` = self.raw_covs(out, mode, learn_eps)).
    -> (dthetab, err, err_ext[, err_sg (B,1), err_gp, err_obs], qc_inv_trajb or None, obscov_inv_trajb, eps_trajb or None): the last three are what
    get_covariances would have built (same values), carrying the graph to `out` like everything else."""
    self._check_inputs(thb, startb, goalb)
    if self.auto_tile: sdfb = self._auto_tiled(sdfb, thb.shape[0])
    out = raw.out
    if out.dtype is not thb.dtype or out.get_device() != thb.get_device() or out.shape[0] != thb.shape[0]:
      raise ValueError('the learn module output must share dtype, device and batch with thb')
    B = thb.shape[0]
    if torch.is_grad_enabled():
      if sdfb is not None and sdfb.requires_grad: sdfb = _expand_base(sdfb)
      ts = (thb, startb, goalb, sdfb, out)
      slots = tuple([i for i in range(5) if ts[i] is not None and ts[i].requires_grad])
      if slots:
        box = []
        res = _GNStepRaw.apply(self, raw, with_errors, slots, ts, box, *[ts[i] for i in slots])
        err = box[0][0]
        k = 5 if with_errors else 2
        sq = list(res[k:])
        qc = sq.pop(0) if raw.n_gp else None
        ow = sq.pop(0)
        eps = sq.pop(0) if raw.learn_eps else None
        self.__dict__['_last'] = (startb, goalb, None if qc is None else qc.detach(), ow.detach(), eps)
        if with_errors: return res[0], err, res[1], res[2].reshape(B, 1), res[3], res[4], qc, ow, eps
        return res[0], err, res[1], qc, ow, eps
    raw.square(self._pc, thb.get_device())
    qc, ow, eps = raw.qc, raw.ow, raw.eps
    self.__dict__['_last'] = (startb, goalb, qc, ow, eps)
    if with_errors:
      dth, err, eex, usg, ugp, uobs = _GNStepErrors.launch(self, raw, thb, startb, goalb, sdfb, out, None, None)[:6]
      return dth, err, eex, usg.reshape(B, 1), ugp, uobs, qc, ow, eps
    dth, err, eex = _GNStep.launch(self, raw, thb, startb, goalb, sdfb, out, None, None)[:3]
    return dth, err, eex, qc, ow, eps

  def _eval_launch(self, thb, sdfb, startb, goalb, qc, ow, eps):
    """One dgp_eval_errors launch -> ([err, err_ext, start_goal_error, gp_error, obs_error] (None where a grid is needed and sdfb is
    None), the contiguous inputs, the marshalled SDF / covariance arguments)."""
    B = thb.shape[0]
    dtype = thb.dtype
    solver = self._solvers.get(dtype) or self._solver(dtype)
    dev = thb.get_device()
    if dev < 0: _require_cuda(thb, 'thb')
    if startb.get_device() != dev or goalb.get_device() != dev: _same_device(dev, startb=startb, goalb=goalb)
    sd = self._sdf_args(sdfb, dtype, B, dev)
    cv = self._cov_args(qc, ow, eps, dtype, B, dev, self.static_flags(qc, ow, eps))
    grid = sdfb is not None
    thc, stc, goc = thb.contiguous(), startb.contiguous(), goalb.contiguous()
    proto = self._err_protos.get((B, dtype, dev))
    if proto is None: proto = self._err_proto(B, dtype, dev, thc)
    outs = [torch.empty_like(proto) if w else None for w in (grid, grid, True, True, grid)]
    _launch(dev, self._pc.eval_errors, solver.h, B, thc.data_ptr(), stc.data_ptr(), goc.data_ptr(), *sd[:7], *cv[:4], _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), _ptr(outs[3]), _ptr(outs[4]), _raw_stream(dev))
    return outs, thc, stc, goc, sd, cv

  def _eval(self, thb, sdfb, startb, goalb, qc, ow, eps):
    """-> [err, err_ext, start_goal_error, gp_error, obs_error]; without a grid (sdfb None) the three that read it are None."""
    return self._eval_launch(thb, sdfb, startb, goalb, qc, ow, eps)[0]

  def errors(self, thb, startb, goalb, sdfb, qc_inv_trajb=None, obscov_inv_trajb=None, eps_trajb=None):
    """(err, err_ext, start_goal_error, gp_error, obs_error), each (B,1,1), in one launch (dgp_eval_errors).  Unlike the
    reference (which reads the means/covariances left behind by the last forward()) everything is an argument."""
    self._check_inputs(thb, startb, goalb)
    if self.auto_tile: sdfb = self._auto_tiled(sdfb, thb.shape[0])
    return tuple(self._eval(thb, sdfb, startb, goalb, qc_inv_trajb, obscov_inv_trajb, eps_trajb))

  def _last_or_raise(self):
    if getattr(self, '_last', None) is None:
      raise RuntimeError('call forward() first: like the reference, the error_* helpers use the start/goal means and the '
                         'covariances set by the last forward() (plan_layer.py:88-94)')
    return self._last

  def error_batch(self, thb, sdfb):
    """plan_layer.py:273-308: normalised factor-graph error at thb, no grad, with the covariances of the last forward()."""
    st, go, qc, ow, eps = self._last_or_raise()
    if self.auto_tile: sdfb = self._auto_tiled(sdfb, thb.shape[0])
    with torch.no_grad():
      return self._eval(thb, sdfb, st, go, qc, ow, eps)[0]

  def _eval_diff(self, thb, sdfb, st, go, eps):
    """(err_ext, start_goal_error, gp_error, obs_error) at thb, each (B,1,1) (None where a grid is needed and sdfb is None), carrying
    the autograd graph the reference's plain torch ops would carry: w.r.t. thb, sdfb, the start / goal means and the current eps."""
    if self.auto_tile: sdfb = self._auto_tiled(sdfb, thb.shape[0])
    if torch.is_grad_enabled():
      if sdfb is not None and sdfb.requires_grad: sdfb = _expand_base(sdfb)
      ts = (thb, st, go, sdfb, eps)
      slots = tuple([i for i in range(5) if ts[i] is not None and ts[i].requires_grad])
      if slots:
        return _EvalErrors.apply(self, slots, ts, *[ts[i] for i in slots])
    o = self._eval(thb, sdfb, st, go, None, None, eps)
    return o[1], o[2], o[3], o[4]

  def error_ext_batch(self, thb, sdfb):
    """plan_layer.py:310-345: same with the FIXED GP / obstacle weights and the current eps.  Differentiable w.r.t. thb, sdfb, the
    start / goal means and the eps tensor of the last forward()."""
    st, go, qc, ow, eps = self._last_or_raise()
    return self._eval_diff(thb, sdfb, st, go, eps)[0]

  def start_goal_error(self, thb):
    """plan_layer.py:384-388 (unweighted; (B,1): the reference's mean(dim=1) drops one of the two singleton dimensions)."""
    st, go, qc, ow, eps = self._last_or_raise()
    return self._eval_diff(thb, None, st, go, None)[1].reshape(thb.shape[0], 1)

  def gp_error(self, thb):
    """plan_layer.py:374-377 (unweighted, mean over the GP factors)."""
    st, go, qc, ow, eps = self._last_or_raise()
    return self._eval_diff(thb, None, st, go, None)[2]

  def obs_error(self, thb, sdfb):
    """plan_layer.py:379-382 (unweighted, mean over states; uses the eps of the last forward())."""
    st, go, qc, ow, eps = self._last_or_raise()
    return self._eval_diff(thb, sdfb, st, go, eps)[3]

  def unweighted_errors(self, thb, sdfb):
    """(start_goal_error (B,1), gp_error (B,1,1), obs_error (B,1,1)) in one launch (and one backward launch)."""
    st, go, qc, ow, eps = self._last_or_raise()
    o = self._eval_diff(thb, sdfb, st, go, eps)
    return o[1].reshape(thb.shape[0], 1), o[2], o[3]


_NO_SDF = (None, 2, 2, 0, 0, 0, None, None)       # the seven fields of DgpSdf + the keep-alive slot
