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
"""
import ctypes

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


_ALL_STATIC_COVS = _capi.DgpCovs(_capi.DGP_QC_STATIC, None, None, None)
_SDF_GRAD_COPIES = 16     # MI355X has 8 XCDs, each with its own L2: two partial grids per XCD (XCD-local atomics, summed afterwards)


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_cuda(t, name):
  if not t.is_cuda:
    raise RuntimeError('dgpmp2_amd: `%s` must be a CUDA/ROCm tensor (got device %s); this build has no CPU path' % (name, t.device))


class _NoGuard(object):
  def __enter__(self): return self
  def __exit__(self, *a): return False


_NO_GUARD = _NoGuard()


def _on_device(dev):
  """Context in which the CURRENT device is `dev` (a launch goes to the current device's stream).  The usual case -- the tensors
  already live on the current device -- costs one integer compare instead of torch.cuda.device()'s get / set / restore."""
  return _NO_GUARD if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


def _same_device(ref, **named):
  """Every tensor argument must live on the device of `thb`: the kernel receives raw addresses."""
  for name, t in named.items():
    if t is not None and torch.is_tensor(t) and t.device != ref.device:
      raise RuntimeError('dgpmp2_amd: `%s` is on %s but `thb` is on %s; all inputs of one call must share a device' % (name, t.device, ref.device))


class _GNStep(torch.autograd.Function):
  """dtheta, err, err_ext = GN step; backward through dgp_gn_step_backward (adjoint block-tridiagonal solve)."""

  @staticmethod
  def launch(layer, static, th, start, goal, sdf, qc, ow, eps):
    """The forward launch itself (no autograd bookkeeping): -> dth, err, eex, and what the backward needs to keep alive."""
    B = th.shape[0]
    solver = layer._solver(th.dtype)
    _same_device(th, startb=start, goalb=goal, sdfb=sdf, qc_inv_trajb=qc, obscov_inv_trajb=ow, eps_trajb=eps)
    sdf_arg, sdf_keep = layer._sdf_arg(solver, sdf, th.dtype, B)
    covs, cov_keep = layer._covs_arg(solver, qc, ow, eps, th.dtype, B, static)
    thc, stc, goc = th.contiguous(), start.contiguous(), goal.contiguous()
    dth = torch.empty_like(thc)
    err = torch.empty(B, 1, 1, dtype=th.dtype, device=th.device)
    eex = torch.empty(B, 1, 1, dtype=th.dtype, device=th.device)
    info = torch.empty(B, dtype=torch.int32, device=th.device)
    with _on_device(th.device):             # the launch goes to the CURRENT device's stream: make that the tensors' device
      solver.gn_step(B, thc.data_ptr(), stc.data_ptr(), goc.data_ptr(), sdf_arg, covs, dth.data_ptr(), err.data_ptr(), eex.data_ptr(),
                     info.data_ptr(), _stream())
    object.__setattr__(layer, 'last_info', info)      # (plain attribute: nn.Module.__setattr__ costs microseconds per call)
    if layer.check_spd and bool(info.any()):
      raise RuntimeError('dgpmp2_amd: A^T K A + delta I is not positive definite for %d of %d trajectories '
                         '(the reference raises from torch.cholesky here)' % (int(info.sum()), B))
    return dth, err, eex, (thc, stc, goc), (sdf_keep, cov_keep)

  @staticmethod
  def forward(ctx, layer, static, th, start, goal, sdf, qc, ow, eps):
    dth, err, eex, (thc, stc, goc), keep = _GNStep.launch(layer, static, th, start, goal, sdf, qc, ow, eps)
    ctx.layer = layer
    ctx.static = static
    ctx.save_for_backward(thc, stc, goc, sdf, qc, ow, eps, dth)
    ctx.keep = keep
    ctx.mark_non_differentiable(err)              # plan_layer.py:275: error_batch runs under no_grad
    ctx.set_materialize_grads(False)              # an unused output arrives as None: no adjoint solve for an err_ext-only loss
    return dth, err, eex

  @staticmethod
  @once_differentiable                            # the backward is a raw kernel: double backward raises instead of returning zeros
  def backward(ctx, g_dth, g_err, g_eex):
    layer = ctx.layer
    th, start, goal, sdf, qc, ow, eps, dth = ctx.saved_tensors
    B, n, d = th.shape
    solver = layer._solver(th.dtype)
    sdf_arg, sdf_keep = layer._sdf_arg(solver, sdf, th.dtype, B)
    covs, cov_keep = layer._covs_arg(solver, qc, ow, eps, th.dtype, B, ctx.static)
    need = (None,) + tuple(ctx.needs_input_grad[2:])      # -> need[1..7] = th, start, goal, sdf, qc, ow, eps
    g_dth = None if g_dth is None else g_dth.contiguous().to(th.dtype)
    g_eex = None if g_eex is None else g_eex.contiguous().to(th.dtype)
    mk = lambda ref, on: torch.empty_like(ref, memory_format=torch.contiguous_format) if on else None
    g_th, g_st, g_go = mk(th, need[1]), mk(start, need[2]), mk(goal, need[3])
    shared = sdf.stride(0) == 0 or sdf.shape[0] == 1
    g_sdf = None
    copies = _SDF_GRAD_COPIES if shared else 1     # shared grid: one partial grid per XCD (XCD-local atomics), summed below
    if need[4]:
      g_sdf = torch.zeros((copies if shared else B, 1) + tuple(sdf.shape[-2:]), dtype=th.dtype, device=th.device)
    static_qc = covs.qc_mode == _capi.DGP_QC_STATIC
    g_qc = torch.empty(qc.shape, dtype=th.dtype, device=th.device) if (need[5] and not static_qc) else None
    g_ow = torch.empty(B, n, dtype=th.dtype, device=th.device) if (need[6] and covs.obs_w) else None
    g_eps = torch.empty(B, n, dtype=th.dtype, device=th.device) if (need[7] and covs.eps) else None
    p = lambda t: None if t is None else t.data_ptr()
    with _on_device(th.device):
      solver.gn_step_backward(B, th.data_ptr(), start.data_ptr(), goal.data_ptr(), sdf_arg, covs, dth.data_ptr(), p(g_dth), p(g_eex), p(g_th), p(g_st),
                              p(g_go), p(g_sdf), 0 if shared else sdf.shape[-1] * sdf.shape[-2], p(g_qc), p(g_ow), p(g_eps), _stream(),
                              g_sdf_copies=copies)
    if g_sdf is not None and shared:
      g_sdf = g_sdf.sum(0, keepdim=True)
    if g_sdf is not None and shared and sdf.shape[0] != 1:
      # the kernel already accumulated all B trajectories into the one shared grid; autograd's expand-backward will sum
      # the B slices of whatever is returned here, so hand it B equal shares
      g_sdf = (g_sdf / sdf.shape[0]).expand(sdf.shape)
    r = lambda g, ref: None if g is None else g.reshape(ref.shape).to(ref.dtype)
    return (None, None, g_th, g_st, g_go, r(g_sdf, sdf) if g_sdf is not None else None, r(g_qc, qc), r(g_ow, ow), r(g_eps, eps))


class _EvalErrors(torch.autograd.Function):
  """err_ext, start_goal_error, gp_error, obs_error at a trajectory = one launch of dgp_eval_errors; backward = one launch of
  dgp_eval_errors_backward.  These are the quantities the reference's training loss differentiates besides dtheta
  (learning/train_planner.py:327,342 -> one_step_loss :75-120): plain torch ops there (plan_layer.py:310-345, :374-388), hence
  differentiable w.r.t. thb, sdfb, the start / goal means and the current eps (all remembered WITH their graphs by the last
  forward(), plan_layer.py:88-94).  None of them depends on qc_inv / obscov_inv (fixed or unit weights)."""

  @staticmethod
  def forward(ctx, layer, th, start, goal, sdf, eps):
    eps_arg = None if (eps is None or getattr(eps, '_dgp_static', False)) else eps
    o = layer._eval(th, sdf, start, goal, None, None, eps_arg)
    ctx.layer = layer
    ctx.has_eps = eps_arg is not None
    ctx.save_for_backward(th, start, goal, sdf, eps_arg)
    ctx.set_materialize_grads(False)
    return o[1], o[2], o[3], o[4]            # (the three that read the grid are None without one)

  @staticmethod
  @once_differentiable
  def backward(ctx, g_eex, g_usg, g_ugp, g_uobs):
    layer = ctx.layer
    th, start, goal, sdf, eps = ctx.saved_tensors
    B, n, d = th.shape
    solver = layer._solver(th.dtype)
    sdf_arg, sdf_keep = layer._sdf_arg(solver, sdf, th.dtype, B)
    covs, cov_keep = layer._covs_arg(solver, None, None, eps, th.dtype, B, (True, True, eps is None))
    need = ctx.needs_input_grad                       # (layer, th, start, goal, sdf, eps)
    cot = [None if g is None else g.contiguous().to(th.dtype) for g in (g_eex, g_usg, g_ugp, g_uobs)]
    thc, stc, goc = th.contiguous(), start.contiguous(), goal.contiguous()
    mk = lambda ref, on: torch.empty_like(ref, memory_format=torch.contiguous_format) if on else None
    g_th, g_st, g_go = mk(thc, need[1]), mk(stc, need[2]), mk(goc, need[3])
    g_sdf, shared, copies = None, True, 1
    if sdf is not None:
      shared = sdf.stride(0) == 0 or sdf.shape[0] == 1
      copies = _SDF_GRAD_COPIES if shared else 1
      if need[4]:
        g_sdf = torch.zeros((copies if shared else B, 1) + tuple(sdf.shape[-2:]), dtype=th.dtype, device=th.device)
    g_eps = torch.empty(B, n, dtype=th.dtype, device=th.device) if (need[5] and eps is not None) else None
    p = lambda t: None if t is None else t.data_ptr()
    with _on_device(th.device):
      solver.eval_errors_backward(B, thc.data_ptr(), stc.data_ptr(), goc.data_ptr(), sdf_arg, covs, p(cot[0]), p(cot[1]), p(cot[2]), p(cot[3]),
                                  p(g_th), p(g_st), p(g_go), p(g_sdf), 0 if (sdf is None or shared) else sdf.shape[-1] * sdf.shape[-2], p(g_eps),
                                  _stream(), g_sdf_copies=copies)
    if g_sdf is not None and shared:
      g_sdf = g_sdf.sum(0, keepdim=True)
      if sdf.shape[0] != 1:                           # expand()ed view: autograd sums the B slices of what is returned (see _GNStep.backward)
        g_sdf = (g_sdf / sdf.shape[0]).expand(sdf.shape)
    r = lambda g, ref: None if g is None else g.reshape(ref.shape).to(ref.dtype)
    return (None, g_th, r(g_st, start), r(g_go, goal), r(g_sdf, sdf) if g_sdf is not None else None, r(g_eps, eps) if g_eps is not None else None)


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
    self.check_spd = check_spd
    self.last_info = None
    self._solvers = {}
    q = gp_params['Q_c_inv']
    self._qc_rows = [[float(v) for v in row] for row in (q.tolist() if torch.is_tensor(q) else q)]
    self._solver(torch.float64)        # validates the configuration now (and fails loudly if the library is missing)

  # -- C-ABI plumbing -------------------------------------------------------------------------------
  def _solver(self, dtype):
    code = _io_code(dtype)
    if code not in self._solvers:
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
      self._solvers[code] = s
    return self._solvers[code]

  def _sdf_arg(self, solver, sdfb, dtype, B):
    """sdfb (B,1,H,W) (only channel 0 is read, obstacle_cost.py:35).  An expand()ed / single grid is passed as shared.
    None: no grid (dgp_eval_errors without obstacle outputs only)."""
    if sdfb is None:
      return solver.sdf_arg(None, 2, 2, 0), None
    _require_cuda(sdfb, 'sdfb')
    if sdfb.dim() != 4: raise ValueError('sdfb must be (B,1,H,W)')
    H, W = sdfb.shape[-2], sdfb.shape[-1]
    shared = sdfb.stride(0) == 0 or sdfb.shape[0] == 1
    if not shared and sdfb.shape[0] != B:       # a per-sample grid tensor with fewer grids than trajectories would be read out of bounds
      raise ValueError('sdfb has %d grids for a batch of %d trajectories (expected %d, or 1 / an expand()ed view for a shared grid)'
                       % (sdfb.shape[0], B, B))
    t = sdfb[0:1, 0:1] if shared else sdfb[:, 0:1]
    if t.dtype != dtype or not t.is_contiguous():
      t = t.to(dtype).contiguous()
    return solver.sdf_arg(t.data_ptr(), H, W, 0 if shared else t.stride(0)), t

  @staticmethod
  def static_flags(qc, ow, eps):
    """(qc, ow, eps) -> which of them stand for the handle's static constants: None, or a tensor that
    DiffGPMP2Planner tagged as the expand()ed view of its own static covariance."""
    return tuple(t is None or getattr(t, '_dgp_static', False) for t in (qc, ow, eps))

  def _covs_arg(self, solver, qc, ow, eps, dtype, B, static=(False, False, False)):
    """Covariance tensors -> DgpCovs.  A static entry selects the constants of the handle (no per-state tensor is streamed)."""
    if static == (True, True, True):
      return _ALL_STATIC_COVS, ()
    n, dof, d = self.num_traj_states, self.dof, self.state_dim
    keep = []

    def prep(t, shape_tail, name, is_static=False):
      if t is None or is_static:
        return None
      _require_cuda(t, name)
      if t.shape[0] != B: raise ValueError('%s has batch %d, expected %d' % (name, t.shape[0], B))
      t = t.to(dtype).contiguous()
      if t.numel() != B * shape_tail: raise ValueError('%s has %d elements, expected %d' % (name, t.numel(), B * shape_tail))
      keep.append(t)
      return t.data_ptr()

    mode = _capi.DGP_QC_STATIC
    qc_p = None
    if qc is not None and not static[0]:
      q_full = self.learn_params is not None and self.dynamics_mode == 'q_full'        # plan_layer.py:90
      mode = _capi.DGP_QC_QFULL if q_full else _capi.DGP_QC_PERSTATE
      qc_p = prep(qc, (n - 1) * (d * d if q_full else dof * dof), 'qc_inv_trajb')
    return solver.covs_arg(mode, qc_p, prep(ow, n * self.nlinks, 'obscov_inv_trajb', static[1]),
                           prep(eps, n * self.nlinks, 'eps_trajb', static[2])), keep

  def _check_inputs(self, thb, startb, goalb):
    _require_cuda(thb, 'thb'); _require_cuda(startb, 'startb'); _require_cuda(goalb, 'goalb')
    B = thb.shape[0]
    if tuple(thb.shape[1:]) != (self.num_traj_states, self.state_dim):
      raise ValueError('thb must be (B,%d,%d), got %s' % (self.num_traj_states, self.state_dim, tuple(thb.shape)))
    if startb.numel() != B * self.state_dim or goalb.numel() != B * self.state_dim:
      raise ValueError('startb/goalb must be (B,1,%d)' % self.state_dim)
    if startb.dtype != thb.dtype or goalb.dtype != thb.dtype:
      raise TypeError('thb, startb, goalb must share one dtype')

  # -- the reference's public surface -----------------------------------------------------------------
  def forward(self, thb, startb, goalb, imb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb):
    """plan_layer.py:87-99.  -> (dthetab (B,n,d), err (B,1,1) [no grad], err_ext (B,1,1) [grad])."""
    self._check_inputs(thb, startb, goalb)
    # like the reference (plan_layer.py:88-94) remember means / covariances for the error_* helpers below
    static = self.static_flags(qc_inv_trajb, obscov_inv_trajb, eps_trajb)
    # (start / goal / eps are kept WITH their graphs, as set_mean / set_eps do: error_ext_batch and the unweighted errors are
    #  differentiable w.r.t. them; qc_inv / obscov_inv only feed error_batch, which runs under no_grad, :275)
    det = lambda t, st: None if (t is None or st) else t.detach()
    object.__setattr__(self, '_last', (startb, goalb, det(qc_inv_trajb, static[0]), det(obscov_inv_trajb, static[1]),
                                       None if (eps_trajb is None or static[2]) else eps_trajb))
    needs_graph = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                  for t in (thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb))
    if not needs_graph:        # planning / validation loops: no autograd node, one launch
      return _GNStep.launch(self, static, thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb)[:3]
    return _GNStep.apply(self, static, thb, startb, goalb, sdfb, qc_inv_trajb, obscov_inv_trajb, eps_trajb)

  def _eval(self, thb, sdfb, startb, goalb, qc, ow, eps):
    """-> [err, err_ext, start_goal_error, gp_error, obs_error]; without a grid (sdfb None) the three that read it are None."""
    B = thb.shape[0]
    solver = self._solver(thb.dtype)
    _same_device(thb, startb=startb, goalb=goalb, sdfb=sdfb, qc_inv_trajb=qc, obscov_inv_trajb=ow, eps_trajb=eps)
    sdf_arg, k1 = self._sdf_arg(solver, sdfb, thb.dtype, B)
    covs, k2 = self._covs_arg(solver, qc, ow, eps, thb.dtype, B, self.static_flags(qc, ow, eps))
    want = [sdfb is not None, sdfb is not None, True, True, sdfb is not None]
    outs = [torch.empty(B, 1, 1, dtype=thb.dtype, device=thb.device) if w else None for w in want]
    thc, stc, goc = thb.contiguous(), startb.contiguous(), goalb.contiguous()
    with _on_device(thb.device):
      solver.eval_errors(B, thc.data_ptr(), stc.data_ptr(), goc.data_ptr(), sdf_arg, covs, *[None if o is None else o.data_ptr() for o in outs],
                         stream=_stream())
    return outs

  def errors(self, thb, startb, goalb, sdfb, qc_inv_trajb=None, obscov_inv_trajb=None, eps_trajb=None):
    """(err, err_ext, start_goal_error, gp_error, obs_error), each (B,1,1), in one launch (dgp_eval_errors).  Unlike the
    reference (which reads the means/covariances left behind by the last forward()) everything is an argument."""
    self._check_inputs(thb, startb, goalb)
    return tuple(self._eval(thb, sdfb, startb, goalb, qc_inv_trajb, obscov_inv_trajb, eps_trajb))

  def _last_or_raise(self):
    if getattr(self, '_last', None) is None:
      raise RuntimeError('call forward() first: like the reference, the error_* helpers use the start/goal means and the '
                         'covariances set by the last forward() (plan_layer.py:88-94)')
    return self._last

  def error_batch(self, thb, sdfb):
    """plan_layer.py:273-308: normalised factor-graph error at thb, no grad, with the covariances of the last forward()."""
    st, go, qc, ow, eps = self._last_or_raise()
    with torch.no_grad():
      return self._eval(thb, sdfb, st, go, qc, ow, eps)[0]

  def _eval_diff(self, thb, sdfb, st, go, eps):
    """(err_ext, start_goal_error, gp_error, obs_error) at thb, each (B,1,1) (None where a grid is needed and sdfb is None), carrying
    the autograd graph the reference's plain torch ops would carry: w.r.t. thb, sdfb, the start / goal means and the current eps."""
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (thb, sdfb, st, go, eps)):
      return _EvalErrors.apply(self, thb, st, go, sdfb, eps)
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
