"""DiffGPMP2Planner -- host-side mirror of the reference's diff_gpmp2.gpmp2.diff_gpmp2_planner.DiffGPMP2Planner
(diff_gpmp2_planner.py:15-299): same constructor dicts, same step() / forward() / error_* signatures and return tuples,
so the reference's outer loops (learning/train_planner.py:297-403, datasets/generate_optimal_paths_gpmp2.py:181-184,
examples/diff_gpmp2_*) run unchanged on top of the HIP solver.

  step()    -> one launch of the fused GN kernel (PlanLayer.forward), differentiable.
  forward() -> the reference loops over samples and iterates each to convergence in Python
               (diff_gpmp2_planner.py:104-156); here the whole batch runs in ONE launch of the fused multi-iteration
               kernel (dgp_gn_solve), every trajectory with its own convergence test, when no autograd graph is needed;
               with requires_grad inputs (the reference keeps the graph across iterations, examples/diff_gpmp2_2d_example.py:77)
               the same launch also records the trajectory history and the backward pass is ONE more launch (dgp_gn_solve_backward);
               learn modules, a plan_time limit or more than 256 states chain differentiable step() calls.

The learned-covariance modules (LearnModuleConv / LearnModuleFCN, stock torch.nn in the reference) are out of this
build's scope: pass your own modules as `learn_module_conv` / `learn_module_fcn`; get_covariances() (the plumbing between
their output vector and the solver's covariance inputs, diff_gpmp2_planner.py:247-290) is provided.
"""
import time
from collections.abc import Sequence

import numpy as np
import torch
import torch.nn as nn

from .plan_layer import PlanLayer, _f, _launch, _raw_stream, _ALL_STATIC, _GNSolve, _NO_COVS, _expand_base, _RawCovs
from ..utils.planner_utils import check_convergence


def _global_module_hooks():
  """True when torch.nn.modules.module has global forward / backward hooks registered (then PlanLayer goes through __call__)."""
  m = torch.nn.modules.module
  return bool(m._global_forward_hooks or m._global_forward_pre_hooks or m._global_backward_hooks or getattr(m, '_global_backward_pre_hooks', None)
              or getattr(m, '_global_forward_hooks_always_called', None))


class _LazySeq(Sequence):
  """Base of the opt-in lazy per-sample results of forward() (DiffGPMP2Planner.lazy_results = True): a read-only collections.abc.Sequence (index / count /
  `in` / reversed come from the mixin) that concatenates, compares and pickles like the python list it stands for."""

  __slots__ = ()

  def tolist(self): return [self[i] for i in range(len(self))]
  def __add__(self, o): return self.tolist() + list(o)
  def __radd__(self, o): return list(o) + self.tolist()
  def __mul__(self, k): return self.tolist() * k
  __rmul__ = __mul__
  def __eq__(self, o):
    if isinstance(o, _LazySeq): o = o.tolist()
    return self.tolist() == (list(o) if isinstance(o, (list, tuple)) else o)
  def __ne__(self, o): return not self.__eq__(o)
  __hash__ = None
  def __reduce__(self): return (list, (self.tolist(),))      # pickles (and deep-copies) as the plain list
  def __repr__(self): return repr(self.tolist())


class _PerSampleHistory(_LazySeq):
  """List-like view of a (B, max_iters) history array: item b is the python list of sample b's first iters[b] entries --
  what the reference accumulates sample by sample (diff_gpmp2_planner.py:140-141,163-164) -- built lazily, because
  materialising 4096 python lists costs more than the whole fused solve."""

  __slots__ = ('_h', '_k')

  def __init__(self, hist, iters):
    self._h, self._k = hist, iters

  def __len__(self):
    return len(self._k)

  def __getitem__(self, b):
    if isinstance(b, slice):
      return [self[i] for i in range(*b.indices(len(self)))]
    return self._h[b, :int(self._k[b])].tolist()

  def __iter__(self):
    return (self[b] for b in range(len(self)))

  def tolist(self):
    return [row[:int(k)].tolist() for row, k in zip(self._h, self._k)]


class _LazyList(_LazySeq):
  """A per-sample result of forward() -- the reference returns python lists with one entry per sample (diff_gpmp2_planner.py:138-174) -- kept as the numpy array
  that came back from the device: indexing, slicing, iteration, len(), ==, +, index(), count(), pickling, np.asarray() and tolist() behave like the list, which is
  only materialised when someone asks for it.  Opt-in (DiffGPMP2Planner.lazy_results): forward() returns real lists by default, as the reference does."""

  __slots__ = ('_a',)

  def __init__(self, a): self._a = a
  def __len__(self): return len(self._a)
  def __getitem__(self, i):
    r = self._a[i]
    return r.tolist() if isinstance(i, slice) else r.item()
  def __iter__(self): return iter(self._a.tolist())
  def __array__(self, dtype=None, copy=None): return self._a if dtype is None else self._a.astype(dtype)
  def tolist(self): return self._a.tolist()


class _SquaredCovs(torch.autograd.Function):
  """get_covariances for single-link robots in the modes whose tensors are plain squares of the learn module's output (diff_gpmp2_planner.py:247-290 with nlinks = 1:
  q q^T and o o^T of 1 x 1 blocks): 'fix_dynamics' and 'diag_identity', with or without learned epsilons -- the same values bit for bit (x * x, and x * x times
  the 0 / 1 entries of the identity), in 3-4 small kernels forward and 4 backward instead of the ~20 that autograd spends on the slices, outer products and
  broadcasts of the literal formulation.  Under HIP-graph replay of a training iteration that glue costs more than the solver."""

  @staticmethod
  def forward(ctx, out, n_gp, n_obs, dof, learn_eps):
    B = out.shape[0]
    v = out[:, 0, :]
    sq = v * v
    ctx.save_for_backward(out)
    ctx.dims = (n_gp, n_obs, dof, learn_eps)
    res = []
    if n_gp:
      s = sq[:, :n_gp].contiguous()
      res.append(s.view(B, n_gp, 1, 1) * torch.eye(dof, device=out.device, dtype=out.dtype))
      res.append(s)                                                    # (the scalars themselves, for the DGP_QC_SCALAR tag; not differentiable)
      ctx.mark_non_differentiable(s)
    if learn_eps and out.shape[2] != n_gp + 2 * n_obs:      # the reference reshapes out[:, 0, n_gp + n_obs:] to (B, n, nl, 1) (:282) and raises on any other length
      raise RuntimeError('get_covariances: the learn module emitted %d values, %d expected with learn_eps' % (out.shape[2], n_gp + 2 * n_obs))
    # contiguous tensors (not strided views into `sq`): PlanLayer hands their addresses to the kernel as they are, and the gradient buffers take their shape
    res.append(sq[:, n_gp:n_gp + n_obs].reshape(B, n_obs, 1, 1).contiguous())
    if learn_eps: res.append(sq[:, n_gp + n_obs:n_gp + 2 * n_obs].reshape(B, n_obs, 1, 1).contiguous())
    return tuple(res)

  @staticmethod
  def backward(ctx, *grads):
    out, = ctx.saved_tensors
    n_gp, n_obs, dof, learn_eps = ctx.dims
    B, W = out.shape[0], out.shape[2]
    parts, i = [], 0
    if n_gp:
      g_qc = grads[0]; i = 2
      parts.append(out.new_zeros(B, n_gp) if g_qc is None else g_qc.diagonal(dim1=-2, dim2=-1).sum(-1))
    for k in range(1 + int(learn_eps)):
      g = grads[i + k]
      parts.append(out.new_zeros(B, n_obs) if g is None else g.reshape(B, n_obs))
    used = n_gp + n_obs * (1 + int(learn_eps))
    if used < W: parts.append(out.new_zeros(B, W - used))
    g_sq = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
    return (2.0 * out[:, 0, :] * g_sq).unsqueeze(1), None, None, None, None


class DiffGPMP2Planner(nn.Module):
  def __init__(self, gp_params, obs_params, planner_params, optim_params, env_params, robot_model, learn_params=None, batch_size=1,
               use_cuda=False, learn_module_conv=None, learn_module_fcn=None):
    super(DiffGPMP2Planner, self).__init__()
    self.use_cuda = torch.cuda.is_available() if use_cuda else False
    if not torch.cuda.is_available():
      raise RuntimeError('dgpmp2_amd.DiffGPMP2Planner needs a ROCm GPU: the solver has no CPU path')
    self.device = torch.device('cuda')
    self.dof = planner_params['dof']
    self.state_dim = planner_params['state_dim']
    self.total_time_sec = planner_params['total_time_sec']
    self.total_time_step = planner_params['total_time_step']
    self.num_traj_states = self.total_time_step + 1
    self.num_gp_factors = self.num_traj_states - 1
    self.num_obs_factors = self.num_traj_states
    self.optim_params, self.gp_params, self.obs_params = optim_params, gp_params, obs_params
    self.robot_model, self.env_params, self.learn_params = robot_model, env_params, learn_params
    self.model_type = 'feed_forward'
    self.non_holonomic = planner_params['non_holonomic'] if 'non_holonomic' in planner_params else False
    self.use_vel_limits = planner_params['use_vel_limits'] if 'use_vel_limits' in planner_params else False
    self.batch_size = batch_size
    nl = robot_model.nlinks
    dd = torch.float64      # the static covariances are exact copies of the config values (the reference runs under a float64 default)
    mk = lambda shape, v: (torch.zeros(*shape, dtype=dd, device=self.device) + torch.as_tensor(v, dtype=dd, device=self.device))
    self._static_views = {}
    self._static_triples = {}
    self.fixed_conv = False
    self.learn_eps = False
    self.dynamics_mode = None
    if learn_params is None:
      # static covariances (diff_gpmp2_planner.py:40-52)
      self.qc_inv_traj = mk((self.num_gp_factors, self.dof, self.dof), gp_params['Q_c_inv'])
      self.obscov_inv_traj = mk((self.num_traj_states, nl, 1), 1.0 / _f(obs_params['cost_sigma']) ** 2.0)
      self.eps_traj = mk((self.num_traj_states, nl, 1), _f(obs_params['epsilon_dist']))
      self.learn_module_conv = None
      self.learn_module_fcn = None
    else:
      # diff_gpmp2_planner.py:53-88
      lp = learn_params
      self.model_type = lp['model']['type'] if 'type' in lp['model'] else False
      self.learn_eps = lp['dgpmp2']['learn_eps'] if 'learn_eps' in lp['dgpmp2'] else False
      self.sdf_predict = lp['dgpmp2']['sdf_predict']
      self.use_dtheta = lp['dgpmp2']['dtheta_predict'] if 'dtheta_predict' in lp['dgpmp2'] else False
      self.dynamics_mode = lp['dgpmp2']['dynamics_mode']
      # like the reference's constructor (:60-76), tell the learn modules -- through the SAME dict -- what they must emit, before they are built
      self.res = self.prepare_learn_params(lp, planner_params, env_params, robot_model)
      if self.dynamics_mode == 'fix_dynamics':
        self.qc_inv_traj = mk((self.num_gp_factors, self.dof, self.dof), gp_params['Q_c_inv'])
      if not self.learn_eps:
        self.eps = obs_params['epsilon_dist']
        self.eps_traj = mk((self.num_traj_states, nl, 1), _f(obs_params['epsilon_dist']))
      self.fixed_conv = lp['dgpmp2']['fixed_conv'] if 'fixed_conv' in lp['dgpmp2'] else False
      if learn_module_fcn is None:
        raise NotImplementedError('learn_params given but no learn modules: the CNN/FCN covariance predictors are stock torch.nn '
                                  'and outside this build; pass them (instances, or the reference\'s classes / any factory with their '
                                  'constructor signature) as learn_module_conv= / learn_module_fcn=')
      # an nn.Module instance is used as it is; anything else callable is a factory with the signature of the reference's classes
      # (diff_gpmp2_planner.py:86-87) and is called now, with learn_params['out_dim'] etc. already in place
      if learn_module_conv is not None and not isinstance(learn_module_conv, nn.Module):
        learn_module_conv = learn_module_conv(lp, env_params, robot_model, use_cuda=self.use_cuda)
      if not isinstance(learn_module_fcn, nn.Module):
        learn_module_fcn = learn_module_fcn(lp, env_params, obs_params, robot_model, use_cuda=self.use_cuda)
      self.learn_module_conv = learn_module_conv
      self.learn_module_fcn = learn_module_fcn
    self.plan_layer = PlanLayer(gp_params, obs_params, planner_params, optim_params, env_params, robot_model, learn_params,
                                self.batch_size, self.use_cuda)
    # plain-attribute aliases for step(): a submodule / None-module attribute is resolved by nn.Module.__getattr__ (~1 us per access)
    self.__dict__['_pl'] = self.plan_layer
    self.__dict__['_learned'] = self.learn_module_fcn is not None

  @staticmethod
  def prepare_learn_params(learn_params, planner_params, env_params, robot_model):
    """What the reference's constructor writes into `learn_params` before it builds LearnModuleConv / LearnModuleFCN from that dict
    (diff_gpmp2_planner.py:60-78): 'num_traj_states' (doubled with dtheta_predict), 'state_dim' and 'out_dim' -- the length of the
    vector get_covariances() slices, per dynamics_mode, + one more block of obstacle factors with learn_eps.  Returns the image
    resolution the reference keeps as `self.res` (:58).  Called by the constructor; public so that a caller who builds the learn
    modules first can prepare the dict the same way."""
    lp = learn_params
    n = int(planner_params['total_time_step']) + 1
    dof, state_dim, nl = int(planner_params['dof']), int(planner_params['state_dim']), robot_model.nlinks
    n_gp, n_obs = n - 1, n * nl
    lp['num_traj_states'] = n
    lp['state_dim'] = planner_params['state_dim']
    if lp['dgpmp2']['dtheta_predict'] if 'dtheta_predict' in lp['dgpmp2'] else False:
      lp['num_traj_states'] = 2 * n
    per_mode = {'fix_dynamics': 0, 'diag_identity': n_gp, 'diag': n_gp * dof, 'qc_full': n_gp * dof, 'q_full': n_gp * state_dim}
    mode = lp['dgpmp2']['dynamics_mode']
    if mode in per_mode:                   # (an unknown mode leaves 'out_dim' alone, as the reference's if / elif chain does)
      lp['out_dim'] = per_mode[mode] + n_obs
    if lp['dgpmp2']['learn_eps'] if 'learn_eps' in lp['dgpmp2'] else False:
      lp['out_dim'] = lp['out_dim'] + n_obs
    return (_f(env_params['x_lims'][1]) - _f(env_params['x_lims'][0])) / (lp['data']['im_size'] * 1.0)

  # -- helpers ------------------------------------------------------------------------------------------
  def _static_view(self, t, B, like):
    """(…)-shaped static covariance -> (B,…) expand()ed view in the dtype of `like`, tagged so that PlanLayer passes the
    handle's constants instead of streaming B copies (the reference materialises them with .repeat, :202-205).  Cached per
    (tensor, batch, dtype): building three views costs more host time than the kernel runs."""
    key = (id(t), B, like.dtype)
    v = self._static_views.get(key)
    if v is None:
      v = t.to(like.dtype).unsqueeze(0).expand(B, *t.shape)
      v._dgp_static = True
      self._static_views[key] = v
    return v

  def _static_covs(self, B, like):
    """The three static covariance views of step() for a batch of B in the dtype of `like` (one dictionary look-up per call)."""
    key = (B, like.dtype)
    v = self._static_triples.get(key)
    if v is None:
      v = self._static_triples[key] = (self._static_view(self.qc_inv_traj, B, like), self._static_view(self.obscov_inv_traj, B, like),
                                       self._static_view(self.eps_traj, B, like))
    return v

  def _predict(self, th_in, conv_out, hiddenb, im_in=None, raw_ok=False):
    """Learned mode: run the user's modules and turn their output into covariances (diff_gpmp2_planner.py:183-199).  raw_ok (step() / step_with_errors()): in the
    modes whose tensors are plain squares of the module output the vector itself is handed on (-> (_RawCovs, None, None, hidden)) and squared inside the kernels."""
    if not self.fixed_conv:
      conv_out, _ = self.learn_module_conv(im_in)
    if self.model_type == 'feed_forward':
      out = self.learn_module_fcn(th_in, conv_out); hidden = None
    else:
      out, hidden = self.learn_module_fcn(th_in, conv_out, hiddenb)
    B = th_in.shape[0]
    eps = None
    if raw_ok and out.dtype is th_in.dtype:
      raw = self.__dict__['_pl'].raw_covs(out, self.dynamics_mode, self.learn_eps)
      if raw is not None: return raw, None, None, hidden
    if self.dynamics_mode == 'fix_dynamics':
      r = self.get_covariances(out, self.dynamics_mode, self.learn_eps)
      obscov, eps = (r if self.learn_eps else (r, None))
      qc = self._static_view(self.qc_inv_traj, B, th_in)
    else:
      r = self.get_covariances(out, self.dynamics_mode, self.learn_eps)
      qc, obscov = r[0], r[1]
      if self.learn_eps: eps = r[2]
    if eps is None:
      eps = self._static_view(self.eps_traj, B, th_in)
    return qc, obscov, eps, hidden

  # -- reference API --------------------------------------------------------------------------------------
  def _step_covariances(self, th_currb, imb, sdfb, conv_out, dtheta_currb, hiddenb, raw_ok=False):
    """The covariance inputs of one step (diff_gpmp2_planner.py:183-205): predicted by the learn modules, or the static ones."""
    if self.__dict__['_learned']:
      im_in = None
      if not self.fixed_conv:
        im_in = torch.cat((imb, sdfb), dim=1) if self.sdf_predict else imb
      th_in = torch.cat((th_currb, dtheta_currb), dim=-1) if self.use_dtheta else th_currb
      return self._predict(th_in, conv_out, hiddenb, im_in, raw_ok)
    return self._static_covs(th_currb.shape[0], th_currb) + (None,)

  def step(self, th_currb, startb, goalb, imb, sdfb, conv_out=None, dtheta_currb=None, hiddenb=None):
    """One iteration of non-linear optimisation on a batch of environments (diff_gpmp2_planner.py:176-211).
    -> (dthetab, hidden_newb, err_oldb, err_ext_oldb, qc_inv_curr, obscov_inv_curr, eps_curr)"""
    pl = self.__dict__['_pl']
    hooks = pl._forward_hooks or pl._forward_pre_hooks or pl._backward_hooks or pl._backward_pre_hooks or _global_module_hooks()
    if self.__dict__['_learned']:
      qc_inv_curr, obscov_inv_curr, eps_curr, hidden = self._step_covariances(th_currb, imb, sdfb, conv_out, dtheta_currb, hiddenb, raw_ok=not hooks)
      if qc_inv_curr.__class__ is _RawCovs:
        # 'diag_identity' / 'fix_dynamics': the module output goes to the kernels as it is (squared there; the backward kernel writes d/d out)
        raw = qc_inv_curr
        dthetab, err_oldb, err_ext_oldb, qc_inv_curr, obscov_inv_curr, eps_curr = pl.forward_raw(th_currb, startb, goalb, imb, sdfb, raw)
        if qc_inv_curr is None: qc_inv_curr = self._static_view(self.qc_inv_traj, th_currb.shape[0], th_currb)
        if eps_curr is None: eps_curr = self._static_view(self.eps_traj, th_currb.shape[0], th_currb)
        return dthetab, (hidden if hiddenb is not None else None), err_oldb, err_ext_oldb, qc_inv_curr, obscov_inv_curr, eps_curr
    else:
      hidden = None
      qc_inv_curr, obscov_inv_curr, eps_curr = self._static_covs(th_currb.shape[0], th_currb)
    # (nn.Module.__call__ costs ~2 us of hook bookkeeping per call; without hooks it does nothing but call forward())
    if hooks:
      dthetab, err_oldb, err_ext_oldb = pl(th_currb, startb, goalb, imb, sdfb, qc_inv_curr, obscov_inv_curr, eps_curr)
    else:
      dthetab, err_oldb, err_ext_oldb = pl.forward(th_currb, startb, goalb, imb, sdfb, qc_inv_curr, obscov_inv_curr, eps_curr)
    hidden_newb = hidden if hiddenb is not None else None
    return dthetab, hidden_newb, err_oldb, err_ext_oldb, qc_inv_curr, obscov_inv_curr, eps_curr

  def step_with_errors(self, th_currb, startb, goalb, imb, sdfb, conv_out=None, dtheta_currb=None, hiddenb=None):
    """step() and, in the same call, unweighted_errors_batch(th_currb + dthetab, sdfb) -- what every iteration of the reference's training
    loop does back to back (learning/train_planner.py:311,313,327).  -> (step()'s 7-tuple, (err_sg (B,1), err_gp (B,1,1), err_obs (B,1,1)));
    one C-ABI call forward, one backward (PlanLayer.forward_with_errors).  No counterpart in the reference: an addition for its outer loop,
        out, (err_sg, err_gp, err_obs) = planner.step_with_errors(th, start, goal, im, sdf, conv_out, dtheta)
    replacing  out = planner.step(...); err_sg, err_gp, err_obs = planner.unweighted_errors_batch(th + out[0], sdf)."""
    pl = self.__dict__['_pl']
    if self.num_traj_states > 256 or pl._forward_hooks or pl._forward_pre_hooks or pl._backward_hooks or pl._backward_pre_hooks or _global_module_hooks():
      # long trajectories (the fused entry points stop at 256 states) and registered module hooks: the two calls this method stands for
      out = self.step(th_currb, startb, goalb, imb, sdfb, conv_out, dtheta_currb, hiddenb)
      return out, self.unweighted_errors_batch(th_currb + out[0], sdfb)
    qc_inv_curr, obscov_inv_curr, eps_curr, hidden = self._step_covariances(th_currb, imb, sdfb, conv_out, dtheta_currb, hiddenb, raw_ok=True)
    if qc_inv_curr.__class__ is _RawCovs:
      raw = qc_inv_curr
      dthetab, err_oldb, err_ext_oldb, e_sg, e_gp, e_obs, qc_inv_curr, obscov_inv_curr, eps_curr = pl.forward_raw(th_currb, startb, goalb, imb, sdfb, raw, with_errors=True)
      if qc_inv_curr is None: qc_inv_curr = self._static_view(self.qc_inv_traj, th_currb.shape[0], th_currb)
      if eps_curr is None: eps_curr = self._static_view(self.eps_traj, th_currb.shape[0], th_currb)
    else:
      dthetab, err_oldb, err_ext_oldb, e_sg, e_gp, e_obs = pl.forward_with_errors(th_currb, startb, goalb, imb, sdfb, qc_inv_curr, obscov_inv_curr, eps_curr)
    hidden_newb = hidden if hiddenb is not None else None
    return (dthetab, hidden_newb, err_oldb, err_ext_oldb, qc_inv_curr, obscov_inv_curr, eps_curr), (e_sg, e_gp, e_obs)

  def forward(self, th_initb, startb, goalb, imb, sdfb, hiddenb=None):
    """Gauss-Newton to convergence for every sample (diff_gpmp2_planner.py:92-174).
    -> (th_currb, hidden_newb, err_initb, err_finalb, err_per_iterb, err_ext_per_iterb, jb, timeb); the err_* / jb / timeb
    entries are python lists with one item per sample, as in the reference."""
    start_t = time.time()
    B = th_initb.shape[0]
    max_iters = int(self.optim_params['max_iters'])
    tol_delta = float(self.optim_params['tol_delta'])
    plan_time = float(self.optim_params['plan_time']) if 'plan_time' in self.optim_params else float('inf')
    needs_graph = torch.is_grad_enabled() and any(t.requires_grad for t in (th_initb, startb, goalb, sdfb))
    if self.learn_module_fcn is None and plan_time == float('inf'):
      if not needs_graph:
        return self._forward_fused(th_initb, startb, goalb, sdfb, max_iters, tol_delta, start_t)
      if self._chain_backward_available(B, max_iters):
        return self._forward_fused(th_initb, startb, goalb, sdfb, max_iters, tol_delta, start_t, with_graph=True)
    return self._forward_stepwise(th_initb, startb, goalb, imb, sdfb, hiddenb, max_iters, tol_delta, plan_time, start_t)

  fused_history_budget = 2 << 30      # bytes of fp64 trajectory history a differentiable fused forward() may hold until its graph is freed
  # forward() returns its per-sample results (err_initb, err_finalb, err_per_iterb, err_ext_per_iterb, jb) as python lists, like the reference.  True: list-like
  # read-only views of the arrays that came back from the device instead (_LazyList / _PerSampleHistory: no 4096-element list construction per call)
  lazy_results = False

  def _chain_backward_available(self, B=1, max_iters=1):
    """dgp_gn_solve_backward covers static covariances (any Q_c_inv: a diagonal one runs the static / Woodbury chain kernels, a non-diagonal one the general ones,
    round 6) and trajectories of up to 256 states; anything else differentiates through chained step() calls (_forward_stepwise).  So does a loop whose history --
    (max_iters, B, n, d) doubles, allocated in full whatever the iterations that run (the reference's YAML default is max_iters = 100) -- would exceed
    `fused_history_budget`: the stepwise path keeps state for the iterations that ran and stops when every trajectory has converged; that fall-back is logged once per
    planner (it changes the cost of a forward() + backward() by an order of magnitude)."""
    if self.num_traj_states > 256: return False
    need = max_iters * B * self.num_traj_states * self.state_dim * 8
    if need > self.fused_history_budget:
      if not self.__dict__.get('_warned_history'):
        self.__dict__['_warned_history'] = True
        import warnings
        warnings.warn('dgpmp2_amd: forward() with a graph: the fp64 trajectory history of the fused loop (max_iters %d x B %d x n %d x d %d = %.1f GiB) exceeds '
                      'fused_history_budget (%.1f GiB): differentiating through chained step() calls instead (slower; raise DiffGPMP2Planner.fused_history_budget or '
                      'lower max_iters)' % (max_iters, B, self.num_traj_states, self.state_dim, need / 2.0 ** 30, self.fused_history_budget / 2.0 ** 30), RuntimeWarning)
      return False
    return True

  def _forward_fused(self, th_initb, startb, goalb, sdfb, max_iters, tol_delta, start_t, with_graph=False):
    """The whole batch in ONE launch of the fused loop.  with_graph: th_currb carries the autograd graph of the loop (w.r.t. th_initb, startb,
    goalb, sdfb) as ONE node whose backward is one more launch (_GNSolve: dgp_gn_solve_traced / dgp_gn_solve_backward)."""
    pl = self.plan_layer
    pl._check_inputs(th_initb, startb, goalb)
    B = th_initb.shape[0]
    if pl.auto_tile: sdfb = pl._auto_tiled(sdfb, B)
    dt, dev = th_initb.dtype, th_initb.device
    solver = pl._solver(dt)
    idx = th_initb.get_device()
    m = max_iters
    if with_graph:
      if sdfb is not None and sdfb.requires_grad: sdfb = _expand_base(sdfb)      # (a shared grid's gradient comes back unexpanded: no B-fold sum in autograd)
      ts = (th_initb, startb, goalb, sdfb)
      slots = tuple([i for i in range(4) if ts[i] is not None and ts[i].requires_grad])
      box = []
      th_out = _GNSolve.apply(pl, max_iters, tol_delta, slots, ts, box, *[ts[i] for i in slots])
      buf, info = box[0]
      st, go = startb, goalb
    else:
      sd = pl._sdf_args(sdfb, dt, B, idx)
      th0, st, go = th_initb.detach().contiguous(), startb.detach().contiguous(), goalb.detach().contiguous()
      th_out = torch.empty_like(th0)
      # the per-sample outputs share ONE device buffer -- err history | err_ext history | final error | iteration counts | SPD flags -- so that one fill (NaN: entries
      # past a sample's last iteration stay untouched) and ONE device-to-host copy serve all four (each separate copy costs a synchronisation of its own)
      buf = torch.full((B * (2 * m + 3),), float('nan'), dtype=dt, device=dev)
      eh, eeh, ef = buf[:B * m], buf[B * m:2 * B * m], buf[2 * B * m:2 * B * m + B]
      iters = buf[2 * B * m + B:].view(torch.int32)[:B]           # int32 counts and SPD flags in the storage of the last 2 B elements
      info = buf[2 * B * m + 2 * B:].view(torch.int32)[:B]
      _launch(idx, pl._pc.gn_solve, solver.h, B, th0.data_ptr(), st.data_ptr(), go.data_ptr(), *sd[:7], *_NO_COVS,
              max_iters, tol_delta, th_out.data_ptr(), iters.data_ptr(), eh.data_ptr(), eeh.data_ptr(), ef.data_ptr(), info.data_ptr(), _raw_stream(idx))
    pl.last_info = info
    pl._last = (st, go, None, None, None)
    # ONE device-to-host copy of everything the reference API returns per sample -- error histories, final errors, iteration counts AND the SPD flags (they live
    # in the tail of the same buffer) -- into pinned host memory (torch's caching host allocator: no page-locking per call), one stream synchronisation
    host = torch.empty(buf.shape, dtype=buf.dtype, pin_memory=True)
    host.copy_(buf, non_blocking=True)
    torch.cuda.current_stream(idx).synchronize()
    flags = host[2 * B * m + 2 * B:].view(torch.int32)[:B].numpy()
    bad = int(np.count_nonzero(flags))
    if bad:      # the reference raises from torch.cholesky at the first non-SPD system (plan_layer.py:226); here the flags ride along with the results: no extra copy
      raise RuntimeError('dgpmp2_amd: A^T K A + delta I is not positive definite for %d of %d trajectories during forward() '
                         '(the reference raises from torch.cholesky here); per-trajectory flags: plan_layer.last_info' % (bad, B))
    eh_c, eeh_c = host[:B * m].view(B, m).numpy(), host[B * m:2 * B * m].view(B, m).numpy()
    ef_c = host[2 * B * m:2 * B * m + B].numpy()
    jb = host[2 * B * m + B:].view(torch.int32)[:B].numpy()
    t = time.time() - start_t
    res = (_LazyList(eh_c[:, 0]), _LazyList(ef_c), _PerSampleHistory(eh_c, jb), _PerSampleHistory(eeh_c, jb), _LazyList(jb))
    if not self.lazy_results: res = tuple(r.tolist() for r in res)      # python lists, one entry per sample, as the reference returns them (:138-174)
    return (th_out, None) + res + ([t] * B,)

  def _forward_stepwise(self, th_initb, startb, goalb, imb, sdfb, hiddenb, max_iters, tol_delta, plan_time, start_t):
    """Differentiable / learned / time-limited variant: chained batched step() calls with a per-trajectory freeze once
    ||dtheta|| < tol_delta (each sample sees exactly the iterations the reference's per-sample loop would run)."""
    B = th_initb.shape[0]
    th = th_initb
    active = torch.ones(B, dtype=torch.bool, device=th.device)
    jb = torch.zeros(B, dtype=torch.int64, device=th.device)
    errs, errs_ext, acts = [], [], []
    hidden = hiddenb
    conv_out = None
    dtheta = torch.zeros_like(th_initb)
    if self.learn_module_fcn is not None and self.fixed_conv:
      conv_out, _ = self.learn_module_conv(torch.cat((imb, sdfb), dim=1) if self.sdf_predict else imb)
    j = 0
    while True:
      dtheta, hidden_new, err_old, err_ext_old, _, _, _ = self.step(th, startb, goalb, imb, sdfb, conv_out, dtheta, hidden)
      if hiddenb is not None: hidden = hidden_new
      errs.append(err_old.detach().reshape(B)); errs_ext.append(err_ext_old.detach().reshape(B)); acts.append(active.clone())
      th = th + dtheta * active.view(B, 1, 1).to(th.dtype)
      j += 1
      jb = jb + active.to(jb.dtype)
      nrm = torch.norm(dtheta.detach().reshape(B, -1), dim=1)
      active = active & ~(nrm < tol_delta)
      if j >= max_iters or not bool(active.any()):
        break
      if time.time() - start_t > plan_time:
        print('Plan time over')
        break
    err_final = self.plan_layer.error_batch(th.detach(), sdfb).reshape(B).cpu().tolist()
    E = torch.stack(errs, 1).cpu().numpy(); EE = torch.stack(errs_ext, 1).cpu().numpy(); jl = jb.cpu().tolist()
    t = time.time() - start_t
    hidden_newb = hidden if hiddenb is not None else None
    he, hee = _PerSampleHistory(E, jl), _PerSampleHistory(EE, jl)
    if not self.lazy_results: he, hee = he.tolist(), hee.tolist()
    return th, hidden_newb, E[:, 0].tolist(), err_final, he, hee, jl, [t] * B

  def graphed_iteration(self, fn, warmup=3, clone_outputs=False):
    """`fn` -- one iteration of an outer loop written against this planner (step() / step_with_errors() / the error helpers, a loss, autograd.grad or backward(),
    optionally the optimiser step), a function of its tensor arguments -- as a callable that records the iteration in a HIP graph at its first call and replays
    it afterwards: the eager loop of learning/train_planner.py:297-403 keeps its shape and loses Python, the autograd engine and ~20 launch enqueues per iteration
    (utils/graph_utils.py: the rules `fn` must obey, and what is returned).  No counterpart in the reference."""
    from ..utils.graph_utils import GraphedIteration
    return GraphedIteration(fn, warmup=warmup, clone_outputs=clone_outputs)

  def error_batch(self, thb, sdfb):
    return self.plan_layer.error_batch(thb, sdfb)

  def error_ext_batch(self, thb, sdfb):
    return self.plan_layer.error_ext_batch(thb, sdfb)

  def unweighted_errors_batch(self, thb, sdfb):
    """diff_gpmp2_planner.py:229-237 -> (err_sg, err_gp, err_obs), each (B,1,1)."""
    return self.plan_layer.unweighted_errors(thb, sdfb)

  def get_covariances(self, out, mode='diag_identity', learn_eps=False):
    """Learn-module output (B,1,out_dim) -> covariance tensors (diff_gpmp2_planner.py:247-290).
    fix_dynamics: obscov[, eps];  diag_identity: q^2 I;  qc_full: q q^T (dof);  q_full: q q^T (state_dim);  'diag' raises."""
    nl = self.robot_model.nlinks
    B = out.shape[0]
    n_obs = self.num_obs_factors * nl
    if nl == 1 and mode in ('fix_dynamics', 'diag_identity') and out.dim() == 3 and out.shape[1] == 1:
      # single-link robot, tensors that are plain squares of the module output: one small autograd node instead of the literal formulation below (same values)
      n_gp = self.num_gp_factors if mode == 'diag_identity' else 0
      res = _SquaredCovs.apply(out, n_gp, n_obs, self.dof, bool(learn_eps))
      if n_gp:
        qc_inv_traj, s = res[0], res[1]
        # the tensor IS q_k^2 I: PlanLayer.forward (and its backward) may hand the kernels the n - 1 scalars instead of the blocks (DGP_QC_SCALAR: the
        # static kernels with scaled lane masks); every other consumer reads the blocks themselves.  (The version: an in-place edit of the blocks voids the tag.)
        qc_inv_traj.__dict__['_dgp_scalar'] = (s, qc_inv_traj._version)
        return (qc_inv_traj,) + tuple(res[2:])
      return res[0] if len(res) == 1 else tuple(res)
    if mode == 'fix_dynamics':
      n_gp = 0
      qc_inv_traj = None
    elif mode == 'diag_identity':
      n_gp = self.num_gp_factors
      q = out[:, 0, 0:n_gp].reshape(B, self.num_gp_factors, 1, 1)
      qc_inv_traj = (q * q.transpose(2, 3)) * torch.eye(self.dof, device=out.device, dtype=out.dtype)
      # the tensor IS q_k^2 I: PlanLayer.forward (and its backward) may hand the kernels the n - 1 scalars instead of the blocks (DGP_QC_SCALAR: the
      # static kernels with scaled lane masks); every other consumer reads the blocks themselves
      qc_inv_traj.__dict__['_dgp_scalar'] = ((q * q).detach().reshape(B, self.num_gp_factors), qc_inv_traj._version)      # (the version: an in-place edit of the blocks voids the tag)
    elif mode == 'diag':
      raise NotImplementedError
    elif mode == 'qc_full':
      n_gp = self.num_gp_factors * self.dof
      q = out[:, 0, 0:n_gp].reshape(B, self.num_gp_factors, self.dof, 1)
      qc_inv_traj = q * q.transpose(2, 3)
    elif mode == 'q_full':
      n_gp = self.num_gp_factors * self.state_dim
      q = out[:, 0, 0:n_gp].reshape(B, self.num_gp_factors, self.state_dim, 1)
      qc_inv_traj = q * q.transpose(2, 3)
    else:
      raise ValueError('unknown dynamics_mode %r' % (mode,))
    o = out[:, 0, n_gp:n_gp + n_obs].reshape(B, self.num_obs_factors, nl, 1)
    obscov_inv_traj = o * o.transpose(2, 3)
    if learn_eps:
      e = out[:, 0, n_gp + n_obs:].reshape(B, self.num_obs_factors, nl, 1)
      eps_traj = e * e.transpose(2, 3)
      if mode == 'fix_dynamics':
        return obscov_inv_traj, eps_traj
      return qc_inv_traj, obscov_inv_traj, eps_traj
    if mode == 'fix_dynamics':
      return obscov_inv_traj
    return qc_inv_traj, obscov_inv_traj

  def get_obs_covariance(self, out):
    """Learn-module output (num_obs_factors values) -> per-state obstacle information blocks (num_obs_factors, nlinks, nlinks):
    out_k^2 on the diagonal (diff_gpmp2_planner.py:293-297)."""
    nl = self.robot_model.nlinks
    w = (out * out).reshape(self.num_obs_factors, 1, 1)
    return w * torch.eye(nl, device=out.device, dtype=out.dtype)
