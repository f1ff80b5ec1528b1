#!/usr/bin/env python
"""Generate golden fixtures by running the REAL reference (/root/reference, read-only) in the build
container.  Only the resulting .npz data files are committed and travel to the GPU box; the reference
sources never do.  Re-run with:   python tests/golden/make_golden.py

Shims (in-process, no file edits): tolerant plt.style.use (env_2d.py:14 asks for the
removed 'seaborn-paper' style) and torch.Tensor.byte -> bool (masks are built with .byte(),
plan_layer.py:392-406, point_robot_2d.py:66-68; torch>=2 rejects uint8 masks).
The reference is fp64-only, so everything here is float64.
"""
import os, sys
sys.dont_write_bytecode = True
import numpy as np
import matplotlib
matplotlib.use('Agg')
import matplotlib.pyplot as plt
_use = plt.style.use
plt.style.use = lambda s: (_use(s) if s in plt.style.available else None)
import torch
import warnings
warnings.filterwarnings('ignore')
torch.Tensor.byte = lambda self: self.bool()
torch.set_default_dtype(torch.float64)

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from diff_gpmp2.robot_models import PointRobot2D
from diff_gpmp2.gpmp2.diff_gpmp2_planner import DiffGPMP2Planner
from diff_gpmp2.gpmp2.plan_layer import PlanLayer
from diff_gpmp2.gpmp2.gp import GPFactor, PriorFactor
from diff_gpmp2.gpmp2.obstacle import ObstacleFactor
from diff_gpmp2.gpmp2.custom_factors import NonHolonomicFactor
from diff_gpmp2.utils.sdf_utils import sdf_2d, bilinear_interpolate
from diff_gpmp2.utils.planner_utils import straight_line_trajb
from oracle.gpmp2_oracle import circles_sdf, C2_CIRCLES

ENV = {'x_lims': [-5.0, 5.0], 'y_lims': [-5.0, 5.0]}


def T(x): return torch.as_tensor(np.asarray(x), dtype=torch.float64)
def N(x): return x.detach().cpu().numpy().copy()


def params_2d(n, reg=0.1, max_iters=10):
  gp = {'Q_c_inv': torch.eye(2), 'K_s': torch.tensor(0.01), 'K_g': torch.tensor(0.01),
        'K_v': torch.tensor(0.01), 'v_x': [1.0], 'v_y': [1.0]}
  obs = {'cost_sigma': torch.tensor(0.01), 'epsilon_dist': torch.tensor(0.4)}
  pl = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': n - 1}
  opt = {'method': 'gauss_newton', 'reg': reg, 'plan_time': float('inf'), 'max_iters': max_iters,
         'tol_err': 1e-3, 'tol_delta': 1e-4}
  return gp, obs, pl, opt


def make_planner(B, n, reg=0.1, max_iters=10):
  gp, obs, pl, opt = params_2d(n, reg, max_iters)
  robot = PointRobot2D(torch.tensor(0.4), B, n)
  return DiffGPMP2Planner(gp, obs, pl, opt, ENV, robot, batch_size=B)


def rand_start_goal(B, seed=0):
  g = torch.Generator().manual_seed(seed)
  s = torch.cat([torch.rand(B, 1, 2, generator=g) * 8 - 4, torch.zeros(B, 1, 2)], -1)
  e = torch.cat([torch.rand(B, 1, 2, generator=g) * 8 - 4, torch.zeros(B, 1, 2)], -1)
  return s, e


def rand_covs(B, n, dof, seed):
  g = torch.Generator().manual_seed(seed)
  a = torch.randn(B, n - 1, dof, dof, generator=g) * 0.3
  qc = torch.matmul(a, a.transpose(-1, -2)) + 0.5 * torch.eye(dof)
  ow = (torch.rand(B, n, 1, 1, generator=g) * 1.5 + 0.25) * 1e4
  eps = torch.rand(B, n, 1, 1, generator=g) * 0.4 + 0.2
  return qc, ow, eps


def save(name, **kw):
  path = os.path.join(HERE, name + '.npz')
  np.savez_compressed(path, **{k: (N(v) if torch.is_tensor(v) else np.asarray(v)) for k, v in kw.items()})
  print('wrote %-28s %8.1f KB' % (name + '.npz', os.path.getsize(path) / 1024.0))


# ------------------------------------------------------------------------------------------------
# G1: factor-level goldens
# ------------------------------------------------------------------------------------------------
def g1_factors():
  B, n, dof, d = 4, 16, 2, 4
  dt = 10.0 / (n - 1)
  g = torch.Generator().manual_seed(11)
  th = torch.randn(B, n, d, generator=g) * 2.5
  # GP factor: gp_factor.py:100-110,65-73
  gpf = GPFactor(dof, dt, n - 1, batch_size=B)
  qc, ow, eps = rand_covs(B, n, dof, 12)
  gpf.set_Q_c_inv(qc)
  e_gp, H1, H2 = gpf.get_error(th)
  # prior factor: prior_factor.py:15-18
  pf = PriorFactor(d, torch.tensor(0.01), batch_size=B)
  mean = torch.randn(B, 1, d, generator=g)
  pf.set_mean(mean)
  e_p, H_p = pf.get_error(th[:, 0:1])
  # obstacle factor on a NON-square random SDF (H=40, W=32): obstacle_factor.py:35-40
  Hh, Ww = 40, 32
  sdf = torch.randn(B, 1, Hh, Ww, generator=g)
  robot = PointRobot2D(torch.tensor(0.4), B, n)
  of = ObstacleFactor(d, n, torch.tensor(0.4), ENV, robot, B)
  of.set_eps(eps)
  th_o = th.clone()
  # edge points: outside the grid, last row/col cell, exact borders
  edge = torch.tensor([[6.0, 1.0], [-7.0, 0.3], [0.3, -5.9], [0.3, 5.5], [4.95, 0.0], [0.0, -4.95],
                       [5.0, 5.0], [-5.0, -5.0], [4.6875, 4.75], [-5.0, 0.0], [0.0, 5.0], [4.99999, -4.99999]])
  th_o[0, :12, 0:2] = edge
  e_o, H_o = of.get_error(th_o, sdf)
  res = 10.0 / Ww
  d_bi, J_bi = bilinear_interpolate(sdf[:, 0], th_o[:, :, 0:2].contiguous(), res, ENV['x_lims'], ENV['y_lims'])
  # hinge tie (Q5): constant SDF equal to eps+r
  sdf_tie = torch.full((B, 1, 16, 16), 0.8)
  eps_tie = torch.full((B, n, 1, 1), 0.4)
  of.set_eps(eps_tie)
  e_t, H_t = of.get_error(th, sdf_tie)
  save('g1_factors_2d', th=th, dt=dt, qc=qc, Q_inv=gpf.get_inv_cov_full(), e_gp=e_gp, H1=H1, H2=H2,
       mean=mean, e_p=e_p, H_p=H_p, sdf=sdf, eps=eps, th_o=th_o, e_o=e_o, H_o=H_o, d_bi=d_bi, J_bi=J_bi,
       sdf_tie=sdf_tie, eps_tie=eps_tie, e_t=e_t, H_t=H_t)


def _patched_vel_factor_cls():
  """VelocityLimitFactor (velocity_limit_factor.py) divides a tensor by 2 with py2 integer semantics;
  emulate py2 by patching '/2' -> '//2' in an in-memory copy of the source (nothing is written)."""
  src = open(os.path.join(REF, 'diff_gpmp2/gpmp2/custom_factors/velocity_limit_factor.py')).read()
  src = src.replace('self.ndims/2', 'int(self.ndims)//2')
  ns = {}
  exec(compile(src, 'velocity_limit_factor_py2', 'exec'), ns)
  return ns['VelocityLimitFactor']


def g1_custom():
  n = 12
  g = torch.Generator().manual_seed(21)
  # velocity limit (unbatched, (n,4)): velocity_limit_factor.py:17-29
  VLF = _patched_vel_factor_cls()
  vf = VLF(4, n, torch.tensor(0.01), 1)
  vf.set_v_traj(torch.tensor([1.0]).unsqueeze(0).expand(n, 1), torch.tensor([1.0]).unsqueeze(0).expand(n, 1))
  tr = torch.randn(n, 4, generator=g) * 1.2
  tr[0, 2:] = torch.tensor([0.5, -2.0]); tr[1, 2:] = torch.tensor([1.0, -1.0]); tr[2, 2:] = torch.tensor([-1.0, 1.0])
  c_v, H_v = vf.get_error_full(tr)
  # non-holonomic (unbatched, (n,6)): nonholonomic_factor.py:16-30
  nh = NonHolonomicFactor(3, torch.tensor(0.01), n, 1)
  tr6 = torch.randn(n, 6, generator=g) * 1.5
  tr6[0] = torch.tensor([0., 0., .3, 1., .5, .1])
  e_d, H_d = nh.get_error_full(tr6)
  save('g1_factors_custom', tr=tr, c_v=c_v, H_v=H_v, w_v=vf.get_inv_cov_full(), tr6=tr6, e_d=e_d, H_d=H_d,
       w_d=nh.get_inv_cov_full())


# ------------------------------------------------------------------------------------------------
# G2: assembled normal equations from the reference's construct_linear_system_batch
# ------------------------------------------------------------------------------------------------
def g2_system():
  for n in (4, 16, 64):
    B = 3
    planner = make_planner(B, n)
    pl = planner.plan_layer
    start, goal = rand_start_goal(B, seed=n)
    g = torch.Generator().manual_seed(100 + n)
    th = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + torch.randn(B, n, 4, generator=g) * 0.3
    sdf_np = circles_sdf(64, C2_CIRCLES)
    sdf = T(sdf_np)[None, None].repeat(B, 1, 1, 1)
    qc, ow, eps = rand_covs(B, n, 2, 200 + n)
    pl.start_prior.set_mean(start); pl.goal_prior.set_mean(goal)
    pl.gp_prior.set_Q_c_inv(qc); pl.obs_factor.set_inv_cov(ow); pl.obs_factor.set_eps(eps)
    A, b, K = pl.construct_linear_system_batch(th, sdf)
    AtK = torch.bmm(A.transpose(1, 2), K)
    LAM = torch.bmm(AtK, A) + 0.1 * torch.eye(pl.N)[None]
    R = torch.bmm(AtK, b)
    d = 4
    Dg = torch.stack([LAM[:, i * d:(i + 1) * d, i * d:(i + 1) * d] for i in range(n)], 1)
    Up = torch.stack([LAM[:, i * d:(i + 1) * d, (i + 1) * d:(i + 2) * d] for i in range(n - 1)], 1)
    save('g2_system_n%d' % n, th=th, start=start, goal=goal, G=64, circles=np.asarray(C2_CIRCLES), qc=qc, ow=ow,
         eps=eps, Dg=Dg, Up=Up, eta=R.view(B, n, d), M=pl.M, bnorm=torch.norm(b), Anorm=torch.norm(A), Knorm=torch.norm(K))


# ------------------------------------------------------------------------------------------------
# G3/G4: one step and 10 teacher-forced steps (planner.step), C1 real map and C2-shaped mini batch
# ------------------------------------------------------------------------------------------------
def c1_inputs(n):
  im = plt.imread(os.path.join(REF, 'diff_gpmp2/env/simple_2d/5.png'))
  if im.ndim > 2: im = np.dot(im[..., :3], [0.299, 0.587, 0.114])
  cell = 10.0 / im.shape[0]
  sdf = sdf_2d(im, res=cell)                           # padlen=1 -> 202x202 (Q3)
  start = torch.tensor([[[-4.0, -4.0, 0.0, 0.0]]]); goal = torch.tensor([[[4.0, 4.0, 0.0, 0.0]]])
  th = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  return T(im)[None, None], T(sdf)[None, None], start, goal, th


def g3_c1():
  out = {}
  for n in (32, 33, 101):
    im, sdf, start, goal, th = c1_inputs(n)
    planner = make_planner(1, n)
    dth, _, err, err_ext, _, _, _ = planner.step(th, start, goal, im, sdf)
    out['n%d_err0' % n] = err; out['n%d_dth0' % n] = dth; out['n%d_errext0' % n] = err_ext
    if n == 32:
      ths = [th]; errs = []; errexts = []; dths = []
      for k in range(10):
        dth, _, err, err_ext, _, _, _ = planner.step(ths[-1], start, goal, im, sdf)
        dths.append(dth); errs.append(err); errexts.append(err_ext); ths.append(ths[-1] + dth)
      out.update(sdf=sdf[0, 0], start=start, goal=goal, th_hist=torch.stack(ths, 0), dth_hist=torch.stack(dths, 0),
                 err_hist=torch.stack(errs, 0), errext_hist=torch.stack(errexts, 0),
                 err_after10=planner.error_batch(ths[-1], sdf))
      usg, ugp, uobs = planner.unweighted_errors_batch(ths[3], sdf)
      out.update(unw_sg=usg, unw_gp=ugp, unw_obs=uobs)
  save('g3_c1', **out)


def g3_c2mini():
  B, n, Gsz = 8, 64, 256
  start, goal = rand_start_goal(B, seed=0)
  th0 = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  sdf = T(circles_sdf(Gsz, C2_CIRCLES))[None, None].repeat(B, 1, 1, 1)
  im = (sdf > 0).double()
  planner = make_planner(B, n)
  out = dict(start=start, goal=goal, G=Gsz, circles=np.asarray(C2_CIRCLES))
  # static covariances, 10 teacher-forced steps
  ths = [th0]; dths = []; errs = []; errexts = []
  for k in range(10):
    dth, _, err, err_ext, qc_s, ow_s, eps_s = planner.step(ths[-1], start, goal, im, sdf)
    dths.append(dth); errs.append(err); errexts.append(err_ext); ths.append(ths[-1] + dth)
  out.update(th_hist=torch.stack(ths, 0), dth_hist=torch.stack(dths, 0), err_hist=torch.stack(errs, 0),
             errext_hist=torch.stack(errexts, 0))
  # random SPD per-state covariances through PlanLayer.forward (plan_layer.py:87-99)
  qc, ow, eps = rand_covs(B, n, 2, 7)
  th1 = ths[2]
  dth, err, err_ext = planner.plan_layer(th1, start, goal, im, sdf, qc, ow, eps)
  out.update(cov_qc=qc, cov_ow=ow, cov_eps=eps, cov_th=th1, cov_dth=dth, cov_err=err, cov_errext=err_ext)
  # per-sample SDFs (3 random circles each, G=96)
  g = torch.Generator().manual_seed(1)
  cc = torch.rand(B, 3, 2, generator=g) * 7 - 3.5
  rr = torch.rand(B, 3, 1, generator=g) * 0.6 + 0.4
  circ = torch.cat([cc, rr], -1)
  sdf_ps = torch.stack([T(circles_sdf(96, [tuple(c) for c in N(circ[b])])) for b in range(B)], 0)[:, None]
  dth, _, err, err_ext, _, _, _ = planner.step(th0, start, goal, (sdf_ps > 0).double(), sdf_ps)
  out.update(ps_circles=circ, ps_G=96, ps_dth=dth, ps_err=err, ps_errext=err_ext)
  save('g3_c2mini', **out)


# ------------------------------------------------------------------------------------------------
# forward() (GN to convergence) on the C1 plumbing and a small batch
# ------------------------------------------------------------------------------------------------
def g4_forward():
  n = 32
  im, sdf, start, goal, th = c1_inputs(n)
  planner = make_planner(1, n, max_iters=12)
  import io, contextlib
  with contextlib.redirect_stdout(io.StringIO()):
    thf, _, e_init, e_final, e_iter, ee_iter, k, _ = planner.forward(th, start, goal, im, sdf)
  out = dict(c1_th_final=thf, c1_err_init=e_init, c1_err_final=e_final, c1_err_iter=np.asarray(e_iter),
             c1_errext_iter=np.asarray(ee_iter), c1_iters=k, c1_max_iters=12, c1_tol_delta=1e-4)
  # obstacle-free map => converges by tol_delta in a few iterations (exercises the early exit)
  B, n2 = 3, 16
  start2, goal2 = rand_start_goal(B, seed=5)
  g = torch.Generator().manual_seed(6)
  th2 = straight_line_trajb(start2[:, :, :2], goal2[:, :, :2], 10.0, n2 - 1, 2) + torch.randn(B, n2, 4, generator=g) * 0.2
  sdf2 = torch.full((B, 1, 32, 32), 3.0)
  gp, obs, plp, opt = params_2d(n2, max_iters=20)
  opt['tol_delta'] = 2e-3
  pl2 = DiffGPMP2Planner(gp, obs, plp, opt, ENV, PointRobot2D(torch.tensor(0.4), 1, n2), batch_size=1)
  with contextlib.redirect_stdout(io.StringIO()):
    thf2, _, ei2, ef2, eit2, eeit2, k2, _ = pl2.forward(th2, start2, goal2, (sdf2 > 0).double(), sdf2)
  maxlen = max(len(e) for e in eit2)
  pad = lambda L: np.asarray([list(e) + [np.nan] * (maxlen - len(e)) for e in L])
  out.update(free_th0=th2, free_start=start2, free_goal=goal2, free_th_final=thf2, free_err_init=ei2, free_err_final=ef2,
             free_err_iter=pad(eit2), free_errext_iter=pad(eeit2), free_iters=k2, free_max_iters=20, free_tol_delta=2e-3)
  save('g4_forward', **out)


# ------------------------------------------------------------------------------------------------
# G5: autograd through one step (reference autograd over plan_layer.py:152-234)
# ------------------------------------------------------------------------------------------------
def g5_grads():
  B, n, Gsz = 4, 16, 48
  start, goal = rand_start_goal(B, seed=3)
  g = torch.Generator().manual_seed(31)
  th = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + torch.randn(B, n, 4, generator=g) * 0.2
  circ = ((-1.0, -1.0, 1.5), (2.0, 1.5, 1.2), (0.5, -2.5, 1.0))
  sdf = T(circles_sdf(Gsz, circ))[None, None].repeat(B, 1, 1, 1)
  qc, ow, eps = rand_covs(B, n, 2, 32)
  gbar = torch.randn(B, n, 4, generator=g)
  gext = torch.randn(B, 1, 1, generator=g)
  planner = make_planner(B, n)
  leaves = [x.clone().requires_grad_(True) for x in (th, sdf, start, goal, qc, ow, eps)]
  dth, err, err_ext = planner.plan_layer(leaves[0], leaves[2], leaves[3], (sdf > 0).double(), leaves[1], leaves[4], leaves[5], leaves[6])
  loss = (gbar * dth).sum()
  grads = torch.autograd.grad(loss, leaves, retain_graph=True, allow_unused=True)
  loss_e = (gext * err_ext).sum()
  grads_e = torch.autograd.grad(loss_e, leaves, allow_unused=True)
  z = lambda gr, x: torch.zeros_like(x) if gr is None else gr
  names = ('th', 'sdf', 'start', 'goal', 'qc', 'ow', 'eps')
  out = dict(th=th, G=Gsz, circles=np.asarray(circ), start=start, goal=goal, qc=qc, ow=ow, eps=eps, gbar=gbar, gext=gext,
             dth=dth, err=err, err_ext=err_ext, err_requires_grad=bool(err.requires_grad))
  for nm, gr, ge, x in zip(names, grads, grads_e, leaves):
    out['g_' + nm] = z(gr, x); out['ge_' + nm] = z(ge, x)
    out['ge_none_' + nm] = ge is None
  save('g5_grads', **out)


# ------------------------------------------------------------------------------------------------
# C3 / C4: the reference's batched path is broken for these; goldens are built from
# the reference's own UNBATCHED factor functions + its own masks (create_factor_masks), per trajectory.
# ------------------------------------------------------------------------------------------------
def _dense_from_ref(pl, th1, start1, goal1, sdf1, e_extra, H_extra, w_extra, kind):
  """Assemble A,b,K for ONE trajectory with the reference's masks; base factors via the reference's
  batched functions on a batch of 1, extra (vel/dyn) factor rows from the unbatched reference outputs."""
  A = torch.zeros(1, pl.M, pl.N); b = torch.zeros(1, pl.M, 1); K = torch.zeros(1, pl.M, pl.M)
  e_p, H_p = pl.start_prior.get_error(th1[:, 0:1]); e_g, H_g = pl.goal_prior.get_error(th1[:, -1:])
  e_gp, H1, H2 = pl.gp_prior.get_error(th1)
  A.masked_scatter_(pl.mask_Astart[None], H_p); b.masked_scatter_(pl.mask_bstart[None], e_p)
  K.masked_scatter_(pl.mask_Kstart[None], pl.start_prior.get_inv_cov()[0:1])
  A.masked_scatter_(pl.mask_A1gp[None], H1); A.masked_scatter_(pl.mask_A2gp[None], H2)
  b.masked_scatter_(pl.mask_bgp[None], e_gp); K.masked_scatter_(pl.mask_Kgp[None], pl.gp_prior.get_inv_cov_full()[0:1])
  A.masked_scatter_(pl.mask_Agoal[None], H_g); b.masked_scatter_(pl.mask_bgoal[None], e_g)
  K.masked_scatter_(pl.mask_Kgoal[None], pl.goal_prior.get_inv_cov()[0:1])
  e_o, H_o = sdf1
  A.masked_scatter_(pl.mask_Aobs[None], H_o); b.masked_scatter_(pl.mask_bobs[None], e_o)
  K.masked_scatter_(pl.mask_Kobs[None], pl.obs_factor.get_inv_cov_full()[0:1])
  mA, mb, mK = (pl.mask_Adyn, pl.mask_bdyn, pl.mask_Kdyn) if kind == 'dyn' else (pl.mask_Avel, pl.mask_bvel, pl.mask_Kvel)
  A.masked_scatter_(mA[None], H_extra); b.masked_scatter_(mb[None], e_extra); K.masked_scatter_(mK[None], w_extra)
  return A, b, K


def g3_c3_vel():
  """2D point robot + velocity-limit factors (config 3), B=4, n=16."""
  B, n = 4, 16
  gp, obs, plp, opt = params_2d(n)
  plp['use_vel_limits'] = True
  VLF = _patched_vel_factor_cls()
  import diff_gpmp2.gpmp2.plan_layer as plmod
  plmod.VelocityLimitFactor = VLF            # py2-semantics copy (in memory only)
  gp['v_x'] = [1.0]; gp['v_y'] = [1.0]
  robot = PointRobot2D(torch.tensor(0.4), 1, n)
  pl = PlanLayer(gp, obs, plp, opt, ENV, robot, None, 1, False)
  start, goal = rand_start_goal(B, seed=9)
  g = torch.Generator().manual_seed(91)
  th = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + torch.randn(B, n, 4, generator=g) * 0.4
  th[:, :, 2:] = th[:, :, 2:] * 2.0          # push some velocities past the 1.0 limit
  sdf = T(circles_sdf(64, C2_CIRCLES))[None, None]
  qc1 = torch.eye(2)[None, None].repeat(1, n - 1, 1, 1)
  ow1 = torch.full((1, n, 1, 1), 1e4); eps1 = torch.full((1, n, 1, 1), 0.4)
  dths, errs = [], []
  for i in range(B):
    t1 = th[i:i + 1]
    pl.start_prior.set_mean(start[i:i + 1]); pl.goal_prior.set_mean(goal[i:i + 1])
    pl.gp_prior.set_Q_c_inv(qc1); pl.obs_factor.set_inv_cov(ow1); pl.obs_factor.set_eps(eps1)
    eo = pl.obs_factor.get_error(t1, sdf)
    c_v, H_v = pl.vel_factor.get_error_full(t1[0])
    A, b, K = _dense_from_ref(pl, t1, None, None, eo, c_v, H_v, pl.vel_factor.get_inv_cov_full(), 'vel')
    dth = pl.solve_linear_system_batch(A, b, K, delta=0.1)
    err = 0.5 * torch.bmm(torch.bmm(b.transpose(1, 2), K), b) / pl.M      # == error_batch's sum (plan_layer.py:273-308)
    dths.append(dth); errs.append(err)
  save('g3_c3_vel', th=th, start=start, goal=goal, G=64, circles=np.asarray(C2_CIRCLES), dth=torch.cat(dths, 0),
       err=torch.cat(errs, 0), M=pl.M)


def g3_c4_xyh():
  """Non-holonomic (x,y,theta) robot, d=6 (config 4), B=4, n=16.  PointRobotXYH has no batched sphere
  model in the reference, so the obstacle rows come from the reference's
  hinge_loss_signed_batch on state[0:2] with H_fk = I_6[0:2,:] (point_robot_xyh.py:28-36)."""
  from diff_gpmp2.gpmp2.obstacle.obstacle_cost import HingeLossObstacleCost
  B, n, dof, d = 4, 16, 3, 6
  gp = {'Q_c_inv': torch.eye(3), 'K_s': torch.tensor(0.01), 'K_g': torch.tensor(0.01), 'K_d': torch.tensor(0.01)}
  obs = {'cost_sigma': torch.tensor(0.01), 'epsilon_dist': torch.tensor(0.2)}
  plp = {'dof': 3, 'state_dim': 6, 'total_time_sec': 10.0, 'total_time_step': n - 1, 'non_holonomic': True}
  opt = {'method': 'gauss_newton', 'reg': 0.0, 'plan_time': float('inf'), 'max_iters': 10, 'tol_err': 1e-4, 'tol_delta': 1e-3}

  class _XYH(object):      # nlinks/sphere radius only -- what PlanLayer.__init__ touches (plan_layer.py:42)
    nlinks = 1
    def get_sphere_radii(self): return torch.tensor(0.4)
  pl = PlanLayer(gp, obs, plp, opt, ENV, _XYH(), None, 1, False)
  g = torch.Generator().manual_seed(41)
  sp = torch.rand(B, 1, 2, generator=g) * 8 - 4; gl = torch.rand(B, 1, 2, generator=g) * 8 - 4
  start = torch.cat([sp, torch.zeros(B, 1, 1), torch.zeros(B, 1, 3)], -1)
  goal = torch.cat([gl, torch.full((B, 1, 1), np.pi / 2), torch.zeros(B, 1, 3)], -1)
  th = straight_line_trajb(start[:, :, :3], goal[:, :, :3], 10.0, n - 1, 3) + torch.randn(B, n, 6, generator=g) * 0.2
  Gsz = 128
  sdf = T(circles_sdf(Gsz, C2_CIRCLES))[None, None]
  hl = HingeLossObstacleCost(ENV)
  qc1 = torch.eye(3)[None, None].repeat(1, n - 1, 1, 1)
  ow1 = torch.full((1, n, 1, 1), 1e4); eps1 = torch.full((1, n, 1, 1), 0.2)
  H_fk = torch.zeros(2, 6); H_fk[0, 0] = 1; H_fk[1, 1] = 1
  dths, errs = [], []
  for i in range(B):
    t1 = th[i:i + 1]
    pl.start_prior.set_mean(start[i:i + 1]); pl.goal_prior.set_mean(goal[i:i + 1])
    pl.gp_prior.set_Q_c_inv(qc1); pl.obs_factor.set_inv_cov(ow1)
    e_o, H_e = hl.hinge_loss_signed_batch(t1[:, :, 0:2].reshape(1, n, 1, 2), torch.tensor(0.4), eps1, sdf)
    H_o = torch.einsum('bsij,jk->bsik', H_e, H_fk)
    e_d, H_d = pl.dyn_factor.get_error_full(t1[0])
    A, b, K = _dense_from_ref(pl, t1, None, None, (e_o, H_o), e_d, H_d, pl.dyn_factor.get_inv_cov_full(), 'dyn')
    dth = pl.solve_linear_system_batch(A, b, K, delta=0.0)
    err = 0.5 * torch.bmm(torch.bmm(b.transpose(1, 2), K), b) / pl.M
    dths.append(dth); errs.append(err)
  save('g3_c4_xyh', th=th, start=start, goal=goal, G=Gsz, circles=np.asarray(C2_CIRCLES), dth=torch.cat(dths, 0),
       err=torch.cat(errs, 0), M=pl.M)


# ------------------------------------------------------------------------------------------------
# G6: caller-side helpers and the data path
#   utils/planner_utils.py:18-56 (check_convergence_batch incl. its overwritten-where quirk, straight_line_traj[b]),
#   utils/sdf_utils.py:6-21 (sdf_2d) and datasets/utils.py:4-18 (the dataset tools' sdf_2d), datasets/planning_dataset.py:15-69
#   reading a mini dataset written with the reference's own writer conventions
#   (generate_2d_im_dataset.py:84-89: plt.imsave(cmap=gray) + np.save(sdf); generate_optimal_paths_gpmp2.py:198-206:
#   np.savez(start, goal, th_opt) + yaml.dump(meta)).
# ------------------------------------------------------------------------------------------------
def g6_helpers():
  from diff_gpmp2.utils.planner_utils import check_convergence_batch, check_convergence, straight_line_traj
  import contextlib, io
  out = {}
  g = torch.Generator().manual_seed(61)
  B = 6
  dthb = torch.randn(B, 16, 4, generator=g) * torch.tensor([1e-6, 1e-3, 1.0, 1e-6, 1e-3, 1.0]).view(B, 1, 1)
  errd = torch.randn(B, 1, 1, generator=g) * torch.tensor([1e-5, 1.0, 1e-5, 1.0, 1e-5, 1.0]).view(B, 1, 1)
  out.update(ccb_dth=dthb, ccb_errd=errd, ccb_tol_err=1e-3, ccb_tol_delta=1e-4, ccb_max_iters=10)
  with contextlib.redirect_stdout(io.StringIO()):
    for j in (3, 10, 11):
      c = check_convergence_batch(dthb, j, errd, 1e-3, 1e-4, 10)
      out['ccb_conv_j%d' % j] = N(c).astype(np.int64); out['ccb_dtype_j%d' % j] = str(c.dtype); out['ccb_shape_j%d' % j] = np.asarray(c.shape)
    out['cc_scalar'] = np.asarray([[int(check_convergence(dthb[b], j, errd[b], 1e-3, 1e-4, 10)) for j in (3, 10)] for b in range(B)])
  s, e = rand_start_goal(5, seed=62)
  for n, dof in ((32, 2), (7, 2), (64, 3)):
    ss = torch.cat([s[:, :, :2], torch.rand(5, 1, dof - 2, generator=g)], -1) if dof > 2 else s[:, :, :2]
    ee = torch.cat([e[:, :, :2], torch.rand(5, 1, dof - 2, generator=g)], -1) if dof > 2 else e[:, :, :2]
    out['sl_start_n%d_dof%d' % (n, dof)] = ss; out['sl_goal_n%d_dof%d' % (n, dof)] = ee
    out['sl_thb_n%d_dof%d' % (n, dof)] = straight_line_trajb(ss, ee, 10.0, n - 1, dof)
    out['sl_th1_n%d_dof%d' % (n, dof)] = straight_line_traj(ss[0], ee[0], 10.0, n - 1, dof)
  # sdf_2d on the real map of BASELINE config 1 and on a small random occupancy image, both padding conventions
  im5 = plt.imread(os.path.join(REF, 'diff_gpmp2/env/simple_2d/5.png'))
  if im5.ndim > 2: im5 = np.dot(im5[..., :3], [0.299, 0.587, 0.114])
  rs = np.random.RandomState(63)
  imr = (rs.rand(24, 40) > 0.3).astype(np.float64)
  out.update(sdf_im5=im5.astype(np.float32), sdf_im5_pad1=sdf_2d(im5, res=10.0 / im5.shape[0]), sdf_imr=imr,
             sdf_imr_pad0=sdf_2d(imr, padlen=0, res=0.25), sdf_imr_pad2=sdf_2d(imr, padlen=2, res=1.0))
  # the learned-covariance plumbing: get_covariances for every runnable mode x learn_eps, get_obs_covariance
  # (diff_gpmp2_planner.py:247-297), on random predictor outputs
  n, Bc = 16, 3
  planner = make_planner(Bc, n)
  gen = torch.Generator().manual_seed(64)
  sizes = {'fix_dynamics': 0, 'diag_identity': n - 1, 'qc_full': (n - 1) * 2, 'q_full': (n - 1) * 4}
  for mode, n_gp in sizes.items():
    for le in (False, True):
      key = 'cov_%s_%d' % (mode, int(le))
      o = torch.randn(Bc, 1, n_gp + n + (n if le else 0), generator=gen)
      res = planner.get_covariances(o, mode, le)
      res = res if isinstance(res, tuple) else (res,)
      out[key + '_in'] = o; out[key + '_count'] = len(res)
      for i, t in enumerate(res): out['%s_out%d' % (key, i)] = t
  o = torch.randn(n, generator=gen)
  out.update(obscov_in=o, obscov_out=planner.get_obs_covariance(o))
  # what the constructor leaves in learn_params before it builds the learn modules (diff_gpmp2_planner.py:58-78), every combination of
  # dynamics_mode x learn_eps x dtheta_predict; the modules themselves replaced by stubs (learn_module_fcn.py:41 needs Python 2)
  import copy
  import diff_gpmp2.gpmp2.diff_gpmp2_planner as pmod
  import torch.nn as nn
  saved = (pmod.LearnModuleConv, pmod.LearnModuleFCN)
  pmod.LearnModuleConv = lambda lp_, *a, **k: nn.Identity()
  pmod.LearnModuleFCN = lambda lp_, *a, **k: nn.Identity()
  try:
    rows = []
    for mode in ('fix_dynamics', 'diag_identity', 'qc_full', 'q_full'):
      for le in (False, True):
        for dp in (False, True):
          lp = copy.deepcopy(TBPTT_LEARN_PARAMS)
          lp['dgpmp2'].update(dynamics_mode=mode, learn_eps=le, dtheta_predict=dp)
          gp_, obs_, pl_, opt_ = params_2d(n)
          pln = DiffGPMP2Planner(gp_, obs_, pl_, opt_, ENV, PointRobot2D(torch.tensor(0.4), 2, n), learn_params=lp, batch_size=2)
          rows.append([lp['num_traj_states'], lp['state_dim'], lp['out_dim']])
    out.update(lp_prepared=np.asarray(rows), lp_res=float(pln.res))
  finally:
    pmod.LearnModuleConv, pmod.LearnModuleFCN = saved
  save('g6_helpers', **out)


def g6_dataset():
  """Writes tests/golden/mini_dataset/ with the reference's writer conventions and stores what the REFERENCE PlanningDataset
  reads back from it (shim: yaml.load without a Loader argument, planning_dataset.py:32, fails on PyYAML >= 6)."""
  import shutil, yaml
  import matplotlib.cm as cm
  from diff_gpmp2.datasets.utils import sdf_2d as ds_sdf_2d
  root = os.path.join(HERE, 'mini_dataset')
  shutil.rmtree(root, ignore_errors=True)
  folder = os.path.join(root, 'train')
  os.makedirs(os.path.join(folder, 'im_sdf'))
  n, num_envs, ppe = 16, 2, 2
  planner = make_planner(1, n, max_iters=5)
  for i, name in enumerate(('5.png', '7.png')):
    im = plt.imread(os.path.join(REF, 'diff_gpmp2/env/simple_2d', name))
    if im.ndim > 2: im = np.dot(im[..., :3], [0.299, 0.587, 0.114])
    im = np.ascontiguousarray(im[::4, ::4])                                        # 50 x 50: keeps the fixture small
    sdf = ds_sdf_2d(im) * (10.0 / im.shape[0])                                     # generate_2d_im_dataset.py:84 (pad 1) in metres
    plt.imsave(os.path.join(folder, 'im_sdf', '%d_im.png' % i), im, cmap=cm.gray)    # generate_2d_im_dataset.py:88
    np.save(os.path.join(folder, 'im_sdf', '%d_sdf.npy' % i), sdf)                   # :89
    for j in range(ppe):
      s, e = rand_start_goal(1, seed=70 + 2 * i + j)
      th_init = straight_line_trajb(s[:, :, :2], e[:, :, :2], 10.0, n - 1, 2)
      imp = torch.tensor(im); sdfp = torch.tensor(sdf)
      import contextlib, io
      with contextlib.redirect_stdout(io.StringIO()):
        res = planner.forward(th_init, s, e, imp.unsqueeze(0).unsqueeze(0), sdfp.unsqueeze(0).unsqueeze(0))
      os.makedirs(os.path.join(folder, 'opt_trajs_gpmp2'), exist_ok=True)
      np.savez(os.path.join(folder, 'opt_trajs_gpmp2', 'env_%d_prob_%d' % (i, j)), start=N(s)[0, 0], goal=N(e)[0, 0],
               th_opt=N(res[0])[0])                                                # generate_optimal_paths_gpmp2.py:192-198
  with open(os.path.join(folder, 'meta.yaml'), 'w') as fp:                         # generate_optimal_paths_gpmp2.py:201-206
    yaml.dump({'num_envs': num_envs, 'probs_per_env': ppe, 'env_params': ENV, 'im_size': 50}, fp)
  _load = yaml.load
  yaml.load = lambda f, Loader=None: _load(f, Loader=Loader or yaml.FullLoader)
  import contextlib, io
  try:
    from diff_gpmp2.datasets.planning_dataset import PlanningDataset
    with contextlib.redirect_stdout(io.StringIO()):
      ds = PlanningDataset(root, mode='train')
      sub = PlanningDataset(root, mode='train', num_envs=1, num_env_probs=1)
    out = {'len': len(ds), 'len_sub': len(sub)}
    for k in range(len(ds)):
      smp = ds[k]
      for key in ('im', 'sdf', 'start', 'goal', 'th_opt'):
        out['s%d_%s' % (k, key)] = smp[key]; out['s%d_%s_dtype' % (k, key)] = str(smp[key].dtype)
  finally:
    yaml.load = _load
  save('g6_dataset', **out)


# ------------------------------------------------------------------------------------------------
# G7: autograd through the errors the reference's TRAINING LOSS differentiates besides dtheta:
#   unweighted_errors_batch (diff_gpmp2_planner.py:229-237 -> plan_layer.py:374-388) and error_ext_batch (:310-345), evaluated at
#   th + dtheta (learning/train_planner.py:313,327), with start / goal / eps remembered WITH their graphs by forward() (:88-94).
#   g7_errors: plan_layer-level, every leaf learnable (incl. eps);  g7_tbptt: one batch of train() (train_planner.py:258-424, the
#   reference's own loop text exec'd from /root/reference) through planner.step() with learn modules -- the stubs of tests/tbptt_driver.py
#   injected in place of LearnModuleConv / LearnModuleFCN (the latter cannot be constructed under Python 3, learn_module_fcn.py:41), in memory only.
# ------------------------------------------------------------------------------------------------
def g7_errors():
  B, n, Gsz = 4, 16, 48
  start, goal = rand_start_goal(B, seed=13)
  g = torch.Generator().manual_seed(71)
  th = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + torch.randn(B, n, 4, generator=g) * 0.2
  th_eval = th + torch.randn(B, n, 4, generator=g) * 0.1
  circ = ((-1.0, -1.0, 1.5), (2.0, 1.5, 1.2), (0.5, -2.5, 1.0))
  sdf = T(circles_sdf(Gsz, circ))[None, None].repeat(B, 1, 1, 1)
  qc, ow, eps = rand_covs(B, n, 2, 72)
  c_sg = torch.randn(B, 1, generator=g); c_gp = torch.randn(B, 1, 1, generator=g); c_obs = torch.randn(B, 1, 1, generator=g)
  c_ee = torch.randn(B, 1, 1, generator=g)
  planner = make_planner(B, n)
  names = ('th', 'sdf', 'start', 'goal', 'qc', 'ow', 'eps')
  z = lambda gr, x: torch.zeros_like(x) if gr is None else gr
  out = dict(th=th, th_eval=th_eval, G=Gsz, circles=np.asarray(circ), start=start, goal=goal, qc=qc, ow=ow, eps=eps,
             c_sg=c_sg, c_gp=c_gp, c_obs=c_obs, c_ee=c_ee)
  # (a) the errors at a trajectory that is a LEAF (isolates the backward of the error evaluation itself)
  leaves = [x.clone().requires_grad_(True) for x in (th, sdf, start, goal, qc, ow, eps)]
  the = th_eval.clone().requires_grad_(True)
  planner.plan_layer(leaves[0], leaves[2], leaves[3], (sdf > 0).double(), leaves[1], leaves[4], leaves[5], leaves[6])
  e_sg, e_gp, e_obs = planner.unweighted_errors_batch(the, leaves[1])
  e_ee = planner.error_ext_batch(the, leaves[1])
  out.update(a_sg=e_sg, a_gp=e_gp, a_obs=e_obs, a_ee=e_ee)
  for tag, loss in (('unw', (c_sg * e_sg).sum() + (c_gp * e_gp).sum() + (c_obs * e_obs).sum()), ('ee', (c_ee * e_ee).sum())):
    gr = torch.autograd.grad(loss, [the] + leaves, retain_graph=True, allow_unused=True)
    out['a_%s_g_th_eval' % tag] = z(gr[0], the)
    for nm, gk, x in zip(names, gr[1:], leaves):
      out['a_%s_g_%s' % (tag, nm)] = z(gk, x); out['a_%s_none_%s' % (tag, nm)] = gk is None
  # (b) the training-loop composition: errors at th + dtheta (train_planner.py:313,327)
  leaves = [x.clone().requires_grad_(True) for x in (th, sdf, start, goal, qc, ow, eps)]
  dth, err, err_ext = planner.plan_layer(leaves[0], leaves[2], leaves[3], (sdf > 0).double(), leaves[1], leaves[4], leaves[5], leaves[6])
  th_new = leaves[0] + dth
  e_sg, e_gp, e_obs = planner.unweighted_errors_batch(th_new, leaves[1])
  e_ee = planner.error_ext_batch(th_new, leaves[1])
  out.update(b_dth=dth, b_sg=e_sg, b_gp=e_gp, b_obs=e_obs, b_ee=e_ee)
  loss = (c_sg * e_sg).sum() + (c_gp * e_gp).sum() + (c_obs * e_obs).sum() + (c_ee * e_ee).sum()
  gr = torch.autograd.grad(loss, leaves, allow_unused=True)
  for nm, gk, x in zip(names, gr, leaves):
    out['b_g_' + nm] = z(gk, x)
  save('g7_errors', **out)


TBPTT_LEARN_PARAMS = {
    'model': {'type': 'feed_forward'},
    'dgpmp2': {'learn_eps': False, 'sdf_predict': True, 'dtheta_predict': False, 'fixed_conv': False, 'T': 4, 'tk': 2, 'tk2': 2,
               'use_inter_loss': True, 'optimize_tk': False},
    'data': {'im_size': 48},
    'optim': {'vel_loss_lambda': 0.5, 'ext_obs_lambda': 2.0, 'ext_loss_weight': 0.3, 'batch_size': 3, 'do_validation': False},
}


def _reference_training_text():
  """(one_step_loss source, the per-batch body of train()) read from the reference's learning/train_planner.py at generation time.
  The file as a whole is Python 2 (print statements elsewhere) and cannot be imported, but these two spans parse under Python 3
  (`xrange` is supplied by the namespace they are exec'd in).  Nothing of the text is written anywhere."""
  import textwrap
  lines = open(os.path.join(REF, 'diff_gpmp2/learning/train_planner.py')).read().split('\n')
  i0 = next(i for i, l in enumerate(lines) if l.startswith('def one_step_loss'))
  i1 = next(i for i, l in enumerate(lines) if i > i0 and l.startswith('def '))
  j0 = next(i for i, l in enumerate(lines) if 'in enumerate(train_loader' in l)
  j1 = next(i for i, l in enumerate(lines) if i > j0 and "print('Updating parameters')" in l) + 1      # ... through the optimizer.step() behind it
  return '\n'.join(lines[i0:i1]), textwrap.dedent('\n'.join(lines[j0:j1 + 1]))


class _NullOptimizer(object):
  """The fixture pins the gradients the loop deposits, not a parameter update: zero_grad() / step() do nothing."""
  def zero_grad(self): pass
  def step(self): pass


def g7_tbptt():
  """One batch of the reference's train() -- its own loop text, exec'd -- on the reference's planner with stub learn modules
  (tests/tbptt_driver.py: the real LearnModuleFCN cannot be constructed under Python 3, learn_module_fcn.py:41): feed-forward model
  in two dynamics modes and a recurrent model (diff_gpmp2_planner.py:192, train_planner.py:282-284,303-309,315,369-371)."""
  import copy, contextlib, io, time
  import torch.nn as nn
  import diff_gpmp2.gpmp2.diff_gpmp2_planner as pmod
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import tbptt_driver as TD
  loss_src, loop_src = _reference_training_text()
  B, n, Gsz = 3, 16, 48
  out = {}
  saved = (pmod.LearnModuleConv, pmod.LearnModuleFCN)
  try:
    for tag, mode, mtype in (('fix_dynamics', 'fix_dynamics', 'feed_forward'), ('qc_full', 'qc_full', 'feed_forward'), ('recurrent', 'fix_dynamics', 'recurrent')):
      lp = copy.deepcopy(TBPTT_LEARN_PARAMS)
      lp['dgpmp2']['dynamics_mode'] = mode
      lp['model']['type'] = mtype
      pmod.LearnModuleConv = lambda lp_, *a, **k: TD.ConvStub()
      pmod.LearnModuleFCN = (lambda lp_, *a, **k: TD.RecurrentFcnStub(lp_['out_dim'])) if mtype == 'recurrent' else (lambda lp_, *a, **k: TD.FcnStub(lp_['out_dim']))
      gp, obs, plp, opt = params_2d(n)
      planner = DiffGPMP2Planner(gp, obs, plp, opt, ENV, PointRobot2D(torch.tensor(0.4), B, n), learn_params=lp, batch_size=B)
      start, goal = rand_start_goal(B, seed=17)
      circ = ((-1.0, -1.0, 1.5), (2.0, 1.5, 1.2), (0.5, -2.5, 1.0))
      sdf = T(circles_sdf(Gsz, circ))[None, None].repeat(B, 1, 1, 1)
      g = torch.Generator().manual_seed(73)
      th_opt = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + torch.randn(B, n, 4, generator=g) * 0.3
      sample = {'im': (sdf > 0).double(), 'sdf': sdf.clone(), 'start': start, 'goal': goal, 'th_opt': th_opt}
      # the free names of the loop text (train_planner.py:203-254 sets them up from the same dicts)
      dg = lp['dgpmp2']
      ns = {'torch': torch, 'nn': nn, 'time': time, 'xrange': range, 'print': lambda *a, **k: None,
            'planner': planner, 'train_loader': [sample], 'device': torch.device('cpu'), 'dof': plp['dof'], 'planner_params': plp, 'learn_params': lp,
            'straight_line_trajb': lambda s, e, tt, ts, dof, dev: straight_line_trajb(s, e, tt, ts, dof),
            'fixed_conv': dg['fixed_conv'], 'optimizer': _NullOptimizer(), 'model_type': mtype, 'batch_size': B, 'T': dg['T'], 'tk': dg['tk'], 'tk2': dg['tk2'],
            'retain_graph': True, 'criterion': None, 'epoch': 0, 'clip_grad': False}      # (retain_graph: train_planner.py:228-229, True with use_inter_loss)
      exec(compile(loss_src, 'train_planner.one_step_loss', 'exec'), ns)
      terms, ref_loss = [], ns['one_step_loss']

      def recording_loss(*a, **k):      # what the loop only accumulates as rounded prints: the loss terms of every step, in full precision
        r = ref_loss(*a, **k)
        terms.append([float(x) for x in r])
        return r
      ns['one_step_loss'] = recording_loss
      with contextlib.redirect_stdout(io.StringIO()):
        exec(compile(loop_src, 'train_planner.train[batch body]', 'exec'), ns)
      if tag == 'fix_dynamics':
        out.update(G=Gsz, circles=np.asarray(circ), start=start, goal=goal, th_opt=th_opt)
      pre = tag + '_'
      grads = {name: p.grad for name, p in planner.named_parameters()}
      for name, gr in grads.items():
        out[pre + 'grad_' + name.replace('.', '_')] = gr
      out[pre + 'param_names'] = np.asarray(sorted(grads.keys()))
      out[pre + 'sdf_grad'] = ns['sdf_b'].grad; out[pre + 'th_final'] = ns['th_new_b']; out[pre + 'err'] = ns['errb']; out[pre + 'err_ext'] = ns['err_extb']
      out[pre + 'th_curr_grad_last'] = ns['th_curr_b'].grad
      out[pre + 'th_init_grad_is_none'] = ns['th_init_b'].grad is None
      out[pre + 'terms'] = np.asarray(terms)          # (T, 8): total, pos, vel, cov, gp, sg, obs, ext of every step (one_step_loss's return order)
      out[pre + 'batch_total_loss'] = ns['batch_total_loss']      # the loop's own running figure (divided by tk at every flush, never cleared)
      out[pre + 'final_loss'] = ns['final_loss'].detach()
  finally:
    pmod.LearnModuleConv, pmod.LearnModuleFCN = saved
  save('g7_tbptt', **out)


# ------------------------------------------------------------------------------------------------
# G8: autograd through DiffGPMP2Planner.forward -- the whole Gauss-Newton loop kept in the graph (diff_gpmp2_planner.py:92-174; consumer:
#   examples/diff_gpmp2_2d_example.py:77) -- w.r.t. the initial trajectory, the grids, the start and goal means.  Samples stop after
#   different numbers of iterations (tol_delta) and one runs into max_iters; the last dtheta of a sample IS applied (planner_utils.py:3-16).
# ------------------------------------------------------------------------------------------------
def g8_forward_grads(qc_inv=None, name='g8_forward_grads'):
  """qc_inv (round 6): a NON-DIAGONAL static Q_c_inv -> fixture g8_forward_grads_qc (the differentiable fused forward() of the general-covariance chain kernels)"""
  import io, contextlib
  B, n, Gsz = 4, 16, 48
  start, goal = rand_start_goal(B, seed=41)
  g = torch.Generator().manual_seed(42)
  th0 = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + torch.randn(B, n, 4, generator=g) * 0.15
  circ = ((-1.0, -1.0, 1.5), (2.0, 1.5, 1.2), (0.5, -2.5, 1.0))
  sdf = T(circles_sdf(Gsz, circ))[None, None].repeat(B, 1, 1, 1)
  sdf[0] = 3.0                                     # an obstacle-free sample: converges by tol_delta after two iterations
  gbar = torch.randn(B, n, 4, generator=g)
  gp, obs, plp, opt = params_2d(n, max_iters=9)
  opt['tol_delta'] = 0.5
  extra = {}
  if qc_inv is not None:
    gp['Q_c_inv'] = torch.as_tensor(qc_inv, dtype=torch.float64)
    extra['Q_c_inv'] = np.asarray(qc_inv, dtype=np.float64)
  planner = DiffGPMP2Planner(gp, obs, plp, opt, ENV, PointRobot2D(torch.tensor(0.4), 1, n), batch_size=1)
  leaves = [x.clone().requires_grad_(True) for x in (th0, sdf, start, goal)]
  with contextlib.redirect_stdout(io.StringIO()):
    thf, _, e_init, e_final, e_iter, ee_iter, k, _ = planner.forward(leaves[0], leaves[2], leaves[3], (sdf > 0).double(), leaves[1])
  gr = torch.autograd.grad((gbar * thf).sum(), leaves)
  maxlen = max(len(e) for e in e_iter)
  pad = lambda L: np.asarray([list(e) + [np.nan] * (maxlen - len(e)) for e in L])
  save(name, th0=th0, G=Gsz, circles=np.asarray(circ), free_sample=0, free_value=3.0, start=start, goal=goal, gbar=gbar, max_iters=9, tol_delta=0.5,
       th_final=thf, iters=np.asarray(k), err_init=np.asarray(e_init), err_final=np.asarray(e_final), err_iter=pad(e_iter), errext_iter=pad(ee_iter),
       g_th0=gr[0], g_sdf=gr[1], g_start=gr[2], g_goal=gr[3], **extra)


if __name__ == '__main__':
  g1_factors(); g1_custom(); g2_system(); g3_c1(); g3_c2mini(); g4_forward(); g5_grads(); g3_c3_vel(); g3_c4_xyh(); g6_helpers(); g6_dataset(); g7_errors(); g7_tbptt(); g8_forward_grads()
  g8_forward_grads(qc_inv=[[1.3, 0.4], [0.4, 0.9]], name='g8_forward_grads_qc')
