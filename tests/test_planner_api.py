"""GPU tests of the host-side mirror of the reference planner API (dgpmp2_amd.gpmp2): constructed with the reference's
param dicts, called with the reference's signatures, compared with the golden fixtures produced by the reference's own
DiffGPMP2Planner.step()/forward() (tests/golden/make_golden.py).  They read like the reference's example scripts
(examples/diff_gpmp2_2d_example.py, diff_gpmp2_2d_batch_step_example.py)."""
import numpy as np
import pytest
import torch
from conftest import rel_err
from oracle import gpmp2_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def ref_params(n, dtype=torch.float64, max_iters=10, tol_delta=1e-4):
  """param dicts exactly as examples/configs/gpmp2_2d_params.yaml + robot_2d.yaml + env_2d_params.yaml load them"""
  t = lambda v: torch.tensor(v, dtype=dtype)     # the reference loads its YAML under a float64 default dtype
  gp_params = {'Q_c_inv': torch.eye(2, dtype=dtype), 'K_s': t(0.01), 'K_g': t(0.01), 'K_v': t(0.01), 'v_x': [1.0], 'v_y': [1.0]}
  obs_params = {'cost_sigma': t(0.01), 'epsilon_dist': t(0.4)}
  planner_params = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': n - 1}
  optim_params = {'method': 'gauss_newton', 'reg': 0.1, 'plan_time': float('inf'), 'max_iters': max_iters, 'tol_err': 1e-3,
                  'tol_delta': tol_delta}
  env_params = {'x_lims': [-5.0, 5.0], 'y_lims': [-5.0, 5.0]}
  return gp_params, obs_params, planner_params, optim_params, env_params


def make_planner(n, B=1, **kw):
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  gp, ob, pp, op, ev = ref_params(n, **kw)
  robot = PointRobot2D(torch.tensor(0.4, dtype=torch.float64), B, n, use_cuda=True)
  return DiffGPMP2Planner(gp, ob, pp, op, ev, robot, batch_size=B, use_cuda=True)


def T(a, dtype=torch.float64):
  return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).to(DEV)


def test_step_matches_reference_c2mini(golden):
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)       # shared grid, as an expand()ed view
  im = (sdf > 0).double()
  th = T(g['th_hist'][0])
  for k in range(10):                       # examples/diff_gpmp2_2d_batch_step_example.py loop
    dth, hidden, err, err_ext, qc, ow, eps = planner.step(th, T(g['start']), T(g['goal']), im, sdf)
    assert hidden is None and dth.shape == (B, n, 4) and err.shape == (B, 1, 1) and err_ext.shape == (B, 1, 1)
    assert qc.shape == (B, n - 1, 2, 2) and ow.shape == (B, n, 1, 1) and eps.shape == (B, n, 1, 1)
    assert rel_err(dth.cpu().numpy(), g['dth_hist'][k]) < 1e-9
    assert rel_err(err.cpu().numpy(), g['err_hist'][k]) < 1e-11 and rel_err(err_ext.cpu().numpy(), g['errext_hist'][k]) < 1e-11
    th = T(g['th_hist'][k + 1])              # teacher forcing
  assert not err.requires_grad


def test_step_loop_is_capturable_in_a_hip_graph(golden):
  """A planning loop of planner.step() calls (examples/diff_gpmp2_2d_batch_step_example.py) captured once in a HIP graph (torch.cuda.CUDAGraph) and replayed:
  every launch goes to the capturing stream, nothing in the call path synchronises or allocates outside torch's allocator -- the replay reproduces the eager
  loop bit for bit, and again after the static input has been overwritten."""
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  im = (sdf > 0).double()
  start, goal = T(g['start']), T(g['goal'])
  th0 = T(g['th_hist'][0])

  def loop(th):
    for _ in range(5):
      th = th + planner.step(th, start, goal, im, sdf)[0]
    return th

  with torch.no_grad():
    eager = loop(th0)
    static_in = th0.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      loop(static_in)                              # warm-up on the side stream, as torch's capture recipe asks
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
      static_out = loop(static_in)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, eager)
    assert rel_err(static_out.cpu().numpy(), g['th_hist'][5]) < 1e-8
    static_in.copy_(T(g['th_hist'][2]))
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_out, loop(T(g['th_hist'][2])))


def test_training_iteration_gradient_subsets(golden):
  """forward_with_errors + backward when only SOME inputs require grad (the one-launch backward with a NULL g_th / NULL covariance gradients / missing cotangents): every
  gradient that is asked for equals the one of the full call."""
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  start, goal = T(g['start']), T(g['goal'])
  gen = torch.Generator(device=DEV).manual_seed(11)
  A = torch.randn(B, n - 1, 2, 2, device=DEV, dtype=torch.float64, generator=gen) * 0.2
  base = dict(th=T(g['th_hist'][1]), qc=torch.eye(2, device=DEV, dtype=torch.float64) + A @ A.transpose(-1, -2),
              ow=torch.rand(B, n, 1, 1, device=DEV, dtype=torch.float64, generator=gen) * 1e4 + 50,
              eps=torch.rand(B, n, 1, 1, device=DEV, dtype=torch.float64, generator=gen) * 0.5 + 0.1, start=start.clone(), goal=goal.clone())
  c_dth = torch.randn(B, n, 4, device=DEV, dtype=torch.float64, generator=gen)
  c_e = torch.randn(B, 1, 1, device=DEV, dtype=torch.float64, generator=gen)

  def run(names, outs):
    L = {k: (v.clone().requires_grad_(True) if k in names else v.clone()) for k, v in base.items()}
    dth, _, eex, sg, gp_, ob = planner.plan_layer.forward_with_errors(L['th'], L['start'], L['goal'], None, sdf, L['qc'], L['ow'], L['eps'])
    o = dict(dth=(dth, c_dth), eex=(eex, c_e), sg=(sg, c_e.view(B, 1)), gp=(gp_, c_e), ob=(ob, c_e))
    loss = sum((o[k][0] * o[k][1]).sum() for k in outs)
    return dict(zip(names, torch.autograd.grad(loss, [L[k] for k in names], allow_unused=True)))
  every = ('th', 'start', 'goal', 'qc', 'ow', 'eps')
  full = run(every, ('dth', 'eex', 'sg', 'gp', 'ob'))
  for names in (('qc', 'ow', 'eps'), ('th',), ('start', 'goal'), ('eps',), ('th', 'qc')):
    part = run(names, ('dth', 'eex', 'sg', 'gp', 'ob'))
    for k in names:
      assert rel_err(part[k].cpu().numpy(), full[k].cpu().numpy()) < 1e-12, (names, k)
  # missing cotangents: only the errors, only dtheta, one error alone -- against the sum rule
  parts = [run(every, (k,)) for k in ('dth', 'eex', 'sg', 'gp', 'ob')]
  for k in every:
    tot = sum(p[k] if p[k] is not None else torch.zeros_like(full[k]) for p in parts)
    assert rel_err(tot.cpu().numpy(), full[k].cpu().numpy()) < 1e-9, k


def test_training_iteration_xyh_robot_two_launch_backward():
  """The (x, y, theta) robot (d = 6, non-holonomic factor): forward_with_errors + backward -- the C entry point keeps two backward launches through a workspace that
  PlanLayer allocates -- against the two calls it replaces (forward() and unweighted_errors(th + dtheta), two autograd nodes), learned per-state covariances."""
  from dgpmp2_amd.robot_models import PointRobotXYH
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  B, n, G = 6, 40, 64
  gp, ob, pp, op, ev = ref_params(n)
  gp.update(Q_c_inv=torch.eye(3, dtype=torch.float64), K_d=torch.tensor(0.05, dtype=torch.float64))
  pp.update(dof=3, state_dim=6, non_holonomic=True)
  robot = PointRobotXYH(torch.tensor(0.4, dtype=torch.float64), use_cuda=True, batch_size=B, num_traj_states=n)
  planner = DiffGPMP2Planner(gp, ob, pp, op, ev, robot, batch_size=B, use_cuda=True)
  pl = planner.plan_layer
  rs = np.random.RandomState(4)
  sdf = T(O.circles_sdf(G, O.C2_CIRCLES))[None, None].expand(B, 1, G, G)
  start = np.zeros((B, 1, 6)); goal = np.zeros((B, 1, 6))
  start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, 2] = rs.uniform(-1, 1, B)
  th = O.straight_line_trajb(start[:, :, :3], goal[:, :, :3], 10.0, n - 1, 3) + 0.05 * rs.randn(B, n, 6)
  A = rs.randn(B, n - 1, 3, 3) * 0.2
  base = dict(th=T(th), qc=T(np.eye(3) + A @ np.swapaxes(A, -1, -2)), ow=T(rs.uniform(50, 2e4, (B, n, 1, 1))), eps=T(rs.uniform(0.1, 0.6, (B, n, 1, 1))))
  start, goal = T(start), T(goal)
  c_dth, c_e = T(rs.randn(B, n, 6)), T(rs.randn(B, 1, 1))

  def leaves(): return {k: v.clone().requires_grad_(True) for k, v in base.items()}
  L = leaves()
  dth, _, eex, sg, gp_, ob_ = pl.forward_with_errors(L['th'], start, goal, None, sdf, L['qc'], L['ow'], L['eps'])
  loss = (dth * c_dth).sum() + (eex * c_e).sum() + (sg * c_e.view(B, 1)).sum() + (gp_ * c_e).sum() + (ob_ * c_e).sum()
  ga = torch.autograd.grad(loss, [L[k] for k in ('th', 'qc', 'ow', 'eps')])
  L2 = leaves()
  d2, _, x2 = pl(L2['th'], start, goal, None, sdf, L2['qc'], L2['ow'], L2['eps'])
  s2, g2, o2 = pl.unweighted_errors(L2['th'] + d2, sdf)
  loss2 = (d2 * c_dth).sum() + (x2 * c_e).sum() + (s2 * c_e.view(B, 1)).sum() + (g2 * c_e).sum() + (o2 * c_e).sum()
  gb = torch.autograd.grad(loss2, [L2[k] for k in ('th', 'qc', 'ow', 'eps')])
  for a_, b_ in ((dth, d2), (eex, x2), (sg, s2), (gp_, g2), (ob_, o2)):
    assert rel_err(a_.detach().cpu().numpy(), b_.detach().cpu().numpy()) < 1e-11
  for k, a_, b_ in zip(('th', 'qc', 'ow', 'eps'), ga, gb):
    assert rel_err(a_.cpu().numpy(), b_.cpu().numpy()) < 1e-9, (k, rel_err(a_.cpu().numpy(), b_.cpu().numpy()))
  # ... and the default learned mode from the module's raw output vector (diag_identity: the scaled d = 6 kernels, dgp_square_covariances[_backward] around them)
  # against the route through get_covariances
  out0 = T(np.concatenate([rs.uniform(0.7, 1.5, (B, 1, n - 1)), rs.uniform(20, 120, (B, 1, n))], axis=2))
  res = []
  for route in ('raw', 'explicit'):
    out = out0.clone().requires_grad_(True); thr = base['th'].clone().requires_grad_(True)
    if route == 'raw':
      raw = pl.raw_covs(out, 'diag_identity', False)
      assert raw is not None
      dth, _, eex, sg, gp_, ob_ = pl.forward_raw(thr, start, goal, None, sdf, raw, with_errors=True)[:6]
    else:
      qc_s, ow_s = planner.get_covariances(out, 'diag_identity')
      dth, _, eex, sg, gp_, ob_ = pl.forward_with_errors(thr, start, goal, None, sdf, qc_s, ow_s, None)
    loss = (dth * c_dth).sum() + (eex * c_e).sum() + (sg.reshape(B, 1) * c_e.view(B, 1)).sum() + (gp_ * c_e).sum() + (ob_ * c_e).sum()
    res.append((dth.detach(), ob_.detach()) + torch.autograd.grad(loss, (thr, out)))
  for a_, b_ in zip(*res):
    assert rel_err(a_.cpu().numpy(), b_.cpu().numpy()) < 1e-9


def test_training_iteration_is_capturable_in_a_hip_graph(golden):
  """One iteration of the training loop -- step_with_errors with learned per-state covariances + the backward of all four outputs w.r.t. the trajectory
  and the three covariance tensors (learning/train_planner.py:311-327, 366) -- captured in a HIP graph: forward AND backward launches are recorded
  (torch's whole-iteration capture), the replay returns the eager gradients bit for bit and follows new values written into the static inputs."""
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  start, goal = T(g['start']), T(g['goal'])
  gen = torch.Generator(device=DEV).manual_seed(3)
  thr = T(g['th_hist'][1]).requires_grad_(True)
  A = torch.randn(B, n - 1, 2, 2, device=DEV, dtype=torch.float64, generator=gen) * 0.2
  qc = (torch.eye(2, device=DEV, dtype=torch.float64) + A @ A.transpose(-1, -2)).requires_grad_(True)
  ow = (torch.rand(B, n, 1, 1, device=DEV, dtype=torch.float64, generator=gen) * 1e4 + 50).requires_grad_(True)
  ep = (torch.rand(B, n, 1, 1, device=DEV, dtype=torch.float64, generator=gen) * 0.5 + 0.1).requires_grad_(True)
  c_dth = torch.randn(B, n, 4, device=DEV, dtype=torch.float64, generator=gen)
  c_e = torch.randn(B, 1, 1, device=DEV, dtype=torch.float64, generator=gen)
  leaves = (thr, qc, ow, ep)

  def iteration():
    dth, _, eex, sg, gp_, ob = planner.plan_layer.forward_with_errors(thr, start, goal, None, sdf, qc, ow, ep)
    return torch.autograd.grad((dth, eex, sg, gp_, ob), leaves, (c_dth, c_e, c_e.view(B, 1), c_e, c_e))

  eager = [t.clone() for t in iteration()]
  side = torch.cuda.Stream()
  side.wait_stream(torch.cuda.current_stream())
  with torch.cuda.stream(side):
    for _ in range(2): iteration()
  torch.cuda.current_stream().wait_stream(side)
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    outs = iteration()
  graph.replay()
  torch.cuda.synchronize()
  for a, b in zip(outs, eager):
    assert torch.equal(a, b)
  with torch.no_grad():
    thr.copy_(T(g['th_hist'][3])); ow.mul_(0.5)
  graph.replay()
  torch.cuda.synchronize()
  replayed = [t.clone() for t in outs]
  for a, b in zip(replayed, iteration()):
    assert torch.equal(a, b)
  assert not torch.equal(replayed[0], eager[0])


def test_diag_identity_covariances_run_the_scaled_static_kernels(golden):
  """dynamics_mode 'diag_identity' (the reference's default learned mode): get_covariances() returns q_k^2 I blocks AND tags them with the scalars, PlanLayer.forward
  hands the scalars to the kernel (DGP_QC_SCALAR) -- same step as with the untagged blocks (the per-state kernels), same gradients (the backward reads the blocks),
  and an in-place edit of the blocks voids the tag."""
  from dgpmp2_amd import _capi
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  pl = planner.plan_layer
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  start, goal, th = T(g['start']), T(g['goal']), T(g['th_hist'][2])
  gen = torch.Generator(device=DEV).manual_seed(9)
  out = (torch.rand(B, 1, (n - 1) + n, device=DEV, dtype=torch.float64, generator=gen) + 0.5).requires_grad_(True)
  out.data[:, :, n - 1:] *= 80.0                                     # obstacle part: o^2 ~ 1e3 .. 1e4

  def run(tagged):
    qc, ow = planner.get_covariances(out, 'diag_identity')
    assert '_dgp_scalar' in qc.__dict__
    if not tagged: qc = qc * 1.0                                     # same values, no tag
    mode = pl._cov_args(qc, ow, None, torch.float64, B, 0, (False, False, True), True)[0]
    dth, err, eex = pl(th, start, goal, None, sdf, qc, ow, None)
    gr, = torch.autograd.grad((dth * dth).sum() + eex.sum(), out)
    return mode, dth.detach(), err, eex.detach(), gr

  def run_iteration(tagged):      # the training iteration: step + unweighted errors at th + dtheta as one node, backward of all of it
    qc, ow = planner.get_covariances(out, 'diag_identity')
    if not tagged: qc = qc * 1.0
    dth, _, eex, sg, gp_, ob = pl.forward_with_errors(th, start, goal, None, sdf, qc, ow, None)
    gr, = torch.autograd.grad((dth * dth).sum() + eex.sum() + 3.0 * sg.sum() + 5.0 * gp_.sum() + 7.0 * ob.sum(), out)
    return dth.detach(), sg.detach(), gr

  i1, i0 = run_iteration(True), run_iteration(False)
  for a_, b_ in zip(i1, i0):
    assert rel_err(a_.cpu().numpy(), b_.cpu().numpy()) < 1e-8
  m1, d1, e1, x1, g1 = run(True)
  m0, d0, e0, x0, g0 = run(False)
  assert m1 == _capi.DGP_QC_SCALAR and m0 == _capi.DGP_QC_PERSTATE
  assert rel_err(d1.cpu().numpy(), d0.cpu().numpy()) < 1e-9 and rel_err(e1.cpu().numpy(), e0.cpu().numpy()) < 1e-12
  assert rel_err(x1.cpu().numpy(), x0.cpu().numpy()) < 1e-12 and rel_err(g1.cpu().numpy(), g0.cpu().numpy()) < 1e-8
  qc, ow = planner.get_covariances(out.detach(), 'diag_identity')
  qc.mul_(2.0)                                                        # the blocks no longer are what the tag says
  assert pl._cov_args(qc, ow, None, torch.float64, B, 0, (False, False, True), True)[0] == _capi.DGP_QC_PERSTATE


def test_step_float32_tensors(golden):
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  f32 = torch.float32
  sdf = T(O.circles_sdf(G, g['circles']), f32)[None, None].expand(B, 1, G, G)
  th, st, go = T(g['th_hist'][4], f32), T(g['start'], f32), T(g['goal'], f32)
  dth, _, err, err_ext, _, _, _ = planner.step(th, st, go, (sdf > 0).float(), sdf)
  assert dth.dtype == f32
  p = O.OracleParams(dof=2, total_time_step=n - 1)
  qc, ow, eps = p.static_covs(B)
  r_dth, r_err, _ = O.plan_layer_forward(th.double().cpu().numpy(), st.double().cpu().numpy(), go.double().cpu().numpy(),
                                         sdf.double().cpu().numpy(), qc, ow, eps, p)
  assert rel_err(dth.cpu().numpy(), r_dth) < 1e-5 and rel_err(err.cpu().numpy(), r_err) < 2e-6


def test_plan_layer_forward_with_covariance_tensors(golden):
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1)          # materialised per-sample copies
  dth, err, err_ext = planner.plan_layer(T(g['cov_th']), T(g['start']), T(g['goal']), (sdf > 0).double(), sdf, T(g['cov_qc']),
                                         T(g['cov_ow']), T(g['cov_eps']))
  assert rel_err(dth.cpu().numpy(), g['cov_dth']) < 1e-9
  assert rel_err(err.cpu().numpy(), g['cov_err']) < 1e-11 and rel_err(err_ext.cpu().numpy(), g['cov_errext']) < 1e-11


def test_forward_c1_plumbing(golden):
  """examples/diff_gpmp2_2d_example.py plumbing: 1 environment (5.png), 1 trajectory, n=32, GN until max_iters."""
  from dgpmp2_amd.utils.planner_utils import straight_line_traj
  g = golden('g4_forward'); c1 = golden('g3_c1')
  planner = make_planner(32, 1, max_iters=int(g['c1_max_iters']), tol_delta=float(g['c1_tol_delta']))
  start, goal = T(c1['start'][0]), T(c1['goal'][0])
  th_init = straight_line_traj(start[:, :2], goal[:, :2], 10.0, 31, 2, DEV)
  sdf = T(c1['sdf'])
  im = (sdf > 0).double()
  th_final, _, err_init, err_final, err_per_iter, err_ext_per_iter, k, time_taken = planner.forward(
      th_init.unsqueeze(0), start.unsqueeze(0), goal.unsqueeze(0), im.unsqueeze(0).unsqueeze(0), sdf.unsqueeze(0).unsqueeze(0))
  assert k == list(g['c1_iters']) and len(time_taken) == 1
  assert rel_err(th_final.cpu().numpy(), g['c1_th_final']) < 1e-7
  assert rel_err(err_per_iter[0], g['c1_err_iter'][0]) < 1e-8 and rel_err(err_ext_per_iter[0], g['c1_errext_iter'][0]) < 1e-8
  assert abs(err_init[0] - float(g['c1_err_init'][0])) < 1e-9 * err_init[0] and rel_err(err_final, g['c1_err_final']) < 1e-8
  # known-answer scalar of the survey: err0 = 372.176512415553
  assert abs(err_init[0] - 372.176512415553) < 1e-8


def test_forward_early_exit_per_trajectory(golden):
  g = golden('g4_forward')
  planner = make_planner(16, 1, max_iters=int(g['free_max_iters']), tol_delta=float(g['free_tol_delta']))
  sdf = torch.full((3, 1, 32, 32), 3.0, dtype=torch.float64, device=DEV)
  out = planner.forward(T(g['free_th0']), T(g['free_start']), T(g['free_goal']), (sdf > 0).double(), sdf)
  assert out[6] == list(g['free_iters'])
  assert rel_err(out[0].cpu().numpy(), g['free_th_final']) < 1e-8 and rel_err(out[3], g['free_err_final']) < 1e-8
  for b in range(3):
    assert rel_err(out[4][b], g['free_err_iter'][b][:out[6][b]]) < 1e-8


def test_error_helpers_and_unweighted_errors(golden):
  g = golden('g3_c1')
  planner = make_planner(32, 1)
  sdf = T(g['sdf'])[None, None]
  st, go = T(g['start']), T(g['goal'])
  planner.step(T(g['th_hist'][3]), st, go, (sdf > 0).double(), sdf)
  th = T(g['th_hist'][3])
  assert rel_err(planner.error_batch(th, sdf).cpu().numpy(), g['err_hist'][3]) < 1e-11
  assert rel_err(planner.error_ext_batch(th, sdf).cpu().numpy(), g['errext_hist'][3]) < 1e-11
  usg, ugp, uobs = planner.unweighted_errors_batch(th, sdf)
  assert usg.shape == (1, 1) and ugp.shape == (1, 1, 1) and uobs.shape == (1, 1, 1)      # the reference's shapes (plan_layer.py:374-388)
  assert rel_err(ugp.cpu().numpy(), g['unw_gp']) < 1e-11 and rel_err(uobs.cpu().numpy(), g['unw_obs']) < 1e-11
  assert abs(float(usg) - float(g['unw_sg'].item())) < 1e-12


def test_unweighted_error_methods_one_by_one(golden):
  """PlanLayer.start_goal_error / gp_error / obs_error called individually, as DiffGPMP2Planner.unweighted_errors_batch does in
  the reference (diff_gpmp2_planner.py:229-237 -> plan_layer.py:374-388), against the reference's values (fixture g3_c1)."""
  g = golden('g3_c1')
  planner = make_planner(32, 1)
  sdf = T(g['sdf'])[None, None]
  planner.step(T(g['th_hist'][3]), T(g['start']), T(g['goal']), (sdf > 0).double(), sdf)
  th = T(g['th_hist'][3])
  pl = planner.plan_layer
  usg, ugp, uobs = pl.start_goal_error(th), pl.gp_error(th), pl.obs_error(th, sdf)
  assert usg.shape == (1, 1) and ugp.shape == (1, 1, 1) and uobs.shape == (1, 1, 1)
  assert abs(float(usg) - float(g['unw_sg'].item())) < 1e-12
  assert rel_err(ugp.cpu().numpy(), g['unw_gp']) < 1e-11 and rel_err(uobs.cpu().numpy(), g['unw_obs']) < 1e-11
  a, b, c = pl.unweighted_errors(th, sdf)
  assert torch.equal(a, usg) and torch.equal(b, ugp) and torch.equal(c, uobs)
  # float32 tensors and a batch: same helpers, same values to fp32 accuracy
  g2 = golden('g3_c2mini')
  B, n, G = 8, 64, int(g2['G'])
  planner = make_planner(n, B)
  sdf = T(O.circles_sdf(G, g2['circles']), torch.float32)[None, None].expand(B, 1, G, G)
  th, st, go = T(g2['th_hist'][2], torch.float32), T(g2['start'], torch.float32), T(g2['goal'], torch.float32)
  planner.step(th, st, go, None, sdf)
  p = O.OracleParams(dof=2, total_time_step=n - 1)
  r_sg, r_gp, r_obs = O.unweighted_errors_batch(th.double().cpu().numpy(), st.double().cpu().numpy(), go.double().cpu().numpy(),
                                                sdf.double().cpu().numpy(), p.static_covs(B)[2], p)
  pl = planner.plan_layer
  assert rel_err(pl.gp_error(th).cpu().numpy(), r_gp) < 2e-6 and rel_err(pl.obs_error(th, sdf).cpu().numpy(), r_obs) < 2e-6
  assert pl.start_goal_error(th).shape == r_sg.shape == (B, 1)
  assert float(np.max(np.abs(pl.start_goal_error(th).double().cpu().numpy() - r_sg))) < 1e-6


def test_rejects_mismatched_sdf_batch_and_double_backward(golden):
  g = golden('g5_grads')
  B, n, G = 4, 16, int(g['G'])
  planner = make_planner(n, B)
  sdf3 = T(O.circles_sdf(G, g['circles']))[None, None].repeat(3, 1, 1, 1)        # 3 grids for 4 trajectories
  with pytest.raises(ValueError):
    planner.step(T(g['th']), T(g['start']), T(g['goal']), None, sdf3)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  th = T(g['th']).requires_grad_(True)
  dth, _, _, err_ext, _, _, _ = planner.step(th, T(g['start']), T(g['goal']), None, sdf)
  (gth,) = torch.autograd.grad((dth ** 2).sum(), th, create_graph=True)
  with pytest.raises(RuntimeError):                 # once_differentiable: no silent zero second-order terms
    gth.sum().backward()
  # an err_ext-only loss needs no adjoint solve (no dtheta cotangent is materialised) and still matches the reference
  th2 = T(g['th']).requires_grad_(True)
  sdfB = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1)
  _, _, ee = planner.plan_layer(th2, T(g['start']), T(g['goal']), None, sdfB, T(g['qc']), T(g['ow']), T(g['eps']))
  (T(g['gext']) * ee).sum().backward()
  assert rel_err(th2.grad.cpu().numpy(), g['ge_th']) < 1e-10


def test_rejects_cpu_tensors_and_bad_shapes():
  planner = make_planner(16, 1)
  th = torch.zeros(1, 16, 4, dtype=torch.float64)
  with pytest.raises(RuntimeError):
    planner.step(th, th[:, :1], th[:, :1], None, torch.zeros(1, 1, 8, 8, dtype=torch.float64))
  thc = th.to(DEV)
  with pytest.raises(ValueError):
    planner.step(thc[:, :15], thc[:, :1], thc[:, :1], None, torch.zeros(1, 1, 8, 8, dtype=torch.float64, device=DEV))


def test_get_covariances_shapes():
  planner = make_planner(16, 2)
  n = 16
  out = torch.randn(2, 1, (n - 1) * 2 + n, device=DEV, dtype=torch.float64)
  qc, ow = planner.get_covariances(out, 'qc_full')
  assert qc.shape == (2, n - 1, 2, 2) and ow.shape == (2, n, 1, 1)
  assert torch.allclose(qc, qc.transpose(2, 3)) and bool((ow >= 0).all())
  out = torch.randn(2, 1, (n - 1) + 2 * n, device=DEV, dtype=torch.float64)
  qc, ow, eps = planner.get_covariances(out, 'diag_identity', learn_eps=True)
  assert qc.shape == (2, n - 1, 2, 2) and eps.shape == (2, n, 1, 1) and float(qc[0, 0, 0, 1]) == 0.0
  o = torch.randn(n, device=DEV, dtype=torch.float64)
  w = planner.get_obs_covariance(o)                                            # diff_gpmp2_planner.py:293-297
  assert w.shape == (n, 1, 1) and torch.equal(w[:, 0, 0], o * o)


# ---- autograd through the planner API: reference = torch autograd over plan_layer.py:152-234 ----------
def test_autograd_matches_reference_grads(golden):
  """Same experiment as tests/golden/make_golden.py::g5_grads, through dgpmp2_amd's PlanLayer and torch.autograd."""
  g = golden('g5_grads')
  B, n, G = 4, 16, int(g['G'])
  planner = make_planner(n, B)
  leaves = {}
  for k in ('th', 'start', 'goal', 'qc', 'ow', 'eps'):
    leaves[k] = T(g[k]).requires_grad_(True)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1).requires_grad_(True)
  dth, err, err_ext = planner.plan_layer(leaves['th'], leaves['start'], leaves['goal'], (sdf.detach() > 0).double(), sdf, leaves['qc'],
                                         leaves['ow'], leaves['eps'])
  assert not err.requires_grad and err_ext.requires_grad and dth.requires_grad
  names = ('th', 'start', 'goal', 'qc', 'ow', 'eps')
  grads = torch.autograd.grad((T(g['gbar']) * dth).sum(), [leaves[k] for k in names] + [sdf], retain_graph=True)
  for k, gr in zip(names + ('sdf',), grads):
    assert rel_err(gr.cpu().numpy(), g['g_' + k]) < 1e-8, k
  grads_e = torch.autograd.grad((T(g['gext']) * err_ext).sum(), [leaves[k] for k in names] + [sdf], allow_unused=True)
  for k, gr in zip(names + ('sdf',), grads_e):
    if bool(g['ge_none_' + k]):
      assert gr is None or float(gr.abs().max()) == 0.0, k       # reference: None (fixed covariances, plan_layer.py:318,330)
    else:
      assert rel_err(gr.cpu().numpy(), g['ge_' + k]) < 1e-10, k


def test_autograd_shared_sdf_expand_and_static_covs(golden):
  """Static covariances (step()), one SDF expand()ed over the batch: the SDF gradient is the sum over the batch."""
  g = golden('g5_grads')
  B, n, G = 4, 16, int(g['G'])
  planner = make_planner(n, B)
  sdf1 = T(O.circles_sdf(G, g['circles']))[None, None].requires_grad_(True)
  th = T(g['th']).requires_grad_(True)
  dth, _, err, err_ext, _, _, _ = planner.step(th, T(g['start']), T(g['goal']), None, sdf1.expand(B, 1, G, G))
  loss = (T(g['gbar']) * dth).sum() + err_ext.sum()
  loss.backward()
  # same thing with materialised per-sample copies
  sdfB = sdf1.detach().repeat(B, 1, 1, 1).requires_grad_(True)
  th2 = T(g['th']).requires_grad_(True)
  dth2, _, _, err_ext2, _, _, _ = planner.step(th2, T(g['start']), T(g['goal']), None, sdfB)
  ((T(g['gbar']) * dth2).sum() + err_ext2.sum()).backward()
  assert rel_err(th.grad.cpu().numpy(), th2.grad.cpu().numpy()) < 1e-12
  assert rel_err(sdf1.grad.cpu().numpy(), sdfB.grad.sum(0, keepdim=True).cpu().numpy()) < 1e-10


def _learn_planner(n, B, lp):
  """dgpmp2_amd planner in learned mode with the stub learn modules of tests/tbptt_driver.py (the ones make_golden.py injected into the
  reference planner), handed over as FACTORIES with the reference classes' constructor signature: like the reference's constructor
  (diff_gpmp2_planner.py:60-87) the planner first writes num_traj_states / state_dim / out_dim into learn_params, then builds the modules."""
  import tbptt_driver as TD
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  gp, ob, pp, op, ev = ref_params(n)
  recurrent = lp['model']['type'] == 'recurrent'
  fcn = (lambda lp_, env, obs, robot, use_cuda=False: (TD.RecurrentFcnStub if recurrent else TD.FcnStub)(lp_['out_dim']).to(DEV))
  conv = lambda lp_, env, robot, use_cuda=False: TD.ConvStub()
  planner = DiffGPMP2Planner(gp, ob, pp, op, ev, PointRobot2D(torch.tensor(0.4, dtype=torch.float64), B, n), learn_params=lp,
                             batch_size=B, use_cuda=True, learn_module_conv=conv, learn_module_fcn=fcn)
  return planner, pp


TBPTT_LEARN_PARAMS = {      # == tests/golden/make_golden.py::TBPTT_LEARN_PARAMS
    'model': {'type': 'feed_forward'},
    'dgpmp2': {'learn_eps': False, 'sdf_predict': True, 'dtheta_predict': False, 'fixed_conv': False, 'T': 4, 'tk': 2, 'tk2': 2,
               'use_inter_loss': True, 'optimize_tk': False},
    'data': {'im_size': 48},
    'optim': {'vel_loss_lambda': 0.5, 'ext_obs_lambda': 2.0, 'ext_loss_weight': 0.3, 'batch_size': 3, 'do_validation': False},
}


@pytest.mark.parametrize('fused', [False, True], ids=['step+errors', 'step_with_errors'])
@pytest.mark.parametrize('tag,mode,mtype', [('fix_dynamics', 'fix_dynamics', 'feed_forward'), ('qc_full', 'qc_full', 'feed_forward'),
                                            ('recurrent', 'fix_dynamics', 'recurrent')])
def test_tbptt_outer_loop_runs(golden, tag, mode, mtype, fused):
  """The solver drops into the reference's outer learning loop.  Fixture g7_tbptt = one batch of the reference's train() -- ITS OWN loop
  text (learning/train_planner.py:258-424 and one_step_loss, exec'd from /root/reference by tests/golden/make_golden.py) on the
  reference's planner.  Here: this build's statement of the same truncated-BPTT procedure (tests/tbptt_driver.py: fresh leaf per GN step,
  loss on the update + the unweighted factor errors at th + dtheta, flush every tk steps through at most tk2 links) through dgpmp2_amd's
  planner -- feed-forward predictor in two dynamics modes, and a recurrent one (hiddenb through step(), diff_gpmp2_planner.py:192,208-210).
  Loss terms of every step, the final trajectory, and the gradients left in the predictor's parameters, in the grid and in the last
  trajectory leaf must agree with the reference's.  fused: every link through planner.step_with_errors (one autograd node per link)
  instead of step() + unweighted_errors_batch()."""
  import copy
  import tbptt_driver as TD
  from dgpmp2_amd.utils.planner_utils import straight_line_trajb
  g = golden('g7_tbptt')
  B, n, G = 3, 16, int(g['G'])
  lp = copy.deepcopy(TBPTT_LEARN_PARAMS)
  lp['dgpmp2']['dynamics_mode'] = mode
  lp['model']['type'] = mtype
  planner, pp = _learn_planner(n, B, lp)
  dg = lp['dgpmp2']
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1)
  batch = {'im': (sdf > 0).double(), 'sdf': sdf.clone().requires_grad_(True), 'start': T(g['start']), 'goal': T(g['goal']), 'th_opt': T(g['th_opt'])}
  th_init = straight_line_trajb(batch['start'][:, :, :2], batch['goal'][:, :, :2], pp['total_time_sec'], pp['total_time_step'], 2, torch.device(DEV))
  th_init.requires_grad_(True)
  r = TD.truncated_bptt(planner, batch, th_init, dg['T'], dg['tk'], dg['tk2'], lp['optim'], recurrent=(mtype == 'recurrent'), fused=fused)
  pre = tag + '_'
  ref_terms = g[pre + 'terms']                    # (T, 8): total, pos, vel, cov (always 0), gp, sg, obs, ext
  mine = np.asarray([[float(x.detach()) if torch.is_tensor(x) else float(x) for x in (t.total, t.pos, t.vel, 0.0, t.gp, t.sg, t.obs, t.ext)] for t in r['terms']])
  assert mine.shape == ref_terms.shape
  for c, name in enumerate(('total', 'pos', 'vel', 'cov', 'gp', 'sg', 'obs', 'ext')):
    assert rel_err(mine[:, c], ref_terms[:, c]) < 1e-9 or float(np.abs(ref_terms[:, c]).max()) == 0.0, name
  grads = {name: p.grad for name, p in planner.named_parameters()}
  assert sorted(grads.keys()) == list(g[pre + 'param_names'])
  assert rel_err(r['th_final'].cpu().numpy(), g[pre + 'th_final']) < 1e-9
  assert rel_err(r['err'].cpu().numpy(), g[pre + 'err']) < 1e-9 and rel_err(r['err_ext'].cpu().numpy(), g[pre + 'err_ext']) < 1e-9
  for name, gr in grads.items():
    assert rel_err(gr.cpu().numpy(), g[pre + 'grad_' + name.replace('.', '_')]) < 1e-8, name
  assert rel_err(batch['sdf'].grad.cpu().numpy(), g[pre + 'sdf_grad']) < 1e-8
  assert rel_err(r['last_input_leaf'].grad.cpu().numpy(), g[pre + 'th_curr_grad_last']) < 1e-8
  assert (th_init.grad is None) == bool(g[pre + 'th_init_grad_is_none'])


def test_constructor_prepares_learn_params_like_the_reference(golden):
  """diff_gpmp2_planner.py:58-78: the constructor writes num_traj_states / state_dim / out_dim into learn_params (per dynamics_mode,
  dtheta_predict, learn_eps) and keeps the image resolution as self.res.  Fixture g6_helpers: what the REFERENCE's constructor left
  in the dict for every combination."""
  import copy
  g = golden('g6_helpers')
  n, B = 16, 2
  combos = [(m, le, dp) for m in ('fix_dynamics', 'diag_identity', 'qc_full', 'q_full') for le in (False, True) for dp in (False, True)]
  got = []
  for mode, learn_eps, dth_pred in combos:
    lp = copy.deepcopy(TBPTT_LEARN_PARAMS)
    lp['dgpmp2'].update(dynamics_mode=mode, learn_eps=learn_eps, dtheta_predict=dth_pred)
    planner, _ = _learn_planner(n, B, lp)
    got.append([lp['num_traj_states'], lp['state_dim'], lp['out_dim']])
    assert planner.res == float(g['lp_res'])
    assert planner.learn_module_fcn.w.numel() == lp['out_dim']       # the factory saw the prepared dict
  assert np.array_equal(np.asarray(got), g['lp_prepared'])


def test_get_covariances_matches_reference_fixture(golden):
  """get_covariances (diff_gpmp2_planner.py:247-290), every mode x learn_eps, and get_obs_covariance (:293-297) on the inputs of
  fixture g6_helpers: bit-identical to what the reference's methods returned."""
  g = golden('g6_helpers')
  n, B = 16, 3
  planner = make_planner(n, B)
  for mode in ('fix_dynamics', 'diag_identity', 'qc_full', 'q_full'):
    for le in (False, True):
      key = 'cov_%s_%d' % (mode, int(le))
      out = T(g[key + '_in'])
      res = planner.get_covariances(out, mode, le)
      res = res if isinstance(res, tuple) else (res,)
      assert len(res) == int(g[key + '_count'])
      for i, t in enumerate(res):
        ref = g['%s_out%d' % (key, i)]
        assert tuple(t.shape) == ref.shape and np.array_equal(t.cpu().numpy(), ref), (key, i)
  with pytest.raises(NotImplementedError):
    planner.get_covariances(T(g['cov_qc_full_0_in']), 'diag')
  w = planner.get_obs_covariance(T(g['obscov_in']))
  assert tuple(w.shape) == g['obscov_out'].shape and np.array_equal(w.cpu().numpy(), g['obscov_out'])


def test_forward_raises_on_non_spd_like_cholesky():
  """forward()'s fused path has synchronised anyway (history copy): a non-SPD system raises, as torch.cholesky does in the reference
  (plan_layer.py:226).  A negative obstacle weight cannot be expressed through the static config, so drive it with reg < 0."""
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  n, B = 16, 2
  gp, ob, pp, op, ev = ref_params(n)
  op['reg'] = -1.0e6
  planner = DiffGPMP2Planner(gp, ob, pp, op, ev, PointRobot2D(torch.tensor(0.4, dtype=torch.float64), B, n, use_cuda=True), batch_size=B, use_cuda=True)
  th = torch.zeros(B, n, 4, dtype=torch.float64, device=DEV)
  sdf = torch.ones(B, 1, 16, 16, dtype=torch.float64, device=DEV)
  with pytest.raises(RuntimeError, match='not positive definite'):
    planner.forward(th, th[:, :1], th[:, :1], None, sdf)
  assert int(planner.plan_layer.last_info.count_nonzero()) == B


def test_forward_with_grad_is_two_launches_and_matches_reference(golden):
  """planner.forward on inputs that require grad (the reference keeps the graph across its Gauss-Newton loop, diff_gpmp2_planner.py:122-156;
  consumer examples/diff_gpmp2_2d_example.py:77): ONE fused launch forward (dgp_gn_solve_traced), ONE launch backward (dgp_gn_solve_backward),
  against the reference's autograd through its own forward() (fixture g8_forward_grads: trajectories that stop after 2, 8, 9, 9 iterations)."""
  import dgpmp2_amd.gpmp2.plan_layer as PL
  g = golden('g8_forward_grads')
  B, n, G = 4, 16, int(g['G'])
  planner = make_planner(n, B, max_iters=int(g['max_iters']), tol_delta=float(g['tol_delta']))
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1)
  sdf[int(g['free_sample'])] = float(g['free_value'])
  L = {'th': T(g['th0']).requires_grad_(True), 'sdf': sdf.requires_grad_(True), 'start': T(g['start']).requires_grad_(True), 'goal': T(g['goal']).requires_grad_(True)}
  pc = planner.plan_layer._pc
  calls = []

  class Spy(object):
    def __getattr__(self, name):
      f = getattr(pc, name)
      def w(*a):
        calls.append(name)
        return f(*a)
      return w
  planner.plan_layer.__dict__['_pc'] = Spy()
  thf, hid, e_init, e_final, e_iter, ee_iter, jb, tb = planner.forward(L['th'], L['start'], L['goal'], (sdf.detach() > 0).double(), L['sdf'])
  assert type(thf.grad_fn).__name__.startswith('_GNSolve') and calls == ['gn_solve_traced']
  assert jb == list(g['iters']) and hid is None
  assert rel_err(thf.detach().cpu().numpy(), g['th_final']) < 1e-8
  assert rel_err(np.asarray(e_init), g['err_init']) < 1e-10 and rel_err(np.asarray(e_final), g['err_final']) < 1e-7
  for b in range(B):
    assert rel_err(np.asarray(e_iter[b]), g['err_iter'][b, :jb[b]]) < 1e-8 and rel_err(np.asarray(ee_iter[b]), g['errext_iter'][b, :jb[b]]) < 1e-8
  gr = torch.autograd.grad((T(g['gbar']) * thf).sum(), [L['th'], L['sdf'], L['start'], L['goal']])
  assert calls == ['gn_solve_traced', 'gn_solve_backward']
  for got, key in zip(gr, ('g_th0', 'g_sdf', 'g_start', 'g_goal')):
    assert rel_err(got.cpu().numpy(), g[key]) < 5e-8, (key, rel_err(got.cpu().numpy(), g[key]))
  # a subset of the inputs (the example differentiates w.r.t. the grid only), a shared grid as an expand()ed view, float32 tensors
  planner.plan_layer.__dict__['_pc'] = pc
  sdf1 = T(O.circles_sdf(G, g['circles']))[None, None].requires_grad_(True)
  thf1 = planner.forward(T(g['th0']), T(g['start']), T(g['goal']), None, sdf1.expand(B, 1, G, G))[0]
  g1, = torch.autograd.grad((T(g['gbar']) * thf1).sum(), [sdf1])
  sdfB = sdf1.detach().repeat(B, 1, 1, 1).requires_grad_(True)
  thf2 = planner.forward(T(g['th0']), T(g['start']), T(g['goal']), None, sdfB)[0]
  g2, = torch.autograd.grad((T(g['gbar']) * thf2).sum(), [sdfB])
  assert torch.equal(thf1, thf2) and rel_err(g1.cpu().numpy(), g2.sum(0, keepdim=True).cpu().numpy()) < 1e-10
  f32 = torch.float32
  th32 = T(g['th0'], f32).requires_grad_(True)
  thf3 = planner.forward(th32, T(g['start'], f32), T(g['goal'], f32), None, sdf.detach().to(f32))[0]
  g3, = torch.autograd.grad((T(g['gbar'], f32) * thf3).sum(), [th32])
  assert thf3.dtype == f32 and g3.dtype == f32 and rel_err(g3.cpu().numpy().astype(np.float64), g['g_th0']) < 5e-3
  with pytest.raises(RuntimeError):               # a raw kernel behind the node: a double backward must raise, not return zeros
    thf4 = planner.forward(T(g['th0']).requires_grad_(True), T(g['start']), T(g['goal']), None, sdf.detach())[0]
    gg, = torch.autograd.grad(thf4.sum(), [thf4.grad_fn.next_functions[0][0].variable], create_graph=True)
    gg.sum().backward()


def test_step_with_errors_matches_reference_and_the_two_calls_it_replaces(golden):
  """planner.step_with_errors == step() followed by unweighted_errors_batch(th + dtheta) (learning/train_planner.py:311,313,327) in ONE autograd
  node: values and every gradient against the reference's autograd through exactly that composition (fixture g7_errors (b); the fixture's loss
  also holds error_ext_batch(th + dtheta), evaluated here through the existing method), and against the two-call sequence."""
  g = golden('g7_errors')
  B, n, G = 4, 16, int(g['G'])
  planner = make_planner(n, B)
  names = ('th', 'sdf', 'start', 'goal', 'qc', 'ow', 'eps')
  c_sg, c_gp, c_obs, c_ee = T(g['c_sg']), T(g['c_gp']), T(g['c_obs']), T(g['c_ee'])

  def leaves():
    L = {k: T(g[k]).requires_grad_(True) for k in ('th', 'start', 'goal', 'qc', 'ow', 'eps')}
    L['sdf'] = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1).requires_grad_(True)
    return L
  L = leaves()
  pl = planner.plan_layer
  dth, err, eex, e_sg, e_gp, e_obs = pl.forward_with_errors(L['th'], L['start'], L['goal'], None, L['sdf'], L['qc'], L['ow'], L['eps'])
  assert e_sg.shape == (B, 1) and e_gp.shape == (B, 1, 1) and e_obs.shape == (B, 1, 1) and not err.requires_grad
  assert dth.grad_fn is e_sg.grad_fn.next_functions[0][0] or type(dth.grad_fn).__name__.startswith('_GNStepErrors')
  for got, key in ((dth, 'b_dth'), (e_sg, 'b_sg'), (e_gp, 'b_gp'), (e_obs, 'b_obs')):
    assert rel_err(got.detach().cpu().numpy(), g[key]) < 1e-9, key
  e_ee = planner.error_ext_batch(L['th'] + dth, L['sdf'])
  loss = (c_sg * e_sg).sum() + (c_gp * e_gp).sum() + (c_obs * e_obs).sum() + (c_ee * e_ee).sum()
  gr = torch.autograd.grad(loss, [L[k] for k in names])
  for k, gk in zip(names, gr):
    assert rel_err(gk.cpu().numpy(), g['b_g_' + k]) < 1e-9, (k, rel_err(gk.cpu().numpy(), g['b_g_' + k]))
  # == the two calls, value for value and gradient for gradient (static covariances through the planner-level method)
  L2 = leaves()
  gb = T(np.random.RandomState(5).randn(B, n, 4))
  out, (s1, g1, o1) = planner.step_with_errors(L2['th'], L2['start'], L2['goal'], None, L2['sdf'])
  l1 = (gb * out[0]).sum() + out[3].sum() + (c_sg * s1).sum() + (c_gp * g1).sum() + (c_obs * o1).sum()
  ga = torch.autograd.grad(l1, [L2[k] for k in ('th', 'sdf', 'start', 'goal')])
  L3 = leaves()
  out3 = planner.step(L3['th'], L3['start'], L3['goal'], None, L3['sdf'])
  s3, g3, o3 = planner.unweighted_errors_batch(L3['th'] + out3[0], L3['sdf'])
  l3 = (gb * out3[0]).sum() + out3[3].sum() + (c_sg * s3).sum() + (c_gp * g3).sum() + (c_obs * o3).sum()
  gc = torch.autograd.grad(l3, [L3[k] for k in ('th', 'sdf', 'start', 'goal')])
  # the fused call runs the step kernels' errors-epilogue twins (a separate compilation of the same source): equal to rounding, not bit for bit
  for a, c in ((out[0], out3[0]), (s1, s3), (g1, g3), (o1, o3), (out[2], out3[2])):
    assert rel_err(a.detach().cpu().numpy(), c.detach().cpu().numpy()) < 1e-12
  for a, c in zip(ga, gc): assert rel_err(a.cpu().numpy(), c.cpu().numpy()) < 1e-9      # (the fixture's own bar above; 1.6e-10 measured on the grid gradient)
  with torch.no_grad():
    out4, (s4, g4, o4) = planner.step_with_errors(L2['th'], L2['start'], L2['goal'], None, L2['sdf'])
  assert out4[0].grad_fn is None and torch.equal(out4[0], out[0]) and torch.equal(s4, s1)


def test_unweighted_errors_and_error_ext_are_differentiable_like_the_reference(golden):
  """unweighted_errors_batch / error_ext_batch carry the graph the reference's plain torch ops carry (plan_layer.py:310-345,
  374-388): w.r.t. the trajectory, the grid and -- remembered WITH their graphs by the last forward(), :88-94 -- the start / goal
  means and the current eps.  Fixture g7_errors = the reference's autograd, (a) at a leaf trajectory, (b) at th + dtheta."""
  g = golden('g7_errors')
  B, n, G = 4, 16, int(g['G'])
  planner = make_planner(n, B)
  names = ('th', 'sdf', 'start', 'goal', 'qc', 'ow', 'eps')
  c_sg, c_gp, c_obs, c_ee = T(g['c_sg']), T(g['c_gp']), T(g['c_obs']), T(g['c_ee'])

  def leaves():
    L = {k: T(g[k]).requires_grad_(True) for k in ('th', 'start', 'goal', 'qc', 'ow', 'eps')}
    L['sdf'] = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1).requires_grad_(True)
    return L
  # (a)
  L = leaves()
  the = T(g['th_eval']).requires_grad_(True)
  planner.plan_layer(L['th'], L['start'], L['goal'], None, L['sdf'], L['qc'], L['ow'], L['eps'])
  e_sg, e_gp, e_obs = planner.unweighted_errors_batch(the, L['sdf'])
  e_ee = planner.error_ext_batch(the, L['sdf'])
  assert e_sg.shape == (B, 1) and e_gp.shape == (B, 1, 1) and e_obs.shape == (B, 1, 1) and e_ee.shape == (B, 1, 1)
  for got, key in ((e_sg, 'a_sg'), (e_gp, 'a_gp'), (e_obs, 'a_obs'), (e_ee, 'a_ee')):
    assert got.requires_grad and rel_err(got.detach().cpu().numpy(), g[key]) < 1e-11, key
  for tag, loss in (('unw', (c_sg * e_sg).sum() + (c_gp * e_gp).sum() + (c_obs * e_obs).sum()), ('ee', (c_ee * e_ee).sum())):
    gr = torch.autograd.grad(loss, [the] + [L[k] for k in names], retain_graph=True, allow_unused=True)
    assert rel_err(gr[0].cpu().numpy(), g['a_%s_g_th_eval' % tag]) < 1e-10, tag
    for k, gk in zip(names, gr[1:]):
      if bool(g['a_%s_none_%s' % (tag, k)]):
        assert gk is None or float(gk.abs().max()) == 0.0, (tag, k)
      else:
        assert gk is not None and rel_err(gk.cpu().numpy(), g['a_%s_g_%s' % (tag, k)]) < 1e-10, (tag, k)
  # the one-by-one methods of PlanLayer carry the same graphs
  pl = planner.plan_layer
  gr = torch.autograd.grad((c_sg * pl.start_goal_error(the)).sum() + (c_gp * pl.gp_error(the)).sum() + (c_obs * pl.obs_error(the, L['sdf'])).sum(),
                           [the, L['start'], L['eps']])
  assert rel_err(gr[0].cpu().numpy(), g['a_unw_g_th_eval']) < 1e-10 and rel_err(gr[1].cpu().numpy(), g['a_unw_g_start']) < 1e-10
  assert rel_err(gr[2].cpu().numpy(), g['a_unw_g_eps']) < 1e-10
  # (b) learning/train_planner.py:313,327: the errors at th + dtheta, every leaf learnable
  L = leaves()
  dth, err, err_ext = planner.plan_layer(L['th'], L['start'], L['goal'], None, L['sdf'], L['qc'], L['ow'], L['eps'])
  th_new = L['th'] + dth
  e_sg, e_gp, e_obs = planner.unweighted_errors_batch(th_new, L['sdf'])
  e_ee = planner.error_ext_batch(th_new, L['sdf'])
  assert rel_err(dth.detach().cpu().numpy(), g['b_dth']) < 1e-9
  for got, key in ((e_sg, 'b_sg'), (e_gp, 'b_gp'), (e_obs, 'b_obs'), (e_ee, 'b_ee')):
    assert rel_err(got.detach().cpu().numpy(), g[key]) < 1e-9, key
  loss = (c_sg * e_sg).sum() + (c_gp * e_gp).sum() + (c_obs * e_obs).sum() + (c_ee * e_ee).sum()
  gr = torch.autograd.grad(loss, [L[k] for k in names])
  for k, gk in zip(names, gr):
    assert rel_err(gk.cpu().numpy(), g['b_g_' + k]) < 1e-8, k
  # under no_grad nothing is recorded
  with torch.no_grad():
    assert not planner.unweighted_errors_batch(the, L['sdf'])[0].requires_grad and not planner.error_ext_batch(the, L['sdf']).requires_grad


def test_forward_with_grad_keeps_graph_across_iterations(golden):
  """examples/diff_gpmp2_2d_example.py:77: th_final.backward(...) through the whole forward()."""
  from dgpmp2_amd.utils.planner_utils import straight_line_traj
  c1 = golden('g3_c1')
  planner = make_planner(32, 1, max_iters=3)
  start, goal = T(c1['start'][0]), T(c1['goal'][0])
  th_init = straight_line_traj(start[:, :2], goal[:, :2], 10.0, 31, 2, DEV).requires_grad_(True)
  sdf = T(c1['sdf'])
  out = planner.forward(th_init.unsqueeze(0), start.unsqueeze(0), goal.unsqueeze(0), None, sdf.unsqueeze(0).unsqueeze(0))
  th_final = out[0]
  assert out[6] == [3] and th_final.requires_grad
  assert rel_err(th_final.detach().cpu().numpy(), c1['th_hist'][3]) < 1e-9
  th_final.backward(torch.randn(th_final.shape, dtype=torch.float64, device=DEV))
  assert th_init.grad is not None and bool(torch.isfinite(th_init.grad).all()) and float(th_init.grad.abs().max()) > 0


# ---- learned-covariance plumbing: user-supplied learn modules -> get_covariances -> solver input modes ----
class _ConvStub(torch.nn.Module):
  def forward(self, im):
    return im.mean(dim=(2, 3)), None


class _FcnStub(torch.nn.Module):
  """Stands in for LearnModuleFCN: (th, conv_out) -> (B,1,out_dim), positive and smooth in its inputs."""

  def __init__(self, out_dim):
    super(_FcnStub, self).__init__()
    self.w = torch.nn.Parameter(torch.linspace(0.5, 1.5, out_dim, dtype=torch.float64))

  def forward(self, th, conv_out):
    s = 1.0 + 0.01 * th.mean(dim=(1, 2), keepdim=True) + 0.0 * conv_out.mean()
    return (self.w.view(1, 1, -1) * s)


@pytest.mark.parametrize('mode,learn_eps', [('fix_dynamics', False), ('diag_identity', True), ('qc_full', False), ('q_full', True)])
def test_learned_covariance_modes(golden, mode, learn_eps):
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  g = golden('g2_system_n16')
  B, n = 3, 16
  gp, ob, pp, op, ev = ref_params(n)
  n_gp = {'fix_dynamics': 0, 'diag_identity': n - 1, 'qc_full': (n - 1) * 2, 'q_full': (n - 1) * 4}[mode]
  out_dim = n_gp + n + (n if learn_eps else 0)
  lp = {'model': {'type': 'feed_forward'}, 'dgpmp2': {'learn_eps': learn_eps, 'sdf_predict': False, 'dtheta_predict': False,
                                                     'dynamics_mode': mode, 'fixed_conv': False}, 'data': {'im_size': 64}}
  planner = DiffGPMP2Planner(gp, ob, pp, op, ev, PointRobot2D(torch.tensor(0.4, dtype=torch.float64), B, n), learn_params=lp,
                             batch_size=B, use_cuda=True, learn_module_conv=_ConvStub(), learn_module_fcn=_FcnStub(out_dim).to(DEV))
  G = int(g['G'])
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  th, st, go = T(g['th']).requires_grad_(True), T(g['start']), T(g['goal'])
  dth, hidden, err, err_ext, qc, ow, eps = planner.step(th, st, go, (sdf > 0).double(), sdf)
  d = 4 if mode == 'q_full' else 2
  assert qc.shape == (B, n - 1, d, d) and ow.shape == (B, n, 1, 1) and eps.shape == (B, n, 1, 1)
  p = PC_P2d(n)
  r_dth, r_err, r_eex = O.plan_layer_forward(g['th'], g['start'], g['goal'], np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)),
                                             qc.detach().cpu().numpy(), ow.detach().cpu().numpy(), eps.detach().cpu().numpy(), p,
                                             q_full=(mode == 'q_full'))
  assert rel_err(dth.detach().cpu().numpy(), r_dth) < 1e-9
  assert rel_err(err.cpu().numpy(), r_err) < 1e-11 and rel_err(err_ext.detach().cpu().numpy(), r_eex) < 1e-11
  # gradients reach the learn module's parameters through the covariance inputs of the solver
  (dth ** 2).sum().backward()
  gw = planner.learn_module_fcn.w.grad
  assert gw is not None and bool(torch.isfinite(gw).all()) and float(gw.abs().max()) > 0


@pytest.mark.parametrize('mode,learn_eps,dtype', [('diag_identity', False, torch.float64), ('diag_identity', True, torch.float32), ('fix_dynamics', True, torch.float64),
                                                  ('fix_dynamics', False, torch.float32)])
def test_step_takes_the_module_output_raw_and_matches_get_covariances(golden, mode, learn_eps, dtype):
  """planner.step() / step_with_errors() in 'diag_identity' / 'fix_dynamics' hand the learn module's output vector to the kernels (DGP_COVS_SQUARED: squared there,
  d/d out written by the backward kernel) -- against the explicit route through get_covariances (diff_gpmp2_planner.py:247-290) + PlanLayer.forward: same update,
  errors, returned covariance tensors and parameter gradients; a cotangent on the returned tensors themselves is chained too."""
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  from dgpmp2_amd.gpmp2 import plan_layer as PLm
  g = golden('g2_system_n16')
  B, n = 3, 16
  gp, ob, pp, op, ev = ref_params(n)
  n_gp = {'fix_dynamics': 0, 'diag_identity': n - 1}[mode]
  out_dim = n_gp + n + (n if learn_eps else 0)
  lp = {'model': {'type': 'feed_forward'}, 'dgpmp2': {'learn_eps': learn_eps, 'sdf_predict': False, 'dtheta_predict': False,
                                                     'dynamics_mode': mode, 'fixed_conv': False}, 'data': {'im_size': 64}}
  fcn = _FcnStub(out_dim).to(DEV).to(dtype)
  with torch.no_grad():
    fcn.w[n_gp:n_gp + n] *= 70.0                              # obstacle weights o^2 ~ 1e3 .. 1e4
    if learn_eps: fcn.w[n_gp + n:] *= 0.6
  planner = DiffGPMP2Planner(gp, ob, pp, op, ev, PointRobot2D(torch.tensor(0.4, dtype=torch.float64), B, n), learn_params=lp,
                             batch_size=B, use_cuda=True, learn_module_conv=_ConvStub(), learn_module_fcn=fcn)
  pl = planner.plan_layer
  G = int(g['G'])
  sdf = T(O.circles_sdf(G, g['circles']), dtype)[None, None].expand(B, 1, G, G)
  im = (sdf > 0).to(dtype)
  th, st, go = T(g['th'], dtype), T(g['start'], dtype), T(g['goal'], dtype)
  gd = torch.randn(B, n, 4, device=DEV, dtype=dtype, generator=torch.Generator(device=DEV).manual_seed(3))
  tol = 1e-11 if dtype is torch.float64 else 2e-5

  def loss(dth, eex, sg, gp_, ob_, qc, ow, eps, with_cov):
    l = (dth * gd).sum() + 2.0 * eex.sum() + 3.0 * sg.sum() + 5.0 * gp_.sum() + 7.0 * ob_.sum()
    if with_cov: l = l + 1e-6 * (ow * ow).sum() + (0.1 * (qc * qc).sum() if qc.requires_grad else 0.0) + (0.3 * eps.sum() if eps.requires_grad else 0.0)
    return l

  def explicit(with_cov):       # the reference's route: module output -> get_covariances -> tensors into the layer
    thr = th.clone().requires_grad_(True)
    conv_out, _ = planner.learn_module_conv(im)
    out = fcn(thr, conv_out)
    r = planner.get_covariances(out, mode, learn_eps)
    if mode == 'fix_dynamics':
      ow, eps = (r if learn_eps else (r, None)); qc = planner._static_view(planner.qc_inv_traj, B, thr)
    else:
      qc, ow = r[0], r[1]; eps = r[2] if learn_eps else None
    if eps is None: eps = planner._static_view(planner.eps_traj, B, thr)
    dth, err, eex, sg, gp_, ob_ = pl.forward_with_errors(thr, st, go, im, sdf, qc, ow, eps)
    gw, gt = torch.autograd.grad(loss(dth, eex, sg, gp_, ob_, qc, ow, eps, with_cov), (fcn.w, thr))
    return dth.detach(), err, eex.detach(), sg.detach(), ob_.detach(), qc.detach(), ow.detach(), eps.detach(), gw, gt

  def raw(with_cov, fused):
    thr = th.clone().requires_grad_(True)
    if fused:
      (dth, _, err, eex, qc, ow, eps), (sg, gp_, ob_) = planner.step_with_errors(thr, st, go, im, sdf)
    else:
      dth, _, err, eex, qc, ow, eps = planner.step(thr, st, go, im, sdf)
      sg, gp_, ob_ = planner.unweighted_errors_batch(thr + dth, sdf)
    gw, gt = torch.autograd.grad(loss(dth, eex, sg.reshape(B, 1), gp_, ob_, qc, ow, eps, with_cov), (fcn.w, thr))
    return dth.detach(), err, eex.detach(), sg.detach().reshape(B, 1), ob_.detach(), qc.detach(), ow.detach(), eps.detach(), gw, gt

  calls = []
  orig = pl.forward_raw
  pl.forward_raw = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
  for with_cov in (False, True):
    e = explicit(with_cov)
    for fused in (True, False):
      r = raw(with_cov, fused)
      for i, (a_, b_) in enumerate(zip(r, e)):
        assert a_.shape == b_.shape or a_.numel() == b_.numel(), i
        assert rel_err(a_.double().cpu().numpy().reshape(-1), b_.double().cpu().numpy().reshape(-1)) < (tol if i < 8 else 50 * tol), (with_cov, fused, i)
  assert len(calls) == 4                                       # the raw route really ran
  # without a graph: same numbers
  with torch.no_grad():
    d0 = planner.step(th, st, go, im, sdf)[0]
  assert rel_err(d0.double().cpu().numpy(), e[0].double().cpu().numpy()) < tol


def test_planner_accepts_tiled_grids(golden):
  """utils.sdf_utils.tile_sdf / sdf_2d_batch(layout='tiled4'): a (B,1,Ht,Wt,4,4) tensor in place of sdfb -- step(), forward(), the error helpers and autograd give
  what the row-major grids give; the gradient w.r.t. the tiled tensor is tiled (untile_sdf -> the row-major gradient); shared (expand()ed) tiled grids too."""
  from dgpmp2_amd.utils.sdf_utils import tile_sdf, untile_sdf
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  rs = np.random.RandomState(3)
  base = O.circles_sdf(G, g['circles'])
  sdf = T(np.stack([base + 0.03 * rs.randn(G, G) for _ in range(B)])[:, None])
  start, goal, th = T(g['start']), T(g['goal']), T(g['th_hist'][2])
  gd = T(rs.randn(B, n, 4))
  sdf_r = sdf.clone().requires_grad_(True)
  sdf_t = tile_sdf(sdf).requires_grad_(True)
  assert sdf_t.shape == (B, 1, G // 4, G // 4, 4, 4) and torch.equal(untile_sdf(sdf_t.detach()), sdf)
  thr = th.clone().requires_grad_(True); tht = th.clone().requires_grad_(True)
  planner.plan_layer.sdf_grad = 'dense'
  # (equal to rounding, not bit for bit: the tiled grids run the tiled translation units -- a separate compilation of the same source)
  close = lambda a_, b_, tol=1e-11: rel_err(a_.detach().cpu().numpy(), b_.detach().cpu().numpy()) < tol
  d_r = planner.step(thr, start, goal, None, sdf_r); d_t = planner.step(tht, start, goal, None, sdf_t)
  assert close(d_r[0], d_t[0]) and close(d_r[2], d_t[2]) and close(d_r[3], d_t[3])
  sg_r, gp_r, ob_r = planner.unweighted_errors_batch(thr + d_r[0], sdf_r); sg_t, gp_t, ob_t = planner.unweighted_errors_batch(tht + d_t[0], sdf_t)
  assert close(ob_r, ob_t)
  ((d_r[0] * gd).sum() + ob_r.sum()).backward(); ((d_t[0] * gd).sum() + ob_t.sum()).backward()
  assert close(thr.grad, tht.grad, 1e-9) and sdf_t.grad.shape == sdf_t.shape
  assert rel_err(untile_sdf(sdf_t.grad, (G, G)).cpu().numpy(), sdf_r.grad.cpu().numpy()) < 1e-9 and float(sdf_r.grad.abs().max()) > 0
  # ... and as sparse taps: a sparse tensor of the TILED tensor's shape (six index rows), no (B,1,H,W)-sized zero fill
  planner.plan_layer.sdf_grad = 'sparse'
  sdf_s = tile_sdf(sdf).requires_grad_(True); ths = th.clone().requires_grad_(True)
  d_s = planner.step(ths, start, goal, None, sdf_s)
  sg_s, gp_s, ob_s = planner.unweighted_errors_batch(ths + d_s[0], sdf_s)
  ((d_s[0] * gd).sum() + ob_s.sum()).backward()
  assert sdf_s.grad.is_sparse and sdf_s.grad.shape == sdf_s.shape
  assert rel_err(untile_sdf(sdf_s.grad.to_dense(), (G, G)).cpu().numpy(), sdf_r.grad.cpu().numpy()) < 1e-9 and close(ths.grad, thr.grad, 1e-9)
  planner.plan_layer.sdf_grad = 'dense'
  # forward(): the fused loop; a shared tiled grid as an expand()ed view, with its gradient
  one = T(base)[None, None]
  smooth = one.repeat(B, 1, 1, 1)      # (per-sample copies of the noise-free grid: ten chained non-converging solves on the noisy ones amplify the rounding differences of the two compilations)
  f_r = planner.forward(th, start, goal, None, smooth); f_t = planner.forward(th, start, goal, None, tile_sdf(smooth))
  assert close(f_r[0], f_t[0], 1e-5) and f_r[6] == f_t[6]
  s_r = one.clone().requires_grad_(True); s_t = tile_sdf(one).requires_grad_(True)
  a_r = planner.step(th, start, goal, None, s_r.expand(B, 1, G, G))[0]; a_t = planner.step(th, start, goal, None, s_t.expand(B, 1, G // 4, G // 4, 4, 4))[0]
  assert close(a_r, a_t)
  (a_r * gd).sum().backward(); (a_t * gd).sum().backward()
  assert rel_err(untile_sdf(s_t.grad, (G, G)).cpu().numpy(), s_r.grad.cpu().numpy()) < 1e-9


def PC_P2d(n):
  return O.OracleParams(dof=2, total_time_step=n - 1)


def test_planner_api_long_trajectory():
  """total_time_step = 299 (n = 300 > 256: the loop kernels of csrc/gn_long.h) through the reference API: step() against the block-tridiagonal C oracle,
  forward() (the fused loop) equal to chained step() calls, and the backward of step() against torch autograd over the dense restatement."""
  from oracle import blocktri as BT, autograd_torch as AT
  B, n, G = 3, 300, 64
  planner = make_planner(n, B, max_iters=3, tol_delta=0.0)
  p = O.OracleParams(dof=2, total_time_step=n - 1)
  rs = np.random.RandomState(5)
  start = np.zeros((B, 1, 4)); goal = np.zeros((B, 1, 4))
  start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2))
  th = O.straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + rs.randn(B, n, 4) * 0.03
  sdf_np = O.circles_sdf(G, O.C2_CIRCLES)[None, None]
  sdf = T(sdf_np).expand(B, 1, G, G)
  dth, _, err, eex, _, _, _ = planner.step(T(th), T(start), T(goal), None, sdf)
  c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf_np)
  assert rel_err(dth.cpu().numpy(), c_dth) < 1e-9 and rel_err(err.cpu().numpy().reshape(-1), c_err) < 1e-11 and not c_info.any()
  assert planner.plan_layer._solver(torch.float64).launch_shape(B) == (64, 5)
  # forward(): three fused iterations == three chained steps
  th_f, _, e0, ef, eh, eeh, k, _ = planner.forward(T(th), T(start), T(goal), None, sdf)
  cur = T(th)
  for _ in range(3): cur = cur + planner.step(cur, T(start), T(goal), None, sdf)[0]
  assert k == [3] * B and rel_err(th_f.cpu().numpy(), cur.cpu().numpy()) < 1e-9
  # backward of one step
  thr = T(th).requires_grad_(True)
  gbar = rs.randn(B, n, 4)
  d2 = planner.step(thr, T(start), T(goal), None, sdf)[0]
  (gth,) = torch.autograd.grad(d2, thr, T(gbar))
  g_o = AT.step_gradients(p, th, start, goal, sdf_np, gbar, np.zeros(B))
  assert rel_err(gth.cpu().numpy(), g_o['th']) < 1e-6


def test_auto_tile_gives_the_rowmajor_results_in_rowmajor_shapes(golden):
  """plan_layer.auto_tile (round 6): a per-sample ROW-MAJOR sdfb -- what an unmodified reference caller passes -- is tiled once per batch of grids and every launch reads
  the tiles; results, dense and sparse gradients come back as for the row-major path, in sdfb's own shape; the tiling pass runs once per (tensor, version)."""
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  pl = planner.plan_layer
  rs = np.random.RandomState(11)
  base = O.circles_sdf(G, g['circles'])
  sdf = T(np.stack([base + 0.03 * rs.randn(G, G) for _ in range(B)])[:, None])
  start, goal, th = T(g['start']), T(g['goal']), T(g['th_hist'][2])
  gd = T(rs.randn(B, n, 4))
  close = lambda a_, b_, tol=1e-11: rel_err(a_.detach().cpu().numpy(), b_.detach().cpu().numpy()) < tol

  def run(leaf):
    thr = th.clone().requires_grad_(True)
    out, (sg, gp, ob) = planner.step_with_errors(thr, start, goal, None, leaf)
    ((out[0] * gd).sum() + ob.sum() + out[3].sum()).backward()
    return out[0].detach(), out[2], ob.detach(), thr.grad, leaf.grad

  ref = run(sdf.clone().requires_grad_(True))
  pl.auto_tile = True
  leaf = sdf.clone().requires_grad_(True)
  got = run(leaf)
  tiles = pl._tile_cache[5]
  assert close(ref[0], got[0]) and close(ref[1], got[1]) and close(ref[2], got[2]) and close(ref[3], got[3], 1e-9)
  assert got[4].shape == sdf.shape and not got[4].is_sparse and close(ref[4], got[4], 1e-9)
  leaf.grad = None
  run(leaf)
  assert pl._tile_cache[5] is tiles                                  # the same batch of grids: no second tiling pass
  pl.sdf_grad = 'sparse'
  leaf2 = sdf.clone().requires_grad_(True)
  got2 = run(leaf2)
  assert leaf2.grad.is_sparse and leaf2.grad.shape == sdf.shape and close(ref[4], leaf2.grad.to_dense(), 1e-9)
  pl.sdf_grad = 'dense'
  # planning calls (no graph) and the fused loop read the tiles too
  with torch.no_grad():
    a = planner.step(th, start, goal, None, sdf)[0]
  assert close(ref[0], a)
  smooth = T(base)[None, None].repeat(B, 1, 1, 1)
  f_t = planner.forward(th, start, goal, None, smooth)
  pl.auto_tile = False
  f_r = planner.forward(th, start, goal, None, smooth)
  assert close(f_r[0], f_t[0], 1e-5) and f_r[6] == f_t[6]
  # forward() returns python lists by default, as the reference does; the lazy views are an opt-in
  assert all(type(x) is list for x in f_r[2:8]) and type(f_r[4][0]) is list and len(f_r[4]) == B
  planner.lazy_results = True
  f_l = planner.forward(th, start, goal, None, smooth)
  assert type(f_l[2]) is not list and f_l[2] == f_r[2] and f_l[4] == f_r[4] and f_l[6] + [1] == f_r[6] + [1]


def test_graphed_iteration_replays_the_eager_iteration(golden):
  """planner.graphed_iteration(fn) (round 6): a training iteration written eagerly against the planner -- step_with_errors with learned per-state covariances, a loss,
  autograd.grad -- is captured in a HIP graph at the first call and replayed afterwards: the results equal the eager call's bit for bit for every new set of inputs,
  one capture serves all calls of a signature, a new batch shape captures again."""
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, B)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  start, goal = T(g['start']), T(g['goal'])
  gen = torch.Generator(device=DEV).manual_seed(5)
  A = torch.randn(B, n - 1, 2, 2, device=DEV, dtype=torch.float64, generator=gen) * 0.2
  qc = (torch.eye(2, device=DEV, dtype=torch.float64) + A @ A.transpose(-1, -2)).requires_grad_(True)
  ow = (torch.rand(B, n, 1, 1, device=DEV, dtype=torch.float64, generator=gen) * 1e4 + 50).requires_grad_(True)
  ep = (torch.rand(B, n, 1, 1, device=DEV, dtype=torch.float64, generator=gen) * 0.5 + 0.1).requires_grad_(True)

  def iteration(th, th_opt, qc_, ow_, ep_):
    dth, _, eex, sg, gp_, ob = planner.plan_layer.forward_with_errors(th, start, goal, None, sdf, qc_, ow_, ep_)
    loss = ((dth - (th_opt - th)) ** 2).sum(-1).mean() + 1e-3 * (sg.mean() + gp_.mean() + ob.mean())
    return (dth, loss) + torch.autograd.grad(loss, (qc_, ow_, ep_))

  it = planner.graphed_iteration(iteration)
  for k in (1, 3, 5):
    th, th_opt = T(g['th_hist'][k]), T(g['th_hist'][9])
    want = [t.clone() for t in iteration(th, th_opt, qc, ow, ep)]
    got = it(th, th_opt, qc, ow, ep)
    torch.cuda.synchronize()
    for a, b in zip(got, want):
      assert torch.equal(a, b)
  assert it.captures == 1
  with torch.no_grad(): ow.mul_(0.5)                                 # new values in a long-lived input: copied into the graph's static tensor
  th, th_opt = T(g['th_hist'][2]), T(g['th_hist'][9])
  want = [t.clone() for t in iteration(th, th_opt, qc, ow, ep)]
  for a, b in zip(it(th, th_opt, qc, ow, ep), want): assert torch.equal(a, b)
  assert it.captures == 1
  # the copy-free form: write into the graph's own input tensors, replay
  th, th_opt = T(g['th_hist'][4]), T(g['th_hist'][9])
  want = [t.clone() for t in iteration(th, th_opt, qc, ow, ep)]
  statics = it.static_inputs(th, th_opt, qc, ow, ep)
  with torch.no_grad():
    statics[0].copy_(th); statics[1].copy_(th_opt)
  for a, b in zip(it.replay(th, th_opt, qc, ow, ep), want): assert torch.equal(a, b)
  assert it.captures == 1
  # another batch size is another signature (the planner itself is batch-agnostic at this level)
  it2 = planner.graphed_iteration(lambda th_: planner.step(th_, start[:4], goal[:4], None, sdf[:4])[0])
  with torch.no_grad():
    a4 = it2(T(g['th_hist'][1])[:4]).clone(); e4 = planner.step(T(g['th_hist'][1])[:4], start[:4], goal[:4], None, sdf[:4])[0]
  assert torch.equal(a4, e4)


def test_tiled_grid_size_travels_with_the_tensor_on_the_gpu(golden):
  """a padded 130 x 130 field (33 x 33 tiles, like 132 x 132) keeps its logical size through .to(device) / .float() / indexing / clone -- the step on the
  tiles equals the step on the row-major field; a plain tensor of the same tiles is refused."""
  from dgpmp2_amd.utils.sdf_utils import tile_sdf
  g = golden('g3_c2mini')
  B, n = 8, 64
  planner = make_planner(n, B)
  rs = np.random.RandomState(2)
  H = 130
  yy, xx = np.meshgrid(np.linspace(5, -5, H), np.linspace(-5, 5, H), indexing='ij')
  field = np.sqrt((xx - 0.5) ** 2 + (yy + 0.3) ** 2) - 1.1
  cpu = torch.from_numpy(np.stack([field + 0.02 * rs.randn(H, H) for _ in range(B)])[:, None])
  tiles = tile_sdf(cpu).to(DEV).clone()[0:B]
  assert tiles.hw == (H, H) and tuple(tiles.shape) == (B, 1, 33, 33, 4, 4)
  start, goal, th = T(g['start']), T(g['goal']), T(g['th_hist'][2])
  with torch.no_grad():
    d_t = planner.step(th, start, goal, None, tiles)[0]
    d_r = planner.step(th, start, goal, None, cpu.to(DEV))[0]
    d_f = planner.step(th.float(), start.float(), goal.float(), None, tiles.float())[0]
  assert rel_err(d_t.cpu().numpy(), d_r.cpu().numpy()) < 1e-11 and rel_err(d_f.double().cpu().numpy(), d_r.cpu().numpy()) < 1e-4
  with pytest.raises(ValueError, match='carries no logical grid size'):
    planner.step(th, start, goal, None, tiles.as_subclass(torch.Tensor))


def test_forward_with_grad_and_a_non_diagonal_qc_is_two_launches(golden):
  """Round 6: gp_params['Q_c_inv'] non-diagonal -- the differentiable forward() is still ONE fused launch + ONE backward launch (the general-covariance
  chain kernels), against the reference's autograd through its own forward() with that Q_c_inv (fixture g8_forward_grads_qc)."""
  from dgpmp2_amd.robot_models import PointRobot2D
  from dgpmp2_amd.gpmp2 import DiffGPMP2Planner
  g = golden('g8_forward_grads_qc')
  B, n, G = 4, 16, int(g['G'])
  gp, ob, pp, op, ev = ref_params(n, max_iters=int(g['max_iters']), tol_delta=float(g['tol_delta']))
  gp['Q_c_inv'] = torch.tensor(g['Q_c_inv'], dtype=torch.float64)
  planner = DiffGPMP2Planner(gp, ob, pp, op, ev, PointRobot2D(torch.tensor(0.4, dtype=torch.float64), B, n, use_cuda=True), batch_size=B, use_cuda=True)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].repeat(B, 1, 1, 1)
  sdf[int(g['free_sample'])] = float(g['free_value'])
  L = [T(g['th0']).requires_grad_(True), sdf.requires_grad_(True), T(g['start']).requires_grad_(True), T(g['goal']).requires_grad_(True)]
  thf, hid, e_init, e_final, e_iter, ee_iter, jb, tb = planner.forward(L[0], L[2], L[3], None, L[1])
  assert type(thf.grad_fn).__name__.startswith('_GNSolve') and jb == list(g['iters'])
  assert rel_err(thf.detach().cpu().numpy(), g['th_final']) < 1e-8 and rel_err(np.asarray(e_init), g['err_init']) < 1e-10
  gr = torch.autograd.grad((T(g['gbar']) * thf).sum(), L)
  for got, key in zip(gr, ('g_th0', 'g_sdf', 'g_start', 'g_goal')):
    assert rel_err(got.cpu().numpy(), g[key]) < 5e-8, (key, rel_err(got.cpu().numpy(), g[key]))
