"""Pin the numpy oracle (oracle/gpmp2_oracle.py) against every golden fixture generated from the real
reference (tests/golden/make_golden.py).  CPU-only; runs under -m "not gpu"."""
import numpy as np
import pytest
from conftest import rel_err
from oracle import gpmp2_oracle as O

TOL = 1e-9       # fp64 restatement vs fp64 reference: different LAPACK/BLAS paths only


def P2d(n, **kw):
  return O.OracleParams(dof=2, total_time_step=n - 1, **kw)


def test_g1_gp_prior_factors_bit_exact(golden):
  g = golden('g1_factors_2d')
  e, H1, H2 = O.gp_factor_error(g['th'], 2, float(g['dt']))
  assert np.array_equal(e, g['e_gp']) or rel_err(e, g['e_gp']) < 1e-15
  assert np.array_equal(H1, g['H1']) and np.array_equal(H2, g['H2'])
  assert rel_err(O.calc_Q_inv_batch(g['qc'], float(g['dt'])), g['Q_inv']) < 1e-15
  ep, Hp = O.prior_error(g['mean'][:, 0], g['th'][:, 0])
  assert np.array_equal(ep, g['e_p']) and np.array_equal(Hp, g['H_p'])


def test_g1_bilinear_and_hinge_bit_exact(golden):
  g = golden('g1_factors_2d')
  res = 10.0 / g['sdf'].shape[-1]
  d, J = O.bilinear_interpolate(g['sdf'][:, 0], g['th_o'][:, :, 0:2], res, (-5., 5.), (-5., 5.))
  assert np.array_equal(d, g['d_bi']) and np.array_equal(J, g['J_bi'])
  # outside the grid / inside the last row-col cell => d == 0 and J == 0 exactly
  assert np.all(d[0, [0, 1, 3]] == 0.0) and np.all(J[0, [0, 1, 3]] == 0.0)
  p = P2d(16)
  e, H = O.obstacle_error(g['th_o'], g['sdf'], g['eps'], p)
  assert np.array_equal(e, g['e_o']) and np.array_equal(H, g['H_o'])
  e, H = O.obstacle_error(g['th'], g['sdf_tie'], g['eps_tie'], p)      # hinge tie d == eps+r (Q5)
  assert np.array_equal(e, g['e_t']) and np.array_equal(H, g['H_t'])


def test_g1_custom_factors(golden):
  g = golden('g1_factors_custom')
  p = O.OracleParams(dof=2, total_time_step=11, use_vel_limits=True)
  c, H = O.vel_limit_error(g['tr'][None], p)
  assert np.array_equal(c[0, :, :, 0], g['c_v']) and np.array_equal(H[0], g['H_v'])
  assert np.allclose(g['w_v'], np.eye(2) * 1e4)
  e, H = O.nonholonomic_error(g['tr6'][None])
  assert np.array_equal(e[0, :, 0], g['e_d']) and np.array_equal(H[0, :, 0], g['H_d'])
  # survey probe values
  assert abs(e[0, 0, 0, 0] - 0.182148) < 1e-6


@pytest.mark.parametrize('n', [4, 16, 64])
def test_g2_normal_equations(golden, n):
  g = golden('g2_system_n%d' % n)
  p = P2d(n)
  B = g['th'].shape[0]
  sdf = np.broadcast_to(O.circles_sdf(int(g['G']), g['circles']), (B, 1, int(g['G']), int(g['G'])))
  Q_inv = O.calc_Q_inv_batch(g['qc'], p.dt)
  A, b, K = O.construct_linear_system_batch(g['th'], g['start'], g['goal'], sdf, Q_inv, g['ow'], g['eps'], p)
  assert p.M == int(g['M'])
  assert abs(np.linalg.norm(A) - float(g['Anorm'])) < 1e-9 * float(g['Anorm'])
  assert abs(np.linalg.norm(b) - float(g['bnorm'])) < 1e-9 * float(g['bnorm'])
  assert abs(np.linalg.norm(K) - float(g['Knorm'])) < 1e-9 * float(g['Knorm'])
  LAM, R = O.normal_equations(A, b, K, p.reg)
  Dg, Up, off = O.triband(LAM, n, 4)
  assert off == 0.0                       # block-tridiagonal structure
  assert rel_err(Dg, g['Dg']) < 1e-13 and rel_err(Up, g['Up']) < 1e-13
  assert rel_err(R.reshape(B, n, 4), g['eta']) < 1e-13


def test_g3_c1_known_answers_and_steps(golden):
  g = golden('g3_c1')
  # known-answer scalars recorded by the survey
  assert abs(float(g['n32_err0'].item()) - 372.176512415553) < 1e-9
  assert abs(np.linalg.norm(g['n32_dth0']) - 7.279045169006) < 1e-9
  assert abs(float(g['n33_err0'].item()) - 369.172164003341) < 1e-9
  assert abs(float(g['n101_err0'].item()) - 330.436499542839) < 1e-9
  assert abs(float(g['err_after10'].item()) - 12.429170126763) < 1e-9
  sdf = g['sdf'][None, None]
  for n in (32, 33, 101):
    p = P2d(n)
    th = O.straight_line_trajb(g['start'][:, :, :2], g['goal'][:, :, :2], 10.0, n - 1, 2)
    qc, ow, eps = p.static_covs(1)
    dth, err, err_ext = O.plan_layer_forward(th, g['start'], g['goal'], sdf, qc, ow, eps, p)
    assert rel_err(dth, g['n%d_dth0' % n]) < TOL
    assert rel_err(err, g['n%d_err0' % n]) < 1e-12 and rel_err(err_ext, g['n%d_errext0' % n]) < 1e-12
  # 10 teacher-forced steps, n=32
  p = P2d(32); qc, ow, eps = p.static_covs(1)
  for k in range(10):
    dth, err, err_ext = O.plan_layer_forward(g['th_hist'][k], g['start'], g['goal'], sdf, qc, ow, eps, p)
    assert rel_err(dth, g['dth_hist'][k]) < TOL, k
    assert rel_err(err, g['err_hist'][k]) < 1e-12 and rel_err(err_ext, g['errext_hist'][k]) < 1e-12
  usg, ugp, uobs = O.unweighted_errors_batch(g['th_hist'][3], g['start'], g['goal'], sdf, eps, p)
  assert usg.shape == g['unw_sg'].shape and ugp.shape == g['unw_gp'].shape and uobs.shape == g['unw_obs'].shape
  assert rel_err(usg, g['unw_sg']) < 1e-12 and rel_err(ugp, g['unw_gp']) < 1e-12 and rel_err(uobs, g['unw_obs']) < 1e-12


def test_g3_c2mini(golden):
  g = golden('g3_c2mini')
  B, n = 8, 64
  p = P2d(n)
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G))
  qc, ow, eps = p.static_covs(B)
  for k in (0, 4, 9):
    dth, err, err_ext = O.plan_layer_forward(g['th_hist'][k], g['start'], g['goal'], sdf, qc, ow, eps, p)
    assert rel_err(dth, g['dth_hist'][k]) < TOL
    assert rel_err(err, g['err_hist'][k]) < 1e-12 and rel_err(err_ext, g['errext_hist'][k]) < 1e-12
  dth, err, err_ext = O.plan_layer_forward(g['cov_th'], g['start'], g['goal'], sdf, g['cov_qc'], g['cov_ow'], g['cov_eps'], p)
  assert rel_err(dth, g['cov_dth']) < TOL and rel_err(err, g['cov_err']) < 1e-12 and rel_err(err_ext, g['cov_errext']) < 1e-12
  Gp = int(g['ps_G'])
  sdf_ps = np.stack([O.circles_sdf(Gp, g['ps_circles'][b]) for b in range(B)], 0)[:, None]
  dth, err, err_ext = O.plan_layer_forward(g['th_hist'][0], g['start'], g['goal'], sdf_ps, qc, ow, eps, p)
  assert rel_err(dth, g['ps_dth']) < TOL and rel_err(err, g['ps_err']) < 1e-12


def test_g4_forward(golden):
  g = golden('g4_forward')
  c1 = golden('g3_c1')
  p = P2d(32)
  th0 = O.straight_line_trajb(c1['start'][:, :, :2], c1['goal'][:, :, :2], 10.0, 31, 2)
  thf, ei, ef, eh, eeh, it = O.planner_forward(th0, c1['start'], c1['goal'], c1['sdf'][None, None], p,
                                               int(g['c1_max_iters']), float(g['c1_tol_delta']))
  assert it == list(g['c1_iters'])
  assert rel_err(thf, g['c1_th_final']) < 1e-8
  assert rel_err(eh, g['c1_err_iter']) < 1e-9 and rel_err(eeh, g['c1_errext_iter']) < 1e-9
  assert rel_err(ei, g['c1_err_init']) < 1e-12 and rel_err(ef, g['c1_err_final']) < 1e-9
  # early exit by tol_delta
  p = P2d(16)
  sdf = np.full((3, 1, 32, 32), 3.0)
  thf, ei, ef, eh, eeh, it = O.planner_forward(g['free_th0'], g['free_start'], g['free_goal'], sdf, p,
                                               int(g['free_max_iters']), float(g['free_tol_delta']))
  assert it == list(g['free_iters']) and max(it) < int(g['free_max_iters'])
  assert rel_err(thf, g['free_th_final']) < 1e-9
  for b in range(3):
    assert rel_err(eh[b], g['free_err_iter'][b][:it[b]]) < 1e-9


def test_g3_c3_velocity_limits(golden):
  g = golden('g3_c3_vel')
  B, n = g['th'].shape[0], g['th'].shape[1]
  p = O.OracleParams(dof=2, total_time_step=n - 1, use_vel_limits=True)
  assert p.M == int(g['M'])
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G))
  qc, ow, eps = p.static_covs(B)
  dth, err, _ = O.plan_layer_forward(g['th'], g['start'], g['goal'], sdf, qc, ow, eps, p)
  assert rel_err(dth, g['dth']) < TOL and rel_err(err, g['err']) < 1e-12


def test_g3_c4_nonholonomic_xyh(golden):
  g = golden('g3_c4_xyh')
  B, n = g['th'].shape[0], g['th'].shape[1]
  p = O.OracleParams(dof=3, total_time_step=n - 1, non_holonomic=True, epsilon_dist=0.2, reg=0.0)
  assert p.M == int(g['M'])
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G))
  qc, ow, eps = p.static_covs(B)
  dth, err, _ = O.plan_layer_forward(g['th'], g['start'], g['goal'], sdf, qc, ow, eps, p)
  assert rel_err(dth, g['dth']) < TOL and rel_err(err, g['err']) < 1e-12


def test_dense_torch_baseline(golden):
  """The PyTorch-CPU dense restatement that bench.py times as cpu_baseline reproduces the reference's outputs."""
  import torch
  from oracle import dense_torch as DT
  g = golden('g3_c2mini')
  B, n = 8, 64
  p = P2d(n)
  G = int(g['G'])
  sdf = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G))))
  T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
  dth, err, err_ext = DT.plan_layer_forward(T(g['cov_th']), T(g['start']), T(g['goal']), sdf, T(g['cov_qc']), T(g['cov_ow']),
                                            T(g['cov_eps']), DT.params_from_oracle(p))
  assert rel_err(dth.numpy(), g['cov_dth']) < TOL
  assert rel_err(err.numpy(), g['cov_err']) < 1e-12 and rel_err(err_ext.numpy(), g['cov_errext']) < 1e-12


def test_blocktri_c_oracle_matches_reference_and_numpy_oracle(golden):
  """oracle/gn_blocktri.c (block-tridiagonal fp64 Thomas) vs the reference's fixtures and the dense numpy oracle."""
  from oracle import blocktri as BT
  g = golden('g3_c2mini')
  B, n = 8, 64
  p = P2d(n)
  G = int(g['G'])
  sdf = O.circles_sdf(G, g['circles'])[None, None]
  for k in (0, 5, 9):
    dth, err, eex, info = BT.gn_step(p, g['th_hist'][k], g['start'], g['goal'], sdf, nthreads=2)
    assert not info.any() and rel_err(dth, g['dth_hist'][k]) < TOL
    assert rel_err(err, g['err_hist'][k].reshape(-1)) < 1e-12 and rel_err(eex, g['errext_hist'][k].reshape(-1)) < 1e-12
  dth, err, eex, _ = BT.gn_step(p, g['cov_th'], g['start'], g['goal'], sdf, qc=g['cov_qc'], ow=g['cov_ow'].reshape(B, n),
                                eps=g['cov_eps'].reshape(B, n))
  assert rel_err(dth, g['cov_dth']) < TOL and rel_err(err, g['cov_err'].reshape(-1)) < 1e-12 and rel_err(eex, g['cov_errext'].reshape(-1)) < 1e-12
  Qf = O.calc_Q_inv_batch(g['cov_qc'], p.dt)
  dth2, _, _, _ = BT.gn_step(p, g['cov_th'], g['start'], g['goal'], sdf, qc=Qf, ow=g['cov_ow'].reshape(B, n), eps=g['cov_eps'].reshape(B, n), q_full=True)
  assert rel_err(dth2, dth) < 1e-12
  c1 = golden('g3_c1')
  p101 = O.OracleParams(dof=2, total_time_step=100)
  th0 = O.straight_line_trajb(c1['start'][:, :, :2], c1['goal'][:, :, :2], 10.0, 100, 2)
  dth, err, _, _ = BT.gn_step(p101, th0, c1['start'], c1['goal'], c1['sdf'][None, None])
  assert rel_err(dth, c1['n101_dth0']) < TOL and abs(err[0] - 330.436499542839) < 1e-9
  g = golden('g3_c3_vel'); n = g['th'].shape[1]
  dth, err, _, _ = BT.gn_step(O.OracleParams(dof=2, total_time_step=n - 1, use_vel_limits=True), g['th'], g['start'], g['goal'],
                              O.circles_sdf(int(g['G']), g['circles'])[None, None])
  assert rel_err(dth, g['dth']) < TOL and rel_err(err, g['err'].reshape(-1)) < 1e-12
  g = golden('g3_c4_xyh'); n = g['th'].shape[1]
  dth, err, _, _ = BT.gn_step(O.OracleParams(dof=3, total_time_step=n - 1, non_holonomic=True, epsilon_dist=0.2, reg=0.0), g['th'],
                              g['start'], g['goal'], O.circles_sdf(int(g['G']), g['circles'])[None, None])
  assert rel_err(dth, g['dth']) < TOL and rel_err(err, g['err'].reshape(-1)) < 1e-12
  # not SPD -> info
  _, _, _, info = BT.gn_step(P2d(16, reg=-1e7), golden('g2_system_n16')['th'], golden('g2_system_n16')['start'],
                             golden('g2_system_n16')['goal'], O.circles_sdf(64, O.C2_CIRCLES)[None, None])
  assert info.all()


def test_autograd_oracle_matches_reference_grads(golden):
  """oracle/autograd_torch.py (the independent gradient oracle of the GPU backward tests and of tests/stress_random_configs.py) against
  the reference's OWN torch autograd: fixture g5_grads (cotangent on dtheta, then on err_ext) and g7_errors part (a) (unweighted
  errors / error_ext_batch at a leaf trajectory)."""
  import torch
  from oracle import autograd_torch as AT
  g = golden('g5_grads')
  B, n = g['th'].shape[:2]
  p = O.OracleParams(dof=2, total_time_step=n - 1)
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)).copy()
  r = AT.step_gradients(p, g['th'], g['start'], g['goal'], sdf, g['gbar'], np.zeros(B), qc=g['qc'], ow=g['ow'], eps=g['eps'])
  assert rel_err(r['dtheta'], g['dth']) < 1e-11
  for k in ('th', 'sdf', 'start', 'goal', 'qc', 'ow', 'eps'):
    assert rel_err(r[k].reshape(g['g_' + k].shape), g['g_' + k]) < 1e-10, k
  r = AT.step_gradients(p, g['th'], g['start'], g['goal'], sdf, np.zeros_like(g['gbar']), g['gext'].reshape(B), qc=g['qc'], ow=g['ow'], eps=g['eps'])
  for k in ('th', 'sdf', 'start', 'goal', 'eps'):
    assert rel_err(r[k].reshape(g['ge_' + k].shape), g['ge_' + k]) < 1e-12, k
  assert np.all(r['qc'] == 0) and np.all(r['ow'] == 0) and bool(g['ge_none_qc']) and bool(g['ge_none_ow'])
  # unweighted errors
  g = golden('g7_errors')
  T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).requires_grad_(True)
  the, st, go, sd, ep = T(g['th_eval']), T(g['start']), T(g['goal']), T(sdf), T(g['eps'])
  sg, gp, ob = AT.unweighted_errors(the, st, go, sd, ep, p)
  assert sg.shape == g['a_sg'].shape and gp.shape == g['a_gp'].shape and ob.shape == g['a_obs'].shape
  loss = (torch.from_numpy(g['c_sg']) * sg).sum() + (torch.from_numpy(g['c_gp']) * gp).sum() + (torch.from_numpy(g['c_obs']) * ob).sum()
  gr = torch.autograd.grad(loss, [the, sd, st, go, ep])
  for got, key in zip(gr, ('th_eval', 'sdf', 'start', 'goal', 'eps')):
    assert rel_err(got.numpy(), g['a_unw_g_' + key]) < 1e-12, key


def test_extended_precision_oracle_build(golden):
  """oracle/gn_blocktri.c built with long-double assembly + solve (the arbiter of tests/stress_random_configs.py): same answers as the
  fp64 build and the reference fixture on a well-conditioned system, closer to the truth on an ill-conditioned one."""
  from oracle import blocktri as BT
  g = golden('g3_c2mini')
  p = O.OracleParams(dof=2, total_time_step=63)
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  th = g['th_hist'][0]
  a = BT.gn_step(p, th, g['start'], g['goal'], sdf, nthreads=2)
  b = BT.gn_step(p, th, g['start'], g['goal'], sdf, nthreads=2, extended=True)
  assert rel_err(b[0], g['dth_hist'][0]) < 1e-11 and rel_err(a[0], b[0]) < 1e-11 and rel_err(a[1], b[1]) < 1e-14 and not b[3].any()
  # weakly regularised (cond ~ 1e7): the fp64 and extended builds differ by cond * 2^-53, and the dense numpy oracle sides with neither exactly
  p2 = O.OracleParams(dof=2, total_time_step=63, reg=1e-5)
  a = BT.gn_step(p2, th, g['start'], g['goal'], sdf)
  b = BT.gn_step(p2, th, g['start'], g['goal'], sdf, extended=True)
  qc, ow, eps = p2.static_covs(8)
  d_dense = O.plan_layer_forward(th, g['start'], g['goal'], np.broadcast_to(sdf, (8, 1) + sdf.shape[2:]), qc, ow, eps, p2)[0]
  assert rel_err(a[0], b[0]) < 1e-7 and rel_err(d_dense, b[0]) < 1e-7
