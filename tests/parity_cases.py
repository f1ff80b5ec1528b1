"""Parity cases shared by the CPU wavefront-emulator tests (tests/test_lane_emulator.py, -m "not gpu")
and the GPU tests proper (tests/test_hip_parity.py, -m gpu).  Each case feeds the same seeded inputs to a
harness.Backend and to the numpy oracle / golden fixtures and compares.

Tolerances (max-norm relative error over the whole tensor, conftest.rel_err, AND per trajectory, conftest.rel_err_per_traj):
  fp64 I/O : 1e-9   (fp64 block-PCR vs fp64 Cholesky + explicit inverses; cond(Lambda) ~ 1e4..2e5)
  fp32 I/O : 1e-5   (BASELINE.json north_star: "within 1e-5 relative fp32"); inputs are rounded to fp32
                    first and the oracle is fed exactly those rounded values.
"""
import numpy as np
from conftest import rel_err, rel_err_per_traj
from oracle import gpmp2_oracle as O

TOL = {'f64': 1e-9, 'f32': 1e-5}
TOL_ERR = {'f64': 1e-11, 'f32': 2e-6}


def rnd(a, io):
  """Round inputs to the I/O dtype (what the kernel will actually read), back to fp64 for the oracle."""
  return np.asarray(a, dtype=np.float32).astype(np.float64) if io == 'f32' else np.asarray(a, dtype=np.float64)


def P2d(n, **kw):
  return O.OracleParams(dof=2, total_time_step=n - 1, **kw)


def check_step(be, p, th, start, goal, sdf, io, qc=None, ow=None, eps=None, q_full=False, ref=None, tag='', oracle_memo=None):
  """oracle_memo: a dict shared by calls on IDENTICAL inputs (e.g. two kernel variants of one configuration): the dense oracle, seconds per call at
  n = 256, then runs once."""
  th, start, goal, sdf = rnd(th, io), rnd(start, io), rnd(goal, io), rnd(sdf, io)
  qc_, ow_, eps_ = [None if c is None else rnd(c, io) for c in (qc, ow, eps)]
  dth, err, eex, info = be.step(p, th, start, goal, sdf, qc=qc_, ow=None if ow_ is None else ow_.reshape(th.shape[0], -1),
                                eps=None if eps_ is None else eps_.reshape(th.shape[0], -1), q_full=q_full, io=io)
  B = th.shape[0]
  sdf_full = np.broadcast_to(sdf, (B,) + sdf.shape[1:])
  sq, so, se = p.static_covs(B)
  o_qc = sq if qc_ is None else qc_
  if qc_ is not None and qc_.ndim == 2: o_qc = qc_[:, :, None, None] * p.Q_c_inv      # DGP_QC_SCALAR: one scalar per GP factor, Q_c^-1 = s_k Q_c_inv
  o_ow = so if ow_ is None else ow_.reshape(so.shape)
  o_eps = se if eps_ is None else eps_.reshape(se.shape)
  if oracle_memo is not None and 'r' in oracle_memo:
    r_dth, r_err, r_eex = oracle_memo['r']
  else:
    r_dth, r_err, r_eex = O.plan_layer_forward(th, start, goal, sdf_full, o_qc, o_ow, o_eps, p, q_full=q_full)
    if oracle_memo is not None: oracle_memo['r'] = (r_dth, r_err, r_eex)
  assert np.all(info == 0), tag
  assert rel_err(dth, r_dth) < TOL[io], (tag, rel_err(dth, r_dth))
  assert rel_err_per_traj(dth, r_dth) < TOL[io], (tag, 'per trajectory', rel_err_per_traj(dth, r_dth))
  assert rel_err(err, r_err.reshape(-1)) < TOL_ERR[io], (tag, rel_err(err, r_err.reshape(-1)))
  assert rel_err(eex, r_eex.reshape(-1)) < TOL_ERR[io], (tag, rel_err(eex, r_eex.reshape(-1)))
  if ref is not None and io == 'f64':      # golden straight from the reference
    g_dth, g_err, g_eex = ref
    assert rel_err(dth, g_dth) < TOL[io], (tag, 'golden', rel_err(dth, g_dth))
    assert rel_err_per_traj(dth, g_dth) < TOL[io], (tag, 'golden, per trajectory', rel_err_per_traj(dth, g_dth))
    if g_err is not None: assert rel_err(err, np.reshape(g_err, -1)) < TOL_ERR[io], (tag, 'golden err')
    if g_eex is not None: assert rel_err(eex, np.reshape(g_eex, -1)) < TOL_ERR[io], (tag, 'golden err_ext')
  return dth, err, eex


# ---------------------------------------------------------------------------------------------------
def case_c2mini_static(be, golden, io, steps=(0, 4, 9), nb=8):
  """C2-shaped mini batch (n=64 -> one trajectory per wavefront), shared 256x256 SDF, static covariances,
  teacher-forced against the reference's own trajectory history."""
  g = golden('g3_c2mini')
  p = P2d(64)
  G = int(g['G'])
  sdf = O.circles_sdf(G, g['circles'])[None, None]
  for k in steps:
    check_step(be, p, g['th_hist'][k][:nb], g['start'][:nb], g['goal'][:nb], sdf, io,
               ref=(g['dth_hist'][k][:nb], g['err_hist'][k][:nb], g['errext_hist'][k][:nb]), tag='c2mini step %d' % k)


def case_c2mini_covs(be, golden, io, nb=8):
  """per-state SPD Q_c^-1, obstacle weights and epsilons (the learned-mode tensor shapes), and the same
  system passed as full Q^-1 ('q_full')."""
  g = golden('g3_c2mini')
  p = P2d(64)
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  ref = (g['cov_dth'][:nb], g['cov_err'][:nb], g['cov_errext'][:nb])
  d1, e1, x1 = check_step(be, p, g['cov_th'][:nb], g['start'][:nb], g['goal'][:nb], sdf, io, qc=g['cov_qc'][:nb],
                          ow=g['cov_ow'][:nb], eps=g['cov_eps'][:nb], ref=ref, tag='covs per-state')
  if io == 'f64':
    Qf = O.calc_Q_inv_batch(g['cov_qc'][:nb], p.dt)
    d2, e2, x2 = check_step(be, p, g['cov_th'][:nb], g['start'][:nb], g['goal'][:nb], sdf, io, qc=Qf, ow=g['cov_ow'][:nb],
                            eps=g['cov_eps'][:nb], q_full=True, ref=ref, tag='covs q_full')
    assert rel_err(d2, d1) < 1e-10, rel_err(d2, d1)      # two kernel variants (Kronecker C_k vs full Q_k^-1 per row): different operation order


def case_c2mini_per_sample_sdf(be, golden, io, nb=8):
  g = golden('g3_c2mini')
  p = P2d(64)
  Gp = int(g['ps_G'])
  sdf_ps = np.stack([O.circles_sdf(Gp, g['ps_circles'][b]) for b in range(nb)], 0)[:, None]
  check_step(be, p, g['th_hist'][0][:nb], g['start'][:nb], g['goal'][:nb], sdf_ps, io,
             ref=(g['ps_dth'][:nb], g['ps_err'][:nb], g['ps_errext'][:nb]), tag='per-sample sdf')


def case_c1(be, golden, io, steps=(0, 3, 9)):
  """BASELINE config 1 plumbing: real map 5.png (202x202 SDF incl. 1px pad), n=32 (two trajectories per wave)."""
  g = golden('g3_c1')
  p = P2d(32)
  sdf = g['sdf'][None, None]
  for k in steps:
    check_step(be, p, g['th_hist'][k], g['start'], g['goal'], sdf, io,
               ref=(g['dth_hist'][k], g['err_hist'][k], g['errext_hist'][k]), tag='c1 step %d' % k)
  if io == 'f64':
    th0 = O.straight_line_trajb(g['start'][:, :, :2], g['goal'][:, :, :2], 10.0, 32, 2)     # n = 33 -> 64 lanes, 31 idle
    check_step(be, O.OracleParams(dof=2, total_time_step=32), th0, g['start'], g['goal'], sdf, io,
               ref=(g['n33_dth0'], g['n33_err0'], g['n33_errext0']), tag='c1 n=33')


def case_small_ragged(be, golden, io):
  """n = 4 and 16 (16 lanes per trajectory, 4 trajectories per wave) with B = 3 (ragged last wave),
  random per-state covariances; fixtures g2_system_*."""
  for n in (4, 16):
    g = golden('g2_system_n%d' % n)
    p = P2d(n)
    sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
    check_step(be, p, g['th'], g['start'], g['goal'], sdf, io, qc=g['qc'], ow=g['ow'], eps=g['eps'], tag='ragged n=%d' % n)
  # B = 5 with n = 20 (32 lanes per trajectory, odd batch)
  rs = np.random.RandomState(5)
  n, B = 20, 5
  p = P2d(n)
  start = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  goal = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  th = O.straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + rs.randn(B, n, 4) * 0.2
  sdf = O.circles_sdf(80, O.C2_CIRCLES)[None, None]
  check_step(be, p, th, start, goal, sdf, io, tag='ragged n=20 B=5')


def case_edges(be, golden, io):
  """Points outside the grid / in the last row-col cell (dist = 0, J = 0 exactly => cost eps+r with
  zero Jacobian), a non-square SDF, and a hinge tie dist == eps + r (Q5)."""
  g = golden('g1_factors_2d')
  n, B = 16, 4
  p = P2d(n)
  rs = np.random.RandomState(3)
  start = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  goal = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  check_step(be, p, g['th_o'], start, goal, g['sdf'], io, eps=g['eps'], tag='edges non-square')
  check_step(be, p, g['th'], start, goal, g['sdf_tie'], io, tag='hinge tie')


def case_c3_vel(be, golden, io):
  g = golden('g3_c3_vel')
  B, n = g['th'].shape[:2]
  p = O.OracleParams(dof=2, total_time_step=n - 1, use_vel_limits=True)
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  check_step(be, p, g['th'], g['start'], g['goal'], sdf, io, ref=(g['dth'], g['err'], None), tag='c3 vel limits')


def case_c4_xyh(be, golden, io):
  g = golden('g3_c4_xyh')
  B, n = g['th'].shape[:2]
  p = O.OracleParams(dof=3, total_time_step=n - 1, non_holonomic=True, epsilon_dist=0.2, reg=0.0)
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  check_step(be, p, g['th'], g['start'], g['goal'], sdf, io, ref=(g['dth'], g['err'], None), tag='c4 xyh')


def case_eval_errors(be, golden, io):
  g = golden('g3_c1')
  p = P2d(32)
  th = rnd(g['th_hist'][3], io); start = rnd(g['start'], io); goal = rnd(g['goal'], io); sdf = rnd(g['sdf'][None, None], io)
  err, eex, usg, ugp, uobs = be.eval_errors(p, th, start, goal, sdf, io=io)
  qc, ow, eps = p.static_covs(1)
  Q = O.calc_Q_inv_batch(qc, p.dt)
  r_err = O.error_batch(th, start, goal, sdf, Q, ow, eps, p)
  r_sg, r_gp, r_obs = O.unweighted_errors_batch(th, start, goal, sdf, eps, p)
  t = TOL_ERR[io]
  assert rel_err(err, r_err.reshape(-1)) < t and rel_err(eex, r_err.reshape(-1)) < t
  assert rel_err(usg, r_sg.reshape(-1)) < max(t, 1e-7 if io == 'f32' else 0) or abs(float(usg[0]) - float(r_sg.item())) < 1e-12
  assert rel_err(ugp, r_gp.reshape(-1)) < t and rel_err(uobs, r_obs.reshape(-1)) < t
  if io == 'f64':
    assert rel_err(err, g['err_hist'][3].reshape(-1)) < t
    assert rel_err(ugp, g['unw_gp'].reshape(-1)) < t and rel_err(uobs, g['unw_obs'].reshape(-1)) < t
  # PlanLayer.gp_error(thb) / start_goal_error(thb) take no grid (plan_layer.py:374-377, :384-388): sdf == NULL, same numbers
  _, _, usg2, ugp2, _ = be.eval_errors(p, th, start, goal, None, io=io)
  assert np.array_equal(usg2, usg) and np.array_equal(ugp2, ugp)
  from dgpmp2_amd import _capi
  import pytest
  solver = _capi.Solver(__import__('harness').config_from_oracle(p, io), api=be.api)
  with pytest.raises(_capi.DgpError) as e:          # ... but an output that reads the grid cannot be requested without one
    solver.eval_errors(1, 0x1000, 0x1000, 0x1000, solver.sdf_arg(None, 2, 2, 0), None, err=0x1000)
  assert e.value.code == _capi.DGP_EINVAL


def case_solve(be, golden, io):
  """Fused GN loop vs DiffGPMP2Planner.forward: C1 to max_iters, and an obstacle-free batch that exits early
  by tol_delta at different iteration counts per trajectory."""
  g = golden('g4_forward'); c1 = golden('g3_c1')
  p = P2d(32)
  th0 = O.straight_line_trajb(c1['start'][:, :, :2], c1['goal'][:, :, :2], 10.0, 31, 2)
  mi = int(g['c1_max_iters'])
  th0r, st, go, sdf = rnd(th0, io), rnd(c1['start'], io), rnd(c1['goal'], io), rnd(c1['sdf'][None, None], io)
  tho, its, eh, eeh, ef, info = be.solve(p, th0r, st, go, sdf, mi, float(g['c1_tol_delta']), io=io)
  assert list(its) == list(g['c1_iters']) and np.all(info == 0)
  if io == 'f64':
    assert rel_err(tho, g['c1_th_final']) < 1e-7
    assert rel_err(eh[0], g['c1_err_iter'][0]) < 1e-8 and rel_err(eeh[0], g['c1_errext_iter'][0]) < 1e-8
    assert rel_err(ef, g['c1_err_final']) < 1e-8
  else:     # 12 un-forced iterations in fp32 I/O: compare with the oracle run on the rounded inputs, loosely
    r_th, _, r_ef, r_eh, _, r_it = O.planner_forward(th0r, st, go, sdf, p, mi, float(g['c1_tol_delta']))
    assert rel_err(tho, r_th) < 1e-4 and rel_err(eh[0], r_eh[0]) < 1e-4
  # early exit
  p = P2d(16)
  sdf = np.full((1, 1, 32, 32), 3.0)
  mi = int(g['free_max_iters'])
  tho, its, eh, eeh, ef, info = be.solve(p, rnd(g['free_th0'], io), rnd(g['free_start'], io), rnd(g['free_goal'], io), sdf, mi,
                                         float(g['free_tol_delta']), io=io)
  if io == 'f64':
    assert list(its) == list(g['free_iters'])
    assert rel_err(tho, g['free_th_final']) < 1e-8
    for b in range(3):
      k = int(its[b])
      assert rel_err(eh[b, :k], g['free_err_iter'][b][:k]) < 1e-8
      assert np.all(np.isnan(eh[b, k:]))                # entries past iters[b] untouched
    assert rel_err(ef, g['free_err_final']) < 1e-8
  else:
    assert np.all(np.abs(its - np.asarray(g['free_iters'])) <= 1)


def case_not_spd(be, golden, io):
  """A strongly negative `reg` makes Lambda indefinite: the reference raises from torch.cholesky; the C-ABI
  reports it per trajectory through info."""
  n, B = 16, 2
  p = P2d(n, reg=-1.0e7)
  rs = np.random.RandomState(1)
  start = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  goal = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  th = O.straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  sdf = np.full((1, 1, 16, 16), 3.0)
  _, _, _, info = be.step(p, th, start, goal, sdf, io=io)
  assert np.all(info == 1)


def case_backward_golden(be, golden, io):
  """Backward of one GN step vs the reference's torch autograd (fixture g5_grads: per-state covariances, per-sample
  SDF copies): cotangent on dtheta, then cotangent on err_ext (whose grads w.r.t. qc / obs_w are None in the reference)."""
  g = golden('g5_grads')
  B, n = g['th'].shape[:2]
  p = P2d(n)
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)).copy()
  tol = 1e-8 if io == 'f64' else 2e-4
  th, st, go, sdf = rnd(g['th'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io)
  qc, ow, eps = rnd(g['qc'], io), rnd(g['ow'].reshape(B, n), io), rnd(g['eps'].reshape(B, n), io)
  dth, _, _, _ = be.step(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)
  r = be.backward(p, th, st, go, sdf, dth, rnd(g['gbar'], io), None, qc=qc, ow=ow, eps=eps, io=io)
  for k in ('th', 'sdf', 'start', 'goal', 'qc', 'ow', 'eps'):
    assert rel_err(r[k].reshape(g['g_' + k].shape), g['g_' + k]) < tol, ('g_' + k, rel_err(r[k].reshape(g['g_' + k].shape), g['g_' + k]))
  r = be.backward(p, th, st, go, sdf, None, None, rnd(g['gext'].reshape(B), io), qc=qc, ow=ow, eps=eps, io=io)
  for k in ('th', 'sdf', 'start', 'goal', 'eps'):
    assert rel_err(r[k].reshape(g['ge_' + k].shape), g['ge_' + k]) < (1e-10 if io == 'f64' else 1e-5), 'ge_' + k
  assert bool(g['ge_none_qc']) and bool(g['ge_none_ow']) and np.all(r['qc'] == 0) and np.all(r['ow'] == 0)
  assert not bool(g['err_requires_grad'])


def _fd_loss(p, th, st, go, sdf, gbar, gext, qc=None, ow=None, eps=None, q_full=False):
  B = th.shape[0]
  sq, so, se = p.static_covs(B)
  dth, _, eex = O.plan_layer_forward(th, st, go, np.broadcast_to(sdf, (B,) + sdf.shape[1:]), sq if qc is None else qc,
                                     so if ow is None else ow.reshape(so.shape), se if eps is None else eps.reshape(se.shape), p, q_full=q_full)
  return float(np.sum(gbar * dth) + np.sum(gext * eex.reshape(-1)))


def case_backward_fd(be, golden, io):
  """Backward vs central finite differences of the ORACLE's loss sum(gbar*dtheta) + sum(gext*err_ext), for the
  configurations the reference cannot differentiate in batch: velocity limits (C3), non-holonomic xyh (C4, d=6),
  static covariances with a shared SDF, and full Q^-1 input (q_full)."""
  if io != 'f64': return
  rs = np.random.RandomState(11)

  def check(p, th, st, go, sdf, qc=None, q_full=False, nprobe=14):
    B, n, d = th.shape
    gbar = rs.randn(B, n, d); gext = rs.randn(B)
    dth, _, _, _ = be.step(p, th, st, go, sdf, qc=qc, q_full=q_full, io='f64')
    r = be.backward(p, th, st, go, sdf, dth, gbar, gext, qc=qc, q_full=q_full, io='f64')
    h = 1e-6
    for name, arr, grad in (('th', th, r['th']), ('start', st, r['start']), ('goal', go, r['goal'])) + \
                           ((('qc', qc, r['qc']),) if qc is not None else ()):
      flat = arr.reshape(-1)
      idx = rs.choice(flat.size, min(nprobe, flat.size), replace=False)
      for i in idx:
        if name == 'qc' and q_full:
          # symmetric perturbation (the kernel reads Q^-1 as a symmetric matrix): move (a,c) and (c,a) together
          a4 = np.unravel_index(i, arr.shape); a4t = a4[:2] + (a4[3], a4[2])
          ap, am = arr.copy(), arr.copy()
          ap[a4] += h; am[a4] -= h
          if a4 != a4t: ap[a4t] += h; am[a4t] -= h
          gsum = grad[a4] + (grad[a4t] if a4 != a4t else 0.0)
        else:
          ap, am = arr.copy().reshape(-1), arr.copy().reshape(-1)
          ap[i] += h; am[i] -= h
          ap, am = ap.reshape(arr.shape), am.reshape(arr.shape)
          gsum = grad.reshape(-1)[i]
        args = dict(th=th, st=st, go=go, qc=qc)
        kp = dict(args); km = dict(args)
        key = {'th': 'th', 'start': 'st', 'goal': 'go', 'qc': 'qc'}[name]
        kp[key] = ap; km[key] = am
        fd = (_fd_loss(p, kp['th'], kp['st'], kp['go'], sdf, gbar, gext, qc=kp['qc'], q_full=q_full) -
              _fd_loss(p, km['th'], km['st'], km['go'], sdf, gbar, gext, qc=km['qc'], q_full=q_full)) / (2 * h)
        scale = max(1.0, float(np.abs(grad).max()))
        assert abs(fd - gsum) < 2e-5 * scale, (name, int(i), fd, float(gsum))

  g = golden('g3_c3_vel')
  n = g['th'].shape[1]
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  check(O.OracleParams(dof=2, total_time_step=n - 1, use_vel_limits=True), g['th'][:2], g['start'][:2], g['goal'][:2], sdf)
  g = golden('g3_c4_xyh')
  n = g['th'].shape[1]
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  check(O.OracleParams(dof=3, total_time_step=n - 1, non_holonomic=True, epsilon_dist=0.2, reg=0.0), g['th'][:2], g['start'][:2],
        g['goal'][:2], sdf)
  g = golden('g5_grads')
  n = g['th'].shape[1]
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  p = P2d(n)
  check(p, g['th'][:2], g['start'][:2], g['goal'][:2], sdf)                                    # static covariances, shared SDF
  Qf = O.calc_Q_inv_batch(g['qc'][:2], p.dt)
  check(p, g['th'][:2], g['start'][:2], g['goal'][:2], sdf, qc=Qf, q_full=True, nprobe=10)     # q_full


def case_unaligned_buffers(be, golden, io):
  """Buffers that are only element-aligned (e.g. a slice of a larger tensor): the kernels fall back from 16-byte vector row
  accesses to scalar ones; forward and backward results must not change."""
  g = golden('g5_grads')
  B, n = g['th'].shape[:2]
  p = P2d(n)
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  th, st, go, sdf = rnd(g['th'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io)
  d0, e0, x0, _ = be.step(p, th, st, go, sdf, io=io)
  r0 = be.backward(p, th, st, go, sdf, d0, rnd(g['gbar'], io), None, io=io)
  be.misalign = True
  try:
    d1, e1, x1, _ = be.step(p, th, st, go, sdf, io=io)
    r1 = be.backward(p, th, st, go, sdf, d0, rnd(g['gbar'], io), None, io=io)
  finally:
    be.misalign = False
  assert np.array_equal(d0, d1) and np.array_equal(e0, e1) and np.array_equal(x0, x1)
  assert np.array_equal(r0['th'], r1['th']) and np.array_equal(r0['start'], r1['start'])
  assert rel_err(r1['sdf'], r0['sdf']) < (1e-12 if io == 'f64' else 1e-5)          # atomics: summation order may differ
  # per-state covariance tensors: their blocks are fetched as 16-byte vectors when aligned, element by element otherwise
  qc, ow, eps = rnd(g['qc'], io), rnd(g['ow'].reshape(B, n), io), rnd(g['eps'].reshape(B, n), io)
  d2, e2, x2, _ = be.step(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)
  be.misalign = True
  try:
    d3, e3, x3, _ = be.step(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)
  finally:
    be.misalign = False
  assert np.array_equal(d2, d3) and np.array_equal(e2, e3) and np.array_equal(x2, x3)


ALL_CASES = [case_c2mini_static, case_c2mini_covs, case_c2mini_per_sample_sdf, case_c1, case_small_ragged, case_edges,
             case_c3_vel, case_c4_xyh, case_eval_errors, case_solve, case_not_spd, case_backward_golden, case_backward_fd, case_unaligned_buffers]


def case_shared_sdf_gradient_partial_copies(be, golden, io):
  """Shared SDF: the gradient accumulated into 8 per-XCD partial grids (XCD-local L2 atomics) and summed equals the
  single-grid device-scope accumulation."""
  g = golden('g5_grads')
  B, n = g['th'].shape[:2]
  p = P2d(n)
  rs = np.random.RandomState(4)
  reps = 40 if be.kind == 'hip' else 1          # on the GPU use enough trajectories to populate every XCD
  th = np.concatenate([g['th'] + rs.randn(B, n, 4) * 0.05 for _ in range(reps)], 0)
  st = np.concatenate([g['start']] * reps, 0); go = np.concatenate([g['goal']] * reps, 0)
  gbar = rs.randn(*th.shape)
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  th, st, go, sdf, gbar = rnd(th, io), rnd(st, io), rnd(go, io), rnd(sdf, io), rnd(gbar, io)
  dth, _, _, _ = be.step(p, th, st, go, sdf, io=io)
  r1 = be.backward(p, th, st, go, sdf, dth, gbar, None, io=io)
  r8 = be.backward(p, th, st, go, sdf, dth, gbar, None, io=io, sdf_copies=8)
  assert r8['sdf'].shape[0] == 8
  if be.kind == 'hip' and reps > 8: assert (np.abs(r8['sdf']).reshape(8, -1).max(1) > 0).sum() >= 2      # really spread over XCDs
  assert rel_err(r8['sdf'].sum(0, keepdims=True), r1['sdf']) < (1e-11 if io == 'f64' else 2e-5)
  assert np.array_equal(r8['th'], r1['th'])
  # fewer copies than XCDs: two XCDs share a copy, so the kernel must stay on device-scope atomics (no lost updates)
  r3 = be.backward(p, th, st, go, sdf, dth, gbar, None, io=io, sdf_copies=3)
  assert r3['sdf'].shape[0] == 3
  assert rel_err(r3['sdf'].sum(0, keepdims=True), r1['sdf']) < (1e-11 if io == 'f64' else 2e-5)
  # more copies than XCDs (what PlanLayer uses: 16): every XCD spreads its wavefronts over copies xcc, xcc + 8
  r16 = be.backward(p, th, st, go, sdf, dth, gbar, None, io=io, sdf_copies=16)
  assert rel_err(r16['sdf'].sum(0, keepdims=True), r1['sdf']) < (1e-11 if io == 'f64' else 2e-5)
  if be.kind == 'hip' and reps > 8: assert (np.abs(r16['sdf']).reshape(16, -1).max(1) > 0).sum() >= 9


def case_tiny_and_odd_sizes(be, golden, io):
  """n = 2 (the minimum: start and goal only), n = 3, n = 5 with B = 1, and a batch that is not a multiple of the number of
  trajectories per wavefront; NaN in one trajectory must not leak into its wave neighbours."""
  rs = np.random.RandomState(21)
  sdf = O.circles_sdf(48, O.C2_CIRCLES)[None, None]
  for n, B in ((2, 1), (3, 2), (5, 7), (17, 3)):
    p = P2d(n)
    start = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
    goal = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
    th = O.straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2) + rs.randn(B, n, 4) * 0.1
    check_step(be, p, th, start, goal, sdf, io, tag='tiny n=%d B=%d' % (n, B))
  n, B = 5, 7
  p = P2d(n)
  start = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  goal = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  th = O.straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  th_nan = th.copy(); th_nan[2, 1, 0] = np.nan
  d0, _, _, _ = be.step(p, rnd(th, io), rnd(start, io), rnd(goal, io), rnd(sdf, io), io=io)
  d1, _, _, _ = be.step(p, rnd(th_nan, io), rnd(start, io), rnd(goal, io), rnd(sdf, io), io=io)
  keep = [b for b in range(B) if b != 2]
  assert np.array_equal(d0[keep], d1[keep]) and np.all(np.isnan(d1[2]))


def case_static_qc_variants(be, golden, io):
  """Static covariances with (a) a diagonal, non-identity Q_c_inv -- the static kernels with unequal per-dof weights --
  and (b) a full symmetric Q_c_inv, which the static kernels do not cover (they skip the structural zeros of a diagonal
  Q_c_inv) and which must therefore run the generic kernels; dof 2 and 3, ragged n."""
  rs = np.random.RandomState(21)
  for dof, n, B, Qc in ((2, 24, 5, np.diag([0.7, 2.5])), (2, 24, 5, np.array([[1.5, 0.4], [0.4, 0.8]])),
                        (3, 13, 3, np.diag([0.5, 1.0, 3.0])), (3, 13, 3, np.array([[1.2, 0.2, -0.1], [0.2, 0.9, 0.3], [-0.1, 0.3, 2.0]]))):
    p = O.OracleParams(dof=dof, total_time_step=n - 1, Q_c_inv=Qc)
    d = 2 * dof
    start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
    start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2))
    th = np.zeros((B, n, d))
    t = np.linspace(0, 1, n)[None, :, None]
    th[:, :, :dof] = start[:, :, :dof] + t * (goal[:, :, :dof] - start[:, :, :dof])
    th[:, :, dof:] = (goal[:, :, :dof] - start[:, :, :dof]) / 10.0
    th = th + rs.randn(B, n, d) * 0.1
    sdf = O.circles_sdf(96, O.C2_CIRCLES)[None, None]
    check_step(be, p, th, start, goal, sdf, io, tag='static Qc dof=%d diag=%s' % (dof, bool(np.all(Qc == np.diag(np.diag(Qc))))))


def case_scalar_covariances(be, golden, io):
  """DGP_QC_SCALAR (the learned mode diag_identity, diff_gpmp2_planner.py:255-258: Q_c^-1 = q_k^2 I): one scalar per GP factor on the static block-elimination
  kernels with scaled lane masks (QK_SCALED) -- against the oracle on the dense tensors s_k Q_c_inv, and against the per-state (Kronecker) kernels on the same
  tensors; identity and diagonal Q_c_inv, dof 2 and 3 (+ non-holonomic, velocity limits), full, ragged and single-lane lengths, with and without per-state
  obstacle weights / epsilons."""
  rs = np.random.RandomState(33)
  confs = [(2, 64, 5, None, {}), (2, 33, 3, np.diag([0.7, 2.5]), {}), (2, 16, 9, None, dict(use_vel_limits=True, K_v=0.01, v_x=1.0, v_y=1.0)),
           (3, 64, 4, None, dict(non_holonomic=True, K_d=0.01)), (3, 21, 3, np.diag([0.5, 1.0, 3.0]), {}), (2, 5, 2, None, {}), (2, 2, 1, None, {})]
  for dof, n, B, Qc, kw in confs:
    p = O.OracleParams(dof=dof, total_time_step=n - 1, Q_c_inv=Qc, **kw)
    d = 2 * dof
    start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
    start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2))
    if dof == 3: goal[:, 0, 2] = rs.uniform(-1.5, 1.5, B)
    th = np.zeros((B, n, d))
    t = np.linspace(0, 1, n)[None, :, None]
    th[:, :, :dof] = start[:, :, :dof] + t * (goal[:, :, :dof] - start[:, :, :dof])
    th[:, :, dof:] = (goal[:, :, :dof] - start[:, :, :dof]) / 10.0
    th = th + rs.randn(B, n, d) * 0.1
    sdf = O.circles_sdf(96, O.C2_CIRCLES)[None, None]
    s_ = rs.uniform(0.3, 3.0, (B, n - 1)) ** 2
    for with_obs in (False, True):
      ow = rs.uniform(50, 2e4, (B, n, 1, 1)) if with_obs else None
      eps = rs.uniform(0.1, 0.6, (B, n, 1, 1)) if with_obs else None
      tag = 'scalar covs dof=%d n=%d %s%s' % (dof, n, 'diag' if Qc is not None else 'identity', ' + obstacle tensors' if with_obs else '')
      d1, e1, x1 = check_step(be, p, th, start, goal, sdf, io, qc=s_, ow=ow, eps=eps, tag=tag)
      dense = rnd(s_, io)[:, :, None, None] * p.Q_c_inv
      d2, e2, x2 = check_step(be, p, th, start, goal, sdf, io, qc=dense, ow=ow, eps=eps, tag=tag + ' (per-state tensors)')
      if io == 'f64':
        assert rel_err(d1, d2) < 1e-9 and rel_err(e1, e2) < 1e-12 and rel_err(x1, x2) < 1e-12, (tag, rel_err(d1, d2), rel_err(e1, e2), rel_err(x1, x2))
      # the backward: the scaled-mask kernels (scalars in) against the Kronecker kernels (blocks in) -- every gradient, incl. the dense (B,n-1,dof,dof) one of the blocks
      gb = rnd(rs.randn(B, n, d), io); ge = rnd(rs.randn(B), io)
      a_ow = None if ow is None else rnd(ow, io).reshape(B, -1); a_eps = None if eps is None else rnd(eps, io).reshape(B, -1)
      args = (p, rnd(th, io), rnd(start, io), rnd(goal, io), rnd(sdf, io), d2, gb, ge)
      r1 = be.backward(*args, qc=rnd(s_, io), ow=a_ow, eps=a_eps, io=io)
      r2 = be.backward(*args, qc=dense, ow=a_ow, eps=a_eps, io=io)
      for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'):
        if r2[key] is None or (key == 'sdf' and io == 'f32'): continue      # (fp32 atomics in memory: order-dependent noise)
        scale = max(np.abs(r2[key]).max(), np.abs(r2['th']).max() if key == 'sdf' else 0.0, 1e-300 if io == 'f64' else 1e-6 * np.abs(r2['th']).max())
        eb = np.abs(r1[key] - r2[key]).max() / scale
        assert eb < (1e-8 if io == 'f64' else 3e-3), (tag, 'backward', key, eb)


ALL_CASES.append(case_scalar_covariances)
ALL_CASES.append(case_static_qc_variants)
ALL_CASES.append(case_tiny_and_odd_sizes)
ALL_CASES.append(case_shared_sdf_gradient_partial_copies)


def case_solve_with_covariances(be, golden, io):
  """The fused GN loop (dgp_gn_solve) with per-state covariance tensors -- the Kronecker kernels (qc_inv (B,n-1,dof,dof)) and the
  general ones (q_full, (B,n-1,d,d)) -- and per-state obstacle weights / epsilons: equals the same number of chained dgp_gn_step
  calls (themselves checked against the oracle and the reference's fixtures), iteration by iteration; d = 4 and d = 6."""
  if io != 'f64': return
  rs = np.random.RandomState(17)
  for dof, n, B, G in ((2, 16, 5, 64), (2, 64, 3, 96), (3, 12, 3, 48)):
    d = 2 * dof
    kw = dict(non_holonomic=True, K_d=0.05, epsilon_dist=0.2) if dof == 3 else {}
    p = O.OracleParams(dof=dof, total_time_step=n - 1, **kw)
    start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
    start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2))
    th0 = O.straight_line_trajb(start[:, :, :dof], goal[:, :, :dof], 10.0, n - 1, dof) + rs.randn(B, n, d) * 0.05
    sdf = O.circles_sdf(G, O.C2_CIRCLES)[None, None]
    a = rs.randn(B, n - 1, dof, dof) * 0.3
    qc = a @ np.swapaxes(a, -1, -2) + 0.5 * np.eye(dof)
    ow = rs.uniform(0.25, 1.75, (B, n)) * 1e4; eps = rs.uniform(0.2, 0.5, (B, n))
    for q_full in (False, True):
      q = O.calc_Q_inv_batch(qc, p.dt) if q_full else qc
      iters = 4
      tho, its, eh, eeh, ef, info = be.solve(p, th0, start, goal, sdf, iters, 0.0, qc=q, ow=ow, eps=eps, q_full=q_full, io='f64')
      assert np.all(its == iters) and np.all(info == 0)
      cur = th0.copy()
      for k in range(iters):
        dth, err, eex, _ = be.step(p, cur, start, goal, sdf, qc=q, ow=ow, eps=eps, q_full=q_full, io='f64')
        assert rel_err(err, eh[:, k]) < 1e-9 and rel_err(eex, eeh[:, k]) < 1e-9, (dof, n, q_full, k)
        cur = cur + dth
      assert rel_err(cur, tho) < 1e-9, (dof, n, q_full, rel_err(cur, tho))
      e_fin = be.eval_errors(p, tho, start, goal, sdf, qc=q, ow=ow, eps=eps, q_full=q_full, io='f64')[0]
      assert rel_err(ef, e_fin) < 1e-10


ALL_CASES.append(case_solve_with_covariances)


def case_woodbury_kernels(be, golden, io, shapes=('16,4', '32,4', '64,4'), nb=3, ragged=True):
  """The QK_WB kernels (gn_woodbury.h: interior rows eliminated through the Woodbury identity on the constant GP block) against the
  oracle, for every shape they are built for (n == LPT * 4), both robots, static and per-state obstacle weights, the single step and the
  fused loop -- and against the block-elimination kernels on the same inputs (DGP_NO_WOODBURY=1): both within tolerance of the oracle,
  NOT bit-identical (which proves the Woodbury kernels are the ones that ran)."""
  import os
  rs = np.random.RandomState(5)
  saved = {k: os.environ.get(k) for k in ('DGP_FORCE_SHAPE', 'DGP_NO_WOODBURY')}
  try:
    # per shape: every row present; then (ragged) the goal row as an interior row of the last lane at positions 2, 1, 0, lanes of padding rows only
    cases = []
    for shape in shapes:
      lpt = int(shape.split(',')[0])
      ns = (lpt * 4,) if not ragged else ((lpt * 4, lpt * 4 - 1, lpt * 4 - 2, lpt * 4 - 3, lpt * 4 - 9, 5) if lpt == 16 else (lpt * 4, lpt * 4 - 6))
      cases += [(shape, n) for n in ns]
    for shape, n in cases:
      os.environ['DGP_FORCE_SHAPE'] = shape
      for dof, kw in ((2, {}), (2, dict(Q_c_inv=2.5 * np.eye(2), cost_sigma=0.05, K_s=0.1, reg=1e-2)),      # (reg = 1e-3 at n = 256: cond 4e7, every solver 2-3e-9 off the dense oracle)
                      (3, dict(non_holonomic=True, K_d=0.01, epsilon_dist=0.2, reg=0.0 if n == 64 else 0.05)),      # reg = 0: BASELINE configs[3] (n = 64); at n = 256 cond(Lambda) > 1e7
                      (3, dict(reg=0.1))):
        p = O.OracleParams(dof=dof, total_time_step=n - 1, **kw)
        sp = rs.uniform(-4, 4, (nb, 1, 2)); gp = rs.uniform(-4, 4, (nb, 1, 2))
        if dof == 3:
          sp = np.concatenate([sp, np.zeros((nb, 1, 1))], -1); gp = np.concatenate([gp, np.full((nb, 1, 1), np.pi / 2)], -1)
        start = np.concatenate([sp, np.zeros((nb, 1, dof))], -1); goal = np.concatenate([gp, np.zeros((nb, 1, dof))], -1)
        th = O.straight_line_trajb(start[:, :, :dof], goal[:, :, :dof], 10.0, n - 1, dof) + 0.03 * rs.randn(nb, n, 2 * dof)
        sdf = O.circles_sdf(96, O.C2_CIRCLES)[None, None]
        os.environ['DGP_NO_WOODBURY'] = '0'
        tag = 'woodbury %s dof %d %s' % (shape, dof, sorted(kw))
        memo = {}      # the two kernel variants see identical inputs: one run of the dense oracle
        d_wb, e_wb, x_wb = check_step(be, p, th, start, goal, sdf, io, tag=tag, oracle_memo=memo)
        os.environ['DGP_NO_WOODBURY'] = '1'
        d_be, e_be, x_be = check_step(be, p, th, start, goal, sdf, io, tag=tag + ' (block elimination)', oracle_memo=memo)
        os.environ['DGP_NO_WOODBURY'] = '0'
        assert rel_err(d_wb, d_be) < 2 * TOL[io], (tag, rel_err(d_wb, d_be))
        if io == 'f64': assert not np.array_equal(d_wb, d_be), tag + ': identical bits -- the Woodbury kernel did not run'
        # backward of the step: the adjoint solve through the Woodbury elimination == through the block elimination (which the
        # reference's autograd fixture and the finite-difference cases pin), every gradient tensor
        gbar = rnd(rs.randn(nb, n, 2 * dof), io); gext = rnd(rs.randn(nb), io)
        th_r, st_r, go_r, sdf_r = rnd(th, io), rnd(start, io), rnd(goal, io), rnd(sdf, io)
        g_wb = be.backward(p, th_r, st_r, go_r, sdf_r, rnd(d_wb, io), gbar, gext, io=io)
        os.environ['DGP_NO_WOODBURY'] = '1'
        g_be = be.backward(p, th_r, st_r, go_r, sdf_r, rnd(d_wb, io), gbar, gext, io=io)
        os.environ['DGP_NO_WOODBURY'] = '0'
        for key in ('th', 'start', 'goal', 'sdf'):
          eb = rel_err(g_wb[key], g_be[key])
          assert eb < (1e-6 if io == 'f64' else 3e-4), (tag, 'backward', key, eb)      # two eliminations of the adjoint system (cond up to 1e6 at n = 256): the bound tests/stress_random_configs.py uses for GPU vs emulator
        if io == 'f64': assert not np.array_equal(g_wb['th'], g_be['th']), tag + ': backward bits identical -- the Woodbury kernel did not run'
        if dof == 2 and not kw:      # per-state obstacle weights / epsilons with static GP covariances (sqrt of the weight on the device)
          ow = rs.uniform(0.5, 2.0, (nb, n, 1, 1)) / p.cost_sigma ** 2
          eps = rs.uniform(0.3, 0.5, (nb, n, 1, 1))
          check_step(be, p, th, start, goal, sdf, io, ow=ow, eps=eps, tag=tag + ' tensor weights')
      # fused loop: dgp_gn_solve == chained steps (the same kernels in MODE_SOLVE)
      p = P2d(n)
      sp = rs.uniform(-4, 4, (nb, 1, 2)); gp = rs.uniform(-4, 4, (nb, 1, 2))
      start = np.concatenate([sp, np.zeros((nb, 1, 2))], -1); goal = np.concatenate([gp, np.zeros((nb, 1, 2))], -1)
      th = O.straight_line_trajb(sp, gp, 10.0, n - 1, 2)
      sdf = O.circles_sdf(96, O.C2_CIRCLES)[None, None]
      th_r, start_r, goal_r, sdf_r = rnd(th, io), rnd(start, io), rnd(goal, io), rnd(sdf, io)
      tho, its, eh, eeh, ef, info = be.solve(p, th_r, start_r, goal_r, sdf_r, 4, 0.0, io=io)
      cur = th_r.copy()
      for k in range(4):
        d, e, x, inf = be.step(p, cur, start_r, goal_r, sdf_r, io=io)
        assert rel_err(eh[:, k], e) < (1e-9 if io == 'f64' else 2e-4), (shape, k, rel_err(eh[:, k], e))
        cur = rnd(cur + d, io) if io == 'f64' else cur + d
      if io == 'f64': assert rel_err(tho, cur) < 1e-9, (shape, rel_err(tho, cur))
      assert np.all(its == 4) and np.all(info == 0)
  finally:
    for k, v in saved.items():
      if v is None: os.environ.pop(k, None)
      else: os.environ[k] = v


ALL_CASES.append(case_woodbury_kernels)


def case_eval_errors_backward(be, golden, io):
  """dgp_eval_errors_backward vs the reference's torch autograd through unweighted_errors_batch / error_ext_batch (fixture g7_errors,
  part (a): the errors at a leaf trajectory; gradients w.r.t. the trajectory, the grid, the start / goal means and the CURRENT eps the
  last forward() left behind, plan_layer.py:88-94,329,374-388), per-sample grids; then central finite differences of the ORACLE for the
  configurations the reference cannot run in batch (velocity limits, xyh robot: err_ext carries those factors) and the grid-less call."""
  g = golden('g7_errors')
  B, n = g['th'].shape[:2]
  p = P2d(n)
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)).copy()
  th, st, go, sdf, eps = rnd(g['th_eval'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io), rnd(g['eps'].reshape(B, n), io)
  tol = 1e-10 if io == 'f64' else 2e-5
  err, eex, usg, ugp, uobs = be.eval_errors(p, th, st, go, sdf, eps=eps, io=io)
  for got, key in ((usg, 'a_sg'), (ugp, 'a_gp'), (uobs, 'a_obs'), (eex, 'a_ee')):
    assert rel_err(got, g[key].reshape(-1)) < (1e-11 if io == 'f64' else 2e-6), key
  r = be.eval_backward(p, th, st, go, sdf, None, rnd(g['c_sg'], io), rnd(g['c_gp'], io), rnd(g['c_obs'], io), eps=eps, io=io)
  for k, key in (('th', 'th_eval'), ('sdf', 'sdf'), ('start', 'start'), ('goal', 'goal'), ('eps', 'eps')):
    ref = g['a_unw_g_' + key]
    assert rel_err(r[k].reshape(ref.shape), ref) < tol, ('unweighted', k, rel_err(r[k].reshape(ref.shape), ref))
  r = be.eval_backward(p, th, st, go, sdf, rnd(g['c_ee'], io), None, None, None, eps=eps, io=io)
  for k, key in (('th', 'th_eval'), ('sdf', 'sdf'), ('start', 'start'), ('goal', 'goal'), ('eps', 'eps')):
    ref = g['a_ee_g_' + key]
    assert rel_err(r[k].reshape(ref.shape), ref) < tol, ('err_ext', k, rel_err(r[k].reshape(ref.shape), ref))
  assert bool(g['a_unw_none_qc']) and bool(g['a_unw_none_ow']) and bool(g['a_ee_none_qc']) and bool(g['a_ee_none_ow'])
  # no grid: start_goal_error / gp_error alone (PlanLayer.gp_error(thb) takes none) -- same th / start / goal gradients as with the
  # obstacle cotangent left out
  r0 = be.eval_backward(p, th, st, go, sdf, None, rnd(g['c_sg'], io), rnd(g['c_gp'], io), None, io=io)
  r1 = be.eval_backward(p, th, st, go, None, None, rnd(g['c_sg'], io), rnd(g['c_gp'], io), None, io=io)
  assert np.array_equal(r0['th'], r1['th']) and np.array_equal(r0['start'], r1['start']) and np.array_equal(r0['goal'], r1['goal'])
  assert np.all(r0['sdf'] == 0)
  if io != 'f64': return
  # shared grid with partial copies == per-sample grids summed
  sdf1 = sdf[:1]
  ra = be.eval_backward(p, th, st, go, sdf1, rnd(g['c_ee'], io), rnd(g['c_sg'], io), rnd(g['c_gp'], io), rnd(g['c_obs'], io), eps=eps, io=io)
  rb = be.eval_backward(p, th, st, go, sdf, rnd(g['c_ee'], io), rnd(g['c_sg'], io), rnd(g['c_gp'], io), rnd(g['c_obs'], io), eps=eps, io=io)
  rc = be.eval_backward(p, th, st, go, sdf1, rnd(g['c_ee'], io), rnd(g['c_sg'], io), rnd(g['c_gp'], io), rnd(g['c_obs'], io), eps=eps, io=io, sdf_copies=8)
  assert rel_err(ra['sdf'], rb['sdf'].sum(0, keepdims=True)) < 1e-12 and rel_err(rc['sdf'].sum(0, keepdims=True), ra['sdf']) < 1e-12
  assert np.array_equal(ra['th'], rb['th']) and np.array_equal(ra['eps'], rb['eps'])
  # finite differences of the oracle: C3 (velocity limits) and C4 (xyh) -- err_ext includes those factors (plan_layer.py:333-343)
  rs = np.random.RandomState(12)
  for name, P in (('g3_c3_vel', lambda n: O.OracleParams(dof=2, total_time_step=n - 1, use_vel_limits=True)),
                  ('g3_c4_xyh', lambda n: O.OracleParams(dof=3, total_time_step=n - 1, non_holonomic=True, epsilon_dist=0.2, reg=0.0))):
    gg = golden(name)
    n2 = gg['th'].shape[1]
    pp = P(n2)
    sdf2 = O.circles_sdf(int(gg['G']), gg['circles'])[None, None]
    th2, st2, go2 = gg['th'][:2], gg['start'][:2], gg['goal'][:2]
    B2, _, d2 = th2.shape
    ce, cs, cg, co = rs.randn(B2), rs.randn(B2), rs.randn(B2), rs.randn(B2)
    eps2 = rs.uniform(0.2, 0.5, (B2, n2))

    def loss(th_, st_, go_, eps_):
      qc_, ow_, _ = pp.static_covs(B2)
      sdfB = np.broadcast_to(sdf2, (B2,) + sdf2.shape[1:])
      ee = O.error_batch(th_, st_, go_, sdfB, O.calc_Q_inv_batch(qc_, pp.dt), ow_, eps_.reshape(B2, n2, 1, 1), pp)      # fixed weights = the static ones
      sg, gp_, ob = O.unweighted_errors_batch(th_, st_, go_, sdfB, eps_.reshape(B2, n2, 1, 1), pp)
      return float(np.sum(ce * ee.reshape(-1)) + np.sum(cs * sg.reshape(-1)) + np.sum(cg * gp_.reshape(-1)) + np.sum(co * ob.reshape(-1)))
    r = be.eval_backward(pp, th2, st2, go2, sdf2, ce, cs, cg, co, eps=eps2, io='f64')
    h = 1e-6
    for nm, arr, grad in (('th', th2, r['th']), ('start', st2, r['start']), ('goal', go2, r['goal']), ('eps', eps2, r['eps'])):
      flat = arr.reshape(-1)
      for i in rs.choice(flat.size, min(12, flat.size), replace=False):
        ap, am = arr.copy().reshape(-1), arr.copy().reshape(-1)
        ap[i] += h; am[i] -= h
        args = dict(th=th2, start=st2, goal=go2, eps=eps2)
        kp = dict(args); km = dict(args)
        kp[nm] = ap.reshape(arr.shape); km[nm] = am.reshape(arr.shape)
        fd = (loss(kp['th'], kp['start'], kp['goal'], kp['eps']) - loss(km['th'], km['start'], km['goal'], km['eps'])) / (2 * h)
        scale = max(1.0, float(np.abs(grad).max()))
        assert abs(fd - grad.reshape(-1)[i]) < 2e-5 * scale, (name, nm, int(i), fd, float(grad.reshape(-1)[i]))


ALL_CASES.append(case_eval_errors_backward)


def case_solve_backward(be, golden, io):
  """dgp_gn_solve_traced + dgp_gn_solve_backward vs the reference's torch autograd through DiffGPMP2Planner.forward, which keeps the graph
  across its Gauss-Newton iterations (fixture g8_forward_grads; diff_gpmp2_planner.py:92-174, consumer examples/diff_gpmp2_2d_example.py:77):
  four trajectories that stop after 2, 8, 9 (= max_iters) and 9 iterations, one of them on an obstacle-free grid; gradients w.r.t. the initial
  trajectory, the grids, the start and goal means.  The loop does not converge on three of them (the hinge switches states on and off), so
  nine chained solves amplify rounding: 6e-9 in fp64 against the reference's own fp64 run.  f32 I/O: the traced loop must equal the plain one
  bit for bit and the history must reproduce the chained steps; the gradients are then only checked against the f64 ones at 2e-3 (the
  iteration counts are the reference's: tol_delta sits far from every |dtheta|)."""
  g = golden('g8_forward_grads')
  B, n = g['th0'].shape[:2]
  G = int(g['G'])
  p = P2d(n)
  K, tol_delta = int(g['max_iters']), float(g['tol_delta'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)).copy()
  sdf[int(g['free_sample'])] = float(g['free_value'])
  th0, st, go, sdf, gbar = rnd(g['th0'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io), rnd(g['gbar'], io)
  tho, its, hist, info = be.solve_traced(p, th0, st, go, sdf, K, tol_delta, io=io)
  assert np.all(info == 0) and np.array_equal(its, g['iters'])
  ref = be.solve(p, th0, st, go, sdf, K, tol_delta, io=io)
  assert np.array_equal(tho, ref[0]) and np.array_equal(its, ref[1])                     # the history store does not touch the loop
  assert rel_err(tho, g['th_final']) < (1e-8 if io == 'f64' else 2e-3)
  for b in range(B):                                  # rows the loop ran are written, the others are not
    k = int(its[b])
    assert np.all(np.isfinite(hist[:k, b])) and np.all(np.isnan(hist[k:, b]))
    assert np.array_equal(hist[0, b], th0[b])
  r = be.solve_backward(p, st, go, sdf, K, hist, tho, its, gbar, io=io)
  if io == 'f64':
    for k, key in (('th', 'g_th0'), ('start', 'g_start'), ('goal', 'g_goal'), ('sdf', 'g_sdf')):
      assert rel_err(r[k], g[key]) < 5e-8, (k, rel_err(r[k], g[key]))
    # the same gradient as chained single-step backward calls (dgp_gn_step_backward, K launches) through the history
    gcur = gbar.copy()
    acc = dict(start=np.zeros_like(st), goal=np.zeros_like(go), sdf=np.zeros_like(sdf))
    for k in range(K - 1, -1, -1):
      on = its > k
      thk = np.where(on[:, None, None], np.nan_to_num(hist[k]), tho)
      nxt = np.where((its > k + 1)[:, None, None], np.nan_to_num(hist[min(k + 1, K - 1)]), tho)
      one = be.backward(p, thk, st, go, sdf, nxt - thk, gcur * on[:, None, None], None, io=io)
      gcur = gcur + one['th'] * on[:, None, None]
      for key in acc: acc[key] += one[key] * on.reshape((B,) + (1,) * (one[key].ndim - 1))
    assert rel_err(r['th'], gcur) < 1e-10 and rel_err(r['start'], acc['start']) < 1e-10 and rel_err(r['goal'], acc['goal']) < 1e-10
    assert rel_err(r['sdf'], acc['sdf']) < 1e-10
    # one grid shared by the batch, partial copies: the per-sample gradients summed
    sdf1 = sdf[1:2]
    t1 = be.solve_traced(p, th0, st, go, sdf1, K, tol_delta, io=io)
    ra = be.solve_backward(p, st, go, sdf1, K, t1[2], t1[0], t1[1], gbar, io=io)
    t2 = be.solve_traced(p, th0, st, go, np.repeat(sdf1, B, 0), K, tol_delta, io=io)
    rb = be.solve_backward(p, st, go, np.repeat(sdf1, B, 0), K, t2[2], t2[0], t2[1], gbar, io=io)
    rc = be.solve_backward(p, st, go, sdf1, K, t1[2], t1[0], t1[1], gbar, io=io, sdf_copies=8)
    assert np.array_equal(ra['th'], rb['th']) and rel_err(ra['sdf'], rb['sdf'].sum(0, keepdims=True)) < 1e-11
    assert rel_err(rc['sdf'].sum(0, keepdims=True), ra['sdf']) < 1e-11
  else:
    for k, key in (('th', 'g_th0'), ('start', 'g_start'), ('goal', 'g_goal')):
      assert rel_err(r[k], g[key]) < 2e-3, (k, rel_err(r[k], g[key]))
  if io != 'f64': return
  # the (x, y, theta) robot with its non-holonomic factor (the reference cannot run it in batch): the chain kernel -- whose d = 6 instantiations
  # keep their accumulators in LDS and re-read means and rows per pass -- against the single-step backward chained by hand
  gg = golden('g3_c4_xyh')
  th6, st6, go6 = gg['th'][:3], gg['start'][:3], gg['goal'][:3]
  n6 = th6.shape[1]
  p6 = O.OracleParams(dof=3, total_time_step=n6 - 1, non_holonomic=True, epsilon_dist=0.2, reg=0.05)
  sdf6 = O.circles_sdf(int(gg['G']), gg['circles'])[None, None]
  d0 = be.step(p6, th6, st6, go6, sdf6, io=io)[0]
  nrm = np.sqrt((d0.reshape(3, -1) ** 2).sum(1))
  K6 = 3
  tho6, its6, hist6, info6 = be.solve_traced(p6, th6, st6, go6, sdf6, K6, float(np.sort(nrm)[0] * 1.01), io=io)       # one trajectory stops after its first iteration
  assert not info6.any() and its6.min() == 1 and its6.max() == K6
  gb6 = np.random.RandomState(8).randn(*th6.shape)
  r6 = be.solve_backward(p6, st6, go6, sdf6, K6, hist6, tho6, its6, gb6, io=io)
  gcur = gb6.copy(); a_s = np.zeros_like(st6); a_g = np.zeros_like(go6); a_f = np.zeros_like(sdf6)
  for k in range(K6 - 1, -1, -1):
    on = its6 > k
    thk = np.where(on[:, None, None], np.nan_to_num(hist6[k]), tho6)
    nxt = np.where((its6 > k + 1)[:, None, None], np.nan_to_num(hist6[min(k + 1, K6 - 1)]), tho6)
    one = be.backward(p6, thk, st6, go6, sdf6, nxt - thk, gcur * on[:, None, None], None, io=io)
    gcur = gcur + one['th'] * on[:, None, None]; a_s += one['start'] * on[:, None, None]; a_g += one['goal'] * on[:, None, None]; a_f += one['sdf']
  assert rel_err(r6['th'], gcur) < 1e-9 and rel_err(r6['start'], a_s) < 1e-9 and rel_err(r6['goal'], a_g) < 1e-9 and rel_err(r6['sdf'], a_f) < 1e-7


ALL_CASES.append(case_solve_backward)


def case_solve_backward_general_qc(be, golden, io):
  """Round 6: dgp_gn_solve_traced + dgp_gn_solve_backward with a NON-DIAGONAL static Q_c_inv (the general-covariance chain kernels) vs the reference's torch
  autograd through DiffGPMP2Planner.forward with gp_params['Q_c_inv'] = [[1.3, 0.4], [0.4, 0.9]] (fixture g8_forward_grads_qc; diff_gpmp2_planner.py:92-174):
  iteration counts, final trajectories, gradients w.r.t. the initial trajectory, the grids, the start and goal means; and the chain against single-step
  backward launches chained by hand."""
  g = golden('g8_forward_grads_qc')
  B, n = g['th0'].shape[:2]
  G = int(g['G'])
  p = P2d(n, Q_c_inv=g['Q_c_inv'])
  K, tol_delta = int(g['max_iters']), float(g['tol_delta'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)).copy()
  sdf[int(g['free_sample'])] = float(g['free_value'])
  th0, st, go, sdf, gbar = rnd(g['th0'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io), rnd(g['gbar'], io)
  tho, its, hist, info = be.solve_traced(p, th0, st, go, sdf, K, tol_delta, io=io)
  assert np.all(info == 0) and np.array_equal(its, g['iters'])
  ref = be.solve(p, th0, st, go, sdf, K, tol_delta, io=io)
  assert np.array_equal(tho, ref[0]) and np.array_equal(its, ref[1])
  assert rel_err(tho, g['th_final']) < (1e-8 if io == 'f64' else 2e-3)
  r = be.solve_backward(p, st, go, sdf, K, hist, tho, its, gbar, io=io)
  tol = 5e-8 if io == 'f64' else 2e-3
  for k, key in (('th', 'g_th0'), ('start', 'g_start'), ('goal', 'g_goal')) + ((('sdf', 'g_sdf'),) if io == 'f64' else ()):
    assert rel_err(r[k], g[key]) < tol, (k, rel_err(r[k], g[key]))
  if io != 'f64': return
  gcur = gbar.copy()
  acc = dict(start=np.zeros_like(st), goal=np.zeros_like(go), sdf=np.zeros_like(sdf))
  for k in range(K - 1, -1, -1):
    on = its > k
    thk = np.where(on[:, None, None], np.nan_to_num(hist[k]), tho)
    nxt = np.where((its > k + 1)[:, None, None], np.nan_to_num(hist[min(k + 1, K - 1)]), tho)
    one = be.backward(p, thk, st, go, sdf, nxt - thk, gcur * on[:, None, None], None, io=io)
    gcur = gcur + one['th'] * on[:, None, None]
    for key in acc: acc[key] += one[key] * on.reshape((B,) + (1,) * (one[key].ndim - 1))
  assert rel_err(r['th'], gcur) < 1e-10 and rel_err(r['start'], acc['start']) < 1e-10 and rel_err(r['goal'], acc['goal']) < 1e-10 and rel_err(r['sdf'], acc['sdf']) < 1e-10


ALL_CASES.append(case_solve_backward_general_qc)


def case_step_errors(be, golden, io):
  """dgp_gn_step_errors / dgp_gn_step_errors_backward: one iteration of the reference's training loop (learning/train_planner.py:311-327 --
  step(), th + dtheta, unweighted_errors_batch) as single calls, against the reference's autograd through exactly that composition (fixture
  g7_errors, part (b), per-state covariances; its loss also holds error_ext_batch at th + dtheta, whose share is added here through the
  existing single-purpose entry points), and bit for bit against the two-call sequence it replaces."""
  g = golden('g7_errors')
  B, n = g['th'].shape[:2]
  p = P2d(n)
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)).copy()
  th, st, go, sdf = rnd(g['th'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io)
  qc, ow, eps = rnd(g['qc'], io), rnd(g['ow'].reshape(B, n), io), rnd(g['eps'].reshape(B, n), io)
  cs, cg, co, ce = rnd(g['c_sg'], io).reshape(B), rnd(g['c_gp'], io).reshape(B), rnd(g['c_obs'], io).reshape(B), rnd(g['c_ee'], io).reshape(B)
  dth, err, eex, info, usg, ugp, uobs = be.step_errors(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)
  # == the two calls it replaces (the sum th + dtheta formed in the I/O type, as torch does)
  d2, e2, x2, i2 = be.step(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)
  npdt = np.float64 if io == 'f64' else np.float32
  th_new = (th.astype(npdt) + d2.astype(npdt)).astype(np.float64)
  _, _, s2, g2, o2 = be.eval_errors(p, th_new, st, go, sdf, eps=eps, io=io)
  # (to rounding, not bit for bit, since round 5: with a row-major grid and up to 128 states the call is ONE launch of the step kernels that carry the errors epilogue -- a
  #  separate compilation of the same source, whose FMA contraction may differ from the standard kernels')
  same = lambda a_, b_: rel_err(a_, b_) < (1e-12 if io == 'f64' else 2e-6)
  assert same(dth, d2) and same(err, e2) and same(eex, x2) and np.array_equal(info, i2)
  assert same(usg, s2) and same(ugp, g2) and same(uobs, o2)
  tolv = 1e-11 if io == 'f64' else 3e-5
  assert rel_err(dth, g['b_dth']) < TOL[io]
  for got, key in ((usg, 'b_sg'), (ugp, 'b_gp'), (uobs, 'b_obs')):
    assert rel_err(got, g[key].reshape(-1)) < tolv, (key, rel_err(got, g[key].reshape(-1)))
  # backward: cotangents of the three unweighted errors through the fused call + the error_ext share through the single-purpose calls
  r = be.step_errors_backward(p, th, st, go, sdf, dth, None, None, cs, cg, co, qc=qc, ow=ow, eps=eps, io=io)
  ee = be.eval_backward(p, th_new, st, go, sdf, g_err_ext=ce, eps=eps, io=io)
  r2 = be.backward(p, th, st, go, sdf, dth, ee['th'], None, qc=qc, ow=ow, eps=eps, io=io)
  tot = dict(th=r['th'] + r2['th'] + ee['th'], start=r['start'] + r2['start'] + ee['start'], goal=r['goal'] + r2['goal'] + ee['goal'],
             sdf=r['sdf'] + r2['sdf'] + ee['sdf'], qc=r['qc'] + r2['qc'], ow=r['ow'] + r2['ow'], eps=r['eps'] + r2['eps'] + ee['eps'])
  tolg = 1e-9 if io == 'f64' else 2e-3
  for k in ('th', 'sdf', 'start', 'goal', 'qc', 'ow', 'eps'):
    ref = g['b_g_' + k].reshape(tot[k].shape)
    assert rel_err(tot[k], ref) < tolg, (k, rel_err(tot[k], ref))
  # ... and the fused backward == its two halves run one after the other by hand (errors' backward at th + dtheta, then the step's with that
  # gradient joined to the dtheta cotangent), with a dtheta cotangent and an err_ext cotangent as well
  gd = rnd(np.random.RandomState(3).randn(B, n, 4), io)
  fused = be.step_errors_backward(p, th, st, go, sdf, dth, gd, ce, cs, cg, co, qc=qc, ow=ow, eps=eps, io=io)
  h1 = be.eval_backward(p, th_new, st, go, sdf, None, cs, cg, co, eps=eps, io=io)
  h2 = be.backward(p, th, st, go, sdf, dth, (gd.astype(npdt) + h1['th'].astype(npdt)).astype(np.float64), ce, qc=qc, ow=ow, eps=eps, io=io)
  ft = 1e-12 if io == 'f64' else 2e-5
  assert rel_err(fused['th'], h2['th'] + h1['th']) < ft and rel_err(fused['start'], h2['start'] + h1['start']) < ft
  assert rel_err(fused['eps'], h2['eps'] + h1['eps']) < ft and rel_err(fused['sdf'], h2['sdf'] + h1['sdf']) < ft
  assert rel_err(fused['qc'], h2['qc']) < ft and rel_err(fused['ow'], h2['ow']) < ft
  # static covariances and a shared grid with partial copies; no error cotangent at all == dgp_gn_step_backward
  s1 = be.step_errors(p, th, st, go, sdf[:1], io=io)
  f1 = be.step_errors_backward(p, th, st, go, sdf[:1], s1[0], gd, None, cs, cg, co, io=io, sdf_copies=8)
  f2 = be.step_errors_backward(p, th, st, go, np.repeat(sdf[:1], B, 0), s1[0], gd, None, cs, cg, co, io=io)
  assert rel_err(f1['th'], f2['th']) < ft and rel_err(f1['sdf'].sum(0, keepdims=True), f2['sdf'].sum(0, keepdims=True)) < (1e-11 if io == 'f64' else 1e-4)
  f3 = be.step_errors_backward(p, th, st, go, sdf, dth, gd, ce, None, None, None, qc=qc, ow=ow, eps=eps, io=io)
  f4 = be.backward(p, th, st, go, sdf, dth, gd, ce, qc=qc, ow=ow, eps=eps, io=io)
  for k in ('th', 'start', 'goal', 'qc', 'ow', 'eps'): assert np.array_equal(f3[k], f4[k]), k
  assert rel_err(f3['sdf'], f4['sdf']) < ft           # (atomic accumulation: the order of the additions is not fixed)


ALL_CASES.append(case_step_errors)


def case_sdf_gradient_delivery(be, golden, io):
  """DgpSdf::grad_mode (round 5).  Per-sample grids: the taps as COO entries (DGP_GSDF_SPARSE: no zero-filled (B,1,H,W) grid, no atomics), scattered back
  into a grid, equal the dense accumulation -- single step, the training iteration's two tap blocks, the fused loop's per-iteration blocks (early stoppers leave
  theirs untouched) and dgp_eval_errors_backward.  Shared grid: FLOAT64 partial copies behind fp32 I/O (DGP_GSDF_DENSE_F64) sum to the fp64 run's gradient far
  below what fp32 atomics in arbitrary order reach."""
  g = golden('g7_errors')
  B, n = g['th'].shape[:2]
  p = P2d(n)
  G = int(g['G'])
  rs = np.random.RandomState(11)
  base = O.circles_sdf(G, g['circles'])
  sdf = np.stack([base + 0.05 * rs.randn(G, G) for _ in range(B)])[:, None]      # B distinct grids
  th, st, go, sdf = rnd(g['th'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io)
  qc, ow, eps = rnd(g['qc'], io), rnd(g['ow'].reshape(B, n), io), rnd(g['eps'].reshape(B, n), io)
  cs, cg, co, ce = rnd(g['c_sg'], io).reshape(B), rnd(g['c_gp'], io).reshape(B), rnd(g['c_obs'], io).reshape(B), rnd(g['c_ee'], io).reshape(B)
  gd = rnd(rs.randn(B, n, 4), io)
  tol = 1e-12 if io == 'f64' else 2e-6      # (the same fp32 / fp64 summands; only the order of a pixel's few additions differs)
  dth = be.step(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)[0]
  d = be.backward(p, th, st, go, sdf, dth, gd, ce, qc=qc, ow=ow, eps=eps, io=io)
  s = be.backward(p, th, st, go, sdf, dth, gd, ce, qc=qc, ow=ow, eps=eps, io=io, sdf_grad='sparse')
  assert np.abs(d['sdf']).max() > 0 and s['sdf'].shape == d['sdf'].shape and rel_err(s['sdf'], d['sdf']) < tol
  for k in ('th', 'start', 'goal', 'qc', 'ow', 'eps'): assert np.array_equal(s[k], d[k]), k
  # static covariances (the Woodbury / static backward kernels)
  dth0 = be.step(p, th, st, go, sdf, io=io)[0]
  d0 = be.backward(p, th, st, go, sdf, dth0, gd, None, io=io)
  s0 = be.backward(p, th, st, go, sdf, dth0, gd, None, io=io, sdf_grad='sparse')
  assert rel_err(s0['sdf'], d0['sdf']) < tol and np.array_equal(s0['th'], d0['th'])
  # the training iteration: taps at th + dtheta and at th
  d2 = be.step_errors_backward(p, th, st, go, sdf, dth, gd, ce, cs, cg, co, qc=qc, ow=ow, eps=eps, io=io)
  s2 = be.step_errors_backward(p, th, st, go, sdf, dth, gd, ce, cs, cg, co, qc=qc, ow=ow, eps=eps, io=io, sdf_grad='sparse')
  assert rel_err(s2['sdf'], d2['sdf']) < tol and rel_err(s2['th'], d2['th']) < tol
  s3 = be.step_errors_backward(p, th, st, go, sdf, dth, gd, ce, None, None, None, qc=qc, ow=ow, eps=eps, io=io, sdf_grad='sparse')      # no error cotangent: one block
  assert rel_err(s3['sdf'], d['sdf']) < tol
  # the errors' backward on its own
  d4 = be.eval_backward(p, th, st, go, sdf, ce, cs, cg, co, eps=eps, io=io)
  s4 = be.eval_backward(p, th, st, go, sdf, ce, cs, cg, co, eps=eps, io=io, sdf_grad='sparse')
  assert rel_err(s4['sdf'], d4['sdf']) < tol
  # the fused loop: one tap block per iteration (the g8 problem: trajectories that stop after 2, 8, 9, 9 iterations -- early stoppers leave their later blocks zero)
  g8 = golden('g8_forward_grads')
  B8, n8, G8 = g8['th0'].shape[0], g8['th0'].shape[1], int(g8['G'])
  p8 = P2d(n8)
  K8 = int(g8['max_iters'])
  sdf8 = np.broadcast_to(O.circles_sdf(G8, g8['circles']), (B8, 1, G8, G8)).copy()
  sdf8[int(g8['free_sample'])] = float(g8['free_value'])
  th8, st8, go8, sdf8, gb8 = rnd(g8['th0'], io), rnd(g8['start'], io), rnd(g8['goal'], io), rnd(sdf8, io), rnd(g8['gbar'], io)
  tho, its, hist, info = be.solve_traced(p8, th8, st8, go8, sdf8, K8, float(g8['tol_delta']), io=io)
  assert its.min() < its.max()
  d5 = be.solve_backward(p8, st8, go8, sdf8, K8, hist, tho, its, gb8, io=io)
  s5 = be.solve_backward(p8, st8, go8, sdf8, K8, hist, tho, its, gb8, io=io, sdf_grad='sparse')
  assert np.abs(d5['sdf']).max() > 0 and rel_err(s5['sdf'], d5['sdf']) < (1e-11 if io == 'f64' else 1e-5) and rel_err(s5['th'], d5['th']) < tol
  # shared grid: double partial copies whatever the I/O type
  sh = sdf[:1]
  dthS = be.step(p, th, st, go, sh, qc=qc, ow=ow, eps=eps, io=io)[0]
  ref = be.backward(p, th, st, go, sh, dthS, gd, ce, qc=qc, ow=ow, eps=eps, io=io)
  w = be.backward(p, th, st, go, sh, dthS, gd, ce, qc=qc, ow=ow, eps=eps, io=io, sdf_copies=16, sdf_grad='f64')
  assert w['sdf'].shape[0] == 16 and rel_err(w['sdf'].sum(0, keepdims=True), ref['sdf']) < (1e-11 if io == 'f64' else 2e-5)
  if io == 'f32':
    # the fp32 summands added in double: independent of the order -- two runs agree to the last bit of a double sum of a few hundred floats
    w2 = be.backward(p, th, st, go, sh, dthS, gd, ce, qc=qc, ow=ow, eps=eps, io=io, sdf_copies=16, sdf_grad='f64')
    assert rel_err(w2['sdf'].sum(0), w['sdf'].sum(0)) < 1e-13


ALL_CASES.append(case_sdf_gradient_delivery)


def case_raw_squared_covariances(be, golden, io):
  """dgp_square_covariances / dgp_square_covariances_backward (round 5): the learn module's raw output vector as covariance input -- one scalar q_k per GP factor
  ('diag_identity': q_k^2 I), raw obstacle weights o_i, raw epsilons e_i, squared by one small launch into the tensors the step takes (diff_gpmp2_planner.py:247-290 for
  a single-link robot) -- against the same step with the squares formed by the caller (DGP_QC_SCALAR + per-state tensors): the forward bit for bit, the backward's
  d/d out = 2 out d/d(out^2).  'fix_dynamics' (static Q_c_inv, n_gp = 0) with and without learned epsilons, trailing unused columns (zero gradient), and the training
  iteration's single-launch backward."""
  g = golden('g7_errors')
  B, n = g['th'].shape[:2]
  p = P2d(n)
  G = int(g['G'])
  sdf = np.broadcast_to(O.circles_sdf(G, g['circles']), (B, 1, G, G)).copy()
  th, st, go, sdf = rnd(g['th'], io), rnd(g['start'], io), rnd(g['goal'], io), rnd(sdf, io)
  rs = np.random.RandomState(17)
  npdt = np.float64 if io == 'f64' else np.float32
  gd = rnd(rs.randn(B, n, 4), io)
  cs, cg, co, ce = rnd(g['c_sg'], io).reshape(B), rnd(g['c_gp'], io).reshape(B), rnd(g['c_obs'], io).reshape(B), rnd(g['c_ee'], io).reshape(B)
  for n_gp, learn_eps, extra in ((n - 1, True, 0), (n - 1, False, 3), (0, True, 0), (0, False, 0)):
    W = n_gp + n * (2 if learn_eps else 1) + extra
    out = np.concatenate([rs.uniform(0.6, 1.4, (B, n_gp)) * rs.choice([-1.0, 1.0], (B, n_gp)),            # raw q_k (signs: only the square enters)
                          rs.uniform(60.0, 140.0, (B, n)) * rs.choice([-1.0, 1.0], (B, n)),               # raw o_i (weights ~ 1e4)
                          rs.uniform(0.4, 0.8, (B, n * (1 if learn_eps else 0) + extra))], 1)              # raw e_i (eps ~ 0.16 .. 0.64) [+ unused columns]
    out = rnd(out, io)
    sq = (out.astype(npdt) * out.astype(npdt)).astype(np.float64)                                          # the squares as torch forms them, in the I/O type
    qc = sq[:, :n_gp] if n_gp else None
    ow = sq[:, n_gp:n_gp + n]
    eps = sq[:, n_gp + n:n_gp + 2 * n] if learn_eps else None
    raw = (out, n_gp, learn_eps)
    a = be.step(p, th, st, go, sdf, raw=raw, io=io)
    b = be.step(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)
    for x, y in zip(a, b): assert np.array_equal(x, y)
    assert np.all(a[3] == 0)
    ra = be.backward(p, th, st, go, sdf, a[0], gd, ce, raw=raw, io=io)
    rb = be.backward(p, th, st, go, sdf, a[0], gd, ce, qc=qc, ow=ow, eps=eps, io=io)
    tol = 1e-13 if io == 'f64' else 3e-6
    for k in ('th', 'start', 'goal'): assert np.array_equal(ra[k], rb[k]), k
    assert rel_err(ra['sdf'], rb['sdf']) < (1e-12 if io == 'f64' else 2e-5)

    def expect(r):
      parts = []
      if n_gp: parts.append(2.0 * out[:, :n_gp] * np.einsum('bkii->bk', r['qc']))                         # the blocks' gradient -> the scalar's: trace (Q_c_inv = I)
      parts.append(2.0 * out[:, n_gp:n_gp + n] * r['ow'])
      if learn_eps: parts.append(2.0 * out[:, n_gp + n:n_gp + 2 * n] * r['eps'])
      return np.concatenate(parts, 1)
    used = W - extra
    assert np.all(ra['out'][:, used:] == 0.0)
    assert rel_err(ra['out'][:, :used], expect(rb)) < tol, (n_gp, learn_eps, rel_err(ra['out'][:, :used], expect(rb)))
    # the training iteration: forward and the single-launch backward
    ea = be.step_errors(p, th, st, go, sdf, raw=raw, io=io)
    eb = be.step_errors(p, th, st, go, sdf, qc=qc, ow=ow, eps=eps, io=io)
    for x, y in zip(ea, eb): assert np.array_equal(x, y)      # (the same launches on the same squared values)
    fa = be.step_errors_backward(p, th, st, go, sdf, ea[0], gd, ce, cs, cg, co, raw=raw, io=io)
    fb = be.step_errors_backward(p, th, st, go, sdf, ea[0], gd, ce, cs, cg, co, qc=qc, ow=ow, eps=eps, io=io)
    for k in ('th', 'start', 'goal'): assert np.array_equal(fa[k], fb[k]), k
    assert rel_err(fa['out'][:, :used], expect(fb)) < tol, (n_gp, learn_eps, 'iteration', rel_err(fa['out'][:, :used], expect(fb)))


ALL_CASES.append(case_raw_squared_covariances)


def case_long_trajectories(be, golden, io, configs=None):
  """n > 256 (gn_long.h: one trajectory per wavefront, ceil(n / 64) rows per lane in a loop, interior state parked in LDS): the reference
  accepts any total_time_step (plan_layer.py:30).  Every entry point -- step, the fused loop, the error evaluation, both backward kernels --
  for both robots, velocity-limit / non-holonomic factors, static / per-state / q_full covariances, shared and per-sample grids, lengths that
  fill the lanes (512 = 64 x 8), leave padding rows inside a lane (300) and padding lanes (257: five rows per lane, 12 lanes empty).
  Forward against oracle/gn_blocktri.c (block-tridiagonal fp64), gradients against torch autograd over the dense restatement."""
  from oracle import blocktri as BT, autograd_torch as AT
  rs = np.random.RandomState(77)
  if configs is None:
    configs = [(2, 300, 'static', {}), (2, 512, 'perstate', dict(use_vel_limits=True, K_v=0.01, v_x=0.5, v_y=0.5)), (3, 257, 'static', dict(non_holonomic=True, K_d=0.05)),
               (3, 320, 'qfull', {}), (2, 1024, 'static', {}), (3, 640, 'perstate', dict(non_holonomic=True, K_d=0.05))]
  for dof, n, cov, kw in configs:
    B, d, G = 2, 2 * dof, 48
    p = O.OracleParams(dof=dof, total_time_step=n - 1, **kw)
    per_sample = cov != 'static'
    sdf = np.stack([O.circles_sdf(G, O.C2_CIRCLES + rs.uniform(-0.3, 0.3, np.shape(O.C2_CIRCLES))) for _ in range(B if per_sample else 1)])[:, None]
    start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
    start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2))
    if dof == 3: goal[:, 0, 2] = rs.uniform(-1, 1, B)
    th = O.straight_line_trajb(start[:, :, :dof], goal[:, :, :dof], 10.0, n - 1, dof) + rs.randn(B, n, d) * 0.05
    qc = ow = eps = None; q_full = False
    if cov != 'static':
      ow = rs.uniform(50, 2e4, (B, n)); eps = rs.uniform(0.1, 0.6, (B, n))
      if cov == 'perstate':
        A = rs.randn(B, n - 1, dof, dof) * 0.2; qc = np.eye(dof) + A @ np.swapaxes(A, -1, -2)
      else:
        A = rs.randn(B, n - 1, d, d) * 0.2; qc = (np.eye(d) + A @ np.swapaxes(A, -1, -2)) * 1.5; q_full = True
    r = lambda a: None if a is None else rnd(a, io)
    th, start, goal, sdf, qc, ow, eps = r(th), r(start), r(goal), r(sdf), r(qc), r(ow), r(eps)
    kwc = dict(qc=qc, ow=ow, eps=eps, q_full=q_full)
    tag = 'long: dof %d n %d %s %s' % (dof, n, cov, io)
    tol = TOL[io] * (10 if n > 512 else 1)               # cond(Lambda) grows with n (dt^-3 in Q^-1): the fp64 oracles themselves differ by that much
    # ---- one step, every trajectory against the block-tridiagonal C oracle
    dth, err, eex, info = be.step(p, th, start, goal, sdf, io=io, **kwc)
    c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, **kwc)
    assert not info.any() and not c_info.any() and np.all(np.isfinite(dth)), tag
    assert rel_err_per_traj(dth, c_dth) < tol, (tag, 'step', rel_err_per_traj(dth, c_dth))
    assert rel_err(err, c_err) < TOL_ERR[io] and rel_err(eex, c_eex) < TOL_ERR[io], (tag, 'errors of the step')
    # ---- the error evaluation against the step's own errors and the oracle's unweighted errors
    e = be.eval_errors(p, th, start, goal, sdf, io=io, **kwc)
    assert rel_err(e[0], c_err) < TOL_ERR[io] and rel_err(e[1], c_eex) < TOL_ERR[io], (tag, 'eval_errors')
    sdfB = np.broadcast_to(sdf, (B,) + sdf.shape[1:])
    import torch
    usg, ugp, uob = AT.unweighted_errors(*[torch.from_numpy(np.array(a, dtype=np.float64)) for a in
                                           (th, start, goal, sdfB, (p.static_covs(B)[2] if eps is None else eps.reshape(B, n, 1, 1)))], p)
    for got, want, name in ((e[2], usg, 'sg'), (e[3], ugp, 'gp'), (e[4], uob, 'obs')):
      assert rel_err(got, want.numpy().reshape(-1)) < TOL_ERR[io] * 10, (tag, 'unweighted ' + name, rel_err(got, want.numpy().reshape(-1)))
    # ---- the fused loop equals chained steps (3 iterations; fp64 I/O: chained fp32 steps would round the state in between)
    if io == 'f64' and n <= 512:
      tho, its, eh, eeh, ef, sinfo = be.solve(p, th, start, goal, sdf, 3, 0.0, io=io, **kwc)
      cur = th.copy()
      for k in range(3):
        d_k, e_k, x_k, i_k = be.step(p, cur, start, goal, sdf, io=io, **kwc)
        assert rel_err(eh[:, k], e_k) < 1e-10 and rel_err(eeh[:, k], x_k) < 1e-10, (tag, 'history', k)
        cur = cur + d_k
      assert np.all(its == 3) and not sinfo.any() and rel_err(tho, cur) < 1e-9, (tag, 'fused loop', rel_err(tho, cur))
      assert rel_err(ef, be.eval_errors(p, cur, start, goal, sdf, io=io, **kwc)[0]) < 1e-9, (tag, 'err_final')
    # ---- backward of the step against torch autograd over the dense restatement (N = n d up to 1 920 here)
    if n <= 320:
      gbar = r(rs.randn(B, n, d)); gext = r(rs.randn(B))
      shared = sdf.shape[0] == 1
      g_h = be.backward(p, th, start, goal, sdf, rnd(dth, io), gbar, gext, io=io, sdf_copies=(16 if shared else 1), **kwc)
      g_o = AT.step_gradients(p, th, start, goal, sdf, gbar, gext, **kwc)
      for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'):
        if g_h[key] is None or (key == 'sdf' and io == 'f32'): continue
        a_ = g_h[key]
        if key == 'sdf' and shared: a_ = a_.sum(0, keepdims=True)
        b_ = g_o[key].reshape(a_.shape)
        eb = np.abs(a_ - b_).max() / max(np.abs(b_).max(), np.abs(g_o['th']).max() if key == 'sdf' else 0.0, 1e-300)
        assert eb < (1e-6 if io == 'f64' else 2e-3), (tag, 'backward', key, eb)
      # ... and of the error evaluation (unweighted errors + err_ext), against autograd over the same restatement
      cot = [r(rs.randn(B)) for _ in range(4)]
      g_e = be.eval_backward(p, th, start, goal, sdf, g_err_ext=cot[0], g_unw_sg=cot[1], g_unw_gp=cot[2], g_unw_obs=cot[3], eps=eps, io=io,
                             sdf_copies=(16 if shared else 1))
      T = lambda a: torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float64)))
      L = dict(th=T(th), start=T(start), goal=T(goal), sdf=T(sdf), eps=T(p.static_covs(B)[2] if eps is None else eps.reshape(B, n, 1, 1)))
      for v in L.values(): v.requires_grad_(True)
      sB = L['sdf'].expand(B, *L['sdf'].shape[1:]) if shared else L['sdf']
      sq, so, se = p.static_covs(B)
      _, _, eext = AT.plan_layer_forward(L['th'], L['start'], L['goal'], sB, T(sq), T(so), L['eps'], p)
      usg, ugp, uob = AT.unweighted_errors(L['th'], L['start'], L['goal'], sB, L['eps'], p)
      loss = (T(cot[0]) * eext.reshape(B)).sum() + (T(cot[1]) * usg.reshape(B)).sum() + (T(cot[2]) * ugp.reshape(B)).sum() + (T(cot[3]) * uob.reshape(B)).sum()
      gr = dict(zip(L.keys(), torch.autograd.grad(loss, list(L.values()), allow_unused=True)))
      for key in ('th', 'start', 'goal', 'sdf', 'eps'):
        if g_e[key] is None or (key == 'sdf' and io == 'f32'): continue
        a_ = g_e[key]
        if key == 'sdf' and shared: a_ = a_.sum(0, keepdims=True)
        b_ = gr[key].numpy().reshape(a_.shape)
        eb = np.abs(a_ - b_).max() / max(np.abs(b_).max(), np.abs(gr['th'].numpy()).max() if key == 'sdf' else 0.0, 1e-300)
        assert eb < (1e-8 if io == 'f64' else 1e-4), (tag, 'eval backward', key, eb)
