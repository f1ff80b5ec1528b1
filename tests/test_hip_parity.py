"""GPU parity tests proper: the HIP path, called through the C-ABI (dgpmp2_amd/lib/libdgpmp2_hip.so), against the
numpy oracle on the same seeded inputs and against the golden fixtures generated from the reference."""
import numpy as np
import pytest
import harness
import parity_cases as PC
from conftest import rel_err
from oracle import gpmp2_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def be():
  import torch
  assert torch.cuda.is_available(), 'the -m gpu tests need a GPU'
  return harness.Backend('hip')


@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('case', PC.ALL_CASES, ids=lambda c: c.__name__)
def test_hip_case(be, golden, case, io):
  case(be, golden, io)


@pytest.mark.parametrize('io', ['f64', 'f32'])
def test_hip_tiled_grids(golden, io):
  """DgpSdf::layout = DGP_SDF_TILED4: parity cases re-run with every grid stored as 4 x 4 tiles (harness.Backend.sdf_tiled) -- per-sample and shared grids, odd-sized
  and non-square ones (padding cells), trajectories leaving the grid, both robots, the fused loop, the backward pass (dense gradients come back tiled and are
  untiled for the comparison), the training iteration."""
  bt = harness.Backend('hip'); bt.sdf_tiled = True
  for case in (PC.case_c2mini_per_sample_sdf, PC.case_c2mini_covs, PC.case_edges, PC.case_c1, PC.case_c4_xyh, PC.case_solve, PC.case_backward_golden,
               PC.case_shared_sdf_gradient_partial_copies, PC.case_sdf_gradient_delivery, PC.case_step_errors, PC.case_solve_backward, PC.case_eval_errors_backward):
    case(bt, golden, io)
  # the 64-lane launch shapes (more than 128 states) are compiled without the tiled branch: refused, not silently read as row-major
  from dgpmp2_amd import _capi
  p = PC.P2d(160)
  th = np.zeros((2, 160, 4)); st = np.zeros((2, 1, 4)); sdf = np.ones((1, 1, 16, 16))
  with pytest.raises(_capi.DgpError) as e:
    bt.step(p, th, st, st, sdf, io=io)
  assert e.value.code == _capi.DGP_EUNSUPPORTED and 'tiled' in str(e.value)


@pytest.mark.parametrize('io', ['f64', 'f32'])
def test_hip_c2_tiled_equals_row_major(io):
  """BASELINE config 2's shape with 1024 DISTINCT grids: the step on the tiled grids equals the step on the row-major ones to rounding (the same taps read from
  another address -- but by the tiled translation units, a separate compilation of the same source whose FMA contraction may differ), and so does the per-sample
  grid gradient once untiled."""
  B, n, G = 1024, 64, 256
  p = PC.P2d(n)
  th, start, goal, _ = _c2_inputs(B, n, G, seed=0, perturb=0.05)
  rs = np.random.RandomState(5)
  base = O.circles_sdf(G, O.C2_CIRCLES)
  sdf = (base[None, None] + 0.02 * rs.randn(B, 1, 1, 1) + 0.01 * rs.randn(1, 1, G, G)).astype(np.float64)      # (distinct grids without 2 GiB of host memory)
  th, start, goal, sdf = PC.rnd(th, io), PC.rnd(start, io), PC.rnd(goal, io), PC.rnd(sdf, io)
  br = harness.Backend('hip'); bt = harness.Backend('hip'); bt.sdf_tiled = True
  a = br.step(p, th, start, goal, sdf, io=io); b = bt.step(p, th, start, goal, sdf, io=io)
  tol = 1e-11 if io == 'f64' else 2e-6
  assert PC.rel_err_per_traj(a[0], b[0]) < tol and rel_err(a[1], b[1]) < tol and rel_err(a[2], b[2]) < tol and np.array_equal(a[3], b[3])
  gd = PC.rnd(rs.randn(B, n, 4), io)
  ga = br.backward(p, th, start, goal, sdf, a[0], gd, None, io=io); gb = bt.backward(p, th, start, goal, sdf, a[0], gd, None, io=io)
  assert rel_err(ga['th'], gb['th']) < 100 * tol and rel_err(ga['sdf'], gb['sdf']) < (1e-9 if io == 'f64' else 1e-4) and np.abs(ga['sdf']).max() > 0


def _c2_inputs(B, n, G, seed=0, perturb=0.0):
  rs = np.random.RandomState(seed)
  start = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  goal = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  th = O.straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  if perturb: th = th + rs.randn(B, n, 4) * perturb
  sdf = O.circles_sdf(G, O.C2_CIRCLES)[None, None]
  return th, start, goal, sdf


@pytest.mark.parametrize('io', ['f64', 'f32'])
def test_hip_c2_full_size_properties(be, io):
  """BASELINE config 2 at full size (B=4096, n=64, 256x256 SDF): the dense oracle cannot run 4096 trajectories in
  seconds, so check (a) a random subset of 64 trajectories against the oracle, (b) batch-independence: the result
  for a trajectory does not depend on its batch neighbours or position, (c) determinism across launches."""
  B, n, G = 4096, 64, 256
  p = PC.P2d(n)
  th, start, goal, sdf = _c2_inputs(B, n, G, seed=0, perturb=0.05)
  th, start, goal, sdf = PC.rnd(th, io), PC.rnd(start, io), PC.rnd(goal, io), PC.rnd(sdf, io)
  dth, err, eex, info = be.step(p, th, start, goal, sdf, io=io)
  assert np.all(info == 0) and np.all(np.isfinite(dth))
  idx = np.random.RandomState(1).choice(B, 64, replace=False)
  qc, ow, eps = p.static_covs(64)
  r_dth, r_err, r_eex = O.plan_layer_forward(th[idx], start[idx], goal[idx], np.broadcast_to(sdf, (64, 1, G, G)), qc, ow, eps, p)
  assert rel_err(dth[idx], r_dth) < PC.TOL[io]
  assert rel_err(err[idx], r_err.reshape(-1)) < PC.TOL_ERR[io] and rel_err(eex[idx], r_eex.reshape(-1)) < PC.TOL_ERR[io]
  # (a') EVERY trajectory against the oracle's block-tridiagonal C restatement (oracle/gn_blocktri.c)
  from oracle import blocktri as BT
  c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, nthreads=4)
  assert not c_info.any()
  per_traj = np.abs(dth - c_dth).reshape(B, -1).max(1) / np.abs(c_dth).reshape(B, -1).max(1)
  assert per_traj.max() < PC.TOL[io], per_traj.max()
  assert rel_err(err, c_err) < PC.TOL_ERR[io] and rel_err(eex, c_eex) < PC.TOL_ERR[io]
  perm = np.random.RandomState(2).permutation(B)
  dth2, err2, _, _ = be.step(p, th[perm], start[perm], goal[perm], sdf, io=io)
  assert np.array_equal(dth2, dth[perm]) and np.array_equal(err2, err[perm])
  dth3, _, _, _ = be.step(p, th, start, goal, sdf, io=io)
  assert np.array_equal(dth3, dth)


@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('config', ['c3_vel_limits', 'c4_xyh'])
def test_hip_c3_c4_full_size_vs_oracles(be, config, io):
  """BASELINE configs 3 (2D point robot + velocity-limit factors) and 4 (non-holonomic (x,y,theta) robot, 6-dim state,
  512x512 SDF) at full size, B=4096 x 64 states: every trajectory against oracle/gn_blocktri.c, a random subset of 48
  against the dense numpy restatement of the reference."""
  from oracle import blocktri as BT
  B, n = 4096, 64
  rs = np.random.RandomState(11)
  if config == 'c3_vel_limits':
    G, dof = 256, 2
    p = O.OracleParams(dof=2, total_time_step=n - 1, use_vel_limits=True, K_v=0.01, v_x=1.0, v_y=1.0)
  else:
    G, dof = 512, 3
    p = O.OracleParams(dof=3, total_time_step=n - 1, non_holonomic=True, K_d=0.01, epsilon_dist=0.2, reg=0.0)
  d = 2 * dof
  start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
  start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2))
  if dof == 3: goal[:, 0, 2] = np.pi / 2                          # examples/diff_gpmp2_nonholonomic_example.py:44-46
  th = O.straight_line_trajb(start[:, :, :dof], goal[:, :, :dof], 10.0, n - 1, dof) + rs.randn(B, n, d) * 0.05
  sdf = O.circles_sdf(G, O.C2_CIRCLES)[None, None]
  th, start, goal, sdf = PC.rnd(th, io), PC.rnd(start, io), PC.rnd(goal, io), PC.rnd(sdf, io)
  dth, err, eex, info = be.step(p, th, start, goal, sdf, io=io)
  assert np.all(info == 0) and np.all(np.isfinite(dth))
  c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, nthreads=4)
  assert not c_info.any()
  per_traj = np.abs(dth - c_dth).reshape(B, -1).max(1) / np.abs(c_dth).reshape(B, -1).max(1)
  assert per_traj.max() < PC.TOL[io], per_traj.max()
  assert rel_err(err, c_err) < PC.TOL_ERR[io] and rel_err(eex, c_eex) < PC.TOL_ERR[io]
  idx = rs.choice(B, 48, replace=False)
  qc, ow, eps = p.static_covs(48)
  r_dth, r_err, r_eex = O.plan_layer_forward(th[idx], start[idx], goal[idx], np.broadcast_to(sdf, (48, 1, G, G)), qc, ow, eps, p)
  assert rel_err(dth[idx], r_dth) < PC.TOL[io]
  assert rel_err(err[idx], r_err.reshape(-1)) < PC.TOL_ERR[io] and rel_err(eex[idx], r_eex.reshape(-1)) < PC.TOL_ERR[io]


def test_hip_c2_ten_iterations_reduce_error(be):
  """10 GN iterations on the C2 workload (fused loop): error decreases overall and equals 10 chained steps."""
  B, n, G = 256, 64, 256
  p = PC.P2d(n)
  th, start, goal, sdf = _c2_inputs(B, n, G, seed=3)
  tho, its, eh, eeh, ef, info = be.solve(p, th, start, goal, sdf, 10, 0.0, io='f64')
  assert np.all(its == 10) and np.all(info == 0)
  assert ef.mean() < 0.25 * eh[:, 0].mean()          # GN without line search: not monotone per trajectory, but converging overall
  # the fused loop and the step kernel are separately compiled programs (different FMA contraction / operation order), and
  # ten undamped GN iterations amplify that rounding noise: agreement is to the fp64 parity tolerance, not to the last bit
  cur = th.copy()
  for k in range(10):
    dth, err, _, _ = be.step(p, cur, start, goal, sdf, io='f64')
    assert rel_err(err, eh[:, k]) < 1e-9
    cur = cur + dth
  assert rel_err(cur, tho) < 1e-9


@pytest.mark.parametrize('B,n', [(65537, 64), (1, 64), (3, 256), (1000, 101), (257, 7), (300, 512), (67, 257), (5, 1024)])
def test_hip_odd_batches_and_lengths_vs_c_oracle(be, B, n):
  """Ragged grids (batch not a multiple of the trajectories per wavefront), the longest trajectory of the unrolled kernels (n = 256:
  64 lanes x 4 states), the reference's default n = 101, and lengths beyond 256 (the loop kernels of gn_long.h: 512 = 64 lanes x 8 rows,
  257 with padding lanes, 1024 = the limit for d = 4), against oracle/gn_blocktri.c on every trajectory."""
  from oracle import blocktri as BT
  p = PC.P2d(n)
  th, start, goal, sdf = _c2_inputs(B, n, 128, seed=B + n, perturb=0.03)
  dth, err, eex, info = be.step(p, th, start, goal, sdf, io='f64')
  c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, nthreads=4)
  assert not info.any() and not c_info.any() and np.all(np.isfinite(dth))
  per_traj = np.abs(dth - c_dth).reshape(B, -1).max(1) / np.abs(c_dth).reshape(B, -1).max(1)
  assert per_traj.max() < (1e-9 if n <= 512 else 1e-8), per_traj.max()      # (cond(Lambda) grows with n)
  assert rel_err(err, c_err) < 1e-11 and rel_err(eex, c_eex) < 1e-11


@pytest.mark.parametrize('io', ['f64', 'f32'])
def test_hip_long_trajectories(be, golden, io):
  """n > 256: every entry point of the loop kernels (gn_long.h), see parity_cases.case_long_trajectories."""
  PC.case_long_trajectories(be, golden, io)


def test_hip_rejects_too_long_trajectory(be):
  from dgpmp2_amd import _capi
  for p in (PC.P2d(1025), O.OracleParams(dof=3, total_time_step=640)):      # LDS capacity of the long-trajectory kernels: n <= 1024 (d = 4), 640 (d = 6)
    with pytest.raises(_capi.DgpError) as e:
      _capi.Solver(harness.config_from_oracle(p, 'f64'))
    assert e.value.code == _capi.DGP_EUNSUPPORTED
  _capi.Solver(harness.config_from_oracle(PC.P2d(1024), 'f64')); _capi.Solver(harness.config_from_oracle(O.OracleParams(dof=3, total_time_step=639), 'f64'))


@pytest.mark.parametrize('covs', ['static', 'perstate'])
@pytest.mark.parametrize('shape', ['16,4', '32,2', '64,1', '32,4', '64,2', '64,4'])
def test_hip_d6_every_launch_shape(be, shape, covs, monkeypatch):
  """Non-holonomic (x,y,theta) robot, d = 6, n = 64: every launch shape that covers 64 states (the library picks (16,4) for
  BASELINE configs[3]; the others serve other lengths / batch sizes), pinned with DGP_FORCE_SHAPE, static and per-state
  covariance kernels, ragged batch, every trajectory against oracle/gn_blocktri.c."""
  from oracle import blocktri as BT
  monkeypatch.setenv('DGP_FORCE_SHAPE', shape)
  B, n, G, d = 37, 64, 128, 6
  rs = np.random.RandomState(31)
  p = O.OracleParams(dof=3, total_time_step=n - 1, non_holonomic=True, K_d=0.01, epsilon_dist=0.2, reg=0.0)
  start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
  start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, 2] = np.pi / 2
  th = O.straight_line_trajb(start[:, :, :3], goal[:, :, :3], 10.0, n - 1, 3) + rs.randn(B, n, d) * 0.05
  sdf = O.circles_sdf(G, O.C2_CIRCLES)[None, None]
  qc = ow = eps = None
  if covs == 'perstate':
    a = rs.randn(B, n - 1, 3, 3) * 0.3
    qc = a @ a.transpose(0, 1, 3, 2) + 0.5 * np.eye(3)
    ow = rs.uniform(0.25, 1.75, (B, n)) * 1e4; eps = rs.uniform(0.1, 0.4, (B, n))
  dth, err, eex, info = be.step(p, th, start, goal, sdf, qc=qc, ow=ow, eps=eps, io='f64')
  c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, qc=qc, ow=ow, eps=eps, nthreads=2)
  assert not info.any() and not c_info.any() and np.all(np.isfinite(dth))
  per_traj = np.abs(dth - c_dth).reshape(B, -1).max(1) / np.abs(c_dth).reshape(B, -1).max(1)
  assert per_traj.max() < 1e-9, per_traj.max()
  assert rel_err(err, c_err) < 1e-11 and rel_err(eex, c_eex) < 1e-11


def _per_sample_circle_sdfs(B, G, seed=1):
  """SURVEY 8(d) per-sample mode: three circles per trajectory, centres ~ U(-3.5,3.5)^2, radii ~ U(0.4,1.0)."""
  rs = np.random.RandomState(seed)
  c = rs.uniform(-3.5, 3.5, (B, 3, 2)); r = rs.uniform(0.4, 1.0, (B, 3))
  xs = np.linspace(-5.0, 5.0, G); ys = np.linspace(5.0, -5.0, G)
  out = np.empty((B, 1, G, G), dtype=np.float32)
  for b in range(B):
    dx = xs[None, None, :] - c[b, :, 0, None, None]; dy = ys[None, :, None] - c[b, :, 1, None, None]
    out[b, 0] = (np.sqrt(dx * dx + dy * dy) - r[b, :, None, None]).min(0)
  return out


def test_hip_c2_full_size_per_sample_sdf(be):
  """BASELINE config 2 in the reference's API shape, sdfb (B,1,H,W) with one DISTINCT 256x256 grid per trajectory (B = 4096:
  1 GiB of fp32 grids): every trajectory against oracle/gn_blocktri.c, a random subset against the dense numpy restatement,
  and against the same trajectories run one grid at a time as a shared grid (the two SDF addressing modes must agree bit
  for bit)."""
  from oracle import blocktri as BT
  B, n, G = 4096, 64, 256
  p = PC.P2d(n)
  th, start, goal, _ = _c2_inputs(B, n, G, seed=5, perturb=0.05)
  th, start, goal = PC.rnd(th, 'f32'), PC.rnd(start, 'f32'), PC.rnd(goal, 'f32')
  sdf32 = _per_sample_circle_sdfs(B, G)                      # float32: exactly what the kernel reads
  dth, err, eex, info = be.step(p, th, start, goal, sdf32, io='f32')
  assert np.all(info == 0) and np.all(np.isfinite(dth))
  c_dth = np.empty_like(dth); c_err = np.empty(B); c_eex = np.empty(B)
  for lo in range(0, B, 512):                                # the C oracle takes fp64 grids: 512 x 256 x 256 x 8 B = 256 MiB per chunk
    sl = slice(lo, lo + 512)
    c_dth[sl], c_err[sl], c_eex[sl], ci = BT.gn_step(p, th[sl], start[sl], goal[sl], sdf32[sl].astype(np.float64), nthreads=4)
    assert not ci.any()
  per_traj = np.abs(dth - c_dth).reshape(B, -1).max(1) / np.abs(c_dth).reshape(B, -1).max(1)
  assert per_traj.max() < PC.TOL['f32'], per_traj.max()
  assert rel_err(err, c_err) < PC.TOL_ERR['f32'] and rel_err(eex, c_eex) < PC.TOL_ERR['f32']
  idx = np.random.RandomState(2).choice(B, 32, replace=False)
  qc, ow, eps = p.static_covs(32)
  r_dth, r_err, r_eex = O.plan_layer_forward(th[idx], start[idx], goal[idx], sdf32[idx].astype(np.float64), qc, ow, eps, p)
  assert rel_err(dth[idx], r_dth) < PC.TOL['f32'] and rel_err(err[idx], r_err.reshape(-1)) < PC.TOL_ERR['f32']
  for b in idx[:6]:                                          # shared-grid addressing of the same trajectory: identical bits
    d1, e1, _, _ = be.step(p, th[b:b + 1], start[b:b + 1], goal[b:b + 1], sdf32[b:b + 1], io='f32')
    assert np.array_equal(d1[0], dth[b]) and e1[0] == err[b]


def test_hip_stress_seed_short():
  """One short seed of tests/stress_random_configs.py (random lengths, batches, robots, factor flags, SDF shapes, covariance modes,
  launch shapes, I/O types; forward vs oracle/gn_blocktri.c with the extended-precision arbiter, fused loop vs chained steps, backward vs
  the autograd oracle), so that the round-end GPU run exercises it; the long seeds are run by hand (profiles/)."""
  import os, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ); env.pop('DGP_FORCE_SHAPE', None)
  r = subprocess.run([sys.executable, os.path.join(root, 'tests', 'stress_random_configs.py'), '--cases', '40', '--seed', '12345', '--maxB', '300'],
                     capture_output=True, text=True, env=env, timeout=900)
  assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
  assert ' 0 failed' in r.stdout.splitlines()[-1], r.stdout[-500:]
