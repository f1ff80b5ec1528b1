"""GPU (-m gpu): EVERY kernel instantiation of the library.  Forward: -- 2 robots x 2 I/O types x 9 launch shapes x 4 covariance
representations (static diagonal Q_c_inv: block elimination or Woodbury; static non-diagonal: general; per-state tensors: Kronecker;
q_full: general) x {single step, fused loop} -- on a small batch against oracle/gn_blocktri.c, with the trajectory length that fills
the shape exactly (n = LPT * C: full-line row I/O, the exact-fit Woodbury kernels) and a ragged one (padding rows, scalar row I/O, the
ragged Woodbury kernels).  Backward: every instantiation against an INDEPENDENT gradient oracle -- torch autograd over the dense
restatement of the reference's step (oracle/autograd_torch.py, itself pinned to the reference's autograd fixtures) -- second test.

Why this exists: hipcc has miscompiled several of the largest d = 6 kernels (wrong results or wild stores, while the same source
is exact on the CPU wavefront emulator and in every other instantiation).  Which instantiation breaks changes with
unrelated edits, so the sampled stress run (tests/stress_random_configs.py) is not enough: this test pins every one, every round.
A failure names the kernel's template arguments."""
import os
import numpy as np
import pytest
import harness
import parity_cases as PC
from oracle import gpmp2_oracle as O, blocktri as BT

pytestmark = pytest.mark.gpu

# Gradients of the fp32-I/O kernels against the fp64 ORACLES: the backward takes the forward's dtheta as an input, rounded to fp32, and the oracle its own fp64 one -- cond(Lambda) x 6e-8,
# measured up to 1e-3 (round 6, DGP_F32_GRAD_SCALE=0.002: gpurun_out/r06d/tight.txt).  These bounds cannot be sharper; test_hip_every_f32_kernel_matches_its_f64_sibling is.
F32_GRAD_SCALE = float(os.environ.get('DGP_F32_GRAD_SCALE', 1.0))
SHAPES = [(l, c) for l in (16, 32, 64) for c in (1, 2, 4)]
COVS = ['static', 'static_full', 'perstate', 'qfull']
ERRS_LENGTHS = (64, 37, 128, 100)      # test_hip_every_step_errors_kernel: the 16- and the 32-lane shape, exact fit and ragged


@pytest.fixture(scope='module')
def be():
  import torch
  assert torch.cuda.is_available(), 'the -m gpu tests need a GPU'
  return harness.Backend('hip')


def _inputs(rs, dof, n, B, cov, io):
  d = 2 * dof
  kw = {}
  if dof == 3: kw.update(non_holonomic=True, K_d=0.1)
  if cov == 'static_full':
    A = rs.randn(dof, dof) * 0.3
    kw['Q_c_inv'] = np.eye(dof) + A @ A.T
  if cov == 'static_diag': kw['Q_c_inv'] = np.diag(1.0 + 0.5 * np.arange(dof))      # diagonal, not c I: the block-elimination static kernels, no Woodbury
  p = O.OracleParams(dof=dof, total_time_step=n - 1, reg=0.1, epsilon_dist=0.3, **kw)
  H, W = 24, 40
  yy, xx = np.meshgrid(np.linspace(5, -5, H), np.linspace(-5, 5, W), indexing='ij')
  cs = rs.uniform(-3, 3, (3, 2)); rr = rs.uniform(0.4, 1.2, 3)
  sdf = np.min(np.sqrt((xx[None] - cs[:, 0, None, None]) ** 2 + (yy[None] - cs[:, 1, None, None]) ** 2) - rr[:, None, None], axis=0)[None, None]
  start = np.zeros((B, 1, d)); goal = np.zeros((B, 1, d))
  start[:, 0, :2] = rs.uniform(-4, 4, (B, 2)); goal[:, 0, :2] = rs.uniform(-4, 4, (B, 2))
  if dof == 3: goal[:, 0, 2] = rs.uniform(-np.pi, np.pi, B)
  th = O.straight_line_trajb(start[:, :, :dof], goal[:, :, :dof], 10.0, n - 1, dof) + rs.randn(B, n, d) * 0.03
  qc = ow = eps = None; q_full = False
  if cov in ('perstate', 'qfull'):
    ow = rs.uniform(50, 2e4, (B, n)); eps = rs.uniform(0.1, 0.6, (B, n))
    if cov == 'perstate':
      A = rs.randn(B, n - 1, dof, dof) * 0.2; qc = np.eye(dof) + A @ np.swapaxes(A, -1, -2)
    else:
      A = rs.randn(B, n - 1, d, d) * 0.2; qc = (np.eye(d) + A @ np.swapaxes(A, -1, -2)) * 1.5; q_full = True
  r = lambda a: None if a is None else PC.rnd(a, io)
  return p, r(th), r(start), r(goal), r(sdf), r(qc), r(ow), r(eps), q_full


@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('dof', [2, 3])
def test_hip_every_forward_kernel_vs_c_oracle(be, dof, io, monkeypatch):
  rs = np.random.RandomState(100 * dof + (io == 'f32'))
  bad = []
  for lpt, c in SHAPES:
    monkeypatch.setenv('DGP_FORCE_SHAPE', '%d,%d' % (lpt, c))
    for cov in COVS:
      for n in (lpt * c, max(2, lpt * c - 3)):
        B = 64 // lpt + 1                      # one full wavefront and a partially filled one
        p, th, start, goal, sdf, qc, ow, eps, q_full = _inputs(rs, dof, n, B, cov, io)
        sh = (B, n, 1, 1)
        okw = dict(qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh), q_full=q_full)
        kw = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
        tag = 'dof %d %s shape (%d,%d) n %d cov %s' % (dof, io, lpt, c, n, cov)
        # ---- single step
        dth, err, eex, info = be.step(p, th, start, goal, sdf, **kw)
        c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, **okw)
        e = PC.rel_err_per_traj(dth, c_dth) if np.all(np.isfinite(dth)) else np.inf
        ee = PC.rel_err(err, c_err) if np.all(np.isfinite(err)) else np.inf
        if not (e < PC.TOL[io] and ee < 10 * PC.TOL_ERR[io] and not info.any()): bad.append((tag, 'step', e, ee))
        # ---- fused loop: three iterations == three chained oracle steps (fp64 I/O; fp32 I/O is held to the fp32 tolerance)
        tho, its, eh, eeh, ef, sinfo = be.solve(p, th, start, goal, sdf, 3, 0.0, **kw)
        cur = th.copy()
        for k in range(3):
          d_k, e_k, _, _ = BT.gn_step(p, cur, start, goal, sdf, **okw)
          cur = cur + d_k
        es = PC.rel_err(tho, cur) if np.all(np.isfinite(tho)) else np.inf
        if not (es < (1e-7 if io == 'f64' else 1e-5) and not sinfo.any() and np.all(its == 3)): bad.append((tag, 'fused loop', es))
  assert not bad, '%d kernel instantiations differ from the C oracle:\n' % len(bad) + '\n'.join(map(str, bad))


@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('dof', [2, 3])
def test_hip_every_scaled_kernel_vs_c_oracle(be, dof, io, monkeypatch):
  """The 36 QK_SCALED step kernels and the 36 backward kernels (DGP_QC_SCALAR: one scalar per GP factor, the learned mode diag_identity): every launch shape, a length that fills it
  and a ragged one, per-state obstacle weights / epsilons next to the scalars -- against the C oracle on the dense tensors s_k I."""
  rs = np.random.RandomState(300 * dof + (io == 'f32'))
  bad = []
  for lpt, c in SHAPES:
    monkeypatch.setenv('DGP_FORCE_SHAPE', '%d,%d' % (lpt, c))
    for n in (lpt * c, max(2, lpt * c - 3)):
      B = 64 // lpt + 1
      p, th, start, goal, sdf, _, ow, eps, _ = _inputs(rs, dof, n, B, 'perstate', io)
      s_ = PC.rnd(rs.uniform(0.3, 3.0, (B, n - 1)) ** 2, io)
      sh = (B, n, 1, 1)
      dth, err, eex, info = be.step(p, th, start, goal, sdf, qc=s_, ow=ow, eps=eps, io=io)
      dense = s_[:, :, None, None] * np.eye(dof)
      c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, qc=dense, ow=ow.reshape(sh), eps=eps.reshape(sh))
      e = PC.rel_err_per_traj(dth, c_dth) if np.all(np.isfinite(dth)) else np.inf
      ee = PC.rel_err(err, c_err) if np.all(np.isfinite(err)) else np.inf
      ex = PC.rel_err(eex, c_eex) if np.all(np.isfinite(eex)) else np.inf
      if not (e < PC.TOL[io] and ee < 10 * PC.TOL_ERR[io] and ex < 10 * PC.TOL_ERR[io] and not info.any()):
        bad.append(('dof %d %s shape (%d,%d) n %d' % (dof, io, lpt, c, n), e, ee, ex))
      # ... and the 36 backward kernels of the same variant against the Kronecker kernels on the dense blocks (themselves pinned to the autograd oracle above)
      gb = PC.rnd(rs.randn(B, n, 2 * dof), io); ge = PC.rnd(rs.randn(B), io)
      # (fp32 I/O: the grid gradient accumulated in FLOAT64 grids, DGP_GSDF_DENSE_F64 -- what PlanLayer uses for a shared grid since round 5: the sum no longer depends on
      #  the order the fp32 atomics land in, so it is compared like every other gradient -- rounds 3-4 skipped it)
      gm = 'f64' if io == 'f32' else 'dense'
      r1 = be.backward(p, th, start, goal, sdf, dth, gb, ge, qc=s_, ow=ow, eps=eps, io=io, sdf_grad=gm)
      r2 = be.backward(p, th, start, goal, sdf, dth, gb, ge, qc=PC.rnd(dense, io), ow=ow, eps=eps, io=io, sdf_grad=gm)
      for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'):
        # (the grid gradient is a sum of signed tap contributions accumulated by atomics: judged against the size of the summands, for which the trajectory gradient stands in)
        scale = max(np.abs(r2[key]).max(), np.abs(r2['th']).max() if key == 'sdf' else 0.0, (1e-300 if io == 'f64' else 1e-6 * np.abs(r2['th']).max()))
        eb = np.abs(r1[key] - r2[key]).max() / scale if np.all(np.isfinite(r1[key])) else np.inf
        if not eb < (1e-7 if io == 'f64' else 5e-3 * F32_GRAD_SCALE): bad.append(('dof %d %s shape (%d,%d) n %d backward' % (dof, io, lpt, c, n), key, eb))
  assert not bad, '%d scaled-mask kernel instantiations differ from the C oracle:\n' % len(bad) + '\n'.join(map(str, bad))


@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('dof', [2, 3])
def test_hip_every_backward_kernel_vs_autograd_oracle(be, dof, io, monkeypatch):
  """Every BACKWARD kernel instantiation (2 robots x 2 I/O types x 9 shapes x static [block elimination / Woodbury, exact fit and ragged] /
  general / per-state) on one small batch each, every gradient tensor, against torch autograd over the dense restatement of the
  reference's step (oracle/autograd_torch.py: dense A, K, Cholesky + two explicit inverses; it shares nothing with the kernels or with
  tests/emul, and reproduces the reference's own autograd fixtures to 1e-12, tests/test_oracle_golden.py) -- so a mathematical error
  common to the kernel source and its host emulator cannot pass.  The non-holonomic robot is included (the reference cannot run it
  in batch; the oracle differentiates through H as torch would)."""
  from oracle import autograd_torch as AT
  rs = np.random.RandomState(200 * dof + (io == 'f32'))
  bad = []
  for lpt, c in SHAPES:
    monkeypatch.setenv('DGP_FORCE_SHAPE', '%d,%d' % (lpt, c))
    variants = [('static', lpt * c), ('static', max(4, lpt * c - 2)), ('static_full', lpt * c), ('perstate', lpt * c - 1), ('qfull', lpt * c)]
    if lpt * c >= 128: del variants[2]      # a non-diagonal static Q_c_inv and q_full tensors run the SAME (general) backward kernel: one dense autograd pass per long shape is enough
    for cov, n in variants:
      B = 2
      p, th, start, goal, sdf, qc, ow, eps, q_full = _inputs(rs, dof, n, B, cov, io)
      d = 2 * dof
      kw = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
      dth = be.step(p, th, start, goal, sdf, **kw)[0]
      gbar = PC.rnd(rs.randn(B, n, d), io); gext = PC.rnd(rs.randn(B), io)
      copies = 16 if (lpt + c) % 3 == 0 else 1
      g_h = be.backward(p, th, start, goal, sdf, PC.rnd(dth, io), gbar, gext, sdf_copies=copies, sdf_grad='f64' if io == 'f32' else 'dense', **kw)      # (fp32 I/O: float64 grids, see the scaled-kernel test)
      g_o = AT.step_gradients(p, th, start, goal, sdf, gbar, gext, qc=qc, ow=ow, eps=eps, q_full=q_full)
      tag = 'dof %d %s shape (%d,%d) n %d cov %s' % (dof, io, lpt, c, n, cov)
      if not PC.rel_err(dth, g_o['dtheta']) < PC.TOL[io]: bad.append((tag, 'dtheta', PC.rel_err(dth, g_o['dtheta'])))
      for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'):
        if g_h[key] is None: continue
        a_ = g_h[key]
        if key == 'sdf' and copies > 1: a_ = a_.sum(0, keepdims=True)
        b_ = g_o[key].reshape(a_.shape)
        if not np.all(np.isfinite(a_)): bad.append((tag, key, 'non-finite')); continue
        eb = np.abs(a_ - b_).max() / max(np.abs(b_).max(), np.abs(g_o['th']).max() if key == 'sdf' else 0.0, 1e-300)
        # f32 I/O: the kernels are handed the forward output dtheta ROUNDED to fp32 and rebuild rho = e - H dtheta from it, so the gradients
        # carry cond(Lambda) * 6e-8 relative to the oracle's (which differentiates its own fp64 dtheta): up to 4e-4 on the random q_full
        # systems here.  The fp64 run pins the mathematics at 1e-6; this one only has to catch code-generation faults (O(1) errors).
        if not eb < (1e-6 if io == 'f64' else 2e-3 * F32_GRAD_SCALE): bad.append((tag, key, eb))
  assert not bad, '%d backward results differ from the autograd oracle:\n' % len(bad) + '\n'.join(map(str, bad))


@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('dof', [2, 3])
def test_hip_every_step_errors_kernel(be, dof, io, monkeypatch):
  """Round 5: the step kernels with the errors epilogue (gn_inst.hip compiled with -DDGP_STEP_ERRS=1: 2 robots x 2 I/O types x the two four-states-per-lane shapes x
  Woodbury exact / ragged, block elimination, scaled, Kronecker, general -- dgp_gn_step_errors as ONE launch) against the two launches they replace (the standard step kernel and
  the error kernel, themselves pinned to the C oracle above), and the errors' backward as the PROLOGUE of every backward kernel shape that holds the trajectory against
  its two halves run by hand (dgp_eval_errors_backward at th + dtheta, then dgp_gn_step_backward with that gradient joined to the dtheta cotangent)."""
  rs = np.random.RandomState(500 * dof + (io == 'f32'))
  npdt = np.float64 if io == 'f64' else np.float32
  bad = []
  for n in ERRS_LENGTHS:
    for cov in ('static', 'static_diag', 'scalar', 'perstate', 'static_full', 'qfull'):      # (the last two: the general family -- twins for d = 4, two launches for d = 6)
      monkeypatch.delenv('DGP_FORCE_SHAPE', raising=False)
      B = 6
      p, th, start, goal, sdf, qc, ow, eps, q_full = _inputs(rs, dof, n, B, 'perstate' if cov == 'scalar' else cov, io)
      if cov == 'scalar': qc = PC.rnd(rs.uniform(0.3, 3.0, (B, n - 1)) ** 2, io)
      kw = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
      tag = 'dof %d %s n %d cov %s' % (dof, io, n, cov)
      fw = be.step_errors(p, th, start, goal, sdf, **kw)
      d2, e2, x2, i2 = be.step(p, th, start, goal, sdf, **kw)
      th_new = (th.astype(npdt) + d2.astype(npdt)).astype(np.float64)
      _, _, s2, g2, o2 = be.eval_errors(p, th_new, start, goal, sdf, eps=eps, io=io)
      tol = 1e-10 if io == 'f64' else 2e-5
      for name, a_, b_ in (('dtheta', fw[0], d2), ('err', fw[1], e2), ('err_ext', fw[2], x2), ('unw_sg', fw[4], s2), ('unw_gp', fw[5], g2), ('unw_obs', fw[6], o2)):
        e = PC.rel_err(a_, b_) if np.all(np.isfinite(a_)) else np.inf
        if not e < tol: bad.append((tag, name, e))
      if fw[3].any(): bad.append((tag, 'info'))
      # ---- the backward prologue, every shape that holds n states
      gd = PC.rnd(rs.randn(B, n, 2 * dof), io); ce = PC.rnd(rs.randn(B), io)
      cs, cg, co = (PC.rnd(rs.randn(B), io) for _ in range(3))
      h1 = be.eval_backward(p, th_new, start, goal, sdf, None, cs, cg, co, eps=eps, io=io)
      h2 = be.backward(p, th, start, goal, sdf, d2, (gd.astype(npdt) + h1['th'].astype(npdt)).astype(np.float64), ce, **kw)
      want = dict(th=h2['th'] + h1['th'], start=h2['start'] + h1['start'], goal=h2['goal'] + h1['goal'], qc=h2['qc'], ow=h2['ow'],
                  eps=None if eps is None else h2['eps'] + h1['eps'])
      for lpt, c in [(None, None)] + [sh for sh in SHAPES if sh[0] * sh[1] >= n]:
        if lpt is not None: monkeypatch.setenv('DGP_FORCE_SHAPE', '%d,%d' % (lpt, c))
        if os.environ.get('DGP_TEST_VERBOSE'): print(tag, 'backward shape', (lpt, c), flush=True)      # (a wild store aborts the process: run with -s to see where)
        r = be.step_errors_backward(p, th, start, goal, sdf, fw[0], gd, ce, cs, cg, co, sdf_grad='none', **kw)
        for key in ('th', 'start', 'goal', 'qc', 'ow', 'eps'):
          if want[key] is None or r[key] is None: continue
          eb = np.abs(r[key] - want[key]).max() / max(np.abs(want[key]).max(), 1e-300) if np.all(np.isfinite(r[key])) else np.inf
          if not eb < (1e-7 if io == 'f64' else 2e-3 * F32_GRAD_SCALE): bad.append((tag, 'backward shape %s' % ((lpt, c),), key, eb))
  assert not bad, '%d step-errors results differ:\n' % len(bad) + '\n'.join(map(str, bad))


@pytest.mark.parametrize('dof', [2, 3])
def test_hip_every_f32_kernel_matches_its_f64_sibling(be, dof, monkeypatch):
  """Round 6: the fp32-I/O kernels against the fp64-I/O kernels of the same family, shape and length on the SAME numbers (inputs that are exact in fp32).  Both run the same fp64
  lane program -- separate compilations of it -- so they agree to the rounding of the fp32 outputs (a few 1e-8 of the largest entry), whatever the conditioning of the system:
  the oracle comparisons above must allow the fp32 kernels 1e-5 (step) and 5e-3 (gradients: the fp32 rounding of dtheta, an INPUT of the backward, times the condition number),
  which is wide enough for a kernel that is wrong in a few per cent of one output (profiles/r06_body_tail_diff.txt).  The fp64 siblings are held to 1e-9 / 1e-7 above, so this
  pins every fp32 instantiation -- step, step + errors, backward, backward with error cotangents, their tiled twins, the fused loop, the traced loop and the chain backward -- at 1e-6."""
  rs = np.random.RandomState(900 + dof)
  bt = harness.Backend(be.kind); bt.sdf_tiled = True
  bad = []
  tol = float(os.environ.get('DGP_SIBLING_TOL', 1e-6))      # (the override shows what the agreement really is: at 1e-9 the list holds fp32 output roundings, 5.6e-8 at most -- profiles/r06_f32_sibling_agreement.txt)

  def cmp(tag, what, a32, a64, t=None, scale_with=None):
    if a32 is None or a64 is None: return
    if not np.all(np.isfinite(a32)): bad.append((tag, what, 'non-finite')); return
    a32 = a32.astype(np.float64)
    if a32.shape != a64.shape and a32.shape[1:] == a64.shape[1:]: a32 = a32.sum(0, keepdims=True)      # (the float64 partial grids of a shared grid's gradient, unsummed)
    e = np.abs(a32 - a64).max() / max(np.abs(a64).max(), 0.0 if scale_with is None else np.abs(scale_with).max(), 1e-300)
    if not e < (t or tol): bad.append((tag, what, e))
  up = lambda a: None if a is None else np.asarray(a, dtype=np.float64)
  for lpt, c in SHAPES:
    monkeypatch.setenv('DGP_FORCE_SHAPE', '%d,%d' % (lpt, c))
    for cov in ('static', 'static_diag', 'static_full', 'perstate', 'qfull', 'scalar'):
      for n in (lpt * c, max(4, lpt * c - 3)):
        B = 64 // lpt + 1
        p, th, start, goal, sdf, qc, ow, eps, q_full = _inputs(rs, dof, n, B, 'perstate' if cov == 'scalar' else cov, 'f32')
        if cov == 'scalar': qc = PC.rnd(rs.uniform(0.3, 3.0, (B, n - 1)) ** 2, 'f32')
        tag = 'dof %d shape (%d,%d) n %d cov %s' % (dof, lpt, c, n, cov)
        k32 = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io='f32')
        k64 = dict(qc=up(qc), ow=up(ow), eps=up(eps), q_full=q_full, io='f64')
        a64 = (up(th), up(start), up(goal), up(sdf))
        for b_, kind in ((be, ''),) + (((bt, ' [tiled]'),) if c == 4 and n <= 128 else ()):
          f32 = b_.step(p, th, start, goal, sdf, **k32); f64 = b_.step(p, *a64, **k64)
          if f32[3].any() or f64[3].any(): bad.append((tag + kind, 'info')); continue
          for i, name in enumerate(('dtheta', 'err', 'err_ext')): cmp(tag + kind, 'step ' + name, f32[i], f64[i])
          dth = f32[0]                       # (fp32-exact: the same dtheta goes into both backward kernels)
          gb = PC.rnd(rs.randn(B, n, 2 * dof), 'f32'); ge = PC.rnd(rs.randn(B), 'f32')
          r32 = b_.backward(p, th, start, goal, sdf, dth, gb, ge, sdf_grad='f64', **k32)
          r64 = b_.backward(p, *a64, up(dth), up(gb), up(ge), sdf_grad='dense', **k64)
          for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'): cmp(tag + kind, 'backward ' + key, r32[key], r64[key], scale_with=r64['th'] if key == 'sdf' else None)
          if cov != 'scalar':
            s32 = b_.solve(p, th, start, goal, sdf, 2, 0.0, **k32); s64 = b_.solve(p, *a64, 2, 0.0, **k64)
            cmp(tag + kind, 'fused loop', s32[0], s64[0])      # (two iterations: the trajectory between them is rounded to fp32 in one kernel only -- a difference of 6e-8 going in)
        if cov in ('static', 'static_diag', 'static_full'):      # the traced fused loop and its backward (chain kernels; the history is float64 whatever the I/O type)
          for b_, kind in ((be, ''),) + (((bt, ' [tiled]'),) if c == 4 and n <= 128 else ()):
            t32 = b_.solve_traced(p, th, start, goal, sdf, 3, 0.0, io='f32'); t64 = b_.solve_traced(p, *a64, 3, 0.0, io='f64')
            cmp(tag + kind, 'traced loop', t32[0], t64[0])
            gb = PC.rnd(rs.randn(B, n, 2 * dof), 'f32')
            c32 = b_.solve_backward(p, start, goal, sdf, 3, t32[2], t32[0], t32[1], gb, io='f32', sdf_grad='f64')
            c64 = b_.solve_backward(p, up(start), up(goal), up(sdf), 3, t32[2], up(t32[0]), t32[1], up(gb), io='f64', sdf_grad='dense')      # (the SAME history and final trajectory)
            for key in ('th', 'start', 'goal', 'sdf'): cmp(tag + kind, 'chain backward ' + key, c32[key], c64[key], scale_with=c64['th'] if key == 'sdf' else None)
        if n <= 128 and c == 4 and cov != 'scalar':      # the step-errors twins and the backward with the errors' cotangents
          ce, cs, cg, co = (PC.rnd(rs.randn(B), 'f32') for _ in range(4))
          w32 = be.step_errors(p, th, start, goal, sdf, **k32); w64 = be.step_errors(p, *a64, **k64)
          # (the errors are taken at th + dtheta SUMMED IN THE I/O TYPE, as torch forms th_curr_b + dthetab: the fp32 kernel's point is 6e-8 away from the fp64 kernel's, and the
          #  start / goal error -- a difference of nearly equal numbers at a trajectory that starts where it should -- moves by up to 4e-4 of itself)
          for i, name in ((0, 'dtheta'), (4, 'unw_sg'), (5, 'unw_gp'), (6, 'unw_obs')): cmp(tag, 'step_errors ' + name, w32[i], w64[i], t=2e-3 if name == 'unw_sg' else None)
          gd = PC.rnd(rs.randn(B, n, 2 * dof), 'f32')
          q32 = be.step_errors_backward(p, th, start, goal, sdf, w32[0], gd, ce, cs, cg, co, sdf_grad='none', **k32)
          q64 = be.step_errors_backward(p, *a64, up(w32[0]), up(gd), up(ce), up(cs), up(cg), up(co), sdf_grad='none', **k64)
          for key in ('th', 'start', 'goal', 'qc', 'ow', 'eps'): cmp(tag, 'step_errors backward ' + key, q32[key], q64[key])
  # the loop kernels of longer trajectories (gn_long.h: 257 ... 1024 / 640 states)
  monkeypatch.delenv('DGP_FORCE_SHAPE', raising=False)
  for n in (257, 384, 1024 if dof == 2 else 640):
    for cov in ('static', 'static_full', 'perstate', 'qfull'):
      B = 3
      p, th, start, goal, sdf, qc, ow, eps, q_full = _inputs(rs, dof, n, B, cov, 'f32')
      tag = 'dof %d n %d cov %s (loop kernels)' % (dof, n, cov)
      k32 = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io='f32')
      k64 = dict(qc=up(qc), ow=up(ow), eps=up(eps), q_full=q_full, io='f64')
      a64 = (up(th), up(start), up(goal), up(sdf))
      f32 = be.step(p, th, start, goal, sdf, **k32); f64 = be.step(p, *a64, **k64)
      if f32[3].any() or f64[3].any(): bad.append((tag, 'info')); continue
      for i, name in enumerate(('dtheta', 'err', 'err_ext')): cmp(tag, 'step ' + name, f32[i], f64[i])
      gb = PC.rnd(rs.randn(B, n, 2 * dof), 'f32'); ge = PC.rnd(rs.randn(B), 'f32')
      r32 = be.backward(p, th, start, goal, sdf, f32[0], gb, ge, sdf_grad='dense', **k32)      # (no float64 partial grids beyond 256 states: the fp32 grid gradient is summed by fp32
      r64 = be.backward(p, *a64, up(f32[0]), up(gb), up(ge), sdf_grad='dense', **k64)            #  atomics in an order that changes from run to run -- left out here)
      for key in ('th', 'start', 'goal', 'qc', 'ow', 'eps'): cmp(tag, 'backward ' + key, r32[key], r64[key])
  assert not bad, '%d fp32 results differ from the fp64 sibling kernels:\n' % len(bad) + '\n'.join(map(str, bad[:60]))


@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('dof', [2, 3])
def test_hip_every_tiled_twin_kernel(be, dof, io, monkeypatch):
  """Round 5: the tiled-grid twins (gn_inst.hip compiled with -DDGP_TL=1: every kernel family in the two four-states-per-lane shapes, 2 robots x 2 I/O types -- step,
  fused loop, error kernel, backward with the grid gradient, chain backward) against the standard kernels on the same grids stored row-major (each of which the tests
  above hold against an oracle).  A separate compilation of the same source: equal to rounding, and a miscompiled twin is wrong by O(1).  Per-sample grids (odd-sized:
  padding cells of the last tile row / column) and a shared one."""
  rs = np.random.RandomState(700 * dof + (io == 'f32'))
  bt = harness.Backend(be.kind); bt.sdf_tiled = True
  monkeypatch.delenv('DGP_FORCE_SHAPE', raising=False)
  bad = []
  tol = 1e-9 if io == 'f64' else 2e-4
  # gradients / fused loops: twin and standard kernel run the same fp64 arithmetic on the same numbers (only the grid's storage order differs) -- measured agreement 6e-10
  # (profiles/r06_body_tail_diff.txt).  Round 6 tightened this from 100 x tol = 2e-2 for fp32 I/O: a kernel with a 3 % error in g_th had passed under the old bound.
  tight = 1e-7 if io == 'f64' else 2e-5
  K = 3

  def cmp(tag, what, a_, b_, scale_with=None, t=None):
    if a_ is None or b_ is None: return
    if not np.all(np.isfinite(a_)): bad.append((tag, what, 'non-finite')); return
    e = np.abs(a_ - b_).max() / max(np.abs(b_).max(), 0.0 if scale_with is None else np.abs(scale_with).max(), 1e-300)
    if not e < (t or tol): bad.append((tag, what, e))
  for n in ERRS_LENGTHS:
    for cov in ('static', 'static_diag', 'static_full', 'scalar', 'perstate', 'qfull'):
      B = 5
      p, th, start, goal, sdf, qc, ow, eps, q_full = _inputs(rs, dof, n, B, 'perstate' if cov == 'scalar' else cov, io)
      if cov == 'scalar': qc = PC.rnd(rs.uniform(0.3, 3.0, (B, n - 1)) ** 2, io)
      if (n + len(cov)) % 2:      # every other configuration: one grid per trajectory, 23 x 37 (neither a multiple of four)
        sdf = PC.rnd(np.stack([sdf[0, :, :23, :37] + 0.05 * rs.randn() for _ in range(B)]), io)
      kw = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
      tag = 'dof %d %s n %d cov %s%s' % (dof, io, n, cov, ' per-sample grids' if sdf.shape[0] > 1 else '')
      a = be.step(p, th, start, goal, sdf, **kw); b = bt.step(p, th, start, goal, sdf, **kw)
      for i, name in enumerate(('dtheta', 'err', 'err_ext')): cmp(tag, 'step ' + name, b[i], a[i])
      if b[3].any() or a[3].any(): bad.append((tag, 'info')); continue
      ea = be.eval_errors(p, th, start, goal, sdf, eps=eps, io=io); eb = bt.eval_errors(p, th, start, goal, sdf, eps=eps, io=io)
      for i, name in enumerate(('err', 'err_ext', 'unw_sg', 'unw_gp', 'unw_obs')): cmp(tag, 'errors ' + name, eb[i], ea[i])
      gb = PC.rnd(rs.randn(B, n, 2 * dof), io); ge = PC.rnd(rs.randn(B), io)
      gm = 'f64' if io == 'f32' else 'dense'
      bkw = dict(kw)
      if cov == 'scalar': bkw['qc'] = PC.rnd(qc[:, :, None, None] * np.eye(dof), io)      # (the scaled backward twin through DGP_QC_SCALAR as well)
      for kq in ((kw, 'backward'),) + (((bkw, 'backward (dense blocks)'),) if cov == 'scalar' else ()):
        ra = be.backward(p, th, start, goal, sdf, a[0], gb, ge, sdf_grad=gm, **kq[0]); rb = bt.backward(p, th, start, goal, sdf, a[0], gb, ge, sdf_grad=gm, **kq[0])
        for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'): cmp(tag, kq[1] + ' ' + key, rb[key], ra[key], scale_with=ra['th'] if key == 'sdf' else None, t=tight)
      if cov != 'scalar':      # the fused loop (DGP_QC_SCALAR is a step-only mode)
        sa = be.solve(p, th, start, goal, sdf, K, 0.0, **kw); sb = bt.solve(p, th, start, goal, sdf, K, 0.0, **kw)
        cmp(tag, 'fused loop', sb[0], sa[0], t=tight)
      if cov in ('static', 'static_diag', 'static_full'):      # ... and its backward (static_full: the general-covariance chain twins, round 6)
        tho, its, hist, info = be.solve_traced(p, th, start, goal, sdf, K, 0.0, io=io)
        tht, itt, hist_t, info_t = bt.solve_traced(p, th, start, goal, sdf, K, 0.0, io=io)
        cmp(tag, 'traced loop', tht, tho, t=tight)
        ca = be.solve_backward(p, start, goal, sdf, K, hist, tho, its, gb, io=io, sdf_grad=gm); cb = bt.solve_backward(p, start, goal, sdf, K, hist, tho, its, gb, io=io, sdf_grad=gm)
        for key in ('th', 'start', 'goal', 'sdf'): cmp(tag, 'chain backward ' + key, cb[key], ca[key], scale_with=ca['th'] if key == 'sdf' else None, t=tight)
  assert not bad, '%d tiled-twin results differ from the row-major kernels:\n' % len(bad) + '\n'.join(map(str, bad))


@pytest.mark.parametrize('cov', ['static', 'static_full'])      # static_full (round 6): a non-diagonal Q_c_inv -- the general-covariance chain kernels
@pytest.mark.parametrize('io', ['f64', 'f32'])
@pytest.mark.parametrize('dof', [2, 3])
def test_hip_every_chain_backward_kernel(be, dof, io, cov, monkeypatch):
  """Every instantiation of the fused loop's backward (dgp_gn_solve_backward: 2 robots x 2 I/O types x 9 shapes x block elimination / Woodbury exact
  fit / Woodbury ragged) and of the traced fused loop in front of it: the history must reproduce the plain loop bit for bit, and the gradients
  must equal the chain of single-step backward launches (dgp_gn_step_backward, each instantiation of which the test above holds against the
  independent autograd oracle) walked by hand through that history.  Three iterations, the second trajectory stopping after the first
  (tol_delta between the two trajectories' first updates), the third batch element in a second, partially filled wavefront."""
  rs = np.random.RandomState(300 * dof + (io == 'f32') + 7 * (cov != 'static'))
  bad = []
  K = 3
  for lpt, c in SHAPES:
    monkeypatch.setenv('DGP_FORCE_SHAPE', '%d,%d' % (lpt, c))
    for n in (lpt * c, max(4, lpt * c - 2)):
      B = 64 // lpt + 1
      p, th, start, goal, sdf, _, _, _, _ = _inputs(rs, dof, n, B, cov, io)
      tag = 'dof %d %s %s shape (%d,%d) n %d' % (dof, io, cov, lpt, c, n)
      d0 = be.step(p, th, start, goal, sdf, io=io)[0]
      nrm = np.sqrt((d0.reshape(B, -1) ** 2).sum(1))
      order = np.argsort(nrm)
      tol = 0.5 * (nrm[order[0]] + nrm[order[1]]) if B > 1 else 0.0      # the trajectory with the smallest first update stops after ONE iteration
      tho, its, hist, info = be.solve_traced(p, th, start, goal, sdf, K, tol, io=io)
      ref = be.solve(p, th, start, goal, sdf, K, tol, io=io)
      if not (np.array_equal(tho, ref[0]) and np.array_equal(its, ref[1]) and not info.any()): bad.append((tag, 'traced loop differs from the plain loop')); continue
      if B > 1 and not (its.min() == 1 and its.max() > 1): bad.append((tag, 'iteration counts', its.tolist())); continue
      gbar = PC.rnd(rs.randn(B, n, 2 * dof), io)
      copies = 16 if (lpt + c) % 3 == 0 else 1
      gm = 'f64' if io == 'f32' else 'dense'      # (fp32 I/O: float64 grids, see the scaled-kernel test)
      r = be.solve_backward(p, start, goal, sdf, K, hist, tho, its, gbar, io=io, sdf_copies=copies, sdf_grad=gm)
      gcur = gbar.copy()
      acc = dict(start=np.zeros_like(start), goal=np.zeros_like(goal), sdf=np.zeros_like(sdf))
      for k in range(K - 1, -1, -1):
        on = its > k
        thk = np.where(on[:, None, None], np.nan_to_num(hist[k]), tho)
        nxt = np.where((its > k + 1)[:, None, None], np.nan_to_num(hist[min(k + 1, K - 1)]), tho)
        one = be.backward(p, thk, start, goal, sdf, nxt - thk, gcur * on[:, None, None], None, io='f64' if io == 'f64' else io, sdf_grad=gm)
        gcur = gcur + one['th'] * on[:, None, None]
        acc['start'] += one['start'] * on[:, None, None]; acc['goal'] += one['goal'] * on[:, None, None]; acc['sdf'] += one['sdf']
      for key, a_, b_ in (('th', r['th'], gcur), ('start', r['start'], acc['start']), ('goal', r['goal'], acc['goal']),
                          ('sdf', r['sdf'].sum(0, keepdims=True) if copies > 1 else r['sdf'], acc['sdf'])):
        if not np.all(np.isfinite(a_)): bad.append((tag, key, 'non-finite')); continue
        eb = np.abs(a_ - b_).max() / max(np.abs(b_).max(), np.abs(gcur).max() if key == 'sdf' else 0.0, 1e-300)
        # f32 I/O: the hand-walked chain rounds th_k, dtheta_k and the running cotangent to fp32 between the launches, the chain kernel keeps them in fp64
        # (the grid gradient is summed by atomics in an order that differs between the one launch and the K: 1e-7 of cancellation noise)
        # (static_full: the general kernels' PCR rounds use explicit block inverses, six rounds deep at 64 lanes -- 1.3e-9 met on <3,64,1,double>: rounding, not a fault)
        if not eb < ((1e-7 if key == 'sdf' else (5e-9 if cov == 'static_full' else 1e-9)) if io == 'f64' else 5e-3 * F32_GRAD_SCALE): bad.append((tag, key, eb))
  assert not bad, '%d chain-backward results differ from the chained single-step backward:\n' % len(bad) + '\n'.join(map(str, bad))
