"""Round 6: stand-alone reproducer of the two step-errors twins that come out wrong on the GPU.

  DGP_LIB_PATH=dgpmp2_amd/lib/libdgpmp2_dev_<variant>.so python profiles/tools/r06_twin_repro.py [d6static|d4general] [repeats]

The library must be a devbuild with -DDGP_TWIN_REPRO=1, --raw: the unrepaired compiler output (profiles/tools/r06_twin_repro.sh builds the variants: optimisation levels, -mllvm switches).
For the case it runs dgp_gn_step_errors (ONE launch: the twin kernel with the errors epilogue) against dgp_gn_step (the standard kernel, pinned to the C oracle by
tests/test_hip_every_kernel.py) and the C oracle itself, `repeats` times on the same inputs, and prints one line per (length, repeat):
  rel. error of the twin's dtheta vs the standard kernel, vs the oracle, and whether two runs of the twin agree bit for bit (determinism).
A second pass calls the twin kernel WITHOUT asking for the errors (the epilogue is compiled in but skipped at run time: DGP_TWIN_NO_ERRS=1 routes dgp_gn_step through the twin unit)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import harness, parity_cases as PC, test_hip_every_kernel as T
from oracle import blocktri as BT

case = sys.argv[1] if len(sys.argv) > 1 else 'd6static'
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dof, io, cov, lengths = {'d6static': (3, 'f32', 'static_diag', (64, 37)), 'd4general': (2, 'f32', 'static_full', (128, 100)),
                         'd6static64': (3, 'f64', 'static_diag', (64, 37)), 'd4qfull': (2, 'f32', 'qfull', (128, 100))}[case]
be = harness.Backend('hip')
rs = np.random.RandomState(7)
worst = 0.0
for n in lengths:
  B = 6
  p, th, start, goal, sdf, qc, ow, eps, q_full = T._inputs(rs, dof, n, B, cov, io)
  kw = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
  sh = (B, n, 1, 1)
  okw = dict(qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh), q_full=q_full)
  c_dth = BT.gn_step(p, th, start, goal, sdf, **okw)[0]
  try: d_std = be.step(p, th, start, goal, sdf, **kw)[0]      # (the standard kernel, when its unit is in the library)
  except Exception: d_std = c_dth
  first = None
  for r in range(reps):
    fw = be.step_errors(p, th, start, goal, sdf, **kw)
    d_tw = fw[0]
    e_std = PC.rel_err(d_tw, d_std) if np.all(np.isfinite(d_tw)) else np.inf
    e_orc = PC.rel_err(d_tw, c_dth) if np.all(np.isfinite(d_tw)) else np.inf
    same = True if first is None else bool(np.array_equal(first, d_tw))
    if first is None: first = d_tw.copy()
    worst = max(worst, e_std)
    bad_traj = [int(b) for b in range(B) if not PC.rel_err(d_tw[b], d_std[b]) < 1e-4]
    print('%s n %3d rep %d: twin vs standard %.3e  twin vs oracle %.3e  standard vs oracle %.3e  same bits as rep 0: %s  wrong trajectories %s' % (
        case, n, r, e_std, e_orc, PC.rel_err(d_std, c_dth), same, bad_traj), flush=True)
print('RESULT %s %s worst %.3e -> %s' % (os.path.basename(os.environ.get('DGP_LIB_PATH', 'product')), case, worst, 'WRONG' if not worst < 1e-4 else 'ok'))
