"""targeted check of the general-covariance kernels (every shape, static_full and qfull) against the C oracle: the every-kernel test restricted to unit <dof>_<io>_g1"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import harness, parity_cases as PC, test_hip_every_kernel as T
from oracle import blocktri as BT
dof, io = int(sys.argv[1]), sys.argv[2]
be = harness.Backend('hip')
rs = np.random.RandomState(100 * dof + (io == 'f32'))
bad = []
for lpt, c in T.SHAPES:
  os.environ['DGP_FORCE_SHAPE'] = '%d,%d' % (lpt, c)
  for cov in T.COVS:
    for n in (lpt * c, max(2, lpt * c - 3)):
      B = 64 // lpt + 1
      p, th, start, goal, sdf, qc, ow, eps, q_full = T._inputs(rs, dof, n, B, cov, io)      # (same random stream as the test)
      if cov not in sys.argv[3].split(','): continue
      sh = (B, n, 1, 1)
      okw = dict(qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh), q_full=q_full)
      dth, err, eex, info = be.step(p, th, start, goal, sdf, qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
      c_dth, c_err, c_eex, c_info = BT.gn_step(p, th, start, goal, sdf, **okw)
      e = PC.rel_err_per_traj(dth, c_dth) if np.all(np.isfinite(dth)) else np.inf
      if not e < PC.TOL[io]: bad.append(('(%d,%d) n %d %s' % (lpt, c, n, cov), e))
print(os.environ.get('DGP_LIB_PATH'), 'bad:', bad)
