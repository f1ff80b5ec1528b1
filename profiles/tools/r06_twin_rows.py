"""Round 6: WHERE is the wrong twin wrong?  Per-state error of dtheta (twin kernel vs C oracle) for one trajectory: rows are printed lane by lane (C = 4 states per lane).
  DGP_LIB_PATH=... python profiles/tools/r06_twin_rows.py [d4general|d6static] [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import harness, parity_cases as PC, test_hip_every_kernel as T
from oracle import blocktri as BT
case = sys.argv[1] if len(sys.argv) > 1 else 'd4general'
dof, io, cov, n0 = {'d6static': (3, 'f32', 'static_diag', 64), 'd4general': (2, 'f32', 'static_full', 128)}[case]
n = int(sys.argv[2]) if len(sys.argv) > 2 else n0
be = harness.Backend('hip')
rs = np.random.RandomState(7)
B = 6
p, th, start, goal, sdf, qc, ow, eps, q_full = T._inputs(rs, dof, n, B, cov, io)
kw = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)
c_dth, c_err, c_eex, _ = BT.gn_step(p, th, start, goal, sdf, qc=qc, q_full=q_full)
fw = be.step_errors(p, th, start, goal, sdf, **kw)
d = fw[0]
scale = np.abs(c_dth).max()
print('err     twin %s oracle %s' % (np.array2string(fw[1], precision=6), np.array2string(c_err.reshape(-1), precision=6)))
print('err_ext twin %s oracle %s' % (np.array2string(fw[2], precision=6), np.array2string(c_eex.reshape(-1), precision=6)))
for b in range(B):
  e = np.abs(d[b] - c_dth[b]).max(axis=1) / scale          # per state
  print('trajectory %d: max rel err %.2e; per lane (4 states each):' % (b, e.max()))
  for l in range(0, n, 16):
    print('   states %3d..%3d  ' % (l, min(n, l + 16) - 1) + ' '.join('%7.0e' % v if v > 1e-5 else '   .   ' for v in e[l:l + 16]))
