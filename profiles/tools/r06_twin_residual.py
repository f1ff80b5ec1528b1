"""Round 6: WHAT did the wrong twin compute?  With Lambda, eta of the dense oracle (oracle/gpmp2_oracle.py) and the twin's dtheta x~:
  r = eta - Lambda x~ is the residual of every block row.  The separator rows' unknowns are right (r06_twin_rows.py), so r != 0 marks the equations the
  interior recovery violated.  If only the term L_0 x_ps of a lane's FIRST interior row is off (x_ps: the previous lane's separator unknown), then r is zero
  in interior rows 1, 2, and in row 0   r_0 = L_0 (x~_ps - x_ps)  ->  x~_ps = x_ps + L_0^-1 r_0, which is compared with the candidates (0, the lane's own
  separator, another lane's).
  DGP_LIB_PATH=... python profiles/tools/r06_twin_residual.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import harness, parity_cases as PC, test_hip_every_kernel as T
from oracle import gpmp2_oracle as O
dof, io, cov, n = 2, 'f32', 'static_full', 128
d, C = 2 * dof, 4
be = harness.Backend('hip')
rs = np.random.RandomState(7)
B = 6
p, th, start, goal, sdf, qc, ow, eps, q_full = T._inputs(rs, dof, n, B, cov, io)
x_tw = be.step_errors(p, th, start, goal, sdf, qc=qc, ow=ow, eps=eps, q_full=q_full, io=io)[0]
qc_s, ow_s, eps_s = p.static_covs(B)
Qi = O.calc_Q_inv_batch(qc_s, p.dt)
A, b, K = O.construct_linear_system_batch(th, start, goal, np.broadcast_to(sdf, (B,) + sdf.shape[1:]), Qi, ow_s, eps_s, p)
LAM, R = O.normal_equations(A, b, K, p.reg)
x_ok = np.linalg.solve(LAM, R).reshape(B, n, d)
Dg, Up, _ = O.triband(LAM, n, d)
for bb in range(2):
  r = (R[bb, :, 0] - LAM[bb] @ x_tw[bb].reshape(-1)).reshape(n, d)
  rn = np.abs(r).max(axis=1) / np.abs(R[bb]).max()
  print('trajectory %d: relative residual per block row (lane by lane: rows 0 1 2 interior, 3 separator)' % bb)
  for l in range(0, 32, 8):
    print('   lanes %2d..%2d:  ' % (l, l + 7) + '  |  '.join(' '.join('%6.0e' % v if v > 1e-6 else '  .   ' for v in rn[4 * j:4 * j + 4]) for j in range(l, l + 8)))
  print('   inferred x~_ps of lanes 1..6 against candidates (max abs difference; |x_ps| ~ %.2e):' % np.abs(x_ok[bb]).max())
  for j in range(1, 7):
    g0 = 4 * j
    L0 = Up[bb, g0 - 1].T                      # block (g0, g0-1) = U_{g0-1}^T
    xps_ok = x_ok[bb, g0 - 1]
    xps_tw = xps_ok - np.linalg.solve(L0, r[g0])      # r_0 = eta_0 - ... - L_0 x_ps(true as stored) ; the kernel used x~_ps:  L_0 (x~_ps - x_ps) = -r_0 ... sign fixed below
    xps_tw2 = xps_ok + np.linalg.solve(L0, r[g0])
    cands = {'0': np.zeros(d), 'own separator': x_ok[bb, g0 + 3], 'separator of lane j-2': x_ok[bb, g0 - 5] if g0 >= 5 else np.zeros(d), 'own row 0': x_ok[bb, g0],
             'true x_ps': xps_ok, 'x_ps with velocity part zeroed': np.concatenate([xps_ok[:dof], np.zeros(dof)]), 'x_ps without dt*v (Phi = I)': xps_ok}
    best = min(((np.abs(v - xx).max(), k, s) for k, v in cands.items() for s, xx in (('-', xps_tw), ('+', xps_tw2))), key=lambda t: t[0])
    print('     lane %d: x_ps %s  inferred(-) %s  inferred(+) %s   closest candidate: %s (%s) diff %.1e' % (
        j, np.array2string(xps_ok, precision=4), np.array2string(xps_tw, precision=4), np.array2string(xps_tw2, precision=4), best[1], best[2], best[0]))
