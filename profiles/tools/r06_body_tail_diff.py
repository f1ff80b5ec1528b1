"""Round 6: did the FIRST version of the exec-join repair change results?  It moved four register saves of a body's exit shuffle in gn_backward_kernel<2,16,4,float,general>[tiled]
behind the moves that reuse their source registers (profiles/r06_compiler_fault.md, last section).  This script runs that kernel (tiled grid, general covariances, fp32 I/O, 64 states:
the (16,4) shape) against the row-major kernel of the same library, for the library given in DGP_LIB_PATH -- once built with the old repair, once with the current one:
  DGP_LIB_PATH=dgpmp2_amd/lib/libdgpmp2_bwdold.so python profiles/tools/r06_body_tail_diff.py     (profiles/tools/devbuild.py 2t_f32_g2 2_f32_g2; the old variant: see the file's end)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import harness, parity_cases as PC, test_hip_every_kernel as T
from oracle import blocktri as BT

be = harness.Backend('hip'); bt = harness.Backend('hip'); bt.sdf_tiled = True
worst = {}
for seed in range(6):
  rs = np.random.RandomState(100 + seed)
  for cov in ('static_full', 'qfull'):
    for n in (64, 61):
      for persample in (False, True):
        B = 9
        p, th, start, goal, sdf, qc, ow, eps, q_full = T._inputs(rs, 2, n, B, cov, 'f32')
        if persample: sdf = PC.rnd(np.stack([sdf[0, :, :23, :37] + 0.05 * rs.randn() for _ in range(B)]), 'f32')
        kw = dict(qc=qc, ow=ow, eps=eps, q_full=q_full, io='f32')
        sh = (B, n, 1, 1)      # (the development libraries hold the backward units only: dtheta from the C oracle)
        a = [PC.rnd(BT.gn_step(p, th, start, goal, sdf, qc=qc, ow=None if ow is None else ow.reshape(sh), eps=None if eps is None else eps.reshape(sh), q_full=q_full)[0], 'f32')]
        gb = PC.rnd(rs.randn(B, n, 4), 'f32'); ge = PC.rnd(rs.randn(B), 'f32')
        ra = be.backward(p, th, start, goal, sdf, a[0], gb, ge, sdf_grad='f64', **kw)
        rb = bt.backward(p, th, start, goal, sdf, a[0], gb, ge, sdf_grad='f64', **kw)
        for key in ('th', 'start', 'goal', 'sdf', 'qc', 'ow', 'eps'):
          if ra[key] is None or rb[key] is None: continue
          e = float(np.abs(rb[key] - ra[key]).max() / max(np.abs(ra[key]).max(), 1e-300)) if np.all(np.isfinite(rb[key])) else float('inf')
          worst[key] = max(worst.get(key, 0.0), e)
print(os.path.basename(os.environ.get('DGP_LIB_PATH', 'product')), 'tiled vs row-major backward, worst relative difference per output over 48 configurations:',
      ' '.join('%s %.2e' % kv for kv in sorted(worst.items())))
