#!/usr/bin/env python
"""cProfile of the Python planner API on the benchmark batch: where the host time of step() and step()+backward() goes."""
import cProfile, io, os, pstats, sys
sys.argv = [sys.argv[0]]
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'api_overhead.py')).read().split("out = {}")[0])
for name, f in (('step (no_grad)', lambda: planner.step(th0, start, goal, None, sdfb)), ('step + backward', None)):
  if f is None:
    thr = th0.clone().requires_grad_(True)
    def f():
      dth = planner.step(thr, start, goal, None, sdfb)[0]
      dth.sum().backward(); thr.grad = None
    ctx = torch.enable_grad()
  else:
    ctx = torch.no_grad()
  with ctx:
    for _ in range(20): f()
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300): f()
    torch.cuda.synchronize(); pr.disable()
  s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
  print('=====', name, '(300 calls)'); print('\n'.join(l[:150] for l in s.getvalue().splitlines()[4:26]))
