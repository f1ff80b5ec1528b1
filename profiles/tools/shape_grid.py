#!/usr/bin/env python
"""Does dgp_host::choose_shape pick the fastest launch shape?  Times every supported (LPT, C) and the automatic choice on a
grid of trajectory lengths and batch sizes (2D point robot, static covariances, f32 I/O).
usage: python profiles/tools/shape_grid.py [--dof 2]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from shape_sweep import time_step

if __name__ == '__main__':
  ap = argparse.ArgumentParser(); ap.add_argument('--dof', type=int, default=2); args = ap.parse_args()
  import __graft_entry__; __graft_entry__.build()
  worst = 1.0
  time_step(4096, 64, 256, torch.float32, 200, None, dof=args.dof)          # clocks up
  for n in (8, 16, 24, 32, 48, 64, 101, 128, 200, 256):
    for B in (256, 1024, 4096, 16384):
      res = {}
      for lpt in (16, 32, 64):
        for c in (1, 2, 4):
          if lpt * c < n: continue
          res['%d,%d' % (lpt, c)] = time_step(B, n, 256, torch.float32, 60, '%d,%d' % (lpt, c), dof=args.dof, warm=0.03)['kernel_us']
      auto = time_step(B, n, 256, torch.float32, 60, None, dof=args.dof, warm=0.03)['kernel_us']
      best = min(res, key=res.get)
      worst = max(worst, auto / res[best])
      print(json.dumps({'n': n, 'B': B, 'auto_us': auto, 'best': best, 'best_us': res[best], 'auto_over_best': round(auto / res[best], 3), 'all': res}), flush=True)
  print(json.dumps({'worst_auto_over_best': round(worst, 3)}))
