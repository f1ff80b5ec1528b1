#!/usr/bin/env python
"""Wall time per call of the Python planner API (DiffGPMP2Planner.step / PlanLayer.forward + backward) vs the bare kernel:
how much host overhead sits on top of the 18 us kernel.   usage: python profiles/tools/api_overhead.py"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs
from dgpmp2_amd.robot_models import PointRobot2D
from dgpmp2_amd.gpmp2 import DiffGPMP2Planner

B, n, G = 4096, 64, 256
dev = torch.device('cuda:0')
th0, start, goal, sdf = make_inputs(B, n, G, dev)
t = lambda v: torch.tensor(v, dtype=torch.float64)
gp = {'Q_c_inv': torch.eye(2, dtype=torch.float64), 'K_s': t(0.01), 'K_g': t(0.01)}
ob = {'cost_sigma': t(0.01), 'epsilon_dist': t(0.4)}
pp = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': n - 1}
op = {'method': 'gauss_newton', 'reg': 0.1, 'plan_time': float('inf'), 'max_iters': 10, 'tol_err': 1e-3, 'tol_delta': 1e-4}
planner = DiffGPMP2Planner(gp, ob, pp, op, {'x_lims': [-5., 5.], 'y_lims': [-5., 5.]}, PointRobot2D(t(0.4), B, n), batch_size=B, use_cuda=True)
sdfb = sdf.expand(B, 1, G, G)


def wall(f, reps):
  for _ in range(5): f()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(reps): f()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / reps * 1e6


out = {}
with torch.no_grad():
  out['step_no_grad_us'] = wall(lambda: planner.step(th0, start, goal, None, sdfb), 200)
  out['forward_10iters_fused_us'] = wall(lambda: planner.forward(th0, start, goal, None, sdfb), 20)
thr = th0.clone().requires_grad_(True)


def fb():
  dth = planner.step(thr, start, goal, None, sdfb)[0]
  dth.sum().backward()
  thr.grad = None


out['step_plus_backward_us'] = wall(fb, 100)
print(json.dumps({k: round(v, 1) for k, v in out.items()}))
