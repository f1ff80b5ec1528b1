#!/usr/bin/env python
"""Time dgp_gn_step for every supported launch shape (LPT lanes per trajectory x C states per lane) -- tuning aid for
dgp_host::choose_shape.  usage: python profiles/tools/shape_sweep.py"""
import ctypes, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs, algorithmic_bytes_per_trajectory
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config


def time_step(B, n, G, dtype, reps, shape, dof=2, warm=0.3):
  dev = torch.device('cuda:0')
  if shape: os.environ['DGP_FORCE_SHAPE'] = shape
  else: os.environ.pop('DGP_FORCE_SHAPE', None)
  th0, start, goal, sdf = make_inputs(B, n, G, dev)
  if dof == 3:
    th0 = torch.cat([th0[:, :, :2], torch.zeros_like(th0[:, :, :1]), th0[:, :, 2:], torch.zeros_like(th0[:, :, :1])], -1).contiguous()
    start = torch.cat([start[:, :, :2], torch.zeros_like(start[:, :, :1]), start[:, :, 2:], torch.zeros_like(start[:, :, :1])], -1).contiguous()
    goal = torch.cat([goal[:, :, :2], torch.zeros_like(goal[:, :, :1]), goal[:, :, 2:], torch.zeros_like(goal[:, :, :1])], -1).contiguous()
  th0, start, goal, sdf = [t.to(dtype) for t in (th0, start, goal, sdf)]
  kw = dict(non_holonomic=True, K_d=0.01, epsilon_dist=0.2, reg=0.0) if dof == 3 else {}
  s = _capi.Solver(solver_config(n, dof, dtype, **kw))
  sa = s.sdf_arg(sdf.data_ptr(), G, G, 0)
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  dth = torch.empty_like(th0); err = torch.empty(B, device=dev, dtype=dtype); eex = torch.empty(B, device=dev, dtype=dtype)
  f = lambda: s.gn_step(B, th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sa, None, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), None, st)
  import time
  t0 = time.time()
  while time.time() - t0 < warm:                 # steady clocks: a cold GPU runs the first milliseconds ~12 % slower
    for _ in range(50): f()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(reps): f()
  e1.record(); torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  by = algorithmic_bytes_per_trajectory(n, 2 * dof, io_bytes=4 if dtype == torch.float32 else 8) * B
  return dict(shape=shape or 'auto', B=B, n=n, dof=dof, dtype=str(dtype).split('.')[-1], kernel_us=round(ms * 1e3, 2), GBs=round(by / (ms * 1e-3) / 1e9, 1),
              chk=float(dth.double().abs().sum()))


if __name__ == '__main__':
  import __graft_entry__; __graft_entry__.build()
  for B in (4096, 32768):
    for shape in ('64,1', '32,2', '16,4', '64,2', '64,4', '32,4', None):
      print(json.dumps(time_step(B, 64, 256, torch.float32, 500 if B == 4096 else 100, shape)), flush=True)
  for B in (256, 4096, 32768):       # d = 6 (BASELINE configs[3]): B = 256 gives the per-wavefront time T6 of dgp_host::choose_shape
    for shape in ('64,1', '32,2', '16,4', '64,2', '32,4', '64,4', None):
      print(json.dumps(time_step(B, 64, 512, torch.float32, 200 if B <= 4096 else 40, shape, dof=3)), flush=True)
  for (n, shape) in ((16, '16,1'), (32, '16,2'), (32, '32,1'), (101, '32,4'), (101, '64,2'), (16, None), (32, None), (101, None)):
    print(json.dumps(time_step(4096, n, 512, torch.float32, 200, shape, dof=3)), flush=True)
  for (n, shape) in ((32, '32,1'), (32, '16,2'), (16, '16,1'), (101, '64,2'), (101, '32,4'), (128, '64,2'), (128, '32,4'), (256, '64,4')):
    print(json.dumps(time_step(4096, n, 256, torch.float32, 300, shape)), flush=True)
