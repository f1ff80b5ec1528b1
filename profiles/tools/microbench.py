#!/usr/bin/env python
"""Kernel-level timing of dgp_gn_step / dgp_gn_solve at several shapes (HIP events), for tuning rounds.
usage: python profiles/tools/microbench.py [--reps 200]"""
import argparse, ctypes, json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import make_inputs, algorithmic_bytes_per_trajectory
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config


def time_step(B, n, G, dtype, reps, dof=2):
  dev = torch.device('cuda:0')
  th0, start, goal, sdf = make_inputs(B, n, G, dev)
  th0, start, goal, sdf = [t.to(dtype) for t in (th0, start, goal, sdf)]
  s = _capi.Solver(solver_config(n, dof, dtype))
  sa = s.sdf_arg(sdf.data_ptr(), G, G, 0)
  st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  dth = torch.empty_like(th0); err = torch.empty(B, device=dev, dtype=dtype); eex = torch.empty(B, device=dev, dtype=dtype)
  f = lambda: s.gn_step(B, th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sa, None, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), None, st)
  for _ in range(10): f()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
  for _ in range(reps): f()
  e1.record(); torch.cuda.synchronize(); wall = time.perf_counter() - t0
  ms = e0.elapsed_time(e1) / reps
  by = algorithmic_bytes_per_trajectory(n, 2 * dof, io_bytes=4 if dtype == torch.float32 else 8) * B
  return dict(B=B, n=n, G=G, dtype=str(dtype), kernel_us=ms * 1e3, wall_us=wall / reps * 1e6, GBs=by / (ms * 1e-3) / 1e9)


if __name__ == '__main__':
  ap = argparse.ArgumentParser(); ap.add_argument('--reps', type=int, default=200); a = ap.parse_args()
  import __graft_entry__; __graft_entry__.build()
  for (B, n, G, dt) in [(4096, 64, 256, torch.float32), (4096, 64, 256, torch.float64), (32768, 64, 256, torch.float32),
                        (1024, 64, 256, torch.float32), (4096, 32, 256, torch.float32), (4096, 16, 256, torch.float32)]:
    print(json.dumps(time_step(B, n, G, dt, a.reps)))
