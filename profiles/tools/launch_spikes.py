#!/usr/bin/env python
"""Host time of every dgp_gn_step call in a back-to-back series (the headline workload): where are the slow ones?  Prints the calls that took more than
3x the median and the distances between them (a period would point at a resource of the runtime that wraps: kernel-argument pool, signal pool, ...).
Round 3, one MI355X box: median 2.8 us per call, and every 75th call (75 x 3 424 B of kernel arguments = one 256 KB chunk of the runtime's pool) blocks for ~480 us --
75 x (9.5 us kernel - 2.8 us host): back-pressure of a host that runs three chunks ahead of the GPU, not a bubble (the default 5 000-step run holds 9.5 us per step).
Not what makes one 20-step run in six slow: those start from an empty queue, where a chunk change has nothing to wait for."""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2.plan_layer import solver_config

def main():
  dev = torch.device('cuda:0'); B, n, G = 4096, 64, 256
  th0, start, goal, sdf = bench.make_inputs(B, n, G, dev, seed=0, dof=2)
  s = _capi.Solver(solver_config(num_states=n, dof=2, io_dtype=torch.float32))
  pc = _capi.get_pycall()
  dth = torch.empty_like(th0); err = torch.empty(B, device=dev); eex = torch.empty(B, device=dev); info = torch.zeros(B, dtype=torch.int32, device=dev)
  args = (s.h, B, th0.data_ptr(), start.data_ptr(), goal.data_ptr(), sdf.data_ptr(), G, G, 0, 0, 0, None, 0, None, None, None, dth.data_ptr(), err.data_ptr(), eex.data_ptr(), info.data_ptr(), torch.cuda.current_stream().cuda_stream)
  for _ in range(3000): pc.gn_step(*args)
  torch.cuda.synchronize()
  N = 6000
  t = np.empty(N + 1)
  pcnt = time.perf_counter
  for i in range(N):
    t[i] = pcnt(); pc.gn_step(*args)
  t[N] = pcnt()
  torch.cuda.synchronize()
  d = np.diff(t) * 1e6
  med = np.median(d)
  idx = np.nonzero(d > 3 * med)[0]
  print('calls %d  median %.2f us  mean %.2f us  p99 %.2f us  max %.2f us;  %d calls above 3x the median' % (N, med, d.mean(), np.percentile(d, 99), d.max(), len(idx)))
  print('slow calls (index: us):', ', '.join('%d: %.0f' % (i, d[i]) for i in idx[:40]))
  if len(idx) > 1: print('distances between them:', np.diff(idx)[:40].tolist())

if __name__ == '__main__':
  main()
