"""Probe the GPU box's host: core count and whether torch.inverse on a batch of triangular 256x256 fp64 matrices works at
various thread counts (the cpu_baseline of bench.py hit 'Pivots given to lu_solve must all be >= 1' there)."""
import os, time, torch
print('cpu_count', os.cpu_count(), 'torch threads', torch.get_num_threads(), torch.__config__.parallel_info().split('\n')[0:6])
os.system("lscpu | egrep 'Model name|Socket|Core|Thread|^CPU\\(s\\)' ; free -g | head -2")
torch.manual_seed(0)
B, N = 256, 256
a = torch.randn(B, N, N, dtype=torch.float64)
lam = torch.bmm(a, a.transpose(1, 2)) + N * torch.eye(N, dtype=torch.float64)
u = torch.linalg.cholesky(lam).mH
for thr in (os.cpu_count(), 64, 32, 16, 8):
  torch.set_num_threads(thr)
  try:
    t0 = time.perf_counter(); inv = torch.inverse(u.transpose(1, 2)); dt = time.perf_counter() - t0
    err = (torch.bmm(inv, u.transpose(1, 2)) - torch.eye(N, dtype=torch.float64)).abs().max().item()
    print('threads', thr, 'inverse ok', round(dt, 3), 's err', err)
  except Exception as e:
    print('threads', thr, 'FAILED', str(e)[:100])
  try:
    t0 = time.perf_counter(); inv = torch.linalg.inv(u.transpose(1, 2).contiguous()); dt = time.perf_counter() - t0
    print('threads', thr, 'inv(contiguous) ok', round(dt, 3))
  except Exception as e:
    print('threads', thr, 'contig FAILED', str(e)[:100])
