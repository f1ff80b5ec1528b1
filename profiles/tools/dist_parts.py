#!/usr/bin/env python
"""What the two collectives of bench.py's distributed branch cost at a world size of one (host wall time, microseconds, median of 30):
  HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29531 profiles/tools/dist_parts.py"""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dgpmp2_amd import parallel


def med(f, n=30):
  ts = []
  for _ in range(n):
    torch.cuda.synchronize(); t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
  return sorted(ts)[len(ts) // 2]


def main():
  torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
  dist.init_process_group('nccl')
  w = dist.get_world_size()
  th = torch.randn(4096, 64, 4, device='cuda')
  ev = torch.cuda.Event()
  for _ in range(3): parallel.all_gather_trajectories(th, 4096 * w); dist.barrier()
  def gather():
    parallel.all_gather_trajectories(th, 4096 * w); ev.record()
    while not ev.query(): pass
  def gather_sync():
    parallel.all_gather_trajectories(th, 4096 * w); torch.cuda.synchronize()
  def barrier():
    dist.barrier(); torch.cuda.synchronize()
  t = torch.zeros(1, device='cuda')
  def allreduce_spin():
    dist.all_reduce(t); ev.record()
    while not ev.query(): pass
  if dist.get_rank() == 0:
    print({'world': w, 'all_gather_4MiB_spin_us': round(med(gather), 1), 'all_gather_4MiB_sync_us': round(med(gather_sync), 1), 'barrier_sync_us': round(med(barrier), 1),
           'all_reduce_1elem_spin_us': round(med(allreduce_spin), 1), 'empty_sync_us': round(med(lambda: None), 1)})
  else:
    for f in (gather, gather_sync, barrier, allreduce_spin, lambda: None): med(f)
  dist.destroy_process_group()


if __name__ == '__main__':
  main()
