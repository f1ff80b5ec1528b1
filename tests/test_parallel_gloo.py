"""N > 1 path on CPU: world_size-2 and -4 gloo processes exercise the batch sharding + final all-gather of dgpmp2_amd.parallel (shared grid and one grid per trajectory).
The per-shard "GPU solve" is stood in for by the wavefront emulator (the same per-lane program the HIP kernel runs), so
the gathered result must equal the single-process result on the whole batch bit for bit."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(B, n, per_sample=False):
  from oracle import gpmp2_oracle as O
  rs = np.random.RandomState(7)
  start = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  goal = np.concatenate([rs.uniform(-4, 4, (B, 1, 2)), np.zeros((B, 1, 2))], -1)
  th = O.straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  if per_sample:      # one grid per trajectory (the reference's API shape): three circles each, so that a rank reading another rank's grids gives another trajectory
    sdf = np.stack([O.circles_sdf(64, [(cx, cy, r) for cx, cy, r in zip(rs.uniform(-3, 3, 3), rs.uniform(-3, 3, 3), rs.uniform(0.5, 1.2, 3))])[None] for _ in range(B)])
  else:
    sdf = O.circles_sdf(64, O.C2_CIRCLES)[None, None]
  return th, start, goal, sdf


def _solve_fn(n, iters):
  import harness
  from oracle import gpmp2_oracle as O
  be = harness.Backend('emul')
  p = O.OracleParams(dof=2, total_time_step=n - 1)

  def fn(th, start, goal, sdf):
    out = be.solve(p, th.numpy(), start.numpy(), goal.numpy(), sdf.numpy(), iters, 0.0, io='f64')
    return torch.from_numpy(out[0])
  return fn


def _worker(rank, world, port, B, n, iters, q, per_sample=False):
  sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
  os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  from dgpmp2_amd import parallel
  th, start, goal, sdf = [torch.from_numpy(a) for a in _inputs(B, n, per_sample)]
  # shared grid: an expand()ed view, must NOT be sliced per rank; per-sample grids: (B,1,H,W), every rank must get exactly its own trajectories' grids
  sdf_b = sdf if per_sample else sdf.expand(B, 1, 64, 64)
  seen = []

  def solve(th_l, st_l, go_l, sdf_l):
    seen.append(tuple(sdf_l.shape))
    return _solve_fn(n, iters)(th_l, st_l, go_l, sdf_l)
  full = parallel.plan_sharded(solve, th, start, goal, sdf_b)
  lo_, hi_ = parallel.shard_range(B, rank, world)
  assert seen == [(hi_ - lo_, 1, 64, 64)]
  lo, hi = parallel.shard_range(B, rank, world)
  # the same gather into a buffer allocated once (what a loop that gathers every outer iteration, and bench.py's timed regions, do)
  buf = parallel.gather_buffer(full[lo:hi], B)
  again = parallel.all_gather_trajectories(full[lo:hi].clone(), B, out=buf)
  assert torch.equal(again, full) and (again.data_ptr() == buf.data_ptr() or B % world != 0)
  try:
    parallel.all_gather_trajectories(full[lo:hi], B, out=buf[:-1])
    raise AssertionError('a wrongly sized out= buffer must be rejected')
  except ValueError:
    pass
  q.put((rank, lo, hi, full.numpy()))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize('B,world,per_sample', [(4, 2, False), (5, 2, False), (6, 4, True)])           # even and ragged splits; world 4 with one grid per trajectory (shards 2,2,1,1)
def test_sharded_plan_equals_single_process(B, world, per_sample):
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  import harness
  harness.build_emulator()
  n, iters = 16, 2
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29500 + (os.getpid() % 2000) + B
  procs = [ctx.Process(target=_worker, args=(r, world, port, B, n, iters, q, per_sample)) for r in range(world)]
  for pr in procs: pr.start()
  res = [q.get(timeout=600) for _ in range(world)]
  for pr in procs: pr.join(timeout=120)
  assert all(pr.exitcode == 0 for pr in procs)
  th, start, goal, sdf = _inputs(B, n, per_sample)
  single = _solve_fn(n, iters)(*[torch.from_numpy(a) for a in (th, start, goal, sdf)]).numpy()
  covered = np.zeros(B, dtype=bool)
  for rank, lo, hi, full in res:
    assert full.shape == single.shape and np.array_equal(full, single)        # every rank holds the whole gathered batch
    covered[lo:hi] = True
  assert covered.all()


def test_shard_range_partitions():
  from dgpmp2_amd.parallel import shard_range
  for B in (1, 7, 8, 4096, 32768):
    for W in (1, 2, 3, 8):
      r = [shard_range(B, k, W) for k in range(W)]
      assert r[0][0] == 0 and r[-1][1] == B and all(r[k][1] == r[k + 1][0] for k in range(W - 1))
      assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_shard_batch_slices_per_sample_grids_and_keeps_tiled_sizes():
  """shard_batch: a per-sample grid tensor (row-major or 4 x 4-tiled) is sliced like the trajectories, a shared one (leading dimension 1 / expand()ed) passes through;
  a tiled shard is still a TiledSdf with its logical size."""
  from dgpmp2_amd.parallel import shard_batch, shard_range
  from dgpmp2_amd.utils.sdf_utils import tile_sdf, tiled_hw
  B = 6
  th = torch.arange(B * 3 * 4, dtype=torch.float64).reshape(B, 3, 4)
  per = torch.randn(B, 1, 10, 13, dtype=torch.float64)
  til = tile_sdf(per)
  one = torch.randn(1, 1, 10, 13, dtype=torch.float64)
  for W in (1, 2, 4):
    for r in range(W):
      lo, hi = shard_range(B, r, W)
      a, b, c, d, e = shard_batch([th, per, til, one, one.expand(B, 1, 10, 13)], r, W)
      assert torch.equal(a, th[lo:hi]) and torch.equal(b, per[lo:hi]) and torch.equal(c, til[lo:hi]) and tiled_hw(c) == (10, 13)
      assert d is one and e.shape[0] == hi - lo and e.stride(0) == 0
