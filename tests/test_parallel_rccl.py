"""GPU: the product's multi-GPU helper (dgpmp2_amd.parallel) executed over RCCL -- backend 'nccl' IS RCCL on ROCm -- on the one
MI355X a test box has: a world-size-1 process group runs parallel.plan_sharded(DiffGPMP2Planner.forward ...) end to end, i.e.
the sharding arithmetic, the HIP solve on the shard and the all_gather_into_tensor collective on device memory.  (World size 2
is covered on CPU with gloo in tests/test_parallel_gloo.py; bench.py --gpus N runs the same helper on N GPUs.)"""
import os
import numpy as np
import pytest
import torch
import torch.distributed as dist
from conftest import rel_err
from oracle import gpmp2_oracle as O

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.fixture(scope='module')
def rccl_group():
  assert torch.cuda.is_available()
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(29600 + os.getpid() % 2000)
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device(DEV))
  yield dist.group.WORLD
  dist.destroy_process_group()


def test_plan_sharded_over_rccl_matches_plain_forward(rccl_group, golden):
  from test_planner_api import make_planner, T
  from dgpmp2_amd import parallel
  from dgpmp2_amd.utils.planner_utils import straight_line_trajb
  assert dist.get_backend(rccl_group) == 'nccl'
  g = golden('g3_c2mini')
  B, n, G = 8, 64, int(g['G'])
  planner = make_planner(n, 1, max_iters=4)
  sdf = T(O.circles_sdf(G, g['circles']))[None, None].expand(B, 1, G, G)
  start, goal = T(g['start']), T(g['goal'])
  th0 = straight_line_trajb(start[:, :, :2], goal[:, :, :2], 10.0, n - 1, 2)
  solve = lambda th, st, go, sd: planner.forward(th, st, go, None, sd)[0]
  full = parallel.plan_sharded(solve, th0, start, goal, sdf, group=rccl_group)
  torch.cuda.synchronize()
  assert full.is_cuda and tuple(full.shape) == (B, n, 4)
  plain = planner.forward(th0, start, goal, None, sdf)[0]
  assert torch.equal(full, plain)
  assert rel_err(full.cpu().numpy(), g['th_hist'][4]) < 1e-8            # the reference's own trajectory after 4 GN iterations


def test_all_gather_trajectories_ragged_padding_on_device(rccl_group):
  from dgpmp2_amd import parallel
  x = torch.randn(5, 64, 4, device=DEV)
  out = parallel.all_gather_trajectories(x, 5, group=rccl_group)
  assert out.is_cuda and torch.equal(out, x)
