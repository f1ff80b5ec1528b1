"""CPU-only: the reference's on-disk dataset layout round-trips through dgpmp2_amd.datasets and batches
into the tensor shapes DiffGPMP2Planner.step()/forward() take."""
import numpy as np
import torch
from torch.utils.data import DataLoader


def test_planning_dataset_roundtrip(tmp_path):
  from dgpmp2_amd.datasets import PlanningDataset, write_environment, write_problem, write_meta
  from dgpmp2_amd.utils.sdf_utils import sdf_2d
  root = str(tmp_path)
  rs = np.random.RandomState(0)
  n, d, G = 16, 4, 32
  for e in range(2):
    im = np.ones((G, G)); im[8 + e:14 + e, 10:20] = 0.0            # one rectangular obstacle
    sdf = sdf_2d(im, padlen=0, res=10.0 / G)
    write_environment(root, 'train', e, im, sdf)
    for pidx in range(3):
      write_problem(root, 'train', e, pidx, rs.randn(d), rs.randn(d), rs.randn(n, d))
  write_meta(root, 'train', 2, 3, {'x_lims': [-5, 5], 'y_lims': [-5, 5]}, G)
  ds = PlanningDataset(root, mode='train')
  assert len(ds) == 6
  s = ds[4]                                                        # env 1, problem 1
  assert s['im'].shape == (1, G, G) and s['sdf'].shape == (1, G, G) and s['start'].shape == (1, d) and s['th_opt'].shape == (n, d)
  assert s['im'].dtype == torch.float64 and set(np.unique(s['im'].numpy())) <= {0.0, 1.0}
  assert float(s['im'][0, 10, 12]) == 0.0 and float(s['sdf'][0, 10, 12]) < 0 and float(s['sdf'][0, 0, 0]) > 0
  batch = next(iter(DataLoader(ds, batch_size=3)))
  assert batch['im'].shape == (3, 1, G, G) and batch['sdf'].shape == (3, 1, G, G) and batch['start'].shape == (3, 1, d)
  assert batch['th_opt'].shape == (3, n, d)
  sub = PlanningDataset(root, mode='train', num_envs=1, num_env_probs=2)
  assert len(sub) == 2
