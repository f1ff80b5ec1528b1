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


def test_dataset_tiled_layout_and_collate():
  """sdf_layout='tiled4' (round 6): sample['sdf'] is the 4 x 4-tiled field (a TiledSdf that carries the logical size), the default collate and
  PlanningDataset.collate stack it into the (B,1,H/4,W/4,4,4) tensor the planner takes in place of sdfb -- also through DataLoader worker processes."""
  import os
  import torch
  from torch.utils.data import DataLoader
  from dgpmp2_amd.datasets.planning_dataset import PlanningDataset
  from dgpmp2_amd.utils.sdf_utils import untile_sdf, tiled_hw
  root = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'mini_dataset')
  rm = PlanningDataset(root, 'train')
  tl = PlanningDataset(root, 'train', sdf_layout='tiled4', keep_rowmajor=True)
  s0, s1 = rm[0], tl[0]
  H, W = s0['sdf'].shape[-2:]
  assert tiled_hw(s1['sdf']) == (H, W) and tuple(s1['sdf'].shape) == (1, (H + 3) // 4, (W + 3) // 4, 4, 4)
  assert torch.equal(untile_sdf(s1['sdf']), s0['sdf']) and torch.equal(s1['sdf_rm'], s0['sdf'])
  ref = next(iter(DataLoader(rm, batch_size=2)))['sdf']
  for coll in (None, PlanningDataset.collate):
    for workers in (0, 2):
      b = next(iter(DataLoader(tl, batch_size=2, collate_fn=coll, num_workers=workers)))
      assert tiled_hw(b['sdf']) == (H, W) and b['sdf'].dim() == 6 and torch.equal(untile_sdf(b['sdf']), ref)
  import pytest
  with pytest.raises(ValueError):
    PlanningDataset(root, 'train', sdf_layout='tiles')


def test_forward_result_wrappers_behave_like_lists():
  """the opt-in lazy per-sample results of forward() are read-only Sequences that add, compare, count, index and pickle like the python lists
  the reference returns (the DEFAULT is real lists: DiffGPMP2Planner.lazy_results = False)."""
  import pickle, json, collections.abc
  import numpy as np
  from dgpmp2_amd.gpmp2.diff_gpmp2_planner import _LazyList, _PerSampleHistory, DiffGPMP2Planner
  assert DiffGPMP2Planner.lazy_results is False
  a = _LazyList(np.array([3.0, 1.0, 2.0, 1.0]))
  assert isinstance(a, collections.abc.Sequence) and len(a) == 4 and a[1] == 1.0 and a[1:3] == [1.0, 2.0]
  assert a + [5.0] == [3.0, 1.0, 2.0, 1.0, 5.0] and [0.0] + a == [0.0, 3.0, 1.0, 2.0, 1.0] and a.count(1.0) == 2 and a.index(2.0) == 2 and 3.0 in a
  assert a == [3.0, 1.0, 2.0, 1.0] and a != [3.0] and pickle.loads(pickle.dumps(a)) == [3.0, 1.0, 2.0, 1.0] and type(pickle.loads(pickle.dumps(a))) is list
  assert json.dumps(a.tolist()) == '[3.0, 1.0, 2.0, 1.0]' and list(reversed(a)) == [1.0, 2.0, 1.0, 3.0]
  h = _PerSampleHistory(np.arange(6.0).reshape(2, 3), np.array([2, 3], dtype=np.int32))
  assert isinstance(h, collections.abc.Sequence) and h[0] == [0.0, 1.0] and h.tolist() == [[0.0, 1.0], [3.0, 4.0, 5.0]] and h + [[9.0]] == [[0.0, 1.0], [3.0, 4.0, 5.0], [9.0]]
  assert pickle.loads(pickle.dumps(h)) == [[0.0, 1.0], [3.0, 4.0, 5.0]] and h == [[0.0, 1.0], [3.0, 4.0, 5.0]]
