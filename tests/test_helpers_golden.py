"""CPU-only: the caller-side helpers and the data path (row f4) against fixtures produced by the
reference's own functions (tests/golden/make_golden.py::g6_helpers / g6_dataset):
  utils/planner_utils.py:3-56   check_convergence, check_convergence_batch (incl. the overwritten torch.where), straight_line_traj[b]
  utils/sdf_utils.py:6-21       sdf_2d (and datasets/utils.py:4-18, the dataset tools' copy)
  datasets/planning_dataset.py  PlanningDataset reading tests/golden/mini_dataset/, which make_golden.py wrote with the
                                reference's writer conventions (generate_2d_im_dataset.py:84-89, generate_optimal_paths_gpmp2.py:198-206)
"""
import os
import numpy as np
import torch
from conftest import GOLDEN


def T(a): return torch.from_numpy(np.ascontiguousarray(a))


def test_check_convergence_batch_matches_reference(golden):
  from dgpmp2_amd.utils.planner_utils import check_convergence_batch, check_convergence
  g = golden('g6_helpers')
  dth, errd = T(g['ccb_dth']), T(g['ccb_errd'])
  tol_err, tol_delta, mi = float(g['ccb_tol_err']), float(g['ccb_tol_delta']), int(g['ccb_max_iters'])
  for j in (3, 10, 11):
    c = check_convergence_batch(dth, j, errd, tol_err, tol_delta, mi)
    assert tuple(c.shape) == tuple(g['ccb_shape_j%d' % j])
    assert np.array_equal(c.numpy().astype(np.int64), g['ccb_conv_j%d' % j])
    # torch.where(..., tensor(1), tensor(0)) -> int64; the max-iters branch is torch.ones(...).byte() (the fixture generator's
    # .byte() -> .bool() shim, SURVEY 8c, records it as bool)
    assert c.dtype == (torch.int64 if str(g['ccb_dtype_j%d' % j]) == 'torch.int64' else torch.uint8)
  # the quirk itself: sample 2 has a LARGE update and a small error change -> "converged"; sample 0 vice versa is also 1
  assert list(g['ccb_conv_j3'].reshape(-1)) == [1, 0, 1, 0, 1, 0]
  sc = [[int(check_convergence(dth[b], j, errd[b], tol_err, tol_delta, mi)) for j in (3, 10)] for b in range(dth.shape[0])]
  assert np.array_equal(np.asarray(sc), g['cc_scalar'])


def test_straight_line_trajectories_match_reference(golden):
  from dgpmp2_amd.utils.planner_utils import straight_line_trajb, straight_line_traj
  g = golden('g6_helpers')
  for n, dof in ((32, 2), (7, 2), (64, 3)):
    tag = '_n%d_dof%d' % (n, dof)
    s, e = T(g['sl_start' + tag]), T(g['sl_goal' + tag])
    thb = straight_line_trajb(s, e, 10.0, n - 1, dof)
    assert thb.dtype == torch.float64 and np.array_equal(thb.numpy(), g['sl_thb' + tag])          # bit-exact: same operation order
    th1 = straight_line_traj(s[0], e[0], 10.0, n - 1, dof)
    assert np.array_equal(th1.numpy(), g['sl_th1' + tag])


def test_sdf_2d_matches_reference_bit_exact(golden):
  from dgpmp2_amd.utils.sdf_utils import sdf_2d
  g = golden('g6_helpers')
  im5 = g['sdf_im5'].astype(np.float64)
  assert np.array_equal(sdf_2d(im5, res=10.0 / im5.shape[0]), g['sdf_im5_pad1'])
  assert np.array_equal(g['sdf_im5_pad1'], golden('g3_c1')['sdf'])              # the grid the C1 parity tests run on
  assert np.array_equal(sdf_2d(g['sdf_imr'], padlen=0, res=0.25), g['sdf_imr_pad0'])
  assert np.array_equal(sdf_2d(g['sdf_imr'], padlen=2, res=1.0), g['sdf_imr_pad2'])


def test_planning_dataset_reads_reference_written_dataset(golden):
  from dgpmp2_amd.datasets import PlanningDataset
  g = golden('g6_dataset')
  root = os.path.join(GOLDEN, 'mini_dataset')
  ds = PlanningDataset(root, mode='train')
  assert len(ds) == int(g['len'])
  assert len(PlanningDataset(root, mode='train', num_envs=1, num_env_probs=1)) == int(g['len_sub'])
  for k in range(len(ds)):
    s = ds[k]
    assert set(s.keys()) == {'im', 'sdf', 'start', 'goal', 'th_opt'}
    for key in s:
      ref = g['s%d_%s' % (k, key)]
      assert str(s[key].dtype) == str(g['s%d_%s_dtype' % (k, key)]), (k, key)
      assert tuple(s[key].shape) == ref.shape and np.array_equal(s[key].numpy(), ref), (k, key)
  # and the planner consumes a batch of it (shapes of DiffGPMP2Planner.step / forward)
  from torch.utils.data import DataLoader
  b = next(iter(DataLoader(PlanningDataset(root, mode='train', num_envs=1, num_env_probs=2), batch_size=2)))
  assert b['im'].shape == (2, 1, 50, 50) and b['sdf'].shape == (2, 1, 52, 52) and b['start'].shape == (2, 1, 4) and b['th_opt'].shape == (2, 16, 4)


def test_writers_reproduce_reference_layout(tmp_path, golden):
  """dgpmp2_amd.datasets' writers produce files the reader (and hence the reference's reader) maps to the same samples."""
  from dgpmp2_amd.datasets import PlanningDataset, write_environment, write_problem, write_meta
  g = golden('g6_dataset')
  root = str(tmp_path)
  for e, k in ((0, 0), (1, 2)):
    write_environment(root, 'train', e, g['s%d_im' % k][0], g['s%d_sdf' % k][0])
    for pidx in range(2):
      kk = k + pidx
      write_problem(root, 'train', e, pidx, g['s%d_start' % kk][0], g['s%d_goal' % kk][0], g['s%d_th_opt' % kk])
  write_meta(root, 'train', 2, 2, {'x_lims': [-5, 5], 'y_lims': [-5, 5]}, 50)
  ds = PlanningDataset(root, mode='train')
  for k in range(4):
    s = ds[k]
    for key in s:
      assert np.array_equal(s[key].numpy(), g['s%d_%s' % (k, key)]), (k, key)
