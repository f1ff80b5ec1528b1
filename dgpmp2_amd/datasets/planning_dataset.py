"""On-disk planning-dataset format of the reference, read and written without the reference:

  <root>/<mode>/meta.yaml                     {num_envs, probs_per_env, env_params, im_size}
  <root>/<mode>/im_sdf/<env>_im.png           occupancy image, free space > 0.75
  <root>/<mode>/im_sdf/<env>_sdf.npy          (H, W) float64 signed distance field in metres
  <root>/<mode>/<label_subdir>/env_<e>_prob_<p>.npz   start (d,), goal (d,), th_opt (n, d)

Reference: datasets/planning_dataset.py:15-69 (reader), datasets/generate_optimal_paths_gpmp2.py:198-206 (writer).
A sample is {'im' (1,H,W) float64 in {0,1}, 'sdf' (1,H,W) float64, 'start' (1,d), 'goal' (1,d), 'th_opt' (n,d)} -- the
tensors a DataLoader batches into the (B,1,H,W) / (B,1,d) / (B,n,d) inputs of DiffGPMP2Planner.step()/forward().
"""
import os

import numpy as np
import torch
import yaml
from torch.utils.data import Dataset


def _imread(path):
  try:
    import matplotlib.pyplot as plt
    return plt.imread(path)
  except ImportError:
    from PIL import Image
    a = np.asarray(Image.open(path), dtype=np.float64)
    return a / 255.0 if a.max() > 1.0 else a


def _imwrite(path, im):
  im8 = (np.clip(np.asarray(im, dtype=np.float64), 0.0, 1.0) * 255.0).astype(np.uint8)
  try:
    from PIL import Image
    Image.fromarray(im8).save(path)
  except ImportError:
    import matplotlib.pyplot as plt
    plt.imsave(path, im8, cmap='gray', vmin=0, vmax=255)


class PlanningDataset(Dataset):
  """Same constructor arguments, length and sample dict as the reference's PlanningDataset."""

  def __init__(self, root_dir, mode='train', num_envs=-1, num_env_probs=-1, label_subdir='opt_trajs_gpmp2', sdf_layout='rowmajor', keep_rowmajor=False):
    """sdf_layout (an addition): 'rowmajor' -- sample['sdf'] is the (1,H,W) field as the reference yields it; 'tiled4' -- the same values as the 4 x 4 tiles the GN
    kernels read with half the memory traffic when every trajectory has its own grid (utils.sdf_utils.tile_sdf: a (1,H/4,W/4,4,4) TiledSdf that carries (H,W)).
    Tiling runs here, in the DataLoader workers, so the training loop never pays for it; the default collate stacks the samples into the (B,1,H/4,W/4,4,4)
    tensor that step() / forward() / the error helpers take in place of sdfb (PlanningDataset.collate does the same explicitly).  The learn modules of the
    reference read `sdf` as an image channel (train_planner.py:273): keep_rowmajor=True keeps the row-major field next to the tiles, as sample['sdf_rm']."""
    if sdf_layout not in ('rowmajor', 'tiled4'): raise ValueError("sdf_layout must be 'rowmajor' or 'tiled4'")
    self.sdf_layout = sdf_layout
    self.keep_rowmajor = bool(keep_rowmajor)
    self.root_dir = os.path.abspath(root_dir)
    self.subdir = os.path.join(root_dir, mode)
    self.imsdf_dir = os.path.join(self.subdir, 'im_sdf')
    self.label_dir = os.path.join(self.subdir, label_subdir)
    with open(os.path.join(self.subdir, 'meta.yaml')) as f:
      self.meta_data = yaml.safe_load(f)            # the reference's bare yaml.load(f) fails on PyYAML >= 6
    if 0 < num_envs <= self.meta_data['num_envs'] and 0 < num_env_probs <= self.meta_data['probs_per_env']:
      self.meta_data['num_envs'] = num_envs
      self.meta_data['probs_per_env'] = num_env_probs
    self.num_files = self.meta_data['num_envs'] * self.meta_data['probs_per_env']

  def __len__(self):
    return self.num_files

  def __getitem__(self, idx):
    ppe = self.meta_data['probs_per_env']
    env_idx, prob_idx = int(idx / ppe), int(idx % ppe)
    im = _imread(os.path.join(self.imsdf_dir, '%d_im.png' % env_idx))
    if im.ndim > 2:
      im = np.dot(im[..., :3], [0.299, 0.587, 0.114])
    im = torch.from_numpy(np.array([im > 0.75], dtype=np.float64))
    sdf = torch.from_numpy(np.asarray(np.load(os.path.join(self.imsdf_dir, '%d_sdf.npy' % env_idx)), dtype=np.float64)[None])
    npf = np.load(os.path.join(self.label_dir, 'env_%d_prob_%d.npz' % (env_idx, prob_idx)))
    sample = {'im': im, 'sdf': sdf, 'start': torch.from_numpy(np.atleast_2d(npf['start'])), 'goal': torch.from_numpy(np.atleast_2d(npf['goal'])),
              'th_opt': torch.from_numpy(np.asarray(npf['th_opt']))}
    if self.sdf_layout == 'tiled4':
      from ..utils.sdf_utils import tile_sdf
      if self.keep_rowmajor: sample['sdf_rm'] = sdf
      sample['sdf'] = tile_sdf(sdf)
    return sample

  @staticmethod
  def collate(samples):
    """collate_fn for torch.utils.data.DataLoader: the default collation, with the tiled fields re-declared as a TiledSdf of the samples' logical size (the default
    collate already keeps it; this one does not depend on that and checks that every sample has the same size)."""
    from torch.utils.data import default_collate
    from ..utils.sdf_utils import tiled_hw, as_tiled
    batch = default_collate(samples)
    hws = set(tiled_hw(s_['sdf']) for s_ in samples)
    if hws != {None}:
      if len(hws) != 1: raise ValueError('PlanningDataset.collate: tiled fields of different logical sizes in one batch: %s' % (sorted(hws, key=str),))
      batch['sdf'] = as_tiled(batch['sdf'], hws.pop())
    return batch


def write_environment(root_dir, mode, env_idx, image, sdf):
  d = os.path.join(root_dir, mode, 'im_sdf')
  os.makedirs(d, exist_ok=True)
  _imwrite(os.path.join(d, '%d_im.png' % env_idx), image)
  np.save(os.path.join(d, '%d_sdf.npy' % env_idx), np.asarray(sdf, dtype=np.float64))


def write_problem(root_dir, mode, env_idx, prob_idx, start, goal, th_opt, label_subdir='opt_trajs_gpmp2'):
  d = os.path.join(root_dir, mode, label_subdir)
  os.makedirs(d, exist_ok=True)
  np.savez(os.path.join(d, 'env_%d_prob_%d.npz' % (env_idx, prob_idx)), start=np.asarray(start), goal=np.asarray(goal), th_opt=np.asarray(th_opt))


def write_meta(root_dir, mode, num_envs, probs_per_env, env_params, im_size):
  os.makedirs(os.path.join(root_dir, mode), exist_ok=True)
  with open(os.path.join(root_dir, mode, 'meta.yaml'), 'w') as f:
    yaml.safe_dump({'num_envs': int(num_envs), 'probs_per_env': int(probs_per_env),
                    'env_params': {k: [float(v) for v in vals] for k, vals in env_params.items()}, 'im_size': int(im_size)}, f)
