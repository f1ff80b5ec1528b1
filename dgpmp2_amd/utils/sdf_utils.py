"""Signed-distance-field preparation (not on the hot path; the bilinear lookup itself lives in the HIP kernel).
Reference: diff_gpmp2/utils/sdf_utils.py (sdf_2d :6-21, rgb2gray :23-24, costmap_2d :26-31).  sdf_2d is the reference's host function
(numpy in, numpy out, scipy underneath); sdf_2d_batch is the same transform for a batch of device images in one C-ABI call."""
import numpy as np


def sdf_2d(image, padlen=1, res=1.0):
  """Signed Euclidean distance transform of an occupancy image (free space > 0.75), padded by `padlen` pixels of free
  space, scaled by `res` (metres per pixel).  Positive in free space, negative inside obstacles."""
  from scipy import ndimage
  im = np.array(image > 0.75, dtype=np.float64)
  if padlen > 0:
    im = np.pad(im, (padlen, padlen), 'constant', constant_values=(1.0, 1.0))
  inside = ndimage.distance_transform_edt(1.0 - im)
  outside = ndimage.distance_transform_edt(im)
  return (outside - inside) * res


def _tiled_cls():
  """The tensor subclass tiled grids travel in (built on first use: this module must stay importable without torch for its numpy helpers)."""
  global _TiledSdf
  if _TiledSdf is not None: return _TiledSdf
  import torch

  class TiledSdf(torch.Tensor):
    """A signed distance field stored as 4 x 4 tiles, (..., Ht, Wt, 4, 4), together with its LOGICAL size `hw` = (H, W).  H and W set the resolution
    (obstacle_cost.py:34: res = (x_max - x_min) / W) and the clamping of the bilinear lookup (sdf_utils.py:64-72), and they cannot be recovered from the tile
    counts (a 130 x 130 padded field and a 132 x 132 one both have 33 x 33 tiles) -- so they ride on the tensor: every torch operation whose result still ends
    in the same (Ht, Wt, 4, 4) tiles (.to(), .float(), .clone(), .detach(), requires_grad_(), indexing / expand() / cat / stack over the leading axes, DataLoader
    collation) returns a TiledSdf with the same `hw`; anything else (a reduction, a reshape of the tile axes, arithmetic with a differently tiled grid) returns
    a plain tensor.  PlanLayer refuses a 6-D grid that carries no size instead of guessing 4 Ht x 4 Wt."""

    @staticmethod
    def wrap(t, hw):
      r = t if isinstance(t, TiledSdf) else t.as_subclass(TiledSdf)
      r._dgp_hw = (int(hw[0]), int(hw[1]))
      return r

    @property
    def hw(self): return self.__dict__.get('_dgp_hw')

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
      with torch._C.DisableTorchFunctionSubclass():      # (not torch.Tensor.__torch_function__: its blanket as_subclass() of every result fails on the sparse
        out = func(*args, **(kwargs or {}))              #  gradients autograd.grad returns for a tiled leaf)
      hw = None
      stack = list(args) + list((kwargs or {}).values())
      while stack and hw is None:                 # the first tiled operand (also inside the list argument of cat / stack)
        a = stack.pop(0)
        if isinstance(a, TiledSdf): hw = a.__dict__.get('_dgp_hw')
        elif isinstance(a, (list, tuple)): stack = list(a) + stack

      def tag(o):
        if not isinstance(o, torch.Tensor): return o
        if o.layout is not torch.strided: return o      # a sparse gradient of a tiled grid: a plain sparse tensor of the tiled shape
        ok = hw is not None and o.dim() >= 4 and tuple(o.shape[-2:]) == (4, 4) and (hw[0] + 3) // 4 == o.shape[-4] and (hw[1] + 3) // 4 == o.shape[-3]
        if ok:
          if not isinstance(o, TiledSdf): o = o.as_subclass(TiledSdf)
          o.__dict__['_dgp_hw'] = hw
          return o
        return o.as_subclass(torch.Tensor) if isinstance(o, TiledSdf) else o
      if isinstance(out, (tuple, list)): return type(out)(tag(o) for o in out)
      return tag(out)

    def __reduce_ex__(self, proto):             # pickling (DataLoader workers, torch.save): the plain tensor + the size
      return (_rebuild_tiled, (self.as_subclass(torch.Tensor), self.__dict__.get('_dgp_hw'), self.requires_grad))

  _TiledSdf = TiledSdf
  return TiledSdf


_TiledSdf = None


def _rebuild_tiled(t, hw, requires_grad):
  r = _tiled_cls().wrap(t, hw) if hw is not None else t
  return r.requires_grad_(True) if requires_grad and not r.requires_grad else r


def tiled_hw(t):
  """The logical (H, W) a tiled grid tensor carries (TiledSdf.hw; a plain tensor tagged by hand with `_dgp_hw` counts too), or None."""
  return t.__dict__.get('_dgp_hw') if hasattr(t, '__dict__') else None


def as_tiled(t, hw):
  """Declare that the (..., Ht, Wt, 4, 4) tensor `t` holds the tiles of H x W grids (e.g. tiles that lost their size on the way through foreign code)."""
  Ht, Wt = t.shape[-4], t.shape[-3]
  if t.dim() < 4 or tuple(t.shape[-2:]) != (4, 4) or (hw[0] + 3) // 4 != Ht or (hw[1] + 3) // 4 != Wt:
    raise ValueError('as_tiled: %s does not hold the 4 x 4 tiles of a %d x %d grid' % (tuple(t.shape), hw[0], hw[1]))
  return _tiled_cls().wrap(t, hw)


def tile_sdf(sdfb):
  """(B, 1, H, W) signed distance fields -> the same values as 4 x 4 TILES, a (B, 1, ceil(H/4), ceil(W/4), 4, 4) tensor (padding cells zero) that every entry point of
  the planner accepts in place of sdfb (DgpSdf::layout = DGP_SDF_TILED4; DESIGN.md section 3 "SDF").  With one grid per trajectory -- the reference's API shape,
  1 GiB per batch of 4096 -- the bilinear taps of a trajectory then touch 29 instead of 70 cache lines: the GN kernels fetch half the bytes and run ~3 us sooner
  (profiles/r05_tile_probe.txt).  The result is a TiledSdf: the logical size (H, W) (it sets the resolution, obstacle_cost.py:34, and the clamping of the lookup,
  sdf_utils.py:64-72) travels with the tensor through .to() / .detach() / indexing / collation.  Works on host tensors too (a single (1, H, W) / (H, W) field as
  well: -> (1, Ht, Wt, 4, 4) / (Ht, Wt, 4, 4)), so a Dataset can tile in its workers.  sdf_2d_batch(..., layout='tiled4') writes this layout directly."""
  import torch
  if sdfb.dim() not in (2, 3, 4) or (sdfb.dim() == 4 and sdfb.shape[1] != 1): raise ValueError('tile_sdf: (B, 1, H, W) expected, got %s' % (tuple(sdfb.shape),))
  lead = tuple(sdfb.shape[:-2])
  H, W = int(sdfb.shape[-2]), int(sdfb.shape[-1])
  Ht, Wt = (H + 3) // 4, (W + 3) // 4
  t = sdfb.as_subclass(torch.Tensor) if type(sdfb) is not torch.Tensor else sdfb
  if Ht * 4 != H or Wt * 4 != W: t = torch.nn.functional.pad(t, (0, Wt * 4 - W, 0, Ht * 4 - H))
  k = len(lead)
  t = t.reshape(lead + (Ht, 4, Wt, 4)).permute(tuple(range(k)) + (k, k + 2, k + 1, k + 3)).contiguous()
  return _tiled_cls().wrap(t, (H, W))


def untile_sdf(t, hw=None):
  """The inverse of tile_sdf: (..., Ht, Wt, 4, 4) tiles (a grid, or the gradient the planner returns for one) -> (..., H, W); hw = the logical size when the tensor
  does not carry one (a gradient does not)."""
  import torch
  if hw is None: hw = tiled_hw(t)
  if hw is None: raise ValueError('untile_sdf: the tensor carries no logical size (not a TiledSdf); pass hw=(H, W)')
  Ht, Wt = t.shape[-4], t.shape[-3]
  H, W = hw
  lead = tuple(t.shape[:-4])
  k = len(lead)
  p = t.as_subclass(torch.Tensor) if type(t) is not torch.Tensor else t
  return p.permute(tuple(range(k)) + (k, k + 2, k + 1, k + 3)).reshape(lead + (Ht * 4, Wt * 4))[..., :H, :W]


def sdf_2d_batch(images, padlen=1, res=1.0, dtype=None, layout='rowmajor'):
  """sdf_2d for a whole batch of occupancy images on the GPU: ONE C-ABI call (dgp_sdf_2d, two launches of csrc/sdf_edt.hip) instead of
  two scipy distance transforms per image on the host.  images: CUDA tensor (B, H, W), (B, 1, H, W) or (H, W), float32 / float64 / uint8, free space
  > 0.75 as in sdf_2d; -> (B, H + 2 padlen, W + 2 padlen) (or without the batch axis for a 2-D input), float64 like the reference unless
  `dtype` says float32.  The float64 result is bit-identical to sdf_2d's (tests/test_sdf_edt.py).  No CPU path: host arrays go through sdf_2d.
  layout='tiled4': the fields as 4 x 4 tiles, (B, 1, ceil(H'/4), ceil(W'/4), 4, 4) with `_dgp_hw` = (H', W') -- exactly tile_sdf() of the row-major result, written
  by the transform itself (no second pass over 1 GiB of grids)."""
  import torch
  from .. import _capi
  if not torch.is_tensor(images) or not images.is_cuda:
    raise RuntimeError('dgpmp2_amd.sdf_2d_batch: `images` must be a CUDA/ROCm tensor (host arrays: use sdf_2d)')
  squeeze = images.dim() == 2
  channel = images.dim() == 4 and images.shape[1] == 1      # (B, 1, H, W), the dataset's image layout (datasets/planning_dataset.py:54): -> (B, 1, H + 2 p, W + 2 p)
  im = images.unsqueeze(0) if squeeze else (images[:, 0] if channel else images)
  if im.dim() != 3 or im.numel() == 0:
    raise ValueError('sdf_2d_batch: images must be (B, H, W), (B, 1, H, W) or (H, W) and non-empty, got %s' % (tuple(images.shape),))
  codes = {torch.float32: _capi.DGP_F32, torch.float64: _capi.DGP_F64, torch.uint8: _capi.DGP_U8}
  if im.dtype not in codes:
    raise TypeError('sdf_2d_batch: float32, float64 or uint8 images, got %s' % im.dtype)
  dtype = torch.float64 if dtype is None else dtype
  if dtype not in (torch.float32, torch.float64):
    raise TypeError('sdf_2d_batch: float32 or float64 output, got %s' % dtype)
  padlen = int(padlen)
  im = im.contiguous()
  B, H, W = im.shape
  api = _capi.get_api()
  if layout not in ('rowmajor', 'tiled4'): raise ValueError("sdf_2d_batch: layout must be 'rowmajor' or 'tiled4'")
  tiled = layout == 'tiled4'
  Hp, Wp = H + 2 * padlen, W + 2 * padlen
  with torch.cuda.device(im.device):
    if tiled: out = torch.zeros((B, 1, (Hp + 3) // 4, (Wp + 3) // 4, 4, 4), dtype=dtype, device=im.device) if (Hp % 4 or Wp % 4) else torch.empty((B, 1, Hp // 4, Wp // 4, 4, 4), dtype=dtype, device=im.device)
    else: out = torch.empty((B, Hp, Wp), dtype=dtype, device=im.device)
    nbytes = api.sdf_2d_workspace_bytes(B, H, W, padlen)
    if nbytes == 0:
      raise ValueError('sdf_2d_batch: padlen must be non-negative')
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.int32, device=im.device)
    api.check(api.sdf_2d(im.data_ptr(), codes[im.dtype], B, H, W, padlen, float(res), out.data_ptr(), codes[dtype], _capi.DGP_SDF_TILED4 if tiled else _capi.DGP_SDF_ROWMAJOR,
                         ws.data_ptr(), ws.numel() * 4, torch.cuda.current_stream(im.device).cuda_stream))
  if tiled: return _tiled_cls().wrap(out, (Hp, Wp))
  return out[0] if squeeze else (out.unsqueeze(1) if channel else out)


def rgb2gray(rgb):
  return np.dot(rgb[..., :3], [0.299, 0.587, 0.114])


def costmap_2d(sdf, eps):
  """Hinge cost map (sdf <= eps) * (eps - sdf) for torch tensors."""
  return (sdf <= eps).to(sdf.dtype) * (-1.0 * sdf + eps)


def circles_sdf(G, circles, x_lims=(-5.0, 5.0), y_lims=(-5.0, 5.0)):
  """Synthetic SDF of SURVEY 8(d): analytic union of circles on a GxG grid, row 0 = y_max, col 0 = x_min (linspace
  endpoints), sdf = min_k(|p - c_k| - r_k), fp64."""
  xs = np.linspace(x_lims[0], x_lims[1], G)
  ys = np.linspace(y_lims[1], y_lims[0], G)
  X, Y = np.meshgrid(xs, ys)
  sdf = np.full((G, G), np.inf)
  for (cx, cy, r) in circles:
    sdf = np.minimum(sdf, np.sqrt((X - cx) ** 2 + (Y - cy) ** 2) - r)
  return sdf


C2_CIRCLES = ((-2.0, -1.0, 1.0), (1.5, 2.0, 0.8), (0.0, 0.0, 0.7))
