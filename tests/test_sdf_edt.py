"""dgp_sdf_2d (csrc/sdf_edt.hip, dgpmp2_amd.utils.sdf_utils.sdf_2d_batch): the reference's sdf_2d (utils/sdf_utils.py:6-21) for a batch of device images.

CPU tests: the brute-force oracle (oracle/edt_oracle.py) against the outputs of the REFERENCE's sdf_2d kept in tests/golden/g6_helpers.npz and against
scipy.ndimage.distance_transform_edt (the reference's third-party dependency) on random and degenerate arrays; host-side argument validation of the
C-ABI entry point (nothing is launched).  GPU tests: the HIP kernels against the golden fixtures, scipy and the oracle -- BIT-EXACT in float64 (integer
squared distances), the float32 output equal to the rounded float64 one -- over ragged sizes, paddings, dtypes, batches and the degenerate images."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import edt_oracle as E            # noqa: E402
from dgpmp2_amd import _capi                  # noqa: E402
from dgpmp2_amd.utils import sdf_utils        # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden', 'g6_helpers.npz')


def _scipy_sdf_2d(image, padlen=1, res=1.0):
  from scipy import ndimage
  im = np.array(np.asarray(image) > 0.75, dtype=np.float64)
  if padlen > 0:
    im = np.pad(im, (padlen, padlen), 'constant', constant_values=(1.0, 1.0))
  return (ndimage.distance_transform_edt(im) - ndimage.distance_transform_edt(1.0 - im)) * res


def _random_images(rs, B, H, W, fill):
  return (rs.rand(B, H, W) > fill).astype(np.float64)


DEGENERATE = [np.ones((3, 4)), np.zeros((3, 4)), np.ones((1, 1)), np.zeros((1, 1)), np.ones((1, 7)), np.zeros((5, 1)),
              np.pad(np.zeros((1, 1)), 3, constant_values=1.0), np.pad(np.ones((2, 2)), 2, constant_values=0.0)]


# ------------------------------------------------------------------ CPU: the oracle is pinned -------------------------------------------------
def test_oracle_matches_the_reference_fixtures():
  g = np.load(GOLD)
  imr = g['sdf_imr']
  np.testing.assert_array_equal(E.sdf_2d(imr, padlen=0, res=0.25), g['sdf_imr_pad0'])
  np.testing.assert_array_equal(E.sdf_2d(imr, padlen=2, res=1.0), g['sdf_imr_pad2'])
  im5 = g['sdf_im5']
  np.testing.assert_array_equal(E.sdf_2d(im5, res=10.0 / im5.shape[0]), g['sdf_im5_pad1'])          # the real map of BASELINE configs[0]: 200 x 200 -> 202 x 202


def test_oracle_matches_scipy_on_random_and_degenerate_arrays():
  from scipy import ndimage
  rs = np.random.RandomState(3)
  cases = [(_random_images(rs, 1, h, w, f)[0]) for (h, w, f) in ((17, 23, 0.1), (40, 9, 0.5), (33, 33, 0.9), (64, 64, 0.02), (5, 70, 0.97))]
  for a in cases + DEGENERATE:
    np.testing.assert_array_equal(E.distance_transform_edt(a), ndimage.distance_transform_edt(a))
    for pad in (0, 1, 3):
      np.testing.assert_array_equal(E.sdf_2d(a, padlen=pad, res=0.05), _scipy_sdf_2d(a, padlen=pad, res=0.05))


def test_host_mirror_sdf_2d_matches_the_reference_fixture():
  g = np.load(GOLD)
  np.testing.assert_array_equal(sdf_utils.sdf_2d(g['sdf_im5'], res=10.0 / g['sdf_im5'].shape[0]), g['sdf_im5_pad1'])


def test_capi_rejects_bad_arguments_without_launching():
  api = _capi.get_api()
  assert api.sdf_2d_workspace_bytes(2, 10, 12, 1) == 256 + 2 * 12 * 14 * 2
  assert api.sdf_2d_workspace_bytes(0, 10, 12, 1) == 0 and api.sdf_2d_workspace_bytes(1, 10, 12, -1) == 0
  buf = (C.c_char * 4096)()
  p = C.addressof(buf)
  ok = dict(image=p, image_dtype=_capi.DGP_F32, batch=1, rows=4, cols=4, padlen=1, res=1.0, out=p, out_dtype=_capi.DGP_F64, ws=p, ws_bytes=4096)

  def call(**kw):
    a = dict(ok); a.update(kw)
    return api.sdf_2d(a['image'], a['image_dtype'], a['batch'], a['rows'], a['cols'], a['padlen'], a['res'], a['out'], a['out_dtype'], a.get('layout', _capi.DGP_SDF_ROWMAJOR),
                      a['ws'], a['ws_bytes'], None)

  assert call(image=None) == _capi.DGP_EINVAL and call(out=None) == _capi.DGP_EINVAL and call(ws=None) == _capi.DGP_EINVAL
  assert call(batch=0) == _capi.DGP_EINVAL and call(rows=0) == _capi.DGP_EINVAL and call(padlen=-1) == _capi.DGP_EINVAL
  assert call(image_dtype=7) == _capi.DGP_EINVAL and call(out_dtype=_capi.DGP_U8) == _capi.DGP_EINVAL and call(layout=5) == _capi.DGP_EINVAL
  assert call(ws_bytes=16) == _capi.DGP_EINVAL and b'workspace' in api.last_error()
  assert call(rows=20000) == _capi.DGP_EUNSUPPORTED and call(batch=70000) == _capi.DGP_EUNSUPPORTED
  assert call(ws=p + 1) == _capi.DGP_EINVAL


def test_batch_wrapper_refuses_host_arrays():
  import torch
  with pytest.raises(RuntimeError, match='CUDA'):
    sdf_utils.sdf_2d_batch(torch.zeros(4, 4))
  with pytest.raises(RuntimeError, match='CUDA'):
    sdf_utils.sdf_2d_batch(np.zeros((4, 4)))


# ------------------------------------------------------------------ GPU: the kernels -----------------------------------------------------------
def _gpu(images, **kw):
  import torch
  return sdf_utils.sdf_2d_batch(torch.as_tensor(images).cuda(), **kw).cpu().numpy()


@pytest.mark.gpu
def test_hip_matches_the_reference_fixtures_bit_for_bit():
  g = np.load(GOLD)
  imr, im5 = g['sdf_imr'], g['sdf_im5']
  np.testing.assert_array_equal(_gpu(imr, padlen=0, res=0.25), g['sdf_imr_pad0'])
  np.testing.assert_array_equal(_gpu(imr, padlen=2, res=1.0), g['sdf_imr_pad2'])
  np.testing.assert_array_equal(_gpu(im5, res=10.0 / im5.shape[0]), g['sdf_im5_pad1'])
  np.testing.assert_array_equal(_gpu(im5.astype(np.float64), res=10.0 / im5.shape[0]), g['sdf_im5_pad1'])


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,fill', [(1, 1, 0.5), (1, 37, 0.5), (41, 1, 0.5), (17, 23, 0.1), (64, 64, 0.02), (63, 65, 0.98), (128, 200, 0.7), (257, 255, 0.995),
                                      (300, 70, 0.3), (2, 2550, 0.9), (3, 3000, 0.9), (2600, 4, 0.95),      # (the last three: the widest row of the padded search (61 KB of LDS), a wider one (clamped search), taller than wide)
                                      # round 6 -- the limits of the tile form of the row pass (8 rows up to 284 padded columns, 4 rows up to 570, one row per workgroup beyond) and of the
                                      # bit-plane column pass (32 rows per word; up to 1024 padded rows), with pad = 1: exactly at and one past each of them
                                      (30, 282, 0.9), (31, 283, 0.9), (62, 568, 0.95), (63, 569, 0.95), (1022, 9, 0.97), (1023, 9, 0.97), (94, 40, 0.5), (95, 41, 0.05)])
@pytest.mark.parametrize('pad', [0, 1, 3])
def test_hip_matches_scipy_and_oracle_on_ragged_batches(H, W, fill, pad):
  rs = np.random.RandomState(H * 1000 + W + pad)
  ims = _random_images(rs, 5, H, W, fill)
  ims[3] = 1.0                       # an image without obstacles
  ims[4] = 0.0                       # ... and one without free space
  out = _gpu(ims, padlen=pad, res=0.04)
  assert out.shape == (5, H + 2 * pad, W + 2 * pad) and out.dtype == np.float64
  for b in range(5):
    np.testing.assert_array_equal(out[b], _scipy_sdf_2d(ims[b], padlen=pad, res=0.04))
  if H * W <= 64 * 64:
    np.testing.assert_array_equal(out[1], E.sdf_2d(ims[1], padlen=pad, res=0.04))
  # input dtypes: float32 and uint8 images of the same occupancy; float32 output = the rounded float64 field
  np.testing.assert_array_equal(_gpu(ims.astype(np.float32), padlen=pad, res=0.04), out)
  np.testing.assert_array_equal(_gpu((ims * 255).astype(np.uint8), padlen=pad, res=0.04), out)
  import torch
  np.testing.assert_array_equal(_gpu(ims, padlen=pad, res=0.04, dtype=torch.float32), out.astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,pad', [(64, 64, 0), (254, 254, 1), (17, 23, 1), (63, 65, 3), (3, 70, 0)])
def test_hip_tiled_output_is_the_tiled_row_major_field(H, W, pad):
  """sdf_2d_batch(layout='tiled4') writes the same field as 4 x 4 tiles (DgpSdf::layout = DGP_SDF_TILED4): untiled, bit-identical to the row-major result; tagged with its
  logical size; equal to tile_sdf() of the row-major tensor (padding cells zero)."""
  import torch
  rs = np.random.RandomState(H + 7 * W)
  ims = torch.as_tensor(_random_images(rs, 4, H, W, 0.7)).cuda()
  for dt in (torch.float64, torch.float32):
    rm = sdf_utils.sdf_2d_batch(ims, padlen=pad, res=0.04, dtype=dt)
    tl = sdf_utils.sdf_2d_batch(ims, padlen=pad, res=0.04, dtype=dt, layout='tiled4')
    Hp, Wp = H + 2 * pad, W + 2 * pad
    assert tl.shape == (4, 1, (Hp + 3) // 4, (Wp + 3) // 4, 4, 4) and tl._dgp_hw == (Hp, Wp)
    assert torch.equal(sdf_utils.untile_sdf(tl)[:, 0], rm)
    assert torch.equal(sdf_utils.tile_sdf(rm.unsqueeze(1)), tl)


@pytest.mark.gpu
def test_hip_degenerate_images_and_2d_input():
  for a in DEGENERATE:
    for pad in (0, 1):
      out = _gpu(a, padlen=pad, res=1.0)
      assert out.shape == (a.shape[0] + 2 * pad, a.shape[1] + 2 * pad)
      np.testing.assert_array_equal(out, _scipy_sdf_2d(a, padlen=pad, res=1.0))


@pytest.mark.gpu
def test_hip_dataset_layout_and_views():
  """(B, 1, H, W) images as the dataset stores them, and a non-contiguous view: same fields, the channel axis kept."""
  import torch
  rs = np.random.RandomState(21)
  ims = _random_images(rs, 3, 40, 56, 0.8)
  ref = np.stack([_scipy_sdf_2d(a, padlen=1, res=0.1) for a in ims])
  d = torch.as_tensor(ims).cuda()
  out4 = sdf_utils.sdf_2d_batch(d[:, None], padlen=1, res=0.1)
  assert out4.shape == (3, 1, 42, 58)
  np.testing.assert_array_equal(out4[:, 0].cpu().numpy(), ref)
  wide = torch.zeros(3, 40, 112, dtype=torch.float64, device='cuda')
  wide[:, :, ::2] = d
  np.testing.assert_array_equal(sdf_utils.sdf_2d_batch(wide[:, :, ::2], padlen=1, res=0.1).cpu().numpy(), ref)
  with pytest.raises(ValueError):
    sdf_utils.sdf_2d_batch(d[:, None, None])


@pytest.mark.gpu
def test_hip_full_size_batch_properties():
  """BASELINE-sized grids (64 images of 512 x 512, sparse obstacles: the longest searches): spot-check against scipy, and size-independent properties
  on the whole batch -- sign = occupancy, |sdf| >= res next to nothing closer, 1-Lipschitz in pixel units along rows and columns."""
  rs = np.random.RandomState(11)
  B, G, res = 64, 512, 10.0 / 512
  ims = np.ones((B, G, G))
  for b in range(B):
    for _ in range(3 + b % 5):
      cy, cx, r = rs.randint(0, G), rs.randint(0, G), rs.randint(4, 60)
      yy, xx = np.ogrid[:G, :G]
      ims[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0.0
  out = _gpu(ims, padlen=1, res=res)
  for b in (0, 17, 63):
    np.testing.assert_array_equal(out[b], _scipy_sdf_2d(ims[b], padlen=1, res=res))
  free = np.pad(ims, ((0, 0), (1, 1), (1, 1)), constant_values=1.0) > 0.75
  assert np.all(out[free] > 0) and np.all(out[~free] < 0) and np.all(np.abs(out) >= res * (1 - 1e-15))
  u = np.abs(out) / res
  su = np.where(free, u, -u)
  # neighbouring pixels: the signed field changes by at most one pixel, except across the boundary, where it jumps from >= 1 to <= -1 (at most 2)
  assert np.max(np.abs(np.diff(su, axis=1))) <= 2 + 1e-12 and np.max(np.abs(np.diff(su, axis=2))) <= 2 + 1e-12
  same = free[:, :, 1:] == free[:, :, :-1]
  assert np.max(np.abs(np.diff(su, axis=2))[same]) <= 1 + 1e-12
