"""Host-side signed-distance-field preparation (not on the hot path; the bilinear lookup itself lives in the HIP
kernel).  Reference: diff_gpmp2/utils/sdf_utils.py (sdf_2d :6-21, rgb2gray :23-24, costmap_2d :26-31)."""
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
