"""dgp_sdf_2d (csrc/sdf_edt.hip) against the host path it replaces (scipy, utils/sdf_utils.py:6-21): time per batch, per image, and the HBM rate implied by
the kernel's algorithmic bytes (image in + field out per padded pixel).  python profiles/tools/edt_bench.py  -> one JSON line per case."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from dgpmp2_amd.utils import sdf_utils
if os.environ.get('DGP_EDT_LIB'):      # tuning aid: a library holding ONLY csrc/sdf_edt.hip (hipcc -shared, seconds to build) instead of the product library
  import ctypes as C
  from dgpmp2_amd import _capi

  class _Shim(object):
    def __init__(self, path):
      self.lib = C.CDLL(path)
      self.sdf_2d_workspace_bytes = self.lib.dgp_sdf_2d_workspace_bytes; self.sdf_2d_workspace_bytes.restype = C.c_size_t
      self.sdf_2d_workspace_bytes.argtypes = [C.c_int32] * 4
      self.sdf_2d = self.lib.dgp_sdf_2d; self.sdf_2d.restype = C.c_int
      self.sdf_2d.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_size_t, C.c_void_p]

    def check(self, rc):
      assert rc == 0, rc

  _shim = _Shim(os.environ['DGP_EDT_LIB'])
  _capi.get_api = lambda: _shim


def images(B, G, seed):
  rs = np.random.RandomState(seed)
  ims = np.ones((B, G, G), dtype=np.float32)
  yy, xx = np.ogrid[:G, :G]
  for b in range(B):
    for _ in range(3 + b % 5):
      cy, cx, r = rs.randint(0, G), rs.randint(0, G), rs.randint(G // 64 + 1, G // 8)
      ims[b][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = 0.0
  return ims


CASES = ((1, 256), (1, 512), (64, 256), (64, 512), (1024, 256), (4096, 256))
if os.environ.get('DGP_EDT_CASES'):      # e.g. "4096x256,64x512"
  CASES = tuple(tuple(int(v) for v in c.split('x')) for c in os.environ['DGP_EDT_CASES'].split(','))
for (B, G) in CASES:
  ims = images(min(B, 64), G, G + B)
  ims = np.tile(ims, (B // ims.shape[0], 1, 1)) if B > ims.shape[0] else ims
  d = torch.as_tensor(ims).cuda()
  for _ in range(3): out = sdf_utils.sdf_2d_batch(d, padlen=1, res=10.0 / G)
  torch.cuda.synchronize()
  reps = 20 if B <= 64 else 5
  t = time.perf_counter()
  for _ in range(reps): out = sdf_utils.sdf_2d_batch(d, padlen=1, res=10.0 / G)
  torch.cuda.synchronize()
  us = (time.perf_counter() - t) / reps * 1e6
  t = time.perf_counter(); ref = sdf_utils.sdf_2d(ims[0], padlen=1, res=10.0 / G); cpu_ms = (time.perf_counter() - t) * 1e3
  same = bool(np.array_equal(out[0].cpu().numpy(), ref))
  alg = B * ((G * G) * 4 + (G + 2) ** 2 * 8)
  print(json.dumps({'batch': B, 'grid': G, 'gpu_us_per_batch': round(us, 1), 'gpu_us_per_image': round(us / B, 2), 'scipy_ms_per_image': round(cpu_ms, 2),
                    'algorithmic_GBs': round(alg / us * 1e-3, 1), 'bit_identical_to_scipy': same, 'variant': os.environ.get('DGP_EDT_TAG', '')}))
