"""Host logic of the torch-facing layer (dgpmp2_amd/gpmp2/plan_layer.py) WITHOUT a GPU: the argument marshalling of PlanLayer.forward /
backward / the error helpers is driven with CPU tensors against a recording stand-in for the METH_FASTCALL trampoline (no kernel runs,
no numbers are checked here -- that is what the -m gpu parity tests do through the real C-ABI).  What is pinned: argument order and
count of every entry point as csrc/dgp_pycall.c unpacks them, the shared / per-sample SDF decision, the static-covariance short cut,
the SDF-argument cache (hit on the same tensor, miss after an in-place change / another tensor / another dtype), the flag-buffer reuse
per stream, gradient shapes and which gradient buffers are requested, and that the real trampoline rejects a wrong argument count."""
import gc

import numpy as np
import pytest
import torch

from dgpmp2_amd import _capi
from dgpmp2_amd.gpmp2 import plan_layer as PL
from dgpmp2_amd.robot_models import PointRobot2D


class FakePycall(object):
  """Records (entry point, args); fills nothing."""

  def __init__(self):
    self.calls = []

  def _rec(self, name, n):
    def f(*a):
      assert len(a) == n, (name, len(a), n)
      self.calls.append((name, a))
      return 0
    return f

  def __getattr__(self, name):
    n = {'gn_step': 21, 'gn_solve': 25, 'eval_errors': 22, 'gn_step_backward': 29, 'eval_errors_backward': 28, 'gn_solve_traced': 26,
         'gn_solve_backward': 23, 'gn_step_errors': 24, 'gn_step_errors_backward': 33, 'sum_partial_grids': 8, 'square_covariances': 13,
         'square_covariances_backward': 13}[name]
    return self._rec(name, n)


@pytest.fixture
def layer(monkeypatch):
  monkeypatch.setattr(PL, '_require_cuda', lambda t, name: None)
  monkeypatch.setattr(PL, '_cur_dev', lambda: -1)
  monkeypatch.setattr(PL, '_raw_stream', lambda i: 77)
  t = lambda v: torch.tensor(v, dtype=torch.float64)
  n = 16
  gp = {'Q_c_inv': torch.eye(2, dtype=torch.float64), 'K_s': t(0.01), 'K_g': t(0.01)}
  ob = {'cost_sigma': t(0.01), 'epsilon_dist': t(0.4)}
  pp = {'dof': 2, 'state_dim': 4, 'total_time_sec': 10.0, 'total_time_step': n - 1}
  op = {'method': 'gauss_newton', 'reg': 0.1, 'max_iters': 10, 'tol_err': 1e-3, 'tol_delta': 1e-4}
  pl = PL.PlanLayer(gp, ob, pp, op, {'x_lims': [-5.0, 5.0], 'y_lims': [-5.0, 5.0]}, PointRobot2D(t(0.4), 1, n))
  pl.__dict__['_pc'] = FakePycall()
  return pl


def _inputs(B=3, n=16, G=8, dtype=torch.float32):
  th = torch.randn(B, n, 4, dtype=dtype)
  st, go = torch.randn(B, 1, 4, dtype=dtype), torch.randn(B, 1, 4, dtype=dtype)
  sdf = torch.randn(1, 1, G, G + 2, dtype=dtype)
  return th, st, go, sdf


def test_forward_static_shared_grid_arguments(layer):
  th, st, go, sdf = _inputs()
  sdfb = sdf.expand(3, 1, 8, 10)
  dth, err, eex = layer(th, st, go, None, sdfb, None, None, None)
  (name, a), = layer._pc.calls
  assert name == 'gn_step'
  h = layer._solvers[torch.float32].h
  assert a[0] == h and a[1] == 3 and a[2:5] == (th.data_ptr(), st.data_ptr(), go.data_ptr())
  assert a[5:12] == (sdf.data_ptr(), 8, 10, 0, _capi.DGP_SDF_ROWMAJOR, 0, None)  # shared grid: stride 0; the seven DgpSdf fields
  assert a[12:16] == (_capi.DGP_QC_STATIC, None, None, None)                      # the four DgpCovs fields
  assert a[16:20] == (dth.data_ptr(), err.data_ptr(), eex.data_ptr(), layer.last_info.data_ptr()) and a[20] == 77
  assert dth.shape == th.shape and err.shape == (3, 1, 1) and eex.shape == (3, 1, 1) and layer.last_info.dtype == torch.int32
  assert dth.grad_fn is None                                                     # nothing requires grad: no autograd node


def test_sdf_cache_hits_and_misses(layer):
  th, st, go, sdf = _inputs()
  sdfb = sdf.expand(3, 1, 8, 10)
  layer(th, st, go, None, sdfb, None, None, None)
  entry = layer._sdf_cache
  assert entry is not None and entry[0]() is sdfb
  layer(th, st, go, None, sdfb, None, None, None)
  assert layer._sdf_cache is entry                                               # same tensor object, storage, shape, strides: hit
  i0 = layer.last_info
  # zero-copy entries cannot go stale: the kernel reads sdfb's own storage, so an in-place write (through the tensor or through .data,
  # which does not bump the version counter -- ADVICE r3) needs no re-marshalling, and the address handed over is still the tensor's
  sdf.add_(1.0); sdf.data.mul_(2.0)
  layer(th, st, go, None, sdfb, None, None, None)
  assert layer._sdf_cache is entry and layer._pc.calls[-1][1][5] == sdf.data_ptr()
  assert layer.last_info is i0                                                   # flag buffer reused per (batch, device, stream) without grad
  # a resize / restride of the same tensor object is a miss
  sdfb2 = sdf.expand(3, 1, 8, 10)
  layer(th, st, go, None, sdfb2, None, None, None)
  assert layer._sdf_cache is not entry
  # a float64 grid with float32 trajectories needs a converted copy: made on EVERY call and never cached (it would not see later writes)
  sdf64 = torch.randn(3, 1, 8, 10, dtype=torch.float64)
  before = layer._sdf_cache
  layer(th, st, go, None, sdf64, None, None, None)
  a = layer._pc.calls[-1][1]
  assert a[5] != sdf64.data_ptr() and a[6:9] == (8, 10, 80)                      # per-sample grids: stride H*W elements
  assert layer._sdf_cache is before
  sdf64.data.zero_()                                                             # invisible to the version counter ...
  sd = layer._sdf_args(sdf64, torch.float32, 3, -1)
  assert float(sd[7].abs().max()) == 0.0                                         # ... but the next call converts again and sees it
  # the cached entry dies with its tensor
  layer(th, st, go, None, sdfb, None, None, None)
  assert layer._sdf_cache is not None
  del sdfb, sdfb2, entry, before
  gc.collect()
  assert layer._sdf_cache is None
  # fewer grids than trajectories would be read out of bounds
  with pytest.raises(ValueError):
    layer(th, st, go, None, torch.randn(2, 1, 8, 10), None, None, None)


def test_info_buffer_is_fresh_under_grad(layer):
  th, st, go, sdf = _inputs()
  sdfb = sdf.expand(3, 1, 8, 10)
  th.requires_grad_(True)
  layer(th, st, go, None, sdfb, None, None, None); i0 = layer.last_info
  layer(th, st, go, None, sdfb, None, None, None); i1 = layer.last_info
  assert i0 is not i1 and i0.data_ptr() != i1.data_ptr()                         # a training step keeps its own flags
  th2 = th.detach()
  layer(th2, st, go, None, sdfb, None, None, None); j0 = layer.last_info
  layer(th2, st, go, None, sdfb, None, None, None); j1 = layer.last_info
  assert j0 is j1                                                                # no autograd node: one buffer per (batch, device, stream)


def test_per_state_covariances_and_backward_arguments(layer):
  B, n = 3, 16
  th, st, go, sdf = _inputs(B)
  th.requires_grad_(True)
  sdfb = sdf.expand(B, 1, 8, 10).clone().requires_grad_(True)                     # per-sample grids with a gradient
  qc = torch.eye(2).expand(B, n - 1, 2, 2).contiguous().requires_grad_(True)
  ow = torch.full((B, n, 1, 1), 1e4, requires_grad=True)
  eps = torch.full((B, n, 1, 1), 0.4)                                             # no gradient asked for eps
  dth, err, eex = layer(th, st, go, None, sdfb, qc, ow, eps)
  name, a = layer._pc.calls[-1]
  assert name == 'gn_step' and a[12:16] == (_capi.DGP_QC_PERSTATE, qc.data_ptr(), ow.data_ptr(), eps.data_ptr())
  assert dth.grad_fn is not None and not err.requires_grad and eex.requires_grad
  (dth.sum() + eex.sum()).backward()
  name, b = layer._pc.calls[-1]
  assert name == 'gn_step_backward'
  assert b[:16] == a[:16]                                                         # same inputs as the forward launch (a small dense per-sample gradient: grad_mode DGP_GSDF_DENSE)
  assert b[16] == dth.data_ptr() and b[17] is not None and b[18] is not None     # dtheta, both cotangents
  assert b[19] is not None and b[20] is None and b[21] is None                   # g_th only (start / goal do not require grad)
  assert b[22] is not None and b[23] == 80 and b[24] == 1                        # per-sample SDF gradient: stride H*W, one copy
  assert b[25] is not None and b[26] is not None and b[27] is None and b[28] == 77   # g_qc, g_ow, no g_eps; stream
  assert th.grad.shape == th.shape and sdfb.grad.shape == sdfb.shape and qc.grad.shape == qc.shape and ow.grad.shape == ow.shape
  assert eps.grad is None


def test_shared_grid_gradient_uses_partial_copies(layer):
  B = 512                                                                      # (B n >= 8192 taps: one partial grid per XCD; a small batch accumulates into a single grid)
  th, st, go, sdf = _inputs(B)
  sdf.requires_grad_(True)
  dth, err, eex = layer(th, st, go, None, sdf.expand(B, 1, 8, 10), None, None, None)
  dth.sum().backward()
  name, b = layer._pc.calls[-1]
  name, b = layer._pc.calls[-2]                                                  # (the last call sums the partial copies)
  assert name == 'gn_step_backward' and b[17] is not None and b[18] is None     # err_ext unused: no cotangent materialised
  assert b[19] is None and b[22] is not None and b[23] == 0 and b[24] == PL._SDF_GRAD_COPIES
  assert b[10] == _capi.DGP_GSDF_DENSE_F64 and b[11] is None                     # double partial grids whatever the I/O type
  name, c = layer._pc.calls[-1]
  # sdfb = sdf.expand(...) of a (1,1,H,W) tensor: the node differentiates w.r.t. the BASE (no B-fold sum in autograd's ExpandBackward): unscaled sum of the copies
  assert name == 'sum_partial_grids' and c[0] == b[22] and c[1:5] == (_capi.DGP_F64, PL._SDF_GRAD_COPIES, 80, 1.0) and c[6] == _capi.DGP_F32 and c[7] == 77
  assert sdf.grad.shape == sdf.shape
  # an expanded view whose base is not a (1,1,H,W) tensor: B equal shares for autograd's expand-backward to sum
  flat = torch.randn(8, 10, requires_grad=True)
  dth, err, eex = layer(th, st, go, None, flat[None, None].expand(B, 1, 8, 10), None, None, None)
  dth.sum().backward()
  assert layer._pc.calls[-1][0] == 'sum_partial_grids' and layer._pc.calls[-1][1][4] == 1.0 / B and flat.grad.shape == flat.shape
  th3, st3, go3, _ = _inputs(3)
  sdf3 = torch.randn(1, 1, 8, 10, requires_grad=True)
  layer(th3, st3, go3, None, sdf3.expand(3, 1, 8, 10), None, None, None)[0].sum().backward()
  assert layer._pc.calls[-2][1][24] == 1 and layer._pc.calls[-1][1][2] == 1      # 48 taps: one grid, device-scope atomics (still float64, still cast by the sum kernel)


def test_per_sample_grid_gradient_as_sparse_taps(layer):
  """layer.sdf_grad = 'sparse' (and 'auto' for large leaf grids): no (B,1,H,W) zero fill -- the launch gets tap value / index arrays (DGP_GSDF_SPARSE) and
  autograd a sparse COO tensor of sdfb's shape."""
  B, n = 3, 16
  th, st, go, sdf = _inputs(B)
  sdfb = sdf.expand(B, 1, 8, 10).clone().requires_grad_(True)
  layer.sdf_grad = 'sparse'
  dth, err, eex = layer(th, st, go, None, sdfb, None, None, None)
  dth.sum().backward()
  name, b = layer._pc.calls[-1]
  assert name == 'gn_step_backward' and b[10] == _capi.DGP_GSDF_SPARSE and b[11] is not None and b[22] is not None and b[23] == 80 and b[24] == 1
  assert sdfb.grad.is_sparse and sdfb.grad.shape == sdfb.shape and sdfb.grad._nnz() == B * n * 4 and sdfb.grad._indices().data_ptr() == b[11]
  # 'auto' keeps the reference's dense layout for a small gradient ...
  layer.sdf_grad = 'auto'
  sdfb.grad = None
  dth, err, eex = layer(th, st, go, None, sdfb, None, None, None)
  dth.sum().backward()
  assert layer._pc.calls[-1][1][10] == _capi.DGP_GSDF_DENSE and not sdfb.grad.is_sparse
  # ... and goes sparse once the dense one would be large (leaf grids only)
  old = PL._SPARSE_MIN_DENSE_BYTES
  try:
    PL._SPARSE_MIN_DENSE_BYTES = 0
    big = torch.randn(B, 1, 64, 64).requires_grad_(True)
    dth, err, eex = layer(th, st, go, None, big, None, None, None)
    dth.sum().backward()
    assert layer._pc.calls[-1][1][10] == _capi.DGP_GSDF_SPARSE and big.grad.is_sparse
    big.grad = None                                  # (the recording stand-in fills nothing: a sparse gradient with uninitialised indices must not be summed)
    nonleaf = big * 1.0
    dth, err, eex = layer(th, st, go, None, nonleaf, None, None, None)
    dth.sum().backward()
    assert layer._pc.calls[-1][1][10] == _capi.DGP_GSDF_DENSE
  finally:
    PL._SPARSE_MIN_DENSE_BYTES = old


def test_error_helpers_arguments(layer):
  B = 3
  th, st, go, sdf = _inputs(B)
  with pytest.raises(RuntimeError):
    layer.error_batch(th, sdf)                                                   # forward() first, like the reference
  eps = torch.full((B, 16, 1, 1), 0.3, requires_grad=True)
  layer(th, st, go, None, sdf, None, None, eps)
  e = layer.error_batch(th, sdf)
  name, a = layer._pc.calls[-1]
  assert name == 'eval_errors' and a[15] == eps.data_ptr() and a[16] == e.data_ptr() and e.grad_fn is None
  g = layer.gp_error(th)
  name, a = layer._pc.calls[-1]
  assert a[5] is None and a[16] is None and a[17] is None and a[20] is None and a[19] == g.data_ptr()    # no grid: none of the outputs that read it
  sg, gp, ob = layer.unweighted_errors(th.clone().requires_grad_(True), sdf)
  assert sg.shape == (B, 1) and gp.shape == (B, 1, 1) and ob.requires_grad
  (sg.sum() + ob.sum()).backward()
  name, b = layer._pc.calls[-1]
  assert name == 'eval_errors_backward' and b[16] is None and b[17] is not None and b[18] is None and b[19] is not None
  assert b[20] is not None and b[23] is None and b[26] is not None              # g_th, no SDF gradient, g_eps (the current eps carries a graph)
  assert eps.grad is not None and eps.grad.shape == eps.shape


def test_input_validation(layer):
  th, st, go, sdf = _inputs()
  with pytest.raises(ValueError):
    layer(th[:, :5], st, go, None, sdf, None, None, None)
  with pytest.raises(TypeError):
    layer(th, st.double(), go, None, sdf, None, None, None)
  with pytest.raises(ValueError):
    layer(th, st, go, None, sdf, torch.eye(2).expand(2, 15, 2, 2), None, None)   # covariance batch != trajectory batch


def test_real_trampoline_argument_counts():
  pc = _capi.get_pycall()
  for name, n in (('gn_step', 21), ('gn_solve', 25), ('eval_errors', 22), ('gn_step_backward', 29), ('eval_errors_backward', 28), ('sum_partial_grids', 8),
                  ('square_covariances', 13), ('square_covariances_backward', 13), ('gn_step_errors', 24), ('gn_step_errors_backward', 33)):
    with pytest.raises(TypeError):
      getattr(pc, name)(*([0] * (n - 1)))
  # a NULL handle comes back as the C-ABI's DGP_EINVAL, not as a crash
  assert pc.gn_step(0, 1, 0, 0, 0, 0, 2, 2, 0, 0, 0, None, 0, None, None, None, 0, 0, 0, 0, 0) == _capi.DGP_EINVAL
  assert b'null' in _capi.get_api().last_error()


def test_inplace_change_between_forward_and_backward_raises(layer):
  """The inputs are not SavedVariables (their unpacking costs more than the launch): the version-counter check autograd would do is
  done by hand."""
  th, st, go, sdf = _inputs()
  th.requires_grad_(True)
  dth, err, eex = layer(th, st, go, None, sdf, None, None, None)
  st.add_(1.0)
  with pytest.raises(RuntimeError, match='modified by an inplace operation'):
    dth.sum().backward()
  dth, err, eex = layer(th, st, go, None, sdf, None, None, None)
  dth.sum().backward()                                                            # untouched inputs: fine
  assert th.grad.shape == th.shape


def test_tiled_grid_needs_its_logical_size(layer):
  """a 6-D tiled grid without a logical size is refused, not read as 4 Ht x 4 Wt.  The size rides on utils.sdf_utils.TiledSdf through .to() / .detach() /
  clone() / indexing / expand() / cat / stack; a plain tensor needs plan_layer.sdf_hw (or as_tiled)."""
  from dgpmp2_amd.utils.sdf_utils import tile_sdf, untile_sdf, as_tiled, tiled_hw
  th, st, go, _ = _inputs()
  sdf = torch.randn(3, 1, 10, 13, dtype=torch.float32)             # 10 x 13: 3 x 4 tiles, which could as well hold 12 x 16
  t = tile_sdf(sdf)
  assert tuple(t.shape) == (3, 1, 3, 4, 4, 4) and t.hw == (10, 13) and torch.equal(untile_sdf(t), sdf)
  for u in (t.to(torch.float64).float(), t.clone(), t.detach(), t[0:3], torch.cat([t[:1], t[1:]], 0), torch.stack([t[0], t[1], t[2]], 0), t.contiguous(), t * 1.0):
    assert tiled_hw(u) == (10, 13), type(u)
  assert tiled_hw(t.sum()) is None and tiled_hw(t.reshape(3, -1)) is None   # not tiles any more: a plain tensor
  layer(th, st, go, None, t, None, None, None)
  a = layer._pc.calls[-1][1]
  assert a[5:10] == (t.data_ptr(), 10, 13, 3 * 4 * 16, _capi.DGP_SDF_TILED4)
  plain = t.as_subclass(torch.Tensor)
  assert tiled_hw(plain) is None
  with pytest.raises(ValueError, match='carries no logical grid size'):
    layer(th, st, go, None, plain, None, None, None)
  layer.sdf_hw = (10, 13)                                           # the explicit size of plain tiled tensors
  layer(th, st, go, None, plain, None, None, None)
  assert layer._pc.calls[-1][1][6:8] == (10, 13)
  layer.sdf_hw = (10, 17)                                           # ... is checked against the tile counts
  with pytest.raises(ValueError, match='does not hold'):
    layer(th, st, go, None, plain, None, None, None)
  layer.sdf_hw = None
  layer(th, st, go, None, as_tiled(plain, (10, 13)), None, None, None)
  assert layer._pc.calls[-1][1][6:8] == (10, 13)
  with pytest.raises(ValueError):
    as_tiled(plain, (12, 17))
  # a shared tiled grid: the expand()ed view keeps the size, and so does its base when the node differentiates w.r.t. it
  one = tile_sdf(sdf[:1]).requires_grad_(True)
  dth, err, eex = layer(th.clone().requires_grad_(True), st, go, None, one.expand(3, 1, 3, 4, 4, 4), None, None, None)
  assert layer._pc.calls[-1][1][5:9] == (one.data_ptr(), 10, 13, 0)


def test_per_sample_grid_gradient_is_dense_by_default(layer):
  """the reference's layout (a dense tensor of sdfb's shape) unless the caller opts into 'sparse' / 'auto'."""
  assert layer.sdf_grad == 'dense'
  th, st, go, _ = _inputs()
  old = PL._SPARSE_MIN_DENSE_BYTES
  try:
    PL._SPARSE_MIN_DENSE_BYTES = 0
    big = torch.randn(3, 1, 64, 64).requires_grad_(True)
    dth, err, eex = layer(th, st, go, None, big, None, None, None)
    dth.sum().backward()
    assert layer._pc.calls[-1][1][10] == _capi.DGP_GSDF_DENSE and not big.grad.is_sparse and big.grad.shape == big.shape
  finally:
    PL._SPARSE_MIN_DENSE_BYTES = old


def test_expanded_grid_view_with_retain_grad_is_differentiated_as_it_is(layer):
  """the node swaps an expand()ed shared grid for its base (no B-fold sum) -- unless the VIEW itself is wanted as a gradient target."""
  th, st, go, sdf = _inputs()
  base = sdf.clone().requires_grad_(True)
  view = base.expand(3, 1, 8, 10)
  assert PL._expand_base(view) is base
  view2 = base.expand(3, 1, 8, 10); view2.retain_grad()
  assert PL._expand_base(view2) is view2
  view3 = base.expand(3, 1, 8, 10); view3.register_hook(lambda g: g)
  assert PL._expand_base(view3) is view3
  dth, err, eex = layer(th.clone().requires_grad_(True), st, go, None, view2, None, None, None)
  dth.sum().backward()
  assert view2.grad is not None and view2.grad.shape == view2.shape and base.grad.shape == base.shape


def test_auto_tile_tiles_a_per_sample_batch_once(layer):
  """plan_layer.auto_tile: a per-sample row-major sdfb is tiled once per (tensor, storage, version) and the launches read the tiles; the gradient comes back in sdfb's
  own row-major shape (dense: un-tiled; sparse: four row-major index rows); shared / already tiled / single grids are left alone."""
  from dgpmp2_amd.utils.sdf_utils import tile_sdf
  th, st, go, sdf = _inputs()
  per = torch.randn(3, 1, 10, 13, dtype=torch.float32)
  layer(th, st, go, None, per, None, None, None)
  assert layer._pc.calls[-1][1][9] == _capi.DGP_SDF_ROWMAJOR and layer._tile_cache is None      # off by default
  layer.auto_tile = True
  layer(th, st, go, None, per, None, None, None)
  a = layer._pc.calls[-1][1]
  tiles = layer._tile_cache[5]
  assert a[5:10] == (tiles.data_ptr(), 10, 13, 3 * 4 * 16, _capi.DGP_SDF_TILED4) and torch.equal(tiles, tile_sdf(per))
  layer(th, st, go, None, per, None, None, None)
  assert layer._tile_cache[5] is tiles and layer._pc.calls[-1][1][5] == tiles.data_ptr()          # same tensor, same version: no second tiling pass
  per.add_(1.0)                                                                                    # an in-place change bumps the version: tiled again
  layer(th, st, go, None, per, None, None, None)
  assert layer._tile_cache[5] is not tiles and torch.equal(layer._tile_cache[5], tile_sdf(per))
  layer.error_batch(th, per)                                                                       # the error helpers read the same tiles
  assert layer._pc.calls[-1][0] == 'eval_errors' and layer._pc.calls[-1][1][9] == _capi.DGP_SDF_TILED4
  layer(th, st, go, None, sdf.expand(3, 1, 8, 10), None, None, None)                              # a shared grid stays as it is (it lives in L2)
  assert layer._pc.calls[-1][1][9] == _capi.DGP_SDF_ROWMAJOR
  # gradients arrive in the row-major shape of the tensor the caller holds
  leaf = per.clone().requires_grad_(True)
  dth, err, eex = layer(th.clone().requires_grad_(True), st, go, None, leaf, None, None, None)
  dth.sum().backward()
  b = layer._pc.calls[-1][1]
  assert layer._pc.calls[-1][0] == 'gn_step_backward' and b[9] == _capi.DGP_SDF_TILED4 and b[10] == _capi.DGP_GSDF_DENSE
  assert leaf.grad.shape == leaf.shape and not leaf.grad.is_sparse
  # the index conversion of a sparse tiled gradient (b,0,y/4,x/4,y%4,x%4) -> (b,0,y,x)  (the recording stand-in fills no indices: the sparse launch itself runs on the GPU tests)
  idx = torch.tensor([[0, 2], [0, 0], [1, 2], [3, 0], [2, 1], [0, 3]])
  g = torch.sparse_coo_tensor(idx, torch.tensor([1.5, -2.0]), (3, 1, 3, 4, 4, 4), check_invariants=False)

  class Ctx(object): hw = (10, 13); shape = (3, 1, 10, 13)
  with torch.no_grad():
    d = PL._AutoTile.backward(Ctx, g)[0].to_dense()
  assert d.shape == (3, 1, 10, 13) and d[0, 0, 6, 12] == 1.5 and d[2, 0, 9, 3] == -2.0 and d.abs().sum() == 3.5
