"""CPU-only checks of the kernel logic: the per-lane program the HIP kernel runs (dgpmp2_amd/csrc/gn_lane.h) is
compiled for the host and executed by tests/emul's 64-thread wavefront emulator, through the same C-ABI
marshalling as the product, and compared with the oracle and the reference's golden fixtures."""
import pytest
import harness
import parity_cases as PC


@pytest.fixture(scope='module')
def be():
  return harness.Backend('emul')


# the emulator runs every lane as a thread: keep the slow cases trimmed
def test_emul_c2mini_static(be, golden): PC.case_c2mini_static(be, golden, 'f64', steps=(0, 9), nb=2)
def test_emul_c2mini_static_f32(be, golden): PC.case_c2mini_static(be, golden, 'f32', steps=(4,), nb=2)
def test_emul_c2mini_covs(be, golden): PC.case_c2mini_covs(be, golden, 'f64', nb=2)
def test_emul_per_sample_sdf(be, golden): PC.case_c2mini_per_sample_sdf(be, golden, 'f64', nb=3)
def test_emul_c1(be, golden): PC.case_c1(be, golden, 'f64', steps=(0, 9))
def test_emul_small_ragged(be, golden): PC.case_small_ragged(be, golden, 'f64')
def test_emul_small_ragged_f32(be, golden): PC.case_small_ragged(be, golden, 'f32')
def test_emul_edges(be, golden): PC.case_edges(be, golden, 'f64')
def test_emul_c3_vel(be, golden): PC.case_c3_vel(be, golden, 'f64')
def test_emul_c4_xyh(be, golden): PC.case_c4_xyh(be, golden, 'f64')
def test_emul_eval_errors(be, golden): PC.case_eval_errors(be, golden, 'f64')
def test_emul_solve(be, golden): PC.case_solve(be, golden, 'f64')
def test_emul_not_spd(be, golden): PC.case_not_spd(be, golden, 'f64')
def test_emul_backward_golden(be, golden): PC.case_backward_golden(be, golden, 'f64')
def test_emul_backward_golden_f32(be, golden): PC.case_backward_golden(be, golden, 'f32')
def test_emul_backward_fd(be, golden): PC.case_backward_fd(be, golden, 'f64')
def test_emul_tiny_sizes(be, golden): PC.case_tiny_and_odd_sizes(be, golden, 'f64')
def test_emul_sdf_grad_copies(be, golden): PC.case_shared_sdf_gradient_partial_copies(be, golden, 'f64')
def test_emul_static_qc_variants(be, golden): PC.case_static_qc_variants(be, golden, 'f64')
def test_emul_scalar_covariances(be, golden): PC.case_scalar_covariances(be, golden, 'f64')
def test_emul_scalar_covariances_f32(be, golden): PC.case_scalar_covariances(be, golden, 'f32')
def test_emul_unaligned(be, golden): PC.case_unaligned_buffers(be, golden, 'f64')
def test_emul_unaligned_f32(be, golden): PC.case_unaligned_buffers(be, golden, 'f32')
def test_emul_solve_with_covariances(be, golden): PC.case_solve_with_covariances(be, golden, 'f64')
def test_emul_eval_errors_backward(be, golden): PC.case_eval_errors_backward(be, golden, 'f64')
def test_emul_eval_errors_backward_f32(be, golden): PC.case_eval_errors_backward(be, golden, 'f32')
def test_emul_solve_backward(be, golden): PC.case_solve_backward(be, golden, 'f64')
def test_emul_solve_backward_f32(be, golden): PC.case_solve_backward(be, golden, 'f32')
def test_emul_solve_backward_general_qc(be, golden): PC.case_solve_backward_general_qc(be, golden, 'f64')
def test_emul_step_errors(be, golden): PC.case_step_errors(be, golden, 'f64')
def test_emul_step_errors_f32(be, golden): PC.case_step_errors(be, golden, 'f32')
def test_emul_sdf_gradient_delivery(be, golden): PC.case_sdf_gradient_delivery(be, golden, 'f64')
def test_emul_sdf_gradient_delivery_f32(be, golden): PC.case_sdf_gradient_delivery(be, golden, 'f32')
def test_emul_tiled_grids(golden):
  """DgpSdf::layout = DGP_SDF_TILED4 (round 5): parity cases re-run with every grid stored as 4 x 4 tiles and the dense grid gradients untiled -- per-sample and
  shared grids, non-square / odd-sized grids (padding cells), trajectories leaving the grid, the backward pass with partial copies and the training iteration."""
  bt = harness.Backend('emul'); bt.sdf_tiled = True
  PC.case_c2mini_per_sample_sdf(bt, golden, 'f64', nb=3)
  PC.case_edges(bt, golden, 'f64')
  PC.case_c1(bt, golden, 'f64', steps=(0,))
  PC.case_backward_golden(bt, golden, 'f64')
  PC.case_backward_golden(bt, golden, 'f32')
  PC.case_shared_sdf_gradient_partial_copies(bt, golden, 'f64')
  PC.case_sdf_gradient_delivery(bt, golden, 'f32')
  PC.case_step_errors(bt, golden, 'f64')
def test_emul_raw_squared_covariances(be, golden): PC.case_raw_squared_covariances(be, golden, 'f64')
def test_emul_raw_squared_covariances_f32(be, golden): PC.case_raw_squared_covariances(be, golden, 'f32')


# ---- launch shapes: LPT lanes per trajectory x C states per lane (local block elimination + PCR over the lanes)
import os
import numpy as np
from conftest import rel_err
from oracle import gpmp2_oracle as O


@pytest.fixture
def force_shape():
  def setter(shape):
    if shape is None: os.environ.pop('DGP_FORCE_SHAPE', None)
    else: os.environ['DGP_FORCE_SHAPE'] = shape
  yield setter
  os.environ.pop('DGP_FORCE_SHAPE', None)


@pytest.mark.parametrize('shape', ['64,1', '32,2', '16,4', '64,2', '32,4', '64,4'])
def test_emul_shapes_n64(be, golden, force_shape, shape):
  force_shape(shape)
  g = golden('g3_c2mini')
  p = PC.P2d(64)
  sdf = O.circles_sdf(int(g['G']), g['circles'])[None, None]
  PC.check_step(be, p, g['th_hist'][3][:2], g['start'][:2], g['goal'][:2], sdf, 'f64',
                ref=(g['dth_hist'][3][:2], g['err_hist'][3][:2], g['errext_hist'][3][:2]), tag='shape ' + shape)


@pytest.mark.parametrize('shape', ['64,2', '32,4', '64,4', None])
def test_emul_n101_reference_default_length(be, golden, force_shape, shape):
  """n = 101 = the reference YAML's total_time_step=100 (examples/configs/gpmp2_2d_params.yaml:6): more states than lanes."""
  force_shape(shape)
  g = golden('g3_c1')
  p = O.OracleParams(dof=2, total_time_step=100)
  th0 = O.straight_line_trajb(g['start'][:, :, :2], g['goal'][:, :, :2], 10.0, 100, 2)
  dth, err, eex, info = be.step(p, th0, g['start'], g['goal'], g['sdf'][None, None], io='f64')
  assert rel_err(dth, g['n101_dth0']) < 1e-9 and rel_err(err, g['n101_err0'].reshape(-1)) < 1e-11 and info[0] == 0
  assert abs(float(err[0]) - 330.436499542839) < 1e-8          # survey known-answer


def test_emul_shapes_xyh_and_solve(be, golden, force_shape):
  force_shape('16,2')
  PC.case_c4_xyh(be, golden, 'f64')
  PC.case_c3_vel(be, golden, 'f64')
  force_shape('16,4')
  PC.case_solve(be, golden, 'f64')
  PC.case_eval_errors(be, golden, 'f64')


def test_force_shape_rejects_unsupported(force_shape):
  from dgpmp2_amd import _capi
  force_shape('16,2')          # 32 rows < n = 64
  with pytest.raises(_capi.DgpError):
    _capi.Solver(harness.config_from_oracle(PC.P2d(64), 'f64'), api=harness.emul_api())


@pytest.mark.parametrize('shape', ['16,2', '16,4', '32,2'])
def test_emul_backward_shapes(be, golden, force_shape, shape):
  force_shape(shape)
  PC.case_backward_golden(be, golden, 'f64')


def test_emul_woodbury_kernels(be, golden):
  """gn_woodbury.h on the emulator: the smallest shape in full, the two larger ones with one robot each (64 threads per wavefront)."""
  PC.case_woodbury_kernels(be, golden, 'f64', shapes=('16,4',), nb=2)


def test_emul_woodbury_kernels_f32_and_wide_shapes(be, golden):
  PC.case_woodbury_kernels(be, golden, 'f32', shapes=('32,4',), nb=1)


def test_emul_long_trajectories(be, golden):
  """n > 256 through the loop kernels of gn_long.h (see parity_cases.case_long_trajectories), trimmed for the thread-per-lane emulator."""
  PC.case_long_trajectories(be, golden, 'f64', configs=[(2, 300, 'perstate', dict(use_vel_limits=True, K_v=0.01, v_x=0.5, v_y=0.5)),
                                                        (3, 257, 'static', dict(non_holonomic=True, K_d=0.05)), (3, 320, 'qfull', {})])


def test_emul_long_trajectories_f32(be, golden):
  PC.case_long_trajectories(be, golden, 'f32', configs=[(2, 300, 'static', {})])
