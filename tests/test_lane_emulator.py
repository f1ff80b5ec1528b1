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
