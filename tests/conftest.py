import os, sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
  cache = {}

  def load(name):
    if name not in cache:
      cache[name] = dict(np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False))
    return cache[name]
  return load


def rel_err(a, b):
  """max-norm relative error used by all parity tests: max|a-b| / max(max|b|, tiny)."""
  a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
  return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-300))


def rel_err_per_traj(a, b):
  """Worst per-trajectory max-norm relative error: max_b [ max|a_b - b_b| / max|b_b| ] over the leading (batch) dimension, so a
  trajectory with a small update is held to the same relative bound as one with a large update."""
  a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
  B = b.shape[0]
  num = np.abs(a - b).reshape(B, -1).max(1)
  den = np.maximum(np.abs(b).reshape(B, -1).max(1), 1e-300)
  return float(np.max(num / den))
