import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
  return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
  cache = {}

  def get(name):
    if name not in cache:
      cache[name] = load_golden(name)
    return cache[name]
  return get


def rel_err(a, b):
  """Normwise relative error ||a-b|| / max(||b||, tiny)."""
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def assert_close_fp32(a, b, tol=1e-5, what=""):
  """north_star tolerance: 1e-5 relative (fp32).  Checked normwise AND elementwise against
  tol * max(1, max|b|) so that near-zero elements of an O(1) vector do not dominate."""
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  assert a.shape == b.shape, (what, a.shape, b.shape)
  if a.size == 0:
    return
  scale = max(1.0, float(np.max(np.abs(b))))
  err = float(np.max(np.abs(a - b)))
  assert err <= tol * scale, f"{what}: max abs err {err:.3e} > {tol * scale:.3e}"
  r = rel_err(a, b)
  assert r <= tol, f"{what}: normwise rel err {r:.3e} > {tol:.1e}"
