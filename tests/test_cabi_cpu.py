"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/orca_hip.h declares,
the ctypes table covers the header, and the host-only MRD planner (no GPU involved) matches the oracle
and the reference's golden plans."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
  text = open(os.path.join(ROOT, "include", "orca_hip.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(orca_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
  from orca_amd import _lib
  names = _header_functions()
  assert len(names) >= 25
  lib = ctypes.CDLL(_lib.LIB_PATH)
  for n in names:
    assert hasattr(lib, n), f"{n} declared in include/orca_hip.h but not exported"
  assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and header disagree"
  assert _lib.load().orca_version() >= 1


def test_no_cpu_fallback():
  """The product path must fail loudly without a HIP device."""
  import torch
  from orca_amd import _lib, ops
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  with pytest.raises(_lib.OrcaError):
    ops.sample_recent(None, torch.zeros(3, dtype=torch.int64), torch.zeros(3, dtype=torch.float64), 5)
  from orca_amd.engine import OrcaEngine
  with pytest.raises(_lib.OrcaError):
    OrcaEngine({}, np.zeros((2, 172), np.float32), 10, 2, 10, device="cpu")


def _plan(src, dst, ts, eidx, n_nodes, bs, K, budget):
  from orca_amd import _lib
  from orca_amd._lib import check, ptr
  nf = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=n_nodes)
  E = len(src)
  nbr = [np.ascontiguousarray(nf.get_temporal_neighbor(x, ts, K)[0]) for x in (src, dst)]
  nb = -(-E // bs)
  plan = np.zeros(max(nb * budget, 1), np.int64)
  plan_len = np.zeros(nb, np.int64)
  stats = np.zeros(2, np.int64)
  avg = np.zeros(1)
  s64, d64 = np.ascontiguousarray(src, np.int64), np.ascontiguousarray(dst, np.int64)
  check(_lib.load().orca_mrd_plan_host(E, bs, K, ptr(s64), ptr(d64), ptr(nbr[0]), ptr(nbr[1]), n_nodes + 1, budget,
                                       ptr(plan), ptr(plan_len), ptr(stats), ptr(avg)), "orca_mrd_plan_host")
  return nf, [plan[b * budget:b * budget + plan_len[b]] for b in range(nb)], stats, avg[0]


@pytest.mark.parametrize("tag", ["tight", "loose", "mid"])
def test_mrd_planner_host(golden, tag):
  g = golden("mrd.npz")
  n_nodes, n_edges, bs, K, budget, num_batch = [int(x) for x in g[f"{tag}_meta"]]
  src, dst, ts, eidx = (g[f"{tag}_{k}"] for k in ("src", "dst", "ts", "eidx"))
  nf, plans, stats, avg = _plan(src, dst, ts, eidx, n_nodes, bs, K, budget)
  assert len(plans) == num_batch
  assert np.array_equal(np.array([len(p) for p in plans]), g[f"{tag}_plan_len"])
  nb, tl, nl, occ = O.read_batches(src, dst, ts, nf, bs, K, n_nodes + 1)
  ref, r, c, a = O.mrd_plan(n_nodes + 1, nb, budget, tl, nl, occ, stable=True)
  for x, y in zip(plans, ref):
    assert np.array_equal(x, y)
  assert (int(stats[0]), int(stats[1])) == (r, c)
  assert abs(avg - a) < 1e-12 or (np.isnan(avg) and np.isnan(a))
  assert int(stats[0] + stats[1]) == int(g[f"{tag}_stats"][0] + g[f"{tag}_stats"][1])
  if tag == "loose":     # tie-free at the cut: identical to the reference's own plan and statistics
    off = np.concatenate([[0], np.cumsum(g[f"{tag}_plan_len"])])
    for i, p in enumerate(plans):
      assert np.array_equal(np.sort(p), np.sort(g[f"{tag}_plan_flat"][off[i]:off[i + 1]]))
    assert [int(stats[0]), int(stats[1])] == list(g[f"{tag}_stats"][:2])
    assert abs(avg - float(g[f"{tag}_avg_reuse_distance"][0])) < 1e-12


def test_mrd_planner_edge_cases():
  from orca_amd import _lib
  e = np.zeros(0, np.int64)
  n32 = np.zeros((0, 5), np.int32)
  rc = _lib.load().orca_mrd_plan_host(0, 10, 5, None, None, None, None, 10, 3, None, None, None, None)
  assert rc == 0
  src = np.array([1, 2, 99], np.int64)
  with pytest.raises(IndexError):
    _plan(src, np.array([2, 3, 1], np.int64), np.array([1.0, 2.0, 3.0]), np.array([1, 2, 3], np.int64), 200, 2, 3, 2)[0] \
        and _lib.check(_lib.load().orca_mrd_plan_host(3, 2, 3, _lib.ptr(src), _lib.ptr(src), _lib.ptr(np.zeros((3, 3), np.int32)),
                                                      _lib.ptr(np.zeros((3, 3), np.int32)), 50, 2,
                                                      _lib.ptr(np.zeros(4, np.int64)), _lib.ptr(np.zeros(2, np.int64)), None, None),
                       "orca_mrd_plan_host")
  # budget 0 -> empty plans
  _, plans, _, _ = _plan(np.array([1, 2, 3, 1], np.int64), np.array([2, 3, 1, 3], np.int64), np.array([1.0, 2.0, 3.0, 4.0]),
                         np.array([1, 2, 3, 4], np.int64), 5, 2, 3, 0)
  assert all(len(p) == 0 for p in plans)


def test_torch_custom_ops_are_registered():
  """orca_amd/torch_ops.py registers the entry points as torch.library custom ops (namespace orca) with autograd
  formulas for the attention and GRU ops; without a HIP device a call must fail loudly (no CPU fallback)."""
  import torch
  from orca_amd import torch_ops
  from orca_amd._lib import OrcaError
  for name in torch_ops.OPS:
    assert hasattr(torch.ops.orca, name), name
  schema = str(torch.ops.orca.tgat_gather.default._schema)
  assert "Tensor[] params" in schema and "Tensor? upd_rows" in schema
  assert "Tensor(a" not in str(torch.ops.orca.gru_cell.default._schema)      # functional: it carries an autograd formula
  x, h = torch.zeros(3, 472), torch.zeros(3, 100)
  w = [torch.zeros(300, 472), torch.zeros(300, 100), torch.zeros(300), torch.zeros(300)]
  try:
    torch.ops.orca.gru_cell(*w, torch.zeros(8), x, h)
  except (OrcaError, RuntimeError, OSError, AssertionError):
    pass
  else:
    raise AssertionError("a CPU call of orca::gru_cell must not succeed")
