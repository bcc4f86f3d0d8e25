"""BASELINE config 4 at its FULL size (1 M nodes, 100 M edges, 172-d edge features, K = 10, 2 layers, Path B,
4095-query batches): the sampler bit-exact against the oracle's C sampler on the same 200 M-entry adjacency, the
native step against the oracle over consecutive batches (embeddings / memory 1e-5, flags and timestamps exact), and
size-independent properties (same batch from the same state twice -> identical bits; eager and per-batch overlay
routes agree).  The 68.8 GB feature table lives in HBM only; the oracle reads its rows on demand.  Needs a real
MI355X (about 80 GB of HBM, 6 GB of host memory)."""
import numpy as np
import pytest
import torch

from conftest import assert_close_fp32
from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch.set_num_threads(1)   # the oracle's History.push is last-write-wins only single-threaded
DEV = torch.device("cuda:0")
N, E, D_E, K, BS = 1_000_000, 100_000_000, 172, 10, 1365


class _DeviceRows:
  """host-side view of a device table for the oracle: rows are copied from HBM on demand (staging only)"""

  def __init__(self, table):
    self.table, self.shape = table, table.shape

  def __getitem__(self, idx):
    return self.table[torch.as_tensor(idx).to(self.table.device)].cpu()


@pytest.fixture(scope="module")
def world():
  from orca_amd import ops, synth
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, DEV)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  ef = synth.edge_features(E, D_E, DEV)
  sd = synth.tgn_state_dict(D_E, 2)
  g = torch.Generator(device=DEV)
  g.manual_seed(11)
  start = int(0.7 * E)
  batches = []
  for b in range(4):
    s = start + b * BS
    batches.append((src[s:s + BS], dst[s:s + BS], torch.randint(1, N + 1, (BS,), generator=g, device=DEV), ts[s:s + BS],
                    eidx[s:s + BS]))
  yield dict(ops=ops, csr=csr, ef=ef, sd=sd, batches=batches, nf=O.OracleNeighborFinder.from_arrays(*csr.arrays_host()))
  torch.cuda.empty_cache()


def test_fullsize_sampler_bit_exact(world):
  ops, csr, nf = world["ops"], world["csr"], world["nf"]
  s, d, n, t, _ = world["batches"][0]
  nodes = torch.cat([s, d, n])
  t3 = torch.cat([t, t, t])
  nbr, eid, et, dt = ops.sample_recent(csr, nodes, t3, K)
  e_nbr, e_eid, e_et = nf.get_temporal_neighbor(nodes.cpu().numpy(), t3.cpu().numpy(), K)
  assert np.array_equal(nbr.cpu().numpy().reshape(-1, K), e_nbr)
  assert np.array_equal(eid.cpu().numpy().reshape(-1, K), e_eid)
  assert np.array_equal(et.cpu().numpy().reshape(-1, K).view(np.uint32), e_et.view(np.uint32))
  e_dt = (t3.cpu().numpy()[:, None] - e_et).astype(np.float32)            # embedding_module.py:128-129
  assert np.array_equal(dt.cpu().numpy().reshape(-1, K).view(np.uint32), e_dt.view(np.uint32))


def test_fullsize_step_vs_oracle_and_properties(world):
  from orca_amd.engine import OrcaEngine
  ops, csr, ef, sd, batches = (world[k] for k in ("ops", "csr", "ef", "sd", "batches"))
  eng = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device=DEV)
  eng.set_neighbor_finder(csr)
  ref = O.OracleTGN(sd, torch.zeros(1, D_E), N + 1, 2, K, True, 0)
  ref.edge_feats = _DeviceRows(ef)
  ref.neighbor_finder = world["nf"]
  for b, bt in enumerate(batches[:3]):
    got = eng.compute_temporal_embeddings(*bt, True).cpu().numpy()
    exp = ref.compute_temporal_embeddings(*(x.cpu().numpy() for x in bt), True).numpy()
    assert_close_fp32(got, exp, 1e-5, f"full-size batch {b} embeddings")
    assert_close_fp32(eng.memory.cpu().numpy(), ref.memory.numpy(), 1e-5, f"full-size batch {b} memory")
    assert np.array_equal(eng.last_update.cpu().numpy(), ref.last_update.numpy())
    assert np.array_equal(eng.pending.cpu().numpy().astype(bool), ref.pending)
    assert_close_fp32(eng.hist[0].cpu().numpy(), ref.hist[0].numpy(), 1e-5, f"full-size batch {b} history")
  # idempotence: the same batch from the same state, twice -> the same bits (the eager route both times)
  from orca_amd import _lib
  assert _lib.load().orca_engine_get_overlay_complete(eng._native), "train-ordered native steps keep the overlay complete"
  bk = eng.backup()
  a = eng.compute_temporal_embeddings(*batches[3], True).clone()
  mem_a = eng.memory.clone()
  eng.restore(bk)          # declares the overlay incomplete: the per-batch work-list route must give the same bits
  c = eng.compute_temporal_embeddings(*batches[3], True).clone()
  assert torch.equal(a, c), "eager and per-batch overlay routes differ at full size"
  assert torch.equal(mem_a, eng.memory)
