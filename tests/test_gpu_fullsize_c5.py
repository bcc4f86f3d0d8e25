"""BASELINE config 5 at its FULL size on one GPU: 10 M nodes, 1 B edges (2 B adjacency entries, 32 GB of CSR), K = 20
seeded-uniform sampling, zero-feature edges (d_e = 1), 2 layers with reuse, 4095-query batches.  The sampler and the
native step are checked against the oracle on the adjacency lists of the queried nodes (copied from HBM row by row:
the whole adjacency would be 32 GB of host arrays), on the rows the batches touch.  Uniform sampling is not in the
reference (parity unpinned; GPU == the oracle's CPU twin bit for bit).  The 8-rank sharding of this config's state
is exercised at small size by tests/test_gpu_multirank.py (uniform K = 20, d_e = 1, 4 ranks); one GPU holds the
replicated state of this config (about 60 GB) with room to spare.  Needs a real MI355X (about 120 GB of HBM at the peak
of the CSR build, 30 GB of host memory)."""
import numpy as np
import pytest
import torch

from conftest import assert_close_fp32
from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch.set_num_threads(1)   # the oracle's History.push is last-write-wins only single-threaded
DEV = torch.device("cuda:0")
N, E, K, BS, SEED = 10_000_000, 1_000_000_000, 20, 1365, 20240511


def test_config5_fullsize_uniform_k20():
  from orca_amd import ops, synth
  from orca_amd.engine import OrcaEngine
  torch.cuda.empty_cache()
  free, _ = torch.cuda.mem_get_info()
  if free < 150 << 30:
    pytest.skip(f"needs 150 GB of free HBM, {free >> 30} GB available")
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, DEV)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  assert csr.nnz == 2 * E
  ef = synth.edge_features(E, 1, DEV, zero=True)
  sd = synth.tgn_state_dict(1, 2, seed=3)
  g = torch.Generator(device=DEV)
  g.manual_seed(13)
  start = int(0.7 * E)
  batches = []
  for b in range(3):
    s = start + b * BS
    batches.append((src[s:s + BS].clone(), dst[s:s + BS].clone(), torch.randint(1, N + 1, (BS,), generator=g, device=DEV),
                    ts[s:s + BS].clone(), eidx[s:s + BS].clone()))
  del src, dst, eidx, ts
  torch.cuda.empty_cache()
  roots = np.concatenate([torch.cat(bt[:3]).cpu().numpy() for bt in batches])
  nf = O.OracleNeighborFinder.from_arrays(*csr.rows_host(roots))
  # ---- sampler: most recent and seeded uniform, bit for bit on the 2 B-entry adjacency
  s, d, n, t, _ = batches[0]
  nodes, t3 = torch.cat([s, d, n]), torch.cat([t, t, t])
  hn, ht = nodes.cpu().numpy(), t3.cpu().numpy()
  nbr, eid, et, dt = ops.sample_recent(csr, nodes, t3, K)
  e_nbr, e_eid, e_et, e_dt = nf.get_temporal_neighbor(hn, ht, K, with_deltas=True)
  assert np.array_equal(nbr.cpu().numpy(), e_nbr) and np.array_equal(eid.cpu().numpy(), e_eid)
  assert np.array_equal(et.cpu().numpy().view(np.uint32), e_et.view(np.uint32))
  assert np.array_equal(dt.cpu().numpy().view(np.uint32), e_dt.view(np.uint32))
  nbr, eid, et, dt = ops.sample_uniform(csr, nodes, t3, K, seed=99)
  e_nbr, e_eid, e_et, e_dt = O.sample_uniform(nf, hn, ht, K, 99)
  assert np.array_equal(nbr.cpu().numpy(), e_nbr) and np.array_equal(eid.cpu().numpy(), e_eid)
  assert np.array_equal(dt.cpu().numpy().view(np.uint32), e_dt.view(np.uint32))
  deg = nf.indptr[hn + 1] - nf.indptr[hn]
  assert deg.max() > 1_000_000, "the power-law hubs of this graph have adjacency lists of millions of entries"
  # ---- the native step (uniform K = 20, d_e = 1, 2 layers, reuse) against the oracle, on the rows the batches touch
  eng = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device=DEV, sampling="uniform", seed=SEED)
  eng.set_neighbor_finder(csr)
  ref = O.OracleTGN(sd, torch.zeros(1, 1), N + 1, 2, K, True, 0)
  ref.edge_feats = torch.zeros(E + 1, 1)        # calloc: only the touched pages are ever committed
  ref.neighbor_finder = nf
  ref.uniform_seed = SEED
  touched = [np.zeros(0, np.int64)]
  for b, bt in enumerate(batches):
    got = eng.compute_temporal_embeddings(*bt, True).cpu().numpy()
    exp = ref.compute_temporal_embeddings(*(x.cpu().numpy() for x in bt), True).numpy()
    assert_close_fp32(got, exp, 1e-5, f"config 5 batch {b} embeddings")
    Q = 3 * BS
    touched.append(np.concatenate([torch.cat(bt[:3]).cpu().numpy(), eng._w["nbr"][:Q * K].cpu().numpy().astype(np.int64)]))
    ids = torch.from_numpy(np.unique(np.concatenate(touched)))
    assert_close_fp32(eng.memory[ids.to(DEV)].cpu().numpy(), ref.memory[ids].numpy(), 1e-5, f"config 5 batch {b} memory rows")
    assert_close_fp32(eng.hist[0][ids.to(DEV)].cpu().numpy(), ref.hist[0][ids].numpy(), 1e-5, f"config 5 batch {b} history rows")
    assert np.array_equal(eng.last_update.cpu().numpy(), ref.last_update.numpy())
    assert np.array_equal(eng.pending.cpu().numpy().astype(bool), ref.pending)
  pos = torch.from_numpy(np.unique(np.concatenate([torch.cat(bt[:2]).cpu().numpy() for bt in batches])))
  assert_close_fp32(eng.messages[pos.to(DEV)].cpu().numpy(), ref.messages[pos].numpy(), 1e-5, "config 5 messages")
  assert int(eng.pending.sum().item()) == len(pos)
  del eng, csr, ef
  torch.cuda.empty_cache()
