"""Pins the oracle (oracle/) against the golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import assert_close_fp32
from oracle import oracle as O

torch.set_num_threads(1)


@pytest.mark.parametrize("tag", ["sorted", "shuf"])
def test_csr_build_and_sampler_bit_exact(golden, tag):
  g = golden("sampler.npz")
  src, dst, ts, eidx = g["src"], g["dst"], g["ts"], g["eidx"]
  if tag == "shuf":
    s = g["shuf"]
    src, dst, ts, eidx = src[s], dst[s], ts[s], eidx[s]
  nf = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=63)
  assert np.array_equal(nf.indptr, g[f"{tag}_indptr"])
  assert np.array_equal(nf.nbr, g[f"{tag}_nbr"])
  assert np.array_equal(nf.eid, g[f"{tag}_eidx"])
  assert np.array_equal(nf.ts, g[f"{tag}_ts"])
  qn, qt = g[f"{tag}_q_nodes"], g[f"{tag}_q_ts"]
  for K in (0, 1, 5, 10, 20):
    nbr, eid, et, dt = nf.get_temporal_neighbor(qn, qt, K, with_deltas=True)
    assert nbr.dtype == np.int32 and eid.dtype == np.int32 and et.dtype == np.float32
    assert np.array_equal(nbr, g[f"{tag}_K{K}_nbr"])
    assert np.array_equal(eid, g[f"{tag}_K{K}_eidx"])
    assert np.array_equal(et.view(np.uint32), g[f"{tag}_K{K}_et"].view(np.uint32))
    assert np.array_equal(dt.view(np.uint32), g[f"{tag}_K{K}_dt"].view(np.uint32))
    n2, e2, t2 = O.sample_recent_numpy(nf.indptr, nf.nbr, nf.eid, nf.ts, qn, qt, K)
    assert np.array_equal(n2, nbr) and np.array_equal(e2, eid) and np.array_equal(t2, et)
  for n, t, ln in zip(g[f"{tag}_fb_nodes"], g[f"{tag}_fb_ts"], g[f"{tag}_fb_len"]):
    assert len(nf.find_before(int(n), float(t))[0]) == ln


def test_sampler_out_of_range_raises():
  nf = O.OracleNeighborFinder([1, 2], [2, 3], [1, 2], [1.0, 2.0])
  with pytest.raises(IndexError):
    nf.get_temporal_neighbor(np.array([9]), np.array([1.0]), 3)


def test_time_encode(golden):
  g = golden("time_encode.npz")
  w = O.time_encode_weights(100)
  assert np.array_equal(w.numpy(), g["w"])
  out = O.time_encode(torch.from_numpy(g["t"]), w).numpy()
  assert_close_fp32(out, g["out"], 1e-6, "time_encode")


@pytest.mark.parametrize("tag", ["de172_K10_", "de1_K10_", "de4_K20_", "de172_K1_"])
def test_attention_layer(golden, tag):
  g = golden("attention.npz")
  P = {k[len(tag + "sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "sd.")}
  mask = torch.from_numpy(g[tag + "mask_in"].copy())
  out, w = O.attention_forward(P, *(torch.from_numpy(g[tag + k]) for k in ("src", "src_t", "nbr", "nbr_t", "edge")), mask)
  assert_close_fp32(out.numpy(), g[tag + "out"], 1e-5, tag + "out")
  assert np.array_equal(mask.numpy(), g[tag + "mask_out"])
  if tag + "weights" in g.files:
    assert_close_fp32(w.numpy(), g[tag + "weights"], 1e-5, tag + "weights")


def test_segment_last_and_mean(golden):
  g = golden("aggregators.npz")
  ids, msgs, ts, q = g["ids"], g["msgs"], g["ts"], g["query_ids"]
  u, idx = O.segment_last(ids)
  keep = np.isin(u, q)
  assert np.array_equal(u[keep], g["last_ids"])
  assert np.array_equal(msgs[idx[keep]], g["last_msgs"])
  assert np.array_equal(ts[idx[keep]], g["last_ts"])
  u2, m2, t2 = O.segment_mean(ids, msgs, ts)
  keep = np.isin(u2, q)
  assert np.array_equal(u2[keep], g["mean_ids"])
  assert_close_fp32(m2[keep], g["mean_msgs"], 1e-6, "mean msgs")
  assert np.array_equal(t2[keep], g["mean_ts"])
  # tgn_model.py:174-176 identity
  rs = np.random.RandomState(0)
  x = rs.randint(0, 20, 200)
  uu, ri = np.unique(np.flip(x), return_index=True)
  a, b = O.segment_last(x)
  assert np.array_equal(a, uu) and np.array_equal(b, len(x) - ri - 1)
  assert len(O.segment_last(np.zeros(0, np.int64))[0]) == 0


def _run_tgn(g, tag, graph):
  meta = g[f"{tag}_meta"]
  L, reuse, budget, bs, K, n_train, n_eval, n_nodes, n_edges = [int(x) for x in meta]
  src, dst, ts, eidx = (g[f"{graph}_{k}"] for k in ("src", "dst", "ts", "eidx"))
  sd = {k[len(graph + "_sd."):]: g[k] for k in g.files if k.startswith(graph + "_sd.")}
  tgn = O.OracleTGN(sd, g[f"{graph}_edge_feats"], n_nodes + 1, L, K, bool(reuse), budget)
  n_tr = n_train * bs
  train_nf = O.OracleNeighborFinder(src[:n_tr], dst[:n_tr], eidx[:n_tr], ts[:n_tr], max_node_idx=n_nodes)
  full_nf = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=n_nodes)
  negs = g[f"{tag}_negs"]
  b = 0
  for phase, nb in (("train", n_train), ("eval", n_eval)):
    tgn.neighbor_finder = train_nf if phase == "train" else full_nf
    for _ in range(nb):
      s, e = b * bs, min(n_edges, (b + 1) * bs)
      plan = g[f"{tag}_plan{b}"] if (budget and phase == "train") else None
      emb = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], negs[s:e], ts[s:e], eidx[s:e], phase == "train", plan)
      what = f"{tag} batch {b} ({phase})"
      assert_close_fp32(emb.numpy(), g[f"{tag}_b{b}_emb"], 1e-5, what + " emb")
      pos, neg = tgn.edge_probabilities(emb, e - s)
      assert_close_fp32(pos.numpy(), g[f"{tag}_b{b}_pos"], 1e-5, what + " pos")
      assert_close_fp32(neg.numpy(), g[f"{tag}_b{b}_neg"], 1e-5, what + " neg")
      assert_close_fp32(tgn.memory.numpy(), g[f"{tag}_b{b}_memory"], 1e-5, what + " memory")
      assert np.array_equal(tgn.last_update.numpy(), g[f"{tag}_b{b}_last_update"])
      assert np.array_equal(tgn.pending, g[f"{tag}_b{b}_pending"])
      if reuse:
        assert_close_fp32(np.stack([h.numpy() for h in tgn.hist]), g[f"{tag}_b{b}_hist"], 1e-5, what + " hist")
        assert np.array_equal(np.stack(tgn.cache_flag).astype(np.uint8), g[f"{tag}_b{b}_cache_flag"])
      b += 1
  assert_close_fp32(tgn.messages.numpy(), g[f"{tag}_final_messages"], 1e-5, tag + " messages")
  assert np.array_equal(tgn.msg_ts.numpy(), g[f"{tag}_final_timestamps"])


@pytest.mark.parametrize("tag,graph", [("A1", "g172"), ("A2", "g172"), ("B2", "g172"), ("C2", "g172"),
                                       ("B2z", "g1"), ("A1z", "g1"), ("C2z", "g1")])
def test_tgn_paths(golden, tag, graph):
  _run_tgn(golden("tgn.npz"), tag, graph)


@pytest.mark.parametrize("tag", ["tight", "loose", "mid"])
def test_mrd_plan(golden, tag):
  g = golden("mrd.npz")
  n_nodes, n_edges, bs, K, budget, num_batch = [int(x) for x in g[f"{tag}_meta"]]
  src, dst, ts, eidx = (g[f"{tag}_{k}"] for k in ("src", "dst", "ts", "eidx"))
  nf = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=n_nodes)
  nb, tl, nl, occ = O.read_batches(src, dst, ts, nf, bs, K, n_nodes + 1)
  assert nb == num_batch
  assert np.array_equal(np.concatenate(tl), g[f"{tag}_target_flat"])
  assert np.array_equal(np.array([len(x) for x in nl]), g[f"{tag}_ngh_len"])
  assert np.array_equal(np.concatenate(nl + [np.zeros(0, np.int64)]), g[f"{tag}_ngh_flat"])
  # NumPy-argsort mode == the fixture exactly (same library the stand-in reference ran on)
  plans, n_reuse, n_recompute, avg = O.mrd_plan(n_nodes + 1, nb, budget, tl, nl, occ, stable=False)
  assert np.array_equal(np.array([len(p) for p in plans]), g[f"{tag}_plan_len"])
  assert np.array_equal(np.concatenate(plans), g[f"{tag}_plan_flat"])
  assert [n_reuse, n_recompute] == list(g[f"{tag}_stats"][:2])
  assert abs(avg - float(g[f"{tag}_avg_reuse_distance"][0])) < 1e-12
  # defined (stable) tie-break: same plan sizes always; same plans where no tie group straddles the cut
  plans_s, r_s, c_s, _ = O.mrd_plan(n_nodes + 1, nb, budget, tl, nl, occ, stable=True)
  assert np.array_equal(np.array([len(p) for p in plans_s]), g[f"{tag}_plan_len"])
  assert r_s + c_s == n_reuse + n_recompute
  if tag == "loose":
    for a, b in zip(plans_s, plans):
      assert np.array_equal(np.sort(a), np.sort(b))
    assert (r_s, c_s) == (n_reuse, n_recompute)
