"""End-to-end parity of the device-resident step (orca_amd/engine.py, all kernels through the C ABI)
against the golden vectors captured from the reference's TGN (tests/golden/tgn.npz) -- Paths A, B
and C, train and eval ordering, d_e = 172 and the zero-feature d_e = 1 case -- and, at the
BASELINE config-4 batch shape (4096 queries, K=10, 2 layers, 172-d), against the CPU oracle plus
size-independent properties.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from conftest import assert_close_fp32
from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch.set_num_threads(1)
DEV = "cuda:0"


def _engine_and_graph(g, tag, graph, slot_history=None):
  from orca_amd import ops
  from orca_amd.engine import OrcaEngine
  meta = g[f"{tag}_meta"]
  L, reuse, budget, bs, K, n_train, n_eval, n_nodes, n_edges = [int(x) for x in meta]
  src, dst, ts, eidx = (g[f"{graph}_{k}"] for k in ("src", "dst", "ts", "eidx"))
  sd = {k[len(graph + "_sd."):]: g[k] for k in g.files if k.startswith(graph + "_sd.")}
  eng = OrcaEngine(sd, g[f"{graph}_edge_feats"], n_nodes + 1, L, K, bool(reuse), budget, device=DEV, slot_history=slot_history)
  n_tr = n_train * bs
  train_csr = ops.CSR.from_edges_host(src[:n_tr], dst[:n_tr], eidx[:n_tr], ts[:n_tr], n_nodes + 1, DEV)
  full_csr = ops.CSR.from_edges_host(src, dst, eidx, ts, n_nodes + 1, DEV)
  return eng, train_csr, full_csr, (L, reuse, budget, bs, K, n_train, n_eval, n_nodes, n_edges), (src, dst, ts, eidx)


@pytest.mark.parametrize("tag,graph,slots", [("A1", "g172", None), ("A2", "g172", None), ("B2", "g172", None),
                                             ("C2", "g172", False), ("C2", "g172", True), ("B2z", "g1", None),
                                             ("A1z", "g1", None), ("C2z", "g1", False), ("C2z", "g1", True)])
def test_tgn_paths_vs_reference_golden(golden, tag, graph, slots):
  """slots (Path C only): the layer-1 history in the budget-bounded slot cache instead of the reference's dense table.  The
  reference's own plans (its MRD planner) drive it; everything the reference computes must come out the same -- embeddings,
  probabilities, memory, flags -- and the history rows of every CACHED id (the dense table also keeps the rows of ids the
  plan dropped, which nothing reads again: those are what the slot cache does not store)."""
  g = golden("tgn.npz")
  eng, train_csr, full_csr, meta, edges = _engine_and_graph(g, tag, graph, slots)
  L, reuse, budget, bs, K, n_train, n_eval, n_nodes, n_edges = meta
  src, dst, ts, eidx = edges
  negs = g[f"{tag}_negs"]
  b = 0
  for phase, nb in (("train", n_train), ("eval", n_eval)):
    eng.set_neighbor_finder(train_csr if phase == "train" else full_csr)
    for _ in range(nb):
      s, e = b * bs, min(n_edges, (b + 1) * bs)
      plan = g[f"{tag}_plan{b}"] if (budget and phase == "train") else None
      pos, neg = eng.compute_edge_probabilities(src[s:e], dst[s:e], negs[s:e], ts[s:e], eidx[s:e], phase == "train", plan)
      what = f"{tag} batch {b} ({phase})"
      emb = eng.last_embeddings.cpu().numpy()
      assert_close_fp32(emb, g[f"{tag}_b{b}_emb"], 1e-5, what + " emb")
      assert_close_fp32(pos.cpu().numpy(), g[f"{tag}_b{b}_pos"], 1e-5, what + " pos")
      assert_close_fp32(neg.cpu().numpy(), g[f"{tag}_b{b}_neg"], 1e-5, what + " neg")
      assert_close_fp32(eng.memory.cpu().numpy(), g[f"{tag}_b{b}_memory"], 1e-5, what + " memory")
      assert np.array_equal(eng.last_update.cpu().numpy(), g[f"{tag}_b{b}_last_update"])
      assert np.array_equal(eng.pending.cpu().numpy().astype(bool), g[f"{tag}_b{b}_pending"])
      if reuse:
        hist = np.stack([h.cpu().numpy() for h in eng.hist])
        exp_hist = g[f"{tag}_b{b}_hist"]
        if slots:
          cached = g[f"{tag}_b{b}_cache_flag"].astype(bool)        # [layers - 1, n_nodes]
          hist, exp_hist = hist[cached], exp_hist[cached]
          assert (eng._hist_dense is None) == (phase == "train"), "slot mode while a plan drives the cache, the table afterwards"
        assert_close_fp32(hist, exp_hist, 1e-5, what + " hist")
        if budget:
          flags = np.stack([c.cpu().numpy() for c in eng.cache_flag])
          assert np.array_equal(flags, g[f"{tag}_b{b}_cache_flag"])
      b += 1
  assert_close_fp32(eng.messages.cpu().numpy(), g[f"{tag}_final_messages"], 1e-5, tag + " messages")
  assert np.array_equal(eng.msg_ts.cpu().numpy(), g[f"{tag}_final_timestamps"])


def _synthetic(seed, n_nodes, n_edges, d_e):
  rs = np.random.RandomState(seed)
  w = 1.0 / np.arange(1, n_nodes + 1) ** 1.2
  w /= w.sum()
  perm = rs.permutation(n_nodes) + 1
  src = perm[rs.choice(n_nodes, n_edges, p=w)].astype(np.int64)
  dst = perm[rs.choice(n_nodes, n_edges, p=w)].astype(np.int64)
  ts = np.sort(rs.uniform(0, 2.6e6, n_edges))
  pos = rs.choice(np.arange(1, n_edges), n_edges // 100, replace=False)
  ts[pos] = ts[pos - 1]
  ts = np.sort(ts)
  eidx = np.arange(1, n_edges + 1, dtype=np.int64)
  ef = rs.randn(n_edges + 1, d_e).astype(np.float32)
  ef[0] = 0
  return src, dst, ts, eidx, ef


@pytest.mark.parametrize("reuse,bs", [(True, 1365), (False, 1365), (True, 2200)])
def test_config4_batch_shape_vs_oracle(golden, reuse, bs):
  """4096-query batches (bs = 1365 interactions + 1 padding query is the bench shape; here bs=1365),
  K = 10, 2 layers, 172-d edge features on a 20k-node / 300k-edge power-law graph: the HIP step
  against the oracle over several consecutive batches (state carried), plus properties.  bs = 2200
  (6600 roots) is past what one LDS election table of the native driver holds, so it covers the
  elections split over two workgroups."""
  from orca_amd import ops
  from orca_amd.engine import OrcaEngine
  g = golden("tgn.npz")
  sd = {k[len("g172_sd."):]: g[k] for k in g.files if k.startswith("g172_sd.")}
  N, E, K = 20000, 300000, 10
  src, dst, ts, eidx, ef = _synthetic(5, N, E, 172)
  n_hist = 280000
  eng = OrcaEngine(sd, ef, N + 1, 2, K, reuse, 0, device=DEV)
  eng.set_neighbor_finder(ops.CSR.from_edges_device(*(torch.from_numpy(x).to(DEV) for x in (src, dst, eidx, ts)), N + 1))
  ref = O.OracleTGN(sd, ef, N + 1, 2, K, reuse, 0)
  ref.neighbor_finder = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=N)
  rs = np.random.RandomState(6)
  n_batches = 4 if reuse else 2
  for b in range(n_batches):
    s, e = n_hist + b * bs, n_hist + (b + 1) * bs
    neg = rs.randint(1, N + 1, bs).astype(np.int64)
    emb = eng.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True).cpu().numpy()
    exp = ref.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True).numpy()
    assert emb.shape == (3 * bs, 100)
    assert_close_fp32(emb, exp, 1e-5, f"batch {b} emb")
    assert_close_fp32(eng.memory.cpu().numpy(), ref.memory.numpy(), 1e-5, f"batch {b} memory")
    assert np.array_equal(eng.last_update.cpu().numpy(), ref.last_update.numpy())
    assert np.array_equal(eng.pending.cpu().numpy().astype(bool), ref.pending)
    if reuse:
      assert_close_fp32(eng.hist[0].cpu().numpy(), ref.hist[0].numpy(), 1e-5, f"batch {b} history")
    # properties: LayerNorm output rows have ~zero mean / unit variance under the affine params
    assert np.isfinite(emb).all()
    # duplicate root (node, t) queries get identical embeddings
    nodes = np.concatenate([src[s:e], dst[s:e], neg])
    t3 = np.concatenate([ts[s:e]] * 3)
    key = np.stack([nodes.astype(np.float64), t3], axis=1)
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    assert np.array_equal(emb, emb[first][inv])
  assert_close_fp32(eng.messages.cpu().numpy(), ref.messages.numpy(), 1e-5, "messages")


def test_election_table_overflow_vs_oracle(golden):
  """The native driver elects last occurrences in LDS hash tables, one per part of the ids (a hash of the id
  picks the part).  Here every id of a 9000-root batch is chosen to fall into part 0 of 2, so that table
  (8192 slots) overflows and the late ids take the exact fallback; the tail of each list repeats ids so
  the fallback has real duplicates to resolve."""
  from orca_amd import ops
  from orca_amd.engine import OrcaEngine
  g = golden("tgn.npz")
  sd = {k[len("g172_sd."):]: g[k] for k in g.files if k.startswith("g172_sd.")}
  N, E, bs, K = 60000, 120000, 3000, 10
  src, dst, ts, eidx, ef = _synthetic(11, N, E, 172)
  ids = np.arange(1, N + 1, dtype=np.int64)
  part0 = ids[((((ids * 0x85EBCA6B) & 0xFFFFFFFF) >> 16) % 2) == 0]
  assert part0.size >= 3 * bs
  rs = np.random.RandomState(12)
  pool = rs.permutation(part0)[:3 * bs]
  b_src, b_dst, b_neg = pool[:bs].copy(), pool[bs:2 * bs].copy(), pool[2 * bs:].copy()
  b_dst[-300:] = np.repeat(b_dst[-300:-150], 2)     # duplicates among the late ids of [src;dst]
  b_neg[-400:] = np.tile(b_neg[-400:-200], 2)       # and of [src;dst;neg]
  b_src[:50] = b_dst[-50:]                          # early ids that recur late
  eng = OrcaEngine(sd, ef, N + 1, 2, K, True, 0, device=DEV)
  eng.set_neighbor_finder(ops.CSR.from_edges_device(*(torch.from_numpy(x).to(DEV) for x in (src, dst, eidx, ts)), N + 1))
  ref = O.OracleTGN(sd, ef, N + 1, 2, K, True, 0)
  ref.neighbor_finder = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=N)
  for b in range(2):       # the second batch reads the messages / history the first one stored
    t = np.sort(rs.uniform(2.7e6 + b * 1e3, 2.7e6 + (b + 1) * 1e3, bs))
    e = rs.randint(1, E + 1, bs).astype(np.int64)
    emb = eng.compute_temporal_embeddings(b_src, b_dst, b_neg, t, e, True).cpu().numpy()
    exp = ref.compute_temporal_embeddings(b_src, b_dst, b_neg, t, e, True).numpy()
    assert_close_fp32(emb, exp, 1e-5, f"batch {b} emb")
    assert_close_fp32(eng.memory.cpu().numpy(), ref.memory.numpy(), 1e-5, f"batch {b} memory")
    assert_close_fp32(eng.hist[0].cpu().numpy(), ref.hist[0].numpy(), 1e-5, f"batch {b} history")
    assert np.array_equal(eng.pending.cpu().numpy().astype(bool), ref.pending)
  assert_close_fp32(eng.messages.cpu().numpy(), ref.messages.numpy(), 1e-5, "messages")
  assert np.array_equal(eng.msg_ts.cpu().numpy(), ref.msg_ts.numpy())


def test_eager_overlay_survives_mode_changes(golden):
  """The native driver applies GRU(message, memory) once per stored message (the ids of the previous step's
  store) instead of building a work-list of the ids each batch reads.  That rests on a state flag which train /
  eval switches, a backup restore, a batch-size change that rebuilds the driver and a Python-dispatched step
  all have to keep honest: after each of them the results must still equal the oracle's."""
  import copy
  from orca_amd import ops
  from orca_amd.engine import OrcaEngine
  g = golden("tgn.npz")
  sd = {k[len("g172_sd."):]: g[k] for k in g.files if k.startswith("g172_sd.")}
  N, E, K = 5000, 80000, 10
  src, dst, ts, eidx, ef = _synthetic(21, N, E, 172)
  eng = OrcaEngine(sd, ef, N + 1, 2, K, True, 0, device=DEV)
  eng.set_neighbor_finder(ops.CSR.from_edges_device(*(torch.from_numpy(x).to(DEV) for x in (src, dst, eidx, ts)), N + 1))
  ref = O.OracleTGN(sd, ef, N + 1, 2, K, True, 0)
  ref.neighbor_finder = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=N)
  rs = np.random.RandomState(22)
  pos = [60000]

  def step(bs, train, what):
    s, e = pos[0], pos[0] + bs
    pos[0] = e
    neg = rs.randint(1, N + 1, bs).astype(np.int64)
    emb = eng.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], train).cpu().numpy()
    exp = ref.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], train).numpy()
    assert_close_fp32(emb, exp, 1e-5, what + " emb")
    assert_close_fp32(eng.memory.cpu().numpy(), ref.memory.numpy(), 1e-5, what + " memory")
    assert_close_fp32(eng.messages.cpu().numpy(), ref.messages.numpy(), 1e-5, what + " messages")
    assert np.array_equal(eng.pending.cpu().numpy().astype(bool), ref.pending), what
    assert np.array_equal(eng.last_update.cpu().numpy(), ref.last_update.numpy()), what

  for b in range(3):
    step(300, True, f"train {b}")
  step(700, True, "train, larger batch (driver rebuilt)")        # drains the outstanding list, carries the flag
  step(700, True, "train after the rebuild")
  bk, ref_bk = eng.backup(), copy.deepcopy(ref)
  bk_pos = pos[0]
  step(250, False, "eval 0 (flushes every pending message)")
  step(250, False, "eval 1")
  step(300, True, "train after eval")
  step(300, True, "train after eval, 2")
  eng.restore(bk)                                                  # pending ids come back without overlay rows
  ref, pos[0] = ref_bk, bk_pos
  step(300, True, "train after restore (work-list route)")
  step(300, True, "train after restore, 2")
  eng.prof = {}                                                    # one Python-dispatched Path B step
  step(300, True, "python-dispatched step")
  eng.prof = None
  step(300, True, "native step after a python-dispatched one")
  eng.reset_state()
  ref.init_memory()
  ref.reset_history()
  step(300, True, "train after reset")
  step(300, True, "train after reset, 2")


def test_native_driver_rejects_bad_calls(golden):
  """Argument checks of the orca_engine_* entry points: a batch beyond the work buffers, a layer outside the
  model, a slice outside the batch -- each refused with ORCA_EINVAL (ValueError in the mirror) and a message,
  leaving the driver usable."""
  import ctypes
  from orca_amd import _lib, ops
  from orca_amd.engine import OrcaEngine
  g = golden("tgn.npz")
  sd = {k[len("g172_sd."):]: g[k] for k in g.files if k.startswith("g172_sd.")}
  N, E, K = 3000, 40000, 10
  src, dst, ts, eidx, ef = _synthetic(31, N, E, 172)
  eng = OrcaEngine(sd, ef, N + 1, 2, K, True, 0, device=DEV)
  eng.set_neighbor_finder(ops.CSR.from_edges_device(*(torch.from_numpy(x).to(DEV) for x in (src, dst, eidx, ts)), N + 1))
  rs = np.random.RandomState(32)
  s, e = 30000, 30100
  neg = rs.randint(1, N + 1, 100).astype(np.int64)
  ok = eng.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True).clone()
  L, h = _lib.load(), eng._native
  t = [torch.from_numpy(x).to(DEV) for x in (src[s:e], dst[s:e], neg, eidx[s:e])] + [torch.from_numpy(ts[s:e]).to(DEV)]
  p = [x.data_ptr() for x in t]
  st = torch.cuda.current_stream().cuda_stream
  big = eng._q_max // 3 + 1
  with pytest.raises(ValueError, match="does not fit"):
    _lib.check(L.orca_engine_begin(h, p[0], p[1], p[2], p[3], p[4], big, 0, 3 * big, 1, st), "orca_engine_begin")
  with pytest.raises(ValueError, match="slice"):
    _lib.check(L.orca_engine_begin(h, p[0], p[1], p[2], p[3], p[4], 100, 10, 400, 1, st), "orca_engine_begin")
  with pytest.raises(ValueError, match="layer"):
    _lib.check(L.orca_engine_layer(h, 3, 0, 300, st), "orca_engine_layer")
  with pytest.raises(ValueError, match="null"):
    _lib.check(L.orca_engine_begin(h, None, p[1], p[2], p[3], p[4], 100, 0, 300, 1, st), "orca_engine_begin")
  # an empty batch is a no-op, and the driver still works afterwards
  assert eng.compute_temporal_embeddings(src[:0], dst[:0], neg[:0], ts[:0], eidx[:0], True).shape == (0, 100)
  eng.reset_state()
  again = eng.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True)
  assert torch.equal(ok, again)


def test_native_driver_long_run_vs_dispatched_path():
  """300 consecutive train-ordered batches of the bench's shape through the native driver (eager overlay lists
  double-buffered across steps, fused persist / push / store) against the Python-dispatched Path B on the same
  inputs (per-batch work-lists, one kernel per operation): a drift or a hazard between steps would show in the
  state tables.  Checked every 100 batches; an eval-ordered batch is mixed in every 50."""
  from orca_amd import ops, synth
  from orca_amd.engine import OrcaEngine
  N, E, bs, K, d_e = 50_000, 2_000_000, 1365, 10, 172
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, "cuda:0", seed=8)
  ef = synth.edge_features(E, d_e, "cuda:0")
  sd = synth.tgn_state_dict(d_e, 2, seed=4)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  native = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device="cuda:0")
  plain = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device="cuda:0")
  plain.prof = {}                     # forces the Python-dispatched path
  for e in (native, plain):
    e.set_neighbor_finder(csr)
  g = torch.Generator(device="cuda:0")
  g.manual_seed(3)
  start = int(0.6 * E)
  for b in range(300):
    s = start + b * bs
    neg = torch.randint(1, N + 1, (bs,), generator=g, device="cuda:0")
    train = b % 50 != 49
    a = native.compute_temporal_embeddings(src[s:s + bs], dst[s:s + bs], neg, ts[s:s + bs], eidx[s:s + bs], train)
    c = plain.compute_temporal_embeddings(src[s:s + bs], dst[s:s + bs], neg, ts[s:s + bs], eidx[s:s + bs], train)
    plain.prof = {}                   # drop the recorded events
    if b % 100 == 99 or b == 299:
      assert_close_fp32(a.cpu().numpy(), c.cpu().numpy(), 1e-5, f"batch {b} embeddings")
      assert_close_fp32(native.memory.cpu().numpy(), plain.memory.cpu().numpy(), 1e-5, f"batch {b} memory")
      assert_close_fp32(native.hist[0].cpu().numpy(), plain.hist[0].cpu().numpy(), 1e-5, f"batch {b} history")
      assert_close_fp32(native.messages.cpu().numpy(), plain.messages.cpu().numpy(), 1e-5, f"batch {b} messages")
      assert torch.equal(native.pending, plain.pending) and torch.equal(native.last_update, plain.last_update)
      assert torch.equal(native.msg_ts, plain.msg_ts)


@pytest.mark.parametrize("d_e", [172, 1])
def test_eager_gru_groups_per_block(monkeypatch, d_e):
  """The eager GRU rides in the step's first kernel as 4-wave groups, one, two or four to a 1024-thread block -- chosen
  per launch from the block count (engine.hip, gru_subs_for) -- or runs as a launch of its own (ORCA_FUSE_GRU=0).  All
  four forms are the same arithmetic: embeddings and every state table bit for bit, over train batches with an
  evaluation batch in between, for both message widths (472 and the zero-feature datasets' 301)."""
  from orca_amd import ops, synth
  from orca_amd.engine import OrcaEngine
  N, E, bs, K = 20_000, 600_000, 700, 10
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, "cuda:0", seed=12)
  ef = synth.edge_features(E, d_e, "cuda:0", zero=d_e == 1)
  sd = synth.tgn_state_dict(d_e, 2, seed=6)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  start = int(0.5 * E)

  def run(env):
    for k in ("ORCA_FUSE_GRU", "ORCA_GRU_SUBS"):
      monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
      monkeypatch.setenv(k, v)
    eng = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device="cuda:0")      # (the handle reads the switches)
    eng.set_neighbor_finder(csr)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(1)
    outs = []
    for b in range(7):
      s = start + b * bs
      neg = torch.randint(1, N + 1, (bs,), generator=g, device="cuda:0")
      outs.append(eng.compute_temporal_embeddings(src[s:s + bs], dst[s:s + bs], neg, ts[s:s + bs], eidx[s:s + bs], b != 4))
    eng.check_errors()
    return outs, [t.clone() for t in (eng.memory, eng.last_update, eng.messages, eng.msg_ts, eng.pending, eng.hist[0])]

  ref_out, ref_tabs = run({"ORCA_FUSE_GRU": "0"})
  for subs in ("1", "2", "4"):
    out, tabs = run({"ORCA_GRU_SUBS": subs})
    for b, (x, y) in enumerate(zip(out, ref_out)):
      assert torch.equal(x, y), f"{subs} groups per block: embeddings of batch {b}"
    for i, (x, y) in enumerate(zip(tabs, ref_tabs)):
      assert torch.equal(x, y), f"{subs} groups per block: state table {i}"


@pytest.mark.parametrize("native", [True, False])
def test_uniform_k20_engine_vs_oracle(golden, native):
  """BASELINE config 5's sampler in the step driver: K = 20 seeded-uniform neighbours (not in the reference:
  parity unpinned, GPU == the oracle's CPU twin), zero-feature edges (d_e = 1), 2 layers with reuse.  The sampled
  ids must equal the twin's bit for bit (checked through the embeddings' inputs: nbr / eid / dt buffers of the
  driver) and embeddings / state agree with the oracle at 1e-5, natively and Python-dispatched."""
  from orca_amd import ops, synth
  from orca_amd.engine import OrcaEngine
  N, E, K, bs = 20000, 300000, 20, 700
  src, dst, ts, eidx, _ = _synthetic(41, N, E, 1)
  ef = np.zeros((E + 1, 1), np.float32)
  sd = {k: v.numpy() for k, v in synth.tgn_state_dict(1, 2, seed=9).items()}
  eng = OrcaEngine(sd, ef, N + 1, 2, K, True, 0, device=DEV, sampling="uniform", seed=1234)
  eng.set_neighbor_finder(ops.CSR.from_edges_device(*(torch.from_numpy(x).to(DEV) for x in (src, dst, eidx, ts)), N + 1))
  if not native:
    eng.prof = {}
  ref = O.OracleTGN(sd, ef, N + 1, 2, K, True, 0)
  ref.neighbor_finder = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=N)
  ref.uniform_seed = 1234
  rs = np.random.RandomState(42)
  n_hist = 250000
  for b in range(3):
    s, e = n_hist + b * bs, n_hist + (b + 1) * bs
    neg = rs.randint(1, N + 1, bs).astype(np.int64)
    emb = eng.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True).cpu().numpy()
    exp = ref.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True).numpy()
    if native:      # the driver's sampler buffers against the twin, bit for bit
      nodes = np.concatenate([src[s:e], dst[s:e], neg])
      seed = (1234 + ref.batch_counter * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
      nbr, eid, et, dt = O.sample_uniform(ref.neighbor_finder, nodes, np.concatenate([ts[s:e]] * 3), K, seed)
      Q = 3 * bs
      assert np.array_equal(eng._w["nbr"][:Q * K].cpu().numpy().reshape(Q, K), nbr)
      assert np.array_equal(eng._w["eid"][:Q * K].cpu().numpy().reshape(Q, K), eid)
      assert np.array_equal(eng._w["dt"][:Q * K].cpu().numpy().view(np.uint32).reshape(Q, K), dt.view(np.uint32))
    eng.prof = {} if not native else None
    assert_close_fp32(emb, exp, 1e-5, f"batch {b} emb")
    assert_close_fp32(eng.memory.cpu().numpy(), ref.memory.numpy(), 1e-5, f"batch {b} memory")
    assert_close_fp32(eng.hist[0].cpu().numpy(), ref.hist[0].numpy(), 1e-5, f"batch {b} history")
    assert np.array_equal(eng.pending.cpu().numpy().astype(bool), ref.pending)
  assert_close_fp32(eng.messages.cpu().numpy(), ref.messages.numpy(), 1e-5, "messages")


def test_refresh_with_new_gru_weights_drops_stale_overlay(golden):
  """overlay rows are GRU(message, memory) computed once per stored message; after the GRU weights
  change (an optimizer step + refresh) every pending id's row is stale and must be recomputed with the new weights,
  as the reference recomputes get_updated_memory on every batch (memory_updater.py:56-78)."""
  from orca_amd import ops
  from orca_amd.engine import OrcaEngine
  g = golden("tgn.npz")
  sd = {k[len("g172_sd."):]: g[k] for k in g.files if k.startswith("g172_sd.")}
  N, E, K, bs = 5000, 80000, 10, 300
  src, dst, ts, eidx, ef = _synthetic(51, N, E, 172)
  csr = ops.CSR.from_edges_device(*(torch.from_numpy(x).to(DEV) for x in (src, dst, eidx, ts)), N + 1)
  eng = OrcaEngine(sd, ef, N + 1, 2, K, True, 0, device=DEV)
  eng.set_neighbor_finder(csr)
  rs = np.random.RandomState(52)
  negs = [rs.randint(1, N + 1, bs).astype(np.int64) for _ in range(4)]

  def batch(b):
    s, e = 60000 + b * bs, 60000 + (b + 1) * bs
    return src[s:e], dst[s:e], negs[b], ts[s:e], eidx[s:e]

  eng.reset_state()
  for b in range(3):
    eng.compute_temporal_embeddings(*batch(b), True)
  bk = eng.backup()
  with torch.no_grad():
    for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
      eng.params["memory_updater.memory_updater." + k].mul_(0.5)
  eng.refresh()
  got = eng.compute_temporal_embeddings(*batch(3), True).clone()
  fresh = OrcaEngine({k: v.cpu() for k, v in eng.params.items()}, ef, N + 1, 2, K, True, 0, device=DEV)
  fresh.set_neighbor_finder(csr)
  fresh.restore(bk)
  exp = fresh.compute_temporal_embeddings(*batch(3), True)
  assert_close_fp32(got.cpu().numpy(), exp.cpu().numpy(), 1e-6, "embeddings after refresh")
  assert_close_fp32(eng.memory.cpu().numpy(), fresh.memory.cpu().numpy(), 1e-6, "memory after refresh")


@pytest.mark.gpu
def test_budget_step_native_equals_dispatched():
  """Path C issued by the native driver (orca_engine_step_budget: the size of the uncached set stays on the device, the
  layer-1 launch is sized for a bound) against the Python-dispatched sequence of the same kernels (which the goldens
  of test_tgn_paths_vs_reference_golden pin to the reference): embeddings, every state table and the cache flags bit
  for bit over train-ordered batches with a plan, without one, and an evaluation batch; and the overflow flag when the
  bound is too small."""
  from orca_amd import ops, synth
  from orca_amd.engine import OrcaEngine
  N, E, K, d_e, bs = 3000, 60000, 10, 172, 200
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, DEV, seed=3)
  ef = synth.edge_features(E, d_e, DEV)
  sd = synth.tgn_state_dict(d_e, 2, seed=4)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  engs = []
  for native in (True, False):
    # (random plans: the slot cache needs plans of the planner, see test_slot_history_equals_dense_table)
    e = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=300, device=DEV, slot_history=False)
    e.native_budget = native
    e.set_neighbor_finder(csr)
    engs.append(e)
  A, B = engs
  rs = np.random.RandomState(1)
  for b in range(8):
    s = 30000 + b * bs
    neg = torch.from_numpy(rs.randint(1, N + 1, bs).astype(np.int64)).to(DEV)
    plan = torch.from_numpy(rs.choice(np.arange(1, N + 1), 300, replace=False).astype(np.int64)).to(DEV) if b % 3 != 2 else None
    train = b != 6
    args = (src[s:s + bs], dst[s:s + bs], neg, ts[s:s + bs], eidx[s:s + bs], train, plan)
    ea = A.compute_temporal_embeddings(*args)
    eb = B.compute_temporal_embeddings(*args)
    assert torch.equal(ea, eb), f"batch {b}: embeddings"
    for name in ("memory", "last_update", "messages", "msg_ts", "pending"):
      assert torch.equal(getattr(A, name), getattr(B, name)), f"batch {b}: {name}"
    assert torch.equal(A.hist[0], B.hist[0]) and torch.equal(A.cache_flag[0], B.cache_flag[0]), f"batch {b}: history"
  assert not A.budget_overflow()
  A.budget_cap = 1                       # one uncached neighbour per batch cannot be enough here
  A.compute_temporal_embeddings(src[40000:40000 + bs], dst[40000:40000 + bs], neg, ts[40000:40000 + bs], eidx[40000:40000 + bs],
                                True, None)
  assert A.budget_overflow()
  # ... and nobody has to ask: the next native step (once the flag has arrived) or check_errors() raises
  import pytest
  from orca_amd._lib import OrcaError
  with pytest.raises(OrcaError, match="budget_cap=1"):
    A.check_errors()
  A.budget_cap = None                    # the safe bound: min(3*bs*K, n_nodes)
  A._bw["err"].zero_()
  A.compute_temporal_embeddings(src[40200:40200 + bs], dst[40200:40200 + bs], neg, ts[40200:40200 + bs], eidx[40200:40200 + bs],
                                True, None)
  A.check_errors()
  assert not A.budget_overflow()


def test_slot_history_equals_dense_table():
  """The budget-bounded slot cache of the native Path C step (include/orca_hip.h: orca_slot_cache) against the dense
  history table on the same batches with the MRD plans of the device planner: embeddings, memory, messages and cache
  flags bit for bit, the history rows of every cached id bit for bit, across train batches, a backup, evaluation batches
  without a plan (the engine moves the slots into the dense table) and the restore back into slot mode."""
  import argparse
  from orca_amd import api, ops, synth
  from orca_amd.engine import OrcaEngine
  N, E, K, d_e, bs, budget = 6000, 120000, 10, 172, 200, 150
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, DEV, seed=6)
  ef = synth.edge_features(E, d_e, DEV)
  sd = synth.tgn_state_dict(d_e, 2, seed=8)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  n_b, start = 24, 60000
  n = n_b * bs
  h = lambda x: x[start:start + n].cpu().numpy()
  data = api.Data(h(src), h(dst), h(ts), h(eidx), np.zeros(n))
  nf = api.NeighborFinder(None, None, None, device=DEV, _csr=csr)
  pa = argparse.Namespace(bs=bs, n_degree=K, budget=budget, planner="device")
  plans = [torch.from_numpy(p).to(DEV) for p in api.get_cache_plan(pa, data, nf, N + 1, "MRD")]
  A = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=budget, device=DEV)                        # slot cache (default)
  B = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=budget, device=DEV, slot_history=False)    # dense table
  for e in (A, B):
    e.set_neighbor_finder(csr)
    e.budget_cap = 800          # launches (and slots) for at most 800 uncached neighbours per batch; more would raise
  rs = np.random.RandomState(2)

  def step(b, train, plan):
    s = start + b * bs
    neg = torch.from_numpy(rs.randint(1, N + 1, bs).astype(np.int64)).to(DEV)
    args = (src[s:s + bs], dst[s:s + bs], neg, ts[s:s + bs], eidx[s:s + bs], train, plan)
    ea, eb = A.compute_temporal_embeddings(*args), B.compute_temporal_embeddings(*args)
    assert torch.equal(ea, eb), f"batch {b}: embeddings"
    for name in ("memory", "last_update", "messages", "msg_ts", "pending"):
      assert torch.equal(getattr(A, name), getattr(B, name)), f"batch {b}: {name}"
    assert torch.equal(A.cache_flag[0], B.cache_flag[0]), f"batch {b}: cache flags"
    cached = B.cache_flag[0].bool()
    assert torch.equal(A.hist[0][cached], B.hist[0][cached]), f"batch {b}: history rows of the cached ids"

  for b in range(12):
    step(b, True, plans[b])
  assert A._hist_dense is None and A._sc is not None, "the engine left slot mode during plan-driven training"
  assert int(A.cache_flag[0].sum()) > 0
  dense_bytes = (N + 1) * 100 * 4
  # rows: the zero row, two plan regions, the batch region (3*bs roots + budget_cap uncached neighbours)
  assert A.history_footprint_bytes() < dense_bytes // 3 and A._sc["rows"].shape[0] == 1 + 2 * budget + 3 * bs + 800
  A.check_errors()
  ba, bb = A.backup(), B.backup()
  for b in range(12, 15):                   # evaluation: no plan, flags accumulate -> the dense table takes over
    step(b, False, None)
  assert A._hist_dense is not None
  A.restore(ba)
  B.restore(bb)
  assert A._hist_dense is None, "a backup taken in slot mode restores into slot mode"
  for b in range(12, 24):
    step(b, True, plans[b])
  A.reset_state()
  B.reset_state()
  for b in range(0, 4):
    step(b, True, plans[b])
