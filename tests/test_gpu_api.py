"""The host-side mirror of the reference's entry points (orca_amd/api: same names, arguments and error
behaviour as LuckyLYM/Orca) against the golden vectors captured from the reference.  These tests are
written the way a test of the reference itself would read.  Needs a real MI355X."""
import argparse

import numpy as np
import pytest
import torch

from conftest import assert_close_fp32
from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch.set_num_threads(1)
DEV = torch.device("cuda:0")


class Data:
  def __init__(self, sources, destinations, timestamps, edge_idxs):
    self.sources, self.destinations, self.timestamps, self.edge_idxs = sources, destinations, timestamps, edge_idxs
    self.n_interactions = len(sources)


@pytest.fixture(scope="module")
def api():
  import orca_amd.api as A
  return A


@pytest.mark.parametrize("tag", ["sorted", "shuf"])
def test_neighbor_finder(api, golden, tag):
  g = golden("sampler.npz")
  src, dst, ts, eidx = g["src"], g["dst"], g["ts"], g["eidx"]
  if tag == "shuf":
    s = g["shuf"]
    src, dst, ts, eidx = src[s], dst[s], ts[s], eidx[s]
  nf = api.get_neighbor_finder(Data(src, dst, ts, eidx), max_node_idx=63, device=DEV)
  assert np.array_equal(np.concatenate(nf.node_to_neighbors), g[f"{tag}_nbr"])
  assert np.array_equal(np.concatenate(nf.node_to_edge_timestamps), g[f"{tag}_ts"])
  qn, qt = g[f"{tag}_q_nodes"], g[f"{tag}_q_ts"]
  for K in (0, 1, 5, 10, 20):
    nbr, ei, et = nf.get_temporal_neighbor(qn, qt, K)
    assert nbr.dtype == np.int32 and ei.dtype == np.int32 and et.dtype == np.float32
    assert np.array_equal(nbr, g[f"{tag}_K{K}_nbr"])
    assert np.array_equal(ei, g[f"{tag}_K{K}_eidx"])
    assert np.array_equal(et.view(np.uint32), g[f"{tag}_K{K}_et"].view(np.uint32))
    deltas = torch.from_numpy(qt[:, np.newaxis] - et).float().numpy()      # embedding_module.py:130-131 on the host
    assert np.array_equal(deltas.view(np.uint32), g[f"{tag}_K{K}_dt"].view(np.uint32))
  for n, t, ln in zip(g[f"{tag}_fb_nodes"], g[f"{tag}_fb_ts"], g[f"{tag}_fb_len"]):
    a, b, c = nf.find_before(int(n), float(t))
    assert len(a) == len(b) == len(c) == ln
  with pytest.raises(IndexError):
    nf.get_temporal_neighbor(np.array([64]), np.array([1.0]), 5)
  # the jitclass constructor form: three ragged lists
  nf2 = api.NeighborFinder(nf.node_to_neighbors, nf.node_to_edge_idxs, nf.node_to_edge_timestamps, device=DEV)
  a = nf2.get_temporal_neighbor(qn, qt, 10)
  assert np.array_equal(a[0], g[f"{tag}_K10_nbr"]) and np.array_equal(a[1], g[f"{tag}_K10_eidx"])


def test_time_encode(api, golden):
  g = golden("time_encode.npz")
  te = api.TimeEncode(100).to(DEV)
  assert np.array_equal(te.w.weight.detach().cpu().numpy().reshape(-1), g["w"])
  assert list(te.state_dict().keys()) == ["w.weight", "w.bias"]
  out = te(torch.from_numpy(g["t"]).to(DEV))
  assert out.shape == (6, 10, 100)
  assert_close_fp32(out.cpu().numpy(), g["out"], 1e-5, "TimeEncode")


@pytest.mark.parametrize("tag", ["de172_K10_", "de1_K10_", "de4_K20_", "de172_K1_"])
def test_temporal_attention_layer(api, golden, tag):
  g = golden("attention.npz")
  d_e = g[tag + "edge"].shape[2]
  layer = api.TemporalAttentionLayer(100, 100, d_e, 100, 100, n_head=2, dropout=0.1)
  layer.load_state_dict({k[len(tag + "sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "sd.")},
                        strict=True)
  layer = layer.to(DEV).eval()
  mask = torch.from_numpy(g[tag + "mask_in"].copy()).to(DEV)
  t = lambda k: torch.from_numpy(g[tag + k]).to(DEV)
  with torch.no_grad():
    out, w = layer(t("src"), t("src_t"), t("nbr"), t("nbr_t"), t("edge"), mask)
  assert_close_fp32(out.cpu().numpy(), g[tag + "out"], 1e-5, tag + "out")
  assert np.array_equal(mask.cpu().numpy(), g[tag + "mask_out"])           # mutated in place, as upstream
  if tag + "weights" in g.files:
    assert_close_fp32(w.cpu().numpy(), g[tag + "weights"], 1e-5, tag + "weights")
  # parameters changed in place -> the packed blob follows
  with torch.no_grad():
    layer.norm.bias.add_(1.0)
  with torch.no_grad():
    out2, _ = layer(t("src"), t("src_t"), t("nbr"), t("nbr_t"), t("edge"), torch.from_numpy(g[tag + "mask_in"].copy()).to(DEV))
  assert_close_fp32(out2.cpu().numpy(), g[tag + "out"] + 1.0, 1e-5, tag + "out after in-place update")


def test_history(api, golden):
  g = golden("history.npz")
  h = api.History(50, 100, DEV, history_budget=10)
  h.push(torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["ids"]).to(DEV))
  assert np.array_equal(h.emb.cpu().numpy(), g["emb"])
  assert np.array_equal(h.pull(torch.from_numpy(g["pull_ids"]).to(DEV)).cpu().numpy(), g["pulled"])
  h.update_flag(None, g["ids"][:5])
  assert np.array_equal(h.cache_flag, g["flag_acc"])
  h.update_flag(g["plan"], g["ids"])
  assert np.array_equal(h.cache_flag, g["flag_plan"])
  h2 = api.History(50, 100, DEV)
  out = h2.push_and_pull(torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["ids"]).to(DEV),
                         torch.from_numpy(g["pull_ids"]).to(DEV))
  assert np.array_equal(out.cpu().numpy(), g["pulled"])
  h2.reset_parameters()
  assert float(h2.emb.abs().sum()) == 0.0


def test_message_aggregators(api, golden):
  g = golden("aggregators.npz")
  from collections import defaultdict
  d = defaultdict(list)
  for i, n in enumerate(g["ids"]):
    d[int(n)].append((torch.from_numpy(g["msgs"][i]), torch.tensor(g["ts"][i])))
  for kind in ("last", "mean"):
    agg = api.get_message_aggregator(kind, DEV)
    u, m, t = agg.aggregate(g["query_ids"], d)
    assert list(u) == list(g[kind + "_ids"])
    assert_close_fp32(m.cpu().numpy(), g[kind + "_msgs"], 1e-6, kind + " msgs")
    assert np.array_equal(t.cpu().numpy(), g[kind + "_ts"])
  with pytest.raises(ValueError):
    api.get_message_aggregator("sum", DEV)
  assert api.get_message_aggregator("last", DEV).aggregate(np.array([20]), d) == ([], [], [])


def test_memory_and_updater(api):
  """Memory / get_memory_updater in the reference's own call pattern (memory_updater.py:25-78)."""
  rs = np.random.RandomState(0)
  N, msg = 300, 472
  mem = api.Memory(N, 100, msg, msg, device=DEV)
  upd = api.get_memory_updater("gru", msg, 100, DEV).to(DEV)
  assert api.get_memory_updater("lstm", msg, 100, DEV) is None
  P = {k: v.detach().cpu() for k, v in upd.memory_updater.state_dict().items()}
  mem.memory = torch.from_numpy(rs.randn(N, 100).astype(np.float32)).to(DEV)
  ids = np.unique(rs.randint(1, N, 80))
  msgs = torch.from_numpy(rs.randn(len(ids), msg).astype(np.float32))
  ts = torch.from_numpy(rs.uniform(0, 1e5, len(ids)).astype(np.float32))
  mem.store_raw_messages(ids, msgs.to(DEV), ts.to(DEV))
  assert np.array_equal(np.nonzero(mem.nodes)[0], ids)
  m0 = mem.memory.cpu().clone()
  index = np.unique(np.concatenate([ids[:30], rs.randint(1, N, 40)]))
  view, last = upd.get_updated_memory(mem, index)
  hit = index[np.isin(index, ids)]
  exp = m0.clone()
  exp[hit] = O.gru_cell(P, msgs[np.searchsorted(ids, hit)], m0[hit])
  assert_close_fp32(view[torch.from_numpy(np.arange(N))].detach().cpu().numpy(), exp.numpy(), 1e-5, "get_updated_memory")
  assert np.array_equal(mem.memory.cpu().numpy(), m0.numpy())               # persistent table untouched
  exp_last = np.zeros(N, np.float32)
  exp_last[hit] = ts.numpy()[np.searchsorted(ids, hit)]
  assert np.array_equal(last.cpu().numpy(), exp_last)
  positives = ids[10:50]
  upd.update_memory(mem, positives)
  mem.clear_messages(positives)
  exp = m0.clone()
  exp[positives] = O.gru_cell(P, msgs[10:50], m0[positives])
  assert_close_fp32(mem.memory.cpu().numpy(), exp.numpy(), 1e-5, "update_memory")
  assert np.array_equal(np.nonzero(mem.nodes)[0], np.setdiff1d(ids, positives))
  bk = mem.backup_memory()
  upd.update_memory_in_test(mem)
  assert not mem.nodes.any()
  rest = np.setdiff1d(ids, positives)
  exp[rest] = O.gru_cell(P, msgs[np.searchsorted(ids, rest)], m0[rest])
  assert_close_fp32(mem.memory.cpu().numpy(), exp.numpy(), 1e-5, "update_memory_in_test")
  mem.restore_memory(bk)
  assert np.array_equal(np.nonzero(mem.nodes)[0], rest)


@pytest.mark.parametrize("tag,graph", [("A1", "g172"), ("A2", "g172"), ("B2", "g172"), ("C2", "g172"),
                                       ("B2z", "g1"), ("A1z", "g1"), ("C2z", "g1")])
def test_tgn_flow_like_the_reference(api, golden, tag, graph):
  """The flow of tests/golden/make_golden.py:run_tgn, with orca_amd.api in place of the reference."""
  g = golden("tgn.npz")
  L, reuse, budget, bs, K, n_train, n_eval, n_nodes, n_edges = [int(x) for x in g[f"{tag}_meta"]]
  src, dst, ts, eidx = (g[f"{graph}_{k}"] for k in ("src", "dst", "ts", "eidx"))
  n_tr = n_train * bs
  args = argparse.Namespace(n_nodes=n_nodes + 1, n_edges=n_edges + 1, reuse=bool(reuse), budget=budget, gradient=False,
                            bs=bs, n_degree=K)
  train_nf = api.get_neighbor_finder(Data(src[:n_tr], dst[:n_tr], ts[:n_tr], eidx[:n_tr]), max_node_idx=n_nodes, device=DEV)
  full_nf = api.get_neighbor_finder(Data(src, dst, ts, eidx), max_node_idx=n_nodes, device=DEV)
  tgn = api.TGN(neighbor_finder=train_nf, node_features=None, edge_features=g[f"{graph}_edge_feats"], device=DEV,
                n_layers=L, n_heads=2, dropout=0.4, use_memory=True, node_dimension=100, time_dimension=100,
                memory_dimension=100, embedding_module_type="graph_attention", message_function="identity",
                aggregator_type="last", memory_updater_type="gru", n_neighbors=K, args=args)
  own = tgn.state_dict()
  sd = {k[len(graph + "_sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(graph + "_sd.")}
  missing = tgn.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
  assert not missing.missing_keys, missing
  tgn.eval()
  negs = g[f"{tag}_negs"]
  b = 0
  for phase, nb in (("train", n_train), ("eval", n_eval)):
    if phase == "eval":
      tgn.set_neighbor_finder(full_nf)
    for _ in range(nb):
      s, e = b * bs, min(n_edges, (b + 1) * bs)
      plan = g[f"{tag}_plan{b}"] if (budget and phase == "train") else None
      with torch.set_grad_enabled(phase == "train"):       # evaluation/evaluation.py:18 runs under no_grad
        pos, neg = tgn.compute_edge_probabilities(src[s:e], dst[s:e], negs[s:e], ts[s:e], eidx[s:e], K, bool(reuse),
                                                  phase == "train", plan)
      what = f"{tag} batch {b} ({phase})"
      assert pos.requires_grad == (phase == "train")        # the train phase builds the autograd graph, as upstream
      assert_close_fp32(pos.detach().cpu().numpy(), g[f"{tag}_b{b}_pos"], 1e-5, what + " pos")
      assert_close_fp32(neg.detach().cpu().numpy(), g[f"{tag}_b{b}_neg"], 1e-5, what + " neg")
      assert_close_fp32(tgn.memory.memory.cpu().numpy(), g[f"{tag}_b{b}_memory"], 1e-5, what + " memory")
      assert np.array_equal(tgn.memory.last_update.cpu().numpy(), g[f"{tag}_b{b}_last_update"])
      assert np.array_equal(tgn.memory.nodes, g[f"{tag}_b{b}_pending"])
      if reuse:
        hs = tgn.embedding_module.histories
        assert_close_fp32(np.stack([h.emb.cpu().numpy() for h in hs]), g[f"{tag}_b{b}_hist"], 1e-5, what + " hist")
        if budget:
          assert np.array_equal(np.stack([h.cache_flag for h in hs]).astype(np.uint8), g[f"{tag}_b{b}_cache_flag"])
      b += 1
  assert_close_fp32(tgn.memory.messages.cpu().numpy(), g[f"{tag}_final_messages"], 1e-5, tag + " messages")
  assert np.array_equal(tgn.memory.timestamps.cpu().numpy(), g[f"{tag}_final_timestamps"])
  with pytest.raises(ValueError):
    api.get_embedding_module("graph_sum", None, None, None, None, None, 1, 100, 172, 100, 100, DEV)


@pytest.mark.parametrize("planner", ["device", "host"])
@pytest.mark.parametrize("tag", ["tight", "loose", "mid"])
def test_get_cache_plan(api, golden, tag, planner, capsys):
  """utils/cache.py:407-428 through the mirror, planned on the device (csrc/plan_dev.hip) and on the host
  (csrc/plan.hip): plan sizes and the tie-free trace against the reference's golden plans, every plan against the
  oracle under the defined (distance, node id) order."""
  g = golden("mrd.npz")
  n_nodes, n_edges, bs, K, budget, num_batch = [int(x) for x in g[f"{tag}_meta"]]
  src, dst, ts, eidx = (g[f"{tag}_{k}"] for k in ("src", "dst", "ts", "eidx"))
  data = Data(src, dst, ts, eidx)
  nf = api.get_neighbor_finder(data, max_node_idx=n_nodes, device=DEV)
  args = argparse.Namespace(n_nodes=n_nodes + 1, bs=bs, n_degree=K, budget=budget, planner=planner)
  plans = api.get_cache_plan(args, data, nf, args.n_nodes, "MRD")
  out = capsys.readouterr().out
  assert "n_reuse" in out and "planning on the host" not in out
  assert len(plans) == num_batch
  assert np.array_equal(np.array([len(p) for p in plans]), g[f"{tag}_plan_len"])
  # exact plans under the defined (distance, node id) order == the oracle in stable mode
  onf = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=n_nodes)
  nb, tl, nl, occ = O.read_batches(src, dst, ts, onf, bs, K, n_nodes + 1)
  ref, n_reuse, n_recompute, avg_rd = O.mrd_plan(n_nodes + 1, nb, budget, tl, nl, occ, stable=True)
  for a, b in zip(plans, ref):
    assert np.array_equal(a, b)
  assert f"n_reuse {n_reuse}, n_recompute {n_recompute}," in out             # cache.py:115,119,427
  assert abs(float(out.split("average_reuse_distance")[1].split()[0]) - avg_rd) <= 1e-9 * max(1.0, abs(avg_rd))
  if tag == "loose":   # no tie group straddles the budget cut: same sets as the reference's own plan
    off = np.concatenate([[0], np.cumsum(g[f"{tag}_plan_len"])])
    for i, p in enumerate(plans):
      assert np.array_equal(np.sort(p), np.sort(g[f"{tag}_plan_flat"][off[i]:off[i + 1]]))
  with pytest.raises(SystemExit):
    api.get_cache_plan(args, data, nf, args.n_nodes, "LRU")


def test_device_planner_equals_host_planner_on_a_long_trace(api, capsys):
  """The two MRD planners (device: csrc/plan_dev.hip, host: csrc/plan.hip) on 1,500 batches of a power-law graph
  with hubs (many ties at the maximum distance, cached ids that are targets again, budget smaller and larger than
  a batch's candidates): identical plans, entry for entry, and identical reuse statistics."""
  from orca_amd import synth
  N, E, bs, K = 4000, 300000, 200, 10
  src, dst, eidx, ts = (x.cpu().numpy() for x in synth.powerlaw_graph(N, E, "cuda:0", seed=17))
  data = Data(src, dst, ts, eidx)
  nf = api.get_neighbor_finder(data, max_node_idx=N, device=DEV)
  # (the last case: batches of 900 interactions with a small budget -- thousands of candidates per batch while the device
  # planner's register sort starts at 1024 keys: its width has to grow twice in mid-trace)
  for budget, bs in ((50, 200), (1000, 200), (6000, 200), (50, 900)):
    res = {}
    for planner in ("device", "host"):
      args = argparse.Namespace(n_nodes=N + 1, bs=bs, n_degree=K, budget=budget, planner=planner)
      plans = api.get_cache_plan(args, data, nf, args.n_nodes, "MRD")
      out = capsys.readouterr().out
      assert "planning on the host" not in out
      res[planner] = (plans, out.split("n_reuse")[1])
    assert len(res["device"][0]) == len(res["host"][0]) == -(-E // bs)
    for a, b in zip(res["device"][0], res["host"][0]):
      assert np.array_equal(a, b)
    assert res["device"][1] == res["host"][1]


def test_eval_edge_prediction_drives_the_hip_model(api, golden):
  """evaluation/evaluation.py:6-47 over orca_amd.api.TGN (Path B): the loop's metrics equal those of the same
  batches scored by hand on an identically initialised model (the loop itself is pinned to the reference in
  tests/test_harness_cpu.py)."""
  from sklearn.metrics import average_precision_score
  g = golden("tgn.npz")
  L, reuse, budget, bs, K, n_train, n_eval, n_nodes, n_edges = [int(x) for x in g["B2_meta"]]
  src, dst, ts, eidx = (g[f"g172_{k}"] for k in ("src", "dst", "ts", "eidx"))
  args = argparse.Namespace(n_nodes=n_nodes + 1, n_edges=n_edges + 1, reuse=True, budget=0, gradient=False, bs=bs, n_degree=K)
  sd = {k[len("g172_sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("g172_sd.")}

  def model():
    nf = api.get_neighbor_finder(Data(src, dst, ts, eidx), max_node_idx=n_nodes, device=DEV)
    tgn = api.TGN(neighbor_finder=nf, node_features=None, edge_features=g["g172_edge_feats"], device=DEV, n_layers=L,
                  n_heads=2, dropout=0.1, use_memory=True, node_dimension=100, time_dimension=100, memory_dimension=100,
                  embedding_module_type="graph_attention", message_function="identity", aggregator_type="last",
                  memory_updater_type="gru", n_neighbors=K, args=args)
    own = tgn.state_dict()
    tgn.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    return tgn.eval()

  lo = n_train * bs
  data = api.Data(src[lo:], dst[lo:], ts[lo:], eidx[lo:], np.zeros(len(src) - lo))
  ap, auc, acc = api.eval_edge_prediction(model(), api.RandEdgeSampler(src, dst, seed=0), data, K, bs, True)
  m2, smp, aps = model(), api.RandEdgeSampler(src, dst, seed=0), []
  for b in range(int(np.ceil(data.n_interactions / bs))):
    r = slice(b * bs, min(data.n_interactions, (b + 1) * bs))
    _, neg = smp.sample(len(data.sources[r]))
    p, n = m2.compute_edge_probabilities(data.sources[r], data.destinations[r], neg, data.timestamps[r], data.edge_idxs[r],
                                         K, True, False, None)
    p, n = p.detach().cpu().numpy(), n.detach().cpu().numpy()
    aps.append(average_precision_score(np.concatenate([np.ones(len(p)), np.zeros(len(n))]), np.concatenate([p, n])))
  assert ap == np.mean(aps) and 0.0 <= auc <= 1.0 and 0.0 <= acc <= 1.0


@pytest.mark.parametrize("reuse,L,hgrad", [(True, 2, False), (False, 2, False), (False, 1, False), (True, 2, True), (True, 3, True)])
def test_training_loop_through_the_drop_in(api, reuse, L, hgrad):
  """The reference's training loop shape (train.py:206-264: compute_edge_probabilities -> BCE on pos / neg ->
  loss.backward() -> Adam step) on orca_amd.api.TGN: the loss back-propagates through the custom ops
  (orca::tgat_gather, orca::gru_cell; orca_amd/torch_ops.py) into the attention layers, the GRU and the link decoder.
  Gradients and the parameters after each Adam step must match OrcaEngine.train_step / step_optimizer (whose
  gradients tests/test_gpu_backward.py checks against torch autograd of the oracle) at 2e-5; the time encoder stays
  frozen as upstream (model/time_encoding.py:20-23).  Dropout 0.  hgrad: args.gradient (train.py:43), the history
  variant whose rows pushed in the same batch keep their graph (orca::history_pull_pushed)."""
  from orca_amd import ops, synth
  from orca_amd.engine import OrcaEngine
  N, E, K, d_e, bs = 1500, 30000, 10, 172, 150
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, "cuda:0", seed=5)
  ef = synth.edge_features(E, d_e, "cuda:0")
  sd = synth.tgn_state_dict(d_e, L, seed=2)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  args = argparse.Namespace(n_nodes=N + 1, n_edges=E + 1, reuse=reuse, budget=0, gradient=hgrad, bs=bs, n_degree=K)
  nf = api.NeighborFinder(None, None, None, device=DEV, _csr=csr)
  tgn = api.TGN(neighbor_finder=nf, node_features=None, edge_features=ef, device=DEV, n_layers=L, n_heads=2, dropout=0.0,
                n_neighbors=K, args=args)
  missing = tgn.load_state_dict({k: v for k, v in sd.items() if k in tgn.state_dict()}, strict=False)
  # (embedding_module.time_encoder is the same module as time_encoder, as upstream: its keys alias the loaded ones)
  assert not [k for k in missing.missing_keys if not k.startswith("embedding_module.time_encoder.")], missing
  tgn.train()
  eng = OrcaEngine(sd, ef, N + 1, L, K, reuse=reuse, budget=0, device="cuda:0", gradient=hgrad)
  eng.set_neighbor_finder(csr)
  opt = torch.optim.Adam(tgn.parameters(), lr=1e-3)               # train.py:180
  eng.optimizer(lr=1e-3)
  crit = torch.nn.BCELoss()                                        # train.py:179
  rs = np.random.RandomState(0)
  h = lambda x: x.cpu().numpy()
  kink_batches = set()
  for b in range(4):
    s, e = 20000 + b * bs, 20000 + (b + 1) * bs
    neg = rs.randint(1, N + 1, bs).astype(np.int64)
    opt.zero_grad()
    pos_p, neg_p = tgn.compute_edge_probabilities(h(src[s:e]), h(dst[s:e]), neg, h(ts[s:e]), h(eidx[s:e]), K, reuse, True, None)
    loss = crit(pos_p.squeeze(), torch.ones(bs, device=DEV)) + crit(neg_p.squeeze(), torch.zeros(bs, device=DEV))   # train.py:231-233
    loss.backward()
    ref_loss, grads = eng.train_step(src[s:e], dst[s:e], torch.from_numpy(neg).to(DEV), ts[s:e], eidx[s:e])
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-5 * max(1.0, abs(float(ref_loss))), f"batch {b} loss"
    named = dict(tgn.named_parameters())
    checked = 0
    for k, gref in grads.items():
      p = named[k]
      if k.endswith("in_proj_bias") and p.grad is not None:
        pass
      assert p.grad is not None, f"batch {b}: no gradient reached {k}"
      # 2e-5: the end-to-end bar of tests/test_gpu_backward.py (two fp32 implementations whose GRU and gather kernels
      # sum in different orders, started from weights an Adam step has moved).  The merger's ReLU makes the gradient
      # discontinuous: the two forwards differ by ~1e-7, and when a pre-activation lies that close to zero its gate
      # differs and everything below it moves by ~1e-3 relative (seen once: one of 45,000 activations of one batch).
      # Such a batch is allowed once, at 5e-3, for the keys below a ReLU gate; everything else is held to 2e-5.
      ga, gr = p.grad.cpu().numpy().reshape(-1), gref.cpu().numpy().reshape(-1)
      top = f"embedding_module.attention_models.{L - 1}."
      smooth = k.startswith("affinity_score.") or k.startswith(top + "norm.") or k.startswith(top + "merger.fc2.")
      try:
        assert_close_fp32(ga, gr, 2e-5, f"batch {b} grad {k}")
      except AssertionError:
        if smooth:
          raise
        assert_close_fp32(ga, gr, 5e-3, f"batch {b} grad {k} (ReLU gate flip)")
        kink_batches.add(b)
      checked += 1
    assert checked >= L * 12 + 4 + 4
    assert named["time_encoder.w.weight"].grad is None and named["time_encoder.w.bias"].grad is None
    opt.step()
    eng.step_optimizer(grads)
    # Adam's first steps move every element by about lr * g / (|g| + 1e-8): where a gradient is at round-off level
    # (|g| ~ 1e-8) the two 1e-5-close gradients give visibly different steps, so the parameters are compared at the
    # size of a step and then made identical again -- the next batch's loss and gradients (1e-5) start from equal weights
    with torch.no_grad():
      for k in grads:
        assert_close_fp32(named[k].detach().cpu().numpy().reshape(-1), eng.params[k].cpu().numpy().reshape(-1), 2e-3,
                          f"batch {b} parameter {k} after the Adam step")
        eng.params[k].copy_(named[k].detach().reshape(eng.params[k].shape))
    eng.refresh()
    assert_close_fp32(tgn.memory.memory.cpu().numpy(), eng.memory.cpu().numpy(), 1e-5, f"batch {b} memory")
    assert_close_fp32(tgn.memory.messages.cpu().numpy(), eng.messages.cpu().numpy(), 1e-5, f"batch {b} messages")
  assert len(kink_batches) <= 1, kink_batches
  for k in grads:        # every parameter the loss reaches has been trained (the GRU from the second batch on)
    assert not torch.equal(named[k].detach().reshape(-1), sd[k].to(DEV).reshape(-1)), f"{k} did not move"


def test_bad_node_ids(api):
  """an id outside the table must never be an out-of-bounds device access.  Host ids (what the reference's
  interface passes) raise IndexError in the mirror, as NumPy indexing does upstream; device ids reach the kernels,
  which skip them (a gather reads zeros)."""
  from orca_amd import ops
  mem = api.Memory(n_nodes=50, memory_dimension=100, input_dimension=0, message_dimension=472, device=DEV)
  with pytest.raises(IndexError):
    mem.get_memory(np.array([3, 50]))
  with pytest.raises(IndexError):
    mem.set_memory(np.array([-51]), torch.zeros(1, 100, device=DEV))
  with pytest.raises(IndexError):
    mem.store_raw_messages(np.array([7, 99]), torch.zeros(2, 472, device=DEV), torch.zeros(2, device=DEV))
  table = torch.arange(50 * 100, dtype=torch.float32, device=DEV).view(50, 100)
  bad = torch.tensor([4, 50, -1, 49, 1 << 40], dtype=torch.int64, device=DEV)
  got = ops.gather_rows(table, bad)
  assert torch.equal(got[0], table[4]) and torch.equal(got[3], table[49])
  assert float(got[1].abs().max()) == 0.0 and float(got[2].abs().max()) == 0.0 and float(got[4].abs().max()) == 0.0
  before = table.clone()
  ops.scatter_rows(table, bad, torch.ones(5, 100, device=DEV))
  assert torch.equal(table[4], torch.ones(100, device=DEV)) and torch.equal(table[49], torch.ones(100, device=DEV))
  keep = torch.ones(50, dtype=torch.bool, device=DEV)
  keep[4] = keep[49] = False
  assert torch.equal(table[keep], before[keep])
  flags = torch.zeros(50, dtype=torch.uint8, device=DEV)
  ops.set_flags(flags, ids64=bad, v=1)
  assert int(flags.sum()) == 2 and int(flags[4]) == 1 and int(flags[49]) == 1
  pos = torch.full((50,), -7, dtype=torch.int32, device=DEV)
  ops.index_positions(torch.tensor([9, 77, -3, 2], dtype=torch.int32, device=DEV), 4, None, pos)
  assert int(pos[9]) == 0 and int(pos[2]) == 3 and int((pos == -7).sum()) == 48
