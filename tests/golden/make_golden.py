#!/usr/bin/env python3
"""Generate the golden parity fixtures in tests/golden/ from the REAL reference.

Run in the build container only (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py

The reference (LuckyLYM/Orca) has no tests and no golden vectors of its own,
so every pin is produced here by importing the reference *unmodified* from /root/reference
and running it on small seeded synthetic inputs.  `numba` is not installable in this image;
a pass-through stand-in (identity decorators; the jitclass body is plain NumPy) is injected
into sys.modules so that the reference modules import.  Only data (inputs + outputs) is
written; no reference source text is stored.

Conventions: torch.set_num_threads(1) (History.push duplicate ids are
last-write-wins only single-threaded), model.eval() so attention-weight dropout is off,
ids >= 1, feature row 0 = zeros, timestamps float64.
"""
import os
import sys
import types
import argparse

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def install_numba_standin():
  nb = types.ModuleType("numba")

  def passthrough(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
      return a[0]
    return lambda f: f

  nb.njit = passthrough
  nb.jit = passthrough
  nb.prange = range
  nb.boolean = np.bool_
  nb.typeof = type
  typed = types.ModuleType("numba.typed")

  class List(list):
    pass

  typed.List = List
  nb.typed = typed
  exp = types.ModuleType("numba.experimental")
  exp.jitclass = lambda spec=None: (lambda cls: cls)
  nb.experimental = exp
  core = types.ModuleType("numba.core")
  errs = types.ModuleType("numba.core.errors")
  for n in ("NumbaDeprecationWarning", "NumbaPendingDeprecationWarning", "NumbaTypeSafetyWarning"):
    setattr(errs, n, type(n, (Warning,), {}))
  core.errors = errs
  nb.core = core
  for k, v in {"numba": nb, "numba.typed": typed, "numba.experimental": exp,
               "numba.core": core, "numba.core.errors": errs}.items():
    sys.modules[k] = v


install_numba_standin()
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import torch  # noqa: E402

torch.set_num_threads(1)

from utils.util import get_neighbor_finder  # noqa: E402
from utils.cache import get_cache_plan  # noqa: E402
from model.tgn_model import TGN  # noqa: E402
from model.time_encoding import TimeEncode  # noqa: E402
from model.temporal_attention import TemporalAttentionLayer  # noqa: E402
from modules.message_aggregator import get_message_aggregator  # noqa: E402
from modules.history import History  # noqa: E402


class Data:
  def __init__(self, sources, destinations, timestamps, edge_idxs):
    self.sources = sources
    self.destinations = destinations
    self.timestamps = timestamps
    self.edge_idxs = edge_idxs
    self.labels = np.zeros(len(sources))
    self.n_interactions = len(sources)


def synth_graph(seed, n_nodes, n_edges, t_max=2.6e6, tie_frac=0.05, bipartite=False, skew=1.1):
  """Power-law-ish temporal graph; ids 1..n_nodes (0 = padding); ts sorted with exact ties."""
  rs = np.random.RandomState(seed)
  w = 1.0 / np.arange(1, n_nodes + 1) ** skew
  w /= w.sum()
  perm = rs.permutation(n_nodes) + 1
  if bipartite:
    half = n_nodes // 2
    wu = w[:half] / w[:half].sum()
    wi = w[:n_nodes - half] / w[:n_nodes - half].sum()
    src = perm[:half][rs.choice(half, n_edges, p=wu)]
    dst = perm[half:][rs.choice(n_nodes - half, n_edges, p=wi)]
  else:
    src = perm[rs.choice(n_nodes, n_edges, p=w)]
    dst = perm[rs.choice(n_nodes, n_edges, p=w)]
  ts = np.sort(rs.uniform(0, t_max, n_edges))
  n_tie = int(tie_frac * n_edges)
  if n_tie:
    pos = rs.choice(np.arange(1, n_edges), n_tie, replace=False)
    ts[pos] = ts[pos - 1]
    ts = np.sort(ts)
  eidx = np.arange(1, n_edges + 1)
  return src.astype(np.int64), dst.astype(np.int64), ts.astype(np.float64), eidx.astype(np.int64)


# ---------------------------------------------------------------------------------------------
def gen_sampler():
  """utils/util.py:91-151 — get_neighbor_finder + get_temporal_neighbor (+ the Δt expression of
  modules/embedding_module.py:130-131)."""
  src, dst, ts, eidx = synth_graph(11, n_nodes=60, n_edges=900, tie_frac=0.15)
  # self loops + a node with many edges at the same time
  src[5] = dst[5]
  src[100:110] = 7
  ts[100:110] = ts[100]
  ts = np.sort(ts)
  # unsorted-input variant exercises the stable per-node sort (H4)
  rs = np.random.RandomState(5)
  shuf = rs.permutation(len(src))
  data_sorted = Data(src, dst, ts, eidx)
  data_shuf = Data(src[shuf], dst[shuf], ts[shuf], eidx[shuf])
  out = dict(src=src, dst=dst, ts=ts, eidx=eidx, shuf=shuf)
  for tag, data in (("sorted", data_sorted), ("shuf", data_shuf)):
    nf = get_neighbor_finder(data, max_node_idx=63)  # nodes 61..63 have degree 0
    lens = np.array([len(x) for x in nf.node_to_neighbors], dtype=np.int64)
    out[f"{tag}_indptr"] = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    out[f"{tag}_nbr"] = np.concatenate([np.asarray(x) for x in nf.node_to_neighbors]).astype(np.int32)
    out[f"{tag}_eidx"] = np.concatenate([np.asarray(x) for x in nf.node_to_edge_idxs]).astype(np.int32)
    out[f"{tag}_ts"] = np.concatenate([np.asarray(x) for x in nf.node_to_edge_timestamps]).astype(np.float64)
    # queries: random nodes incl. 0 and degree-0 nodes; times: random, exact edge times, before-all, after-all
    q_nodes = np.concatenate([rs.randint(0, 64, 300), [0, 61, 62, 63, 7, 7, 7]]).astype(np.int64)
    q_ts = rs.uniform(0, 2.7e6, len(q_nodes))
    exact = rs.choice(len(ts), 120)
    q_ts[:120] = data.timestamps[exact]  # t equal to an edge time: strictly-before semantics (H1)
    q_nodes[:60] = data.sources[exact[:60]]
    q_nodes[60:120] = data.destinations[exact[60:120]]
    q_ts[120:130] = 0.0
    q_ts[130:140] = 1e9
    q_ts[-3] = ts[100]
    q_ts[-2] = np.nextafter(ts[100], np.inf)
    q_ts[-1] = 1e9
    out[f"{tag}_q_nodes"] = q_nodes
    out[f"{tag}_q_ts"] = q_ts
    for K in (0, 1, 5, 10, 20):
      nbr, ei, et = nf.get_temporal_neighbor(q_nodes, q_ts, K)
      assert nbr.dtype == np.int32 and ei.dtype == np.int32 and et.dtype == np.float32
      deltas = q_ts[:, np.newaxis] - et                        # embedding_module.py:130
      deltas_f32 = torch.from_numpy(deltas).float().numpy()    # embedding_module.py:131
      out[f"{tag}_K{K}_nbr"] = nbr
      out[f"{tag}_K{K}_eidx"] = ei
      out[f"{tag}_K{K}_et"] = et
      out[f"{tag}_K{K}_dt"] = deltas_f32
    # find_before on a few
    fb_nodes = q_nodes[:8]
    fb_ts = q_ts[:8]
    out[f"{tag}_fb_nodes"] = fb_nodes
    out[f"{tag}_fb_ts"] = fb_ts
    out[f"{tag}_fb_len"] = np.array([len(nf.find_before(int(n), float(t))[0]) for n, t in zip(fb_nodes, fb_ts)])
  np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)
  print("sampler.npz", {k: v.shape for k, v in list(out.items())[:6]})


# ---------------------------------------------------------------------------------------------
def gen_time_encode():
  """model/time_encoding.py:17-28."""
  te = TimeEncode(100)
  rs = np.random.RandomState(3)
  t = np.concatenate([rs.uniform(0, 2.6e6, 40), rs.uniform(0, 10, 10), [0.0, 1.0, 1e8, 3.1415927, 12345.678],
                      rs.uniform(0, 1e5, 5)]).astype(np.float32).reshape(6, 10)
  out = te(torch.from_numpy(t)).numpy()
  np.savez_compressed(os.path.join(OUT, "time_encode.npz"), t=t, out=out,
                      w=te.w.weight.detach().numpy().reshape(-1), b=te.w.bias.detach().numpy())
  print("time_encode.npz", out.shape)


# ---------------------------------------------------------------------------------------------
def sd_to_np(sd, prefix=""):
  return {prefix + k: v.detach().cpu().numpy() for k, v in sd.items()}


def gen_attention():
  """model/temporal_attention.py:24-56 (dense-input drop-in boundary), d_e = 172 and d_e = 1 (H12)."""
  out = {}
  for d_e, K, B, seed in ((172, 10, 37, 0), (1, 10, 21, 1), (4, 20, 9, 2), (172, 1, 5, 3)):
    torch.manual_seed(seed)
    layer = TemporalAttentionLayer(100, 100, d_e, 100, 100, n_head=2, dropout=0.1).eval()
    # non-trivial biases / LN affine so every parameter is exercised
    with torch.no_grad():
      for p in layer.parameters():
        if p.dim() == 1:
          p.add_(0.1 * torch.randn_like(p))
    rs = np.random.RandomState(seed)
    src = rs.randn(B, 100).astype(np.float32)
    src_t = np.cos(rs.uniform(0, 1, (B, 1, 100))).astype(np.float32)
    src_t[: B // 2] = 1.0   # cos(0) rows as in the real flow
    nbr = rs.randn(B, K, 100).astype(np.float32)
    nbr_t = np.cos(rs.uniform(0, 100, (B, K, 100))).astype(np.float32)
    edge = rs.randn(B, K, d_e).astype(np.float32)
    mask = rs.rand(B, K) < 0.4
    mask[0] = True            # all-padded rows (H6)
    mask[B - 1] = True
    mask[1] = False
    if K > 1:
      mask[2, :-1] = True
      mask[2, -1] = False
    mask_in = mask.copy()
    mask_t = torch.from_numpy(mask.copy())
    with torch.no_grad():
      o, wts = layer(torch.from_numpy(src), torch.from_numpy(src_t), torch.from_numpy(nbr),
                     torch.from_numpy(nbr_t), torch.from_numpy(edge), mask_t)
    tag = f"de{d_e}_K{K}_"
    out.update(sd_to_np(layer.state_dict(), tag + "sd."))
    out.update({tag + "src": src, tag + "src_t": src_t, tag + "nbr": nbr, tag + "nbr_t": nbr_t,
                tag + "edge": edge, tag + "mask_in": mask_in, tag + "mask_out": mask_t.numpy(),
                tag + "out": o.numpy()})
    if K > 1:   # at K == 1 the reference's .squeeze() drops the K axis and the weights come out [B,B]
      out[tag + "weights"] = wts.numpy().reshape(B, K)
  np.savez_compressed(os.path.join(OUT, "attention.npz"), **out)
  print("attention.npz", len(out), "arrays")


# ---------------------------------------------------------------------------------------------
def gen_aggregators():
  """modules/message_aggregator.py:38-54 (last), :61-81 (mean)."""
  rs = np.random.RandomState(9)
  M, D = 40, 12
  ids = rs.randint(1, 9, M).astype(np.int64)
  msgs = rs.randn(M, D).astype(np.float32)
  ts = np.sort(rs.uniform(0, 100, M)).astype(np.float32)
  from collections import defaultdict
  d = defaultdict(list)
  for i, n in enumerate(ids):
    d[int(n)].append((torch.from_numpy(msgs[i]), torch.tensor(ts[i])))
  out = dict(ids=ids, msgs=msgs, ts=ts)
  query_ids = np.array([1, 2, 3, 3, 5, 8, 20], dtype=np.int64)  # 20 has no message
  out["query_ids"] = query_ids
  for kind in ("last", "mean"):
    agg = get_message_aggregator(kind, "cpu")
    u, m, t = agg.aggregate(query_ids, d)
    out[kind + "_ids"] = np.array(u, dtype=np.int64)
    out[kind + "_msgs"] = m.numpy()
    out[kind + "_ts"] = t.numpy()
  np.savez_compressed(os.path.join(OUT, "aggregators.npz"), **out)
  print("aggregators.npz")


# ---------------------------------------------------------------------------------------------
def gen_history():
  """modules/history.py:25-46 — push with duplicate ids (last write wins, 1 thread), pull, flags."""
  rs = np.random.RandomState(4)
  h = History(50, 100, "cpu", history_budget=10)
  ids = rs.randint(0, 50, 80).astype(np.int64)
  x = rs.randn(80, 100).astype(np.float32)
  h.push(torch.from_numpy(x), torch.from_numpy(ids))
  pull_ids = rs.randint(0, 50, 64).astype(np.int64)
  pulled = h.pull(torch.from_numpy(pull_ids)).numpy()
  plan = np.array([3, 9, 27], dtype=np.int64)
  h.update_flag(None, ids[:5])
  flag_acc = h.cache_flag.copy()
  h.update_flag(plan, ids)
  flag_plan = h.cache_flag.copy()
  np.savez_compressed(os.path.join(OUT, "history.npz"), ids=ids, x=x, emb=h.emb.numpy(), pull_ids=pull_ids,
                      pulled=pulled, plan=plan, flag_acc=flag_acc, flag_plan=flag_plan)
  print("history.npz")


# ---------------------------------------------------------------------------------------------
def make_args(n_nodes, n_edges, reuse, budget, bs, K):
  return argparse.Namespace(n_nodes=n_nodes + 1, n_edges=n_edges + 1, reuse=reuse, budget=budget,
                            gradient=False, bs=bs, n_degree=K)


def run_tgn(tag, out, graph, edge_feats, sd_full, n_layers, reuse, budget, bs, K, n_train, n_eval, neg_seed):
  """model/tgn_model.py:95-157 in train ordering (train=True, module in eval() so dropout is off),
  then eval ordering (train=False).  Captures embeddings, probabilities and the state tables."""
  src, dst, ts, eidx = graph
  n_nodes = int(max(src.max(), dst.max()))
  n_edges = len(src)
  n_tr = n_train * bs
  train_data = Data(src[:n_tr], dst[:n_tr], ts[:n_tr], eidx[:n_tr])
  full_data = Data(src, dst, ts, eidx)
  args = make_args(n_nodes, n_edges, reuse, budget, bs, K)
  train_nf = get_neighbor_finder(train_data, max_node_idx=n_nodes)
  full_nf = get_neighbor_finder(full_data, max_node_idx=n_nodes)
  plans = None
  if reuse and budget > 0:
    plans = get_cache_plan(args, train_data, train_nf, args.n_nodes, "MRD")
    for i, p in enumerate(plans):
      out[f"{tag}_plan{i}"] = np.asarray(p, dtype=np.int64)
  torch.manual_seed(0)
  tgn = TGN(neighbor_finder=train_nf, node_features=None, edge_features=edge_feats, device=torch.device("cpu"),
            n_layers=n_layers, n_heads=2, dropout=0.4, use_memory=True, node_dimension=100, time_dimension=100,
            memory_dimension=100, embedding_module_type="graph_attention", message_function="identity",
            aggregator_type="last", memory_updater_type="gru", n_neighbors=K, args=args)
  missing = tgn.load_state_dict({k: torch.from_numpy(v) for k, v in sd_full.items()
                                 if k in tgn.state_dict()}, strict=False)
  assert not missing.missing_keys, missing
  tgn.eval()
  rs = np.random.RandomState(neg_seed)
  negs = rs.randint(1, n_nodes + 1, n_edges).astype(np.int64)
  out[f"{tag}_negs"] = negs
  out[f"{tag}_meta"] = np.array([n_layers, int(reuse), budget, bs, K, n_train, n_eval, n_nodes, n_edges])
  with torch.no_grad():
    b = 0
    for phase, nb in (("train", n_train), ("eval", n_eval)):
      if phase == "eval":
        tgn.set_neighbor_finder(full_nf)
      for _ in range(nb):
        s, e = b * bs, min(n_edges, (b + 1) * bs)
        cache_plan = plans[b] if (plans is not None and phase == "train") else None
        se, de, ne = tgn.compute_temporal_embeddings(src[s:e], dst[s:e], negs[s:e], ts[s:e], eidx[s:e], K,
                                                     reuse, phase == "train", cache_plan)
        n = e - s
        score = tgn.affinity_score(torch.cat([se, se], dim=0), torch.cat([de, ne])).squeeze(dim=0)
        out[f"{tag}_b{b}_emb"] = torch.cat([se, de, ne]).numpy()
        out[f"{tag}_b{b}_pos"] = score[:n].sigmoid().numpy()
        out[f"{tag}_b{b}_neg"] = score[n:].sigmoid().numpy()
        out[f"{tag}_b{b}_memory"] = tgn.memory.memory.numpy().copy()
        out[f"{tag}_b{b}_last_update"] = tgn.memory.last_update.numpy().copy()
        out[f"{tag}_b{b}_pending"] = tgn.memory.nodes.copy()
        if reuse:
          hs = tgn.embedding_module.histories
          out[f"{tag}_b{b}_hist"] = np.stack([h.emb.numpy().copy() for h in hs])
          out[f"{tag}_b{b}_cache_flag"] = np.stack([h.cache_flag.copy() for h in hs]).astype(np.uint8)
        b += 1
  out[f"{tag}_final_messages"] = tgn.memory.messages.numpy().copy()
  out[f"{tag}_final_timestamps"] = tgn.memory.timestamps.numpy().copy()


def gen_tgn():
  out = {}
  # one weight set (L=2, d_e=172) shared by paths A(L=1), A(L=2), B, C; one (L=2, d_e=1) for H12
  for d_e in (172, 1):
    n_nodes, n_edges = (90, 520) if d_e == 172 else (120, 640)
    graph = synth_graph(21 + d_e, n_nodes, n_edges, tie_frac=0.05, skew=0.9)
    rs = np.random.RandomState(100 + d_e)
    if d_e == 1:
      edge_feats = np.zeros((n_edges + 1, 1))                   # train.py:141
    else:
      edge_feats = rs.randn(n_edges + 1, d_e).astype(np.float32)
      edge_feats[0] = 0
    args = make_args(n_nodes, n_edges, True, 0, 40, 10)
    torch.manual_seed(7 + d_e)
    proto = TGN(neighbor_finder=None, node_features=None, edge_features=edge_feats, device=torch.device("cpu"),
                n_layers=2, n_heads=2, dropout=0.4, use_memory=True, node_dimension=100, time_dimension=100,
                memory_dimension=100, embedding_module_type="graph_attention", message_function="identity",
                aggregator_type="last", memory_updater_type="gru", n_neighbors=10, args=args)
    with torch.no_grad():
      for name, p in proto.named_parameters():
        if p.dim() == 1 and "time_encoder" not in name:
          p.add_(0.05 * torch.randn_like(p))
    sd = sd_to_np(proto.state_dict())
    g = f"g{d_e}"
    out.update({f"{g}_sd.{k}": v for k, v in sd.items()})
    out.update({f"{g}_src": graph[0], f"{g}_dst": graph[1], f"{g}_ts": graph[2], f"{g}_eidx": graph[3],
                f"{g}_edge_feats": edge_feats.astype(np.float32)})
    if d_e == 172:
      run_tgn("A1", out, graph, edge_feats, sd, n_layers=1, reuse=False, budget=0, bs=40, K=10, n_train=6, n_eval=2, neg_seed=1)
      run_tgn("A2", out, graph, edge_feats, sd, n_layers=2, reuse=False, budget=0, bs=40, K=10, n_train=5, n_eval=2, neg_seed=2)
      run_tgn("B2", out, graph, edge_feats, sd, n_layers=2, reuse=True, budget=0, bs=40, K=10, n_train=8, n_eval=2, neg_seed=3)
      run_tgn("C2", out, graph, edge_feats, sd, n_layers=2, reuse=True, budget=12, bs=40, K=10, n_train=8, n_eval=0, neg_seed=4)
    else:
      run_tgn("B2z", out, graph, edge_feats, sd, n_layers=2, reuse=True, budget=0, bs=50, K=10, n_train=8, n_eval=2, neg_seed=5)
      run_tgn("A1z", out, graph, edge_feats, sd, n_layers=1, reuse=False, budget=0, bs=50, K=5, n_train=4, n_eval=1, neg_seed=6)
      # BASELINE config 3's shape: zero features, two layers, budgeted reuse with an MRD plan
      run_tgn("C2z", out, graph, edge_feats, sd, n_layers=2, reuse=True, budget=15, bs=50, K=10, n_train=8, n_eval=0, neg_seed=7)
  np.savez_compressed(os.path.join(OUT, "tgn.npz"), **out)
  sz = os.path.getsize(os.path.join(OUT, "tgn.npz"))
  print("tgn.npz", len(out), "arrays", sz // 1024, "KiB")


# ---------------------------------------------------------------------------------------------
def gen_mrd():
  """utils/cache.py:19-89 (read_batches) + :92-166 (MRD_numba) through get_cache_plan :407-428.
  H11: plan membership under tied reuse distances is sort-implementation-defined; the 'loose'
  trace has budget >= candidates-with-finite-distance so the cut never splits a finite tie group."""
  import contextlib
  import io
  import utils.cache as rc
  out = {}
  for tag, (n_nodes, n_edges, bs, K, budget, seed) in {
      "tight": (80, 1200, 30, 5, 6, 31), "loose": (60, 900, 30, 5, 400, 32), "mid": (200, 2000, 50, 10, 40, 33)}.items():
    src, dst, ts, eidx = synth_graph(seed, n_nodes, n_edges, skew=1.0)
    data = Data(src, dst, ts, eidx)
    nf = get_neighbor_finder(data, max_node_idx=n_nodes)
    args = make_args(n_nodes, n_edges, True, budget, bs, K)
    num_batch, target_list, ngh_list, occur_list, *stats = rc.read_batches(args, data, nf, args.n_nodes)
    plan, n_reuse, n_recompute, avg = rc.MRD_numba(args.n_nodes, num_batch, budget, target_list, ngh_list, occur_list)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
      plan2 = get_cache_plan(args, data, nf, args.n_nodes, "MRD")
    assert all(np.array_equal(a, b) for a, b in zip(plan, plan2))
    out.update({f"{tag}_src": src, f"{tag}_dst": dst, f"{tag}_ts": ts, f"{tag}_eidx": eidx,
                f"{tag}_meta": np.array([n_nodes, n_edges, bs, K, budget, num_batch]),
                f"{tag}_stats": np.array([n_reuse, n_recompute] + list(stats), dtype=np.int64),
                f"{tag}_avg_reuse_distance": np.array([avg])})
    out[f"{tag}_plan_len"] = np.array([len(p) for p in plan])
    out[f"{tag}_plan_flat"] = np.concatenate([np.asarray(p, dtype=np.int64) for p in plan])
    out[f"{tag}_target_len"] = np.array([len(x) for x in target_list])
    out[f"{tag}_target_flat"] = np.concatenate([np.asarray(x, dtype=np.int64) for x in target_list])
    out[f"{tag}_ngh_len"] = np.array([len(x) for x in ngh_list])
    out[f"{tag}_ngh_flat"] = np.concatenate([np.asarray(x, dtype=np.int64) for x in ngh_list] + [np.zeros(0, np.int64)])
  np.savez_compressed(os.path.join(OUT, "mrd.npz"), **out)
  print("mrd.npz")


def gen_harness():
  """The callers either side of the path: utils/data_processing.py:31-120
  (get_data on the ml_{name}.csv / .npy layout), utils/util.py:68-87 (RandEdgeSampler) and
  evaluation/evaluation.py:6-47 (eval_edge_prediction, here with a deterministic stand-in model so that only the
  loop and the metrics are pinned)."""
  import contextlib
  import io
  import tempfile
  import pandas as pd
  from utils.data_processing import get_data, load_feat
  from utils.util import RandEdgeSampler
  from evaluation.evaluation import eval_edge_prediction
  out = {}
  # ---- get_data on a synthetic bipartite interaction file ------------------------------------------
  rs = np.random.RandomState(41)
  n_users, n_items, n_edges = 300, 120, 6000
  u = rs.randint(1, n_users + 1, n_edges)
  i = rs.randint(n_users + 1, n_users + n_items + 1, n_edges)
  ts = np.sort(rs.uniform(0, 2.6e6, n_edges))
  label = (rs.uniform(size=n_edges) < 0.02).astype(np.int64)
  idx = np.arange(1, n_edges + 1)
  feats = rs.randn(n_edges + 1, 8).astype(np.float64)
  cwd = os.getcwd()
  with tempfile.TemporaryDirectory() as tmp:
    os.makedirs(os.path.join(tmp, "data", "toy"))
    os.makedirs(os.path.join(tmp, "run"))
    pd.DataFrame({"u": u, "i": i, "ts": ts, "label": label, "idx": idx}).to_csv(
        os.path.join(tmp, "data", "toy", "ml_toy.csv"))
    np.save(os.path.join(tmp, "data", "toy", "ml_toy.npy"), feats)
    os.chdir(os.path.join(tmp, "run"))
    try:
      with contextlib.redirect_stdout(io.StringIO()):
        full, train, val, test, nn_val, nn_test, n_nodes, n_e = get_data("toy")
      node_f, edge_f = load_feat("toy")
    finally:
      os.chdir(cwd)
  assert node_f is None and np.array_equal(edge_f, feats)
  out.update(data_u=u, data_i=i, data_ts=ts, data_label=label, data_idx=idx, data_feats=feats,
             data_counts=np.array([n_nodes, n_e]))
  for name, d in dict(full=full, train=train, val=val, test=test, nn_val=nn_val, nn_test=nn_test).items():
    out[f"data_{name}_edge_idxs"] = np.asarray(d.edge_idxs)
    out[f"data_{name}_sources"] = np.asarray(d.sources)
    out[f"data_{name}_n_unique_nodes"] = np.array([d.n_unique_nodes])
  # ---- RandEdgeSampler -----------------------------------------------------------------------------
  smp = RandEdgeSampler(full.sources, full.destinations, seed=3)
  a1 = smp.sample(50)
  a2 = smp.sample(7)
  smp.reset_random_state()
  a3 = smp.sample(50)
  assert np.array_equal(a1[0], a3[0]) and np.array_equal(a1[1], a3[1])
  out.update(neg_src_50=a1[0], neg_dst_50=a1[1], neg_src_7=a2[0], neg_dst_7=a2[1])
  # ---- eval_edge_prediction with a deterministic stand-in model ------------------------------------
  class Model:
    def eval(self):
      return self

    def compute_edge_probabilities(self, s, d, n, t, e, n_neighbors, reuse, train, cache):
      pos = 1.0 / (1.0 + np.exp(-np.sin(0.37 * s + 0.11 * d + 1e-6 * t)))
      neg = 1.0 / (1.0 + np.exp(-np.sin(0.37 * s + 0.11 * n + 1e-6 * t) + 0.4))
      return torch.from_numpy(pos.reshape(-1, 1)), torch.from_numpy(neg.reshape(-1, 1))

  smp = RandEdgeSampler(full.sources, full.destinations, seed=0)
  ap, auc, acc = eval_edge_prediction(Model(), smp, val, 10, 200)
  out["eval_metrics"] = np.array([ap, auc, acc])
  np.savez_compressed(os.path.join(OUT, "harness.npz"), **out)
  print("harness.npz")


if __name__ == "__main__":
  which = sys.argv[1:] or ["sampler", "time", "attention", "agg", "history", "tgn", "mrd", "harness"]
  fns = dict(sampler=gen_sampler, time=gen_time_encode, attention=gen_attention, agg=gen_aggregators,
             history=gen_history, tgn=gen_tgn, mrd=gen_mrd, harness=gen_harness)
  for w in which:
    fns[w]()
