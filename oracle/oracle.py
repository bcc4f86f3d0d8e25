"""oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatement of Orca's temporal hot path (sample -> TGAT aggregate -> GRU memory / raw
messages -> embedding-reuse history), used ONLY as the checker in tests/, in
__graft_entry__.smoke() and as the `cpu_baseline` ("port") leg of bench.py.  Nothing under
orca_amd/ imports it.

Integer / index work goes through the C file next to this one (orca_oracle.c, ctypes); floating
point work is stock PyTorch CPU fp32 ops sequenced as the reference sequences them.  Every
function cites the reference lines (relative to the LuckyLYM/Orca tree) it follows.

Pinned: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz,
which were produced by running the reference itself (tests/golden/make_golden.py).
Residual "parity unpinned" items: MRD plan membership when a tie group straddles the budget cut
(SURVEY H11), and uniform sampling (not in the reference at all).
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liborca_oracle.so")
_lib_handle = None

c_i64p = ctypes.POINTER(ctypes.c_int64)
c_i32p = ctypes.POINTER(ctypes.c_int32)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_f32p = ctypes.POINTER(ctypes.c_float)


def build():
  subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
  global _lib_handle
  if _lib_handle is None:
    if not os.path.exists(_SO):
      build()
    L = ctypes.CDLL(_SO)
    L.orc_csr_build.restype = ctypes.c_int
    L.orc_find_before.restype = ctypes.c_int64
    L.orc_find_before.argtypes = [c_i64p, c_f64p, ctypes.c_int64, ctypes.c_double]
    L.orc_sample_recent.restype = ctypes.c_int
    L.orc_sample_uniform.restype = ctypes.c_int
    L.orc_segment_last.restype = ctypes.c_int64
    L.orc_segment_mean.restype = ctypes.c_int64
    _lib_handle = L
  return _lib_handle


def _p(a, t):
  return a.ctypes.data_as(t)


# ------------------------------------------------------------------------------------------
# sampler  (utils/util.py:91-151)
# ------------------------------------------------------------------------------------------
class OracleNeighborFinder:
  """utils/util.py:91-105 (build) and :121-151 (NeighborFinder) on a flattened CSR."""

  def __init__(self, sources, destinations, edge_idxs, timestamps, max_node_idx=None):
    src = np.ascontiguousarray(sources, dtype=np.int64)
    dst = np.ascontiguousarray(destinations, dtype=np.int64)
    eid = np.ascontiguousarray(edge_idxs, dtype=np.int64)
    ts = np.ascontiguousarray(timestamps, dtype=np.float64)
    if max_node_idx is None:
      max_node_idx = int(max(src.max(), dst.max()))     # util.py:92
    self.n_rows = int(max_node_idx) + 1
    E = len(src)
    self.indptr = np.zeros(self.n_rows + 1, dtype=np.int64)
    self.nbr = np.zeros(2 * E, dtype=np.int32)
    self.eid = np.zeros(2 * E, dtype=np.int32)
    self.ts = np.zeros(2 * E, dtype=np.float64)
    rc = lib().orc_csr_build(ctypes.c_int64(E), _p(src, c_i64p), _p(dst, c_i64p), _p(eid, c_i64p), _p(ts, c_f64p),
                             ctypes.c_int64(self.n_rows), _p(self.indptr, c_i64p), _p(self.nbr, c_i32p),
                             _p(self.eid, c_i32p), _p(self.ts, c_f64p))
    if rc != 0:
      raise IndexError("node id out of range in get_neighbor_finder")

  @classmethod
  def from_arrays(cls, indptr, nbr, eid, ts):
    """NeighborFinder(node_to_neighbors, node_to_edge_idxs, node_to_edge_timestamps) (utils/util.py:123-126)
    on already-flattened, time-sorted lists."""
    self = cls.__new__(cls)
    self.indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    self.nbr = np.ascontiguousarray(nbr, dtype=np.int32)
    self.eid = np.ascontiguousarray(eid, dtype=np.int32)
    self.ts = np.ascontiguousarray(ts, dtype=np.float64)
    self.n_rows = len(self.indptr) - 1
    return self

  def find_before(self, src_idx, cut_time):
    """utils/util.py:129-131."""
    i = lib().orc_find_before(_p(self.indptr, c_i64p), _p(self.ts, c_f64p), int(src_idx), float(cut_time))
    s = self.indptr[src_idx]
    return self.nbr[s:s + i], self.eid[s:s + i], self.ts[s:s + i]

  def get_temporal_neighbor(self, source_nodes, timestamps, n_neighbors, with_deltas=False):
    """utils/util.py:134-151; with_deltas adds embedding_module.py:130-131."""
    nodes = np.ascontiguousarray(source_nodes, dtype=np.int64)
    qts = np.ascontiguousarray(timestamps, dtype=np.float64)
    Q = len(nodes)
    W = n_neighbors if n_neighbors > 0 else 1
    nbr = np.zeros((Q, W), dtype=np.int32)
    eid = np.zeros((Q, W), dtype=np.int32)
    et = np.zeros((Q, W), dtype=np.float32)
    dt = np.zeros((Q, W), dtype=np.float32)
    rc = lib().orc_sample_recent(_p(self.indptr, c_i64p), _p(self.nbr, c_i32p), _p(self.eid, c_i32p),
                                 _p(self.ts, c_f64p), ctypes.c_int64(self.n_rows), ctypes.c_int64(Q),
                                 _p(nodes, c_i64p), _p(qts, c_f64p), ctypes.c_int(int(n_neighbors)),
                                 _p(nbr, c_i32p), _p(eid, c_i32p), _p(et, c_f32p), _p(dt, c_f32p))
    if rc != 0:
      raise IndexError("node id out of range in get_temporal_neighbor")
    if with_deltas:
      return nbr, eid, et, dt
    return nbr, eid, et


def sample_uniform(nf, source_nodes, timestamps, K, seed):
  """Uniform sampling -- not in the reference ("parity unpinned"); CPU twin of orca_sample_uniform."""
  nodes = np.ascontiguousarray(source_nodes, dtype=np.int64)
  qts = np.ascontiguousarray(timestamps, dtype=np.float64)
  Q = len(nodes)
  nbr = np.zeros((Q, K), dtype=np.int32)
  eid = np.zeros((Q, K), dtype=np.int32)
  et = np.zeros((Q, K), dtype=np.float32)
  dt = np.zeros((Q, K), dtype=np.float32)
  rc = lib().orc_sample_uniform(_p(nf.indptr, c_i64p), _p(nf.nbr, c_i32p), _p(nf.eid, c_i32p), _p(nf.ts, c_f64p),
                                ctypes.c_int64(nf.n_rows), ctypes.c_int64(Q), _p(nodes, c_i64p), _p(qts, c_f64p),
                                ctypes.c_int(int(K)), ctypes.c_uint64(int(seed)), _p(nbr, c_i32p), _p(eid, c_i32p),
                                _p(et, c_f32p), _p(dt, c_f32p))
  if rc != 0:
    raise IndexError("orc_sample_uniform failed")
  return nbr, eid, et, dt


def sample_recent_numpy(indptr, nbr, eid, ts, nodes, qts, K):
  """Pure-NumPy twin of orc_sample_recent (utils/util.py:134-151), for cross-checking the C."""
  W = K if K > 0 else 1
  Q = len(nodes)
  o_n = np.zeros((Q, W), np.int32)
  o_e = np.zeros((Q, W), np.int32)
  o_t = np.zeros((Q, W), np.float32)
  for q in range(Q):
    s, e = indptr[nodes[q]], indptr[nodes[q] + 1]
    i = np.searchsorted(ts[s:e], qts[q])
    if i > 0 and K > 0:
      n = min(i, K)
      o_n[q, K - n:] = nbr[s + i - n:s + i]
      o_e[q, K - n:] = eid[s + i - n:s + i]
      o_t[q, K - n:] = ts[s + i - n:s + i]
  return o_n, o_e, o_t


# ------------------------------------------------------------------------------------------
# segment reduce  (model/tgn_model.py:174-176, modules/message_aggregator.py)
# ------------------------------------------------------------------------------------------
def segment_last(ids):
  ids = np.ascontiguousarray(ids, dtype=np.int64)
  M = len(ids)
  u = np.zeros(M, np.int64)
  idx = np.zeros(M, np.int64)
  U = lib().orc_segment_last(ctypes.c_int64(M), _p(ids, c_i64p), _p(u, c_i64p), _p(idx, c_i64p))
  return u[:U], idx[:U]


def segment_mean(ids, msgs, ts):
  ids = np.ascontiguousarray(ids, dtype=np.int64)
  msgs = np.ascontiguousarray(msgs, dtype=np.float32)
  ts = np.ascontiguousarray(ts, dtype=np.float32)
  M, D = msgs.shape
  u = np.zeros(M, np.int64)
  om = np.zeros((M, D), np.float32)
  ot = np.zeros(M, np.float32)
  U = lib().orc_segment_mean(ctypes.c_int64(M), ctypes.c_int64(D), _p(ids, c_i64p), _p(msgs, c_f32p), _p(ts, c_f32p),
                             _p(u, c_i64p), _p(om, c_f32p), _p(ot, c_f32p))
  return u[:U], om[:U], ot[:U]


# ------------------------------------------------------------------------------------------
# time encoder  (model/time_encoding.py:17-28)
# ------------------------------------------------------------------------------------------
def time_encode_weights(dim=100):
  """time_encoding.py:18: w = 1 / 10**linspace(0, 9, dim) computed in float32; bias 0."""
  return torch.from_numpy(1 / 10 ** np.linspace(0, 9, dim, dtype=np.float32))


def time_encode(t, w, b=None):
  """time_encoding.py:26-27: cos(Linear(1,dim)(t.unsqueeze(2))).  With a bias the argument is the rounded
  product plus the bias (two roundings).  torch's CPU linear is NOT used for this: it rounds once or twice
  depending on the host's MKL code path, which moves single values by O(0.1) at t ~ 1e6.  The reference's
  fixtures all have b = 0, where the order cannot matter: the b != 0 case is parity-unpinned."""
  if b is None:
    return torch.cos(t.unsqueeze(-1) * w)
  return torch.cos(t.unsqueeze(-1) * w + b)


# ------------------------------------------------------------------------------------------
# TGAT attention layer  (model/temporal_attention.py:24-56, utils/util.py:12-24)
# ------------------------------------------------------------------------------------------
def attention_forward(P, src, src_t, nbr, nbr_t, edge, mask, n_head=2, drop_mult=None):
  """P: dict of the layer's state_dict tensors (keys as in SURVEY §8b).  mask is mutated in place
  exactly as temporal_attention.py:46-47 does.  Returns (out [B,D], weights [B,K]).
  drop_mult [B, n_head, K] (0 or 1/(1-p)): nn.MultiheadAttention's dropout on the softmax output
  (temporal_attention.py:17-21) with the keep decisions given explicitly -- torch's own dropout RNG stream cannot be
  reproduced by another implementation, so dropout parity is pinned on the masked arithmetic, not on the stream."""
  B, K = mask.shape
  query = torch.cat([src.unsqueeze(1), src_t], dim=2).reshape(B, -1)            # :41-42
  key = torch.cat([nbr, edge, nbr_t], dim=2)                                    # :43
  invalid = mask.all(dim=1, keepdim=True)                                       # :46
  mask[invalid.squeeze(1), 0] = False                                           # :47
  E = query.shape[1]
  hd = E // n_head
  b = P["multi_head_target.in_proj_bias"]
  q = query @ P["multi_head_target.q_proj_weight"].t() + b[:E]                  # nn.MultiheadAttention, kdim != embed_dim
  k = key @ P["multi_head_target.k_proj_weight"].t() + b[E:2 * E]
  v = key @ P["multi_head_target.v_proj_weight"].t() + b[2 * E:]
  q = (q * math.sqrt(1.0 / hd)).view(B, n_head, 1, hd)
  k = k.view(B, K, n_head, hd).permute(0, 2, 1, 3)
  v = v.view(B, K, n_head, hd).permute(0, 2, 1, 3)
  s = q @ k.transpose(-1, -2)                                                   # [B,H,1,K]
  s = s.masked_fill(mask.view(B, 1, 1, K), float("-inf"))
  a = torch.softmax(s, dim=-1)
  if drop_mult is not None:
    a = a * drop_mult.view(B, n_head, 1, K)
  o = (a @ v).reshape(B, E)
  o = o @ P["multi_head_target.out_proj.weight"].t() + P["multi_head_target.out_proj.bias"]
  w = a.mean(dim=1).reshape(B, K)
  o = o.masked_fill(invalid, 0)                                                 # :52
  w = w.masked_fill(invalid, 0)                                                 # :53
  x = torch.cat([o, src], dim=1)                                                # util.py:22
  h = torch.relu(x @ P["merger.fc1.weight"].t() + P["merger.fc1.bias"])
  y = h @ P["merger.fc2.weight"].t() + P["merger.fc2.bias"]
  y = torch.nn.functional.layer_norm(y, (y.shape[1],), P["norm.weight"], P["norm.bias"], 1e-5)  # :55
  return y, w


def gru_cell(P, x, h):
  """torch.nn.GRUCell as used at modules/memory_updater.py:86 (gate order r,z,n)."""
  gi = x @ P["weight_ih"].t() + P["bias_ih"]
  gh = h @ P["weight_hh"].t() + P["bias_hh"]
  H = h.shape[1]
  r = torch.sigmoid(gi[:, :H] + gh[:, :H])
  z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
  n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
  return (1 - z) * n + z * h


# ------------------------------------------------------------------------------------------
# one TGN step  (model/tgn_model.py:95-191, modules/embedding_module.py:103-285,
#                modules/memory.py, modules/memory_updater.py, modules/history.py)
# ------------------------------------------------------------------------------------------
class OracleTGN:
  """Forward-only restatement of TGN with the graph-attention embedding, identity message
  function, 'last' aggregation and GRU updater -- the reference's defaults (train.py:35-38)."""

  def __init__(self, sd, edge_feats, n_nodes, n_layers, K, reuse, budget, D=100):
    self.sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v).float() for k, v in sd.items()}
    self.edge_feats = torch.as_tensor(edge_feats, dtype=torch.float32)
    self.d_e = self.edge_feats.shape[1]
    self.N = int(n_nodes)                 # args.n_nodes = unique + 1 (train.py:129)
    self.L, self.K, self.reuse, self.budget, self.D = n_layers, K, reuse, budget, D
    self.msg_dim = 2 * D + self.d_e + D   # tgn_model.py:49-50
    self.w_time = self.sd["time_encoder.w.weight"].reshape(-1)
    b = self.sd.get("time_encoder.w.bias")
    self.b_time = b.reshape(-1) if b is not None and bool((b != 0).any()) else None
    self.att = [{k[len(f"embedding_module.attention_models.{i}."):]: v for k, v in self.sd.items()
                 if k.startswith(f"embedding_module.attention_models.{i}.")} for i in range(n_layers)]
    self.gru = {k[len("memory_updater.memory_updater."):]: v for k, v in self.sd.items()
                if k.startswith("memory_updater.memory_updater.")}
    self.aff = {k[len("affinity_score."):]: v for k, v in self.sd.items() if k.startswith("affinity_score.")}
    self.neighbor_finder = None
    self.uniform_seed = None              # set to an int: seeded-uniform sampling (NOT in the reference; see sample_uniform)
    self.batch_counter = 0
    self.test_mode = False
    self.init_memory()
    self.reset_history()

  # modules/memory.py:17-23
  def init_memory(self):
    self.memory = torch.zeros(self.N, self.D)
    self.last_update = torch.zeros(self.N)
    self.pending = np.zeros(self.N, dtype=bool)
    self.messages = torch.zeros(self.N, self.msg_dim)
    self.msg_ts = torch.zeros(self.N)

  # modules/history.py:11-20
  def reset_history(self):
    self.hist = [torch.zeros(self.N, self.D) for _ in range(self.L - 1)] if self.reuse else []
    self.cache_flag = [np.zeros(self.N) for _ in range(self.L - 1)] if self.reuse else []

  # modules/memory_updater.py:56-78
  def get_updated_memory(self, index):
    ids = index[self.pending[index]]
    mem = self.memory.clone()
    if len(ids):
      mem[ids] = gru_cell(self.gru, self.messages[ids], mem[ids])
    return mem

  # modules/memory_updater.py:25-39
  def update_memory(self, positives):
    ids = positives[self.pending[positives]]
    if len(ids) == 0:
      return
    self.last_update[ids] = self.msg_ts[ids]
    self.memory[ids] = gru_cell(self.gru, self.messages[ids], self.memory[ids])

  # modules/memory_updater.py:42-53
  def update_memory_in_test(self):
    ids = np.where(self.pending)[0]
    if len(ids) == 0:
      return
    self.last_update[ids] = self.msg_ts[ids]
    self.memory[ids] = gru_cell(self.gru, self.messages[ids], self.memory[ids])
    self.pending[ids] = False

  def _sample(self, nodes, ts):
    if self.uniform_seed is not None:     # per-batch seed as OrcaEngine._step_seed; the draw key is the query's position
      seed = (self.uniform_seed + self.batch_counter * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
      nbr, eid, et, dt = sample_uniform(self.neighbor_finder, nodes, ts, self.K, seed)
    else:
      nbr, eid, et, dt = self.neighbor_finder.get_temporal_neighbor(nodes, ts, self.K, with_deltas=True)
    return nbr, torch.from_numpy(nbr.astype(np.int64)), torch.from_numpy(eid.astype(np.int64)), torch.from_numpy(dt)

  def _aggregate(self, layer, src_feat, nbr_feat, dt, eid_t, mask):
    """embedding_module.py:318-329 with the time encodings of :126,132 and the gather of :133."""
    B = src_feat.shape[0]
    src_t = time_encode(torch.zeros(B, 1), self.w_time, self.b_time)
    nbr_t = time_encode(dt, self.w_time, self.b_time)
    edge = self.edge_feats[eid_t]
    out, _ = attention_forward(self.att[layer - 1], src_feat, src_t, nbr_feat, nbr_t, edge, mask)
    return out

  # modules/embedding_module.py:116-147
  def compute_embedding(self, memory, nodes, ts, n_layers, train):
    if n_layers == 0:
      if train:
        memory = self.get_updated_memory(np.unique(nodes))
      return memory[torch.from_numpy(nodes)]
    nbr, nbr_t, eid_t, dt = self._sample(nodes, ts)
    mask = nbr_t == 0
    comb_nodes = np.hstack((nodes, nbr.flatten()))
    comb_ts = np.hstack((ts, np.repeat(ts, self.K)))
    comb = self.compute_embedding(memory, comb_nodes, comb_ts, n_layers - 1, train)
    n = len(nodes)
    return self._aggregate(n_layers, comb[:n], comb[n:].view(n, self.K, -1), dt, eid_t, mask)

  # modules/embedding_module.py:154-182 (+ :103-113, modules/history.py:33-40)
  def compute_embedding_reuse(self, memory, nodes, ts, train):
    nbr, nbr_t, eid_t, dt = self._sample(nodes, ts)
    if train:
      memory = self.get_updated_memory(np.unique(np.concatenate((nbr, nodes[:, None]), 1)))
    mask = nbr_t == 0
    flat = nbr_t.flatten()
    n = len(nodes)
    nbr_emb = memory[flat].view(n, self.K, -1)
    src_emb = memory[torch.from_numpy(nodes)]
    for layer in range(1, self.L):
      src_emb = self._aggregate(layer, src_emb, nbr_emb, dt, eid_t, mask.clone())
      self.hist[layer - 1][torch.from_numpy(nodes)] = src_emb.clone()   # push: last write wins (1 thread)
      nbr_emb = self.hist[layer - 1].index_select(0, flat).view(n, self.K, -1)
    return self._aggregate(self.L, src_emb, nbr_emb, dt, eid_t, mask)

  # modules/embedding_module.py:200-212
  def find_uncached(self, nodes, nbr, n_layers):
    if n_layers == 1:
      return np.zeros(0, dtype=np.int64)
    u = np.unique(nbr)
    out = u[~np.isin(u, nodes)]
    flag = self.cache_flag[0][out]
    return out[np.where(flag == 0)[0]][1:]

  # modules/embedding_module.py:215-285 (+ :186-197, modules/history.py:25-30)
  def compute_embedding_budget_reuse(self, memory, nodes, ts, n_layers, cache_plan, train):
    if n_layers == 0:
      if train:
        memory = self.get_updated_memory(np.unique(nodes))
      return memory[torch.from_numpy(nodes)]
    nbr, nbr_t, eid_t, dt = self._sample(nodes, ts)
    mask = nbr_t == 0
    n = len(nodes)
    if n_layers == 1:
      comb_nodes = np.hstack((nodes, nbr.flatten()))
      comb_ts = np.hstack((ts, np.repeat(ts, self.K)))
      comb = self.compute_embedding_budget_reuse(memory, comb_nodes, comb_ts, 0, cache_plan, train)
      return self._aggregate(1, comb[:n], comb[n:].view(n, self.K, -1), dt, eid_t, mask)
    unc = self.find_uncached(nodes, nbr, n_layers)
    if len(unc):
      comb_nodes = np.hstack((nodes, unc)).astype(np.int64)
      comb_ts = np.hstack((ts, np.repeat(np.max(ts), len(unc))))
    else:
      comb_nodes, comb_ts = nodes, ts
    comb = self.compute_embedding_budget_reuse(memory, comb_nodes, comb_ts, n_layers - 1, cache_plan, train)
    h = n_layers - 2
    self.hist[h][torch.from_numpy(comb_nodes)] = comb.clone()
    nbr_emb = self.hist[h].index_select(0, nbr_t.flatten()).view(n, self.K, -1)
    if cache_plan is not None:                               # history.py:26-28
      self.cache_flag[h] = np.zeros(self.N)
      self.cache_flag[h][cache_plan] = 1
    else:                                                    # history.py:30
      self.cache_flag[h][comb_nodes] = 1
    return self._aggregate(n_layers, comb[:n], nbr_emb, dt, eid_t, mask)

  # model/tgn_model.py:172-191 + modules/memory.py:25-28
  def store_raw_messages(self, src, dst, ts, eidx):
    ids = np.concatenate([src, dst])
    partner = np.concatenate([dst, src])
    t2 = np.concatenate([ts, ts])
    e2 = np.concatenate([eidx, eidx])
    uniq, index = segment_last(ids)
    up = partner[index]
    et = torch.from_numpy(t2[index]).float()
    ef = self.edge_feats[torch.from_numpy(e2[index])]
    delta = et - self.last_update[uniq]
    enc = time_encode(delta.unsqueeze(1), self.w_time, self.b_time).view(len(uniq), -1)
    msg = torch.cat([self.memory[uniq], self.memory[up], ef, enc], dim=1)
    self.pending[uniq] = True
    self.messages[uniq] = msg
    self.msg_ts[uniq] = et

  # model/tgn_model.py:95-147
  def compute_temporal_embeddings(self, src, dst, neg, ts, eidx, train, cache_plan=None):
    self.batch_counter += 1
    n = len(src)
    nodes = np.concatenate([src, dst, neg]).astype(np.int64)
    positives = np.unique(np.concatenate([src, dst]))
    t3 = np.concatenate([ts, ts, ts])
    if train:
      self.test_mode = False
    elif not self.test_mode:
      self.update_memory_in_test()
      self.test_mode = True
    memory = self.memory
    with torch.no_grad():
      if self.reuse:
        if self.budget != 0:
          emb = self.compute_embedding_budget_reuse(memory, nodes, t3, self.L, cache_plan, train)
        else:
          emb = self.compute_embedding_reuse(memory, nodes, t3, train)
      else:
        emb = self.compute_embedding(memory, nodes, t3, self.L, train)
      if train:
        self.update_memory(positives)
        self.pending[positives] = False
      self.store_raw_messages(src, dst, ts, eidx)
      if not train:
        self.update_memory(positives)
        self.pending[positives] = False
    return emb

  # model/tgn_model.py:150-157
  def edge_probabilities(self, emb, n):
    se, de, ne = emb[:n], emb[n:2 * n], emb[2 * n:]
    x = torch.cat([torch.cat([se, se]), torch.cat([de, ne])], dim=1)
    h = torch.relu(x @ self.aff["fc1.weight"].t() + self.aff["fc1.bias"])
    s = (h @ self.aff["fc2.weight"].t() + self.aff["fc2.bias"])
    return s[:n].sigmoid(), s[n:].sigmoid()


# ------------------------------------------------------------------------------------------
# MRD cache planner  (utils/cache.py:19-89, 92-166)
# ------------------------------------------------------------------------------------------
def read_batches(src, dst, ts, finder, bs, K, num_embeddings):
  """utils/cache.py:19-89: per batch the unique targets, the non-unique out-of-batch sampled
  neighbours, and per node the list of (batch, kind) occurrences (0 target, 1 neighbour)."""
  n = len(src)
  num_batch = math.ceil(n / bs)
  target_list, ngh_list = [], []
  occur = [[] for _ in range(num_embeddings)]
  for b in range(num_batch):
    s, e = b * bs, min(n, (b + 1) * bs)
    nodes = np.concatenate([src[s:e], dst[s:e]])
    t2 = np.concatenate([ts[s:e], ts[s:e]])
    nbr, _, _ = finder.get_temporal_neighbor(nodes, t2, K)
    nbr = nbr[nbr != 0]
    ut = np.unique(nodes)
    un = np.unique(nbr)
    uin = np.intersect1d(ut, un)
    out = nbr[~np.isin(nbr, uin)]
    target_list.append(ut)
    ngh_list.append(out)
    for t in ut:
      occur[t].append((b, 0))
    for g in np.unique(out):
      occur[g].append((b, 1))
  return num_batch, target_list, ngh_list, occur


def mrd_plan(num_embeddings, num_batch, budget, target_list, ngh_list, occur, stable=True):
  """utils/cache.py:92-166.  Ordering of equal reuse distances: the reference calls np.argsort
  (unstable introsort; Numba's differs again -- SURVEY H11).  stable=True defines the tie-break as
  ascending node id (candidates are sorted-unique), which is what the product implements;
  stable=False reproduces NumPy's default argsort and equals the golden fixtures exactly."""
  MAXD = 100000000
  cache_flag = np.zeros(num_embeddings, dtype=bool)
  time_flag = np.zeros(num_embeddings)
  pos = np.zeros(num_embeddings, dtype=np.int64)
  n_reuse = n_recompute = 0
  total_rd = 0.0
  plans = []
  for b in range(num_batch):
    target, ngh = target_list[b], ngh_list[b]
    c = cache_flag[ngh]
    unc = ngh[~c]
    cached_ngh = ngh[c]
    n_recompute += len(unc)
    n_reuse += len(cached_ngh)
    total_rd += float(np.sum(b - time_flag[cached_ngh]))
    cached = np.where(cache_flag)[0]
    new_computed = np.unique(np.concatenate((unc, target)))
    cand = np.unique(np.concatenate((unc, cached, target)))
    dist = np.empty(len(cand), dtype=np.int64)
    for i, node in enumerate(cand):
      oc = occur[node]
      p = pos[node]
      while p < len(oc) and oc[p][0] <= b:
        p += 1
      pos[node] = p
      if p == len(oc) or oc[p][1] == 0:
        dist[i] = MAXD + 1
      else:
        dist[i] = oc[p][0] - b
    order = np.argsort(dist, kind="stable") if stable else np.argsort(dist)
    to_cache = cand[order][:budget]
    cache_flag[:] = False
    cache_flag[to_cache] = True
    plans.append(to_cache)
    new_nodes = to_cache[np.isin(to_cache, new_computed)]
    time_flag[new_nodes] = b
  return plans, n_reuse, n_recompute, (total_rd / n_reuse if n_reuse else float("nan"))
