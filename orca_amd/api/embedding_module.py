"""modules/embedding_module.py:64-330,364-427 -- GraphAttentionEmbedding: the recursion / reuse control of
the three embedding algorithms, re-expressed on device tensors (ids, flags and tables stay in HBM; the
gathers, time encodings and attention of one layer are ONE fused kernel launch).  Under autograd (the reference's
training loop, train.py:225-226) every attention evaluation is an orca::tgat_gather node: gradients flow into the
layer parameters, into the previous layer's output rows and into the updated-memory rows (hence the GRU); history
rows are detached as upstream (history.py:32-40)."""
import numpy as np
import torch
from torch import nn

from .. import ops
from .history import History
from .temporal_attention import TemporalAttentionLayer


class GraphAttentionEmbedding(nn.Module):
  def __init__(self, node_features, edge_features, memory, neighbor_finder, time_encoder, n_layers, n_node_features,
               n_edge_features, n_time_features, embedding_dimension, device, n_heads=2, dropout=0.1, use_memory=True,
               reuse=False, history_budget=-1, args=None, num_nodes=-1):
    super().__init__()
    self.node_features = node_features
    self.edge_features = edge_features
    self.neighbor_finder = neighbor_finder
    self.time_encoder = time_encoder
    self.n_layers = n_layers
    self.n_node_features = n_node_features
    self.n_edge_features = n_edge_features
    self.n_time_features = n_time_features
    self.dropout = dropout
    self.embedding_dimension = embedding_dimension
    self.device = device
    self.use_memory = use_memory
    self.args = args
    self.budget = history_budget
    if reuse:                                                       # embedding_module.py:78-81
      self.histories = nn.ModuleList([History(num_nodes, embedding_dimension, device, history_budget)
                                      for _ in range(n_layers - 1)])
    self.attention_models = nn.ModuleList([                          # :305-315
        TemporalAttentionLayer(n_node_features=n_node_features, n_neighbors_features=n_node_features,
                               n_edge_features=n_edge_features, time_dim=n_time_features, n_head=n_heads,
                               dropout=dropout, output_dimension=n_node_features) for _ in range(n_layers)])

  # :83-101
  def backup_history(self):
    return [self.histories[i].emb.clone() for i in range(self.n_layers - 1)]

  def restore_history(self, backup):
    for i in range(self.n_layers - 1):
      self.histories[i].emb = backup[i]

  def reset_history(self):
    for i in range(self.n_layers - 1):
      self.histories[i].reset_parameters()

  def detach_history(self):
    for i in range(self.n_layers - 1):
      self.histories[i].detach_history()

  # ---- helpers --------------------------------------------------------------------------------
  def _dev(self, x, dtype):
    if not isinstance(x, torch.Tensor):
      x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=self.device, dtype=dtype).contiguous()

  def _edge_table(self):
    ef = self.edge_features
    if not (isinstance(ef, torch.Tensor) and ef.is_cuda and ef.dtype == torch.float32 and ef.is_contiguous()):
      ef = self.edge_features = self._dev(ef, torch.float32)
    return ef

  def _sample(self, nodes, ts, K):
    nbr, eid, _, dt = self.neighbor_finder.get_temporal_neighbor_device(nodes, ts, K)
    return nbr, eid, dt

  def _memory_tables(self, memory, ids32, ids64, memory_updater, train):
    """:119-123 / :159-161 / :217-221.  train: `memory` is the Memory module and the rows read by this
    batch get GRU(message, memory) through the overlay; eval: `memory` is the raw table."""
    if train:
      view, _ = memory_updater.get_updated_memory(memory, torch.cat(
          [ids64] + ([ids32.flatten().to(torch.int64)] if ids32 is not None else [])))
      return dict(table=view.table, overlay=view.overlay, flags=view.flags, upd_rows=view.upd_rows, pos=view.pos)
    table = memory.memory if hasattr(memory, "memory") else memory
    return dict(table=table.contiguous(), overlay=None, flags=None)

  def _attend(self, layer, Q, K, nbr, eid, dt, src, src_idx, nb, nbr_idx):
    te = self.time_encoder.w
    att = self.attention_models[layer - 1]
    tw, tb = te.weight.detach().reshape(-1).contiguous(), te.bias.detach().contiguous()
    if torch.is_grad_enabled() and (any(p.requires_grad for p in att.parameters()) or src["table"].requires_grad or
                                    nb["table"].requires_grad or src.get("upd_rows") is not None):
      # the training loop (train.py:225-226): the layer as an autograd node (orca::tgat_gather, torch_ops.py)
      from .. import torch_ops  # noqa: F401
      named = dict(att.named_parameters())
      m = src if src.get("flags") is not None else nb
      p, seed = att.dropout_args()
      return torch.ops.orca.tgat_gather([named[k] for k in ops.TGAT_KEYS], att.packed(), self.n_edge_features, Q, K,
                                        src["table"], src_idx, nb["table"], nbr_idx, nbr, self._edge_table(), eid, dt,
                                        tw, tb, m.get("overlay"), m.get("flags"), m.get("upd_rows"), m.get("pos"), False,
                                        p, seed)
    kw = dict(src_table=src["table"], src_idx=src_idx, nbr_table=nb["table"], nbr_idx=nbr_idx, nbr_ids=nbr,
              edge_table=self._edge_table(), edge_idx=eid, dt=dt, time_w=tw, time_b=tb)
    if src.get("flags") is not None:
      kw.update(src_overlay=src["overlay"], src_flags=src["flags"])
    if nb.get("flags") is not None:
      kw.update(nbr_overlay=nb["overlay"], nbr_flags=nb["flags"])
    p, seed = att.dropout_args()
    with torch.no_grad():
      out, _ = ops.tgat_forward(att.packed(), self.n_edge_features, Q, K, dropout_p=p, dropout_seed=seed, **kw)
    return out

  # ---- Path A: :116-147 --------------------------------------------------------------------------
  def compute_embedding(self, memory, source_nodes, timestamps, n_layers, n_neighbors, memory_updater=None, train=True):
    assert (n_layers >= 0)
    nodes = self._dev(source_nodes, torch.int64)
    ts = self._dev(timestamps, torch.float64)
    if n_layers == 0:
      m = self._memory_tables(memory, None, nodes, memory_updater, train)
      from .memory_updater import UpdatedMemoryView
      return UpdatedMemoryView(m["table"], m["overlay"], m["flags"], m.get("upd_rows"), m.get("pos"))[nodes]
    K = n_neighbors
    f_nodes, f_ts, samples = [nodes], [ts], []
    for _ in range(n_layers):
      nbr, eid, dt = self._sample(f_nodes[-1], f_ts[-1], K)
      samples.append((nbr, eid, dt))
      f_nodes.append(torch.cat([f_nodes[-1], nbr.flatten().to(torch.int64)]))
      f_ts.append(torch.cat([f_ts[-1], f_ts[-1].repeat_interleave(K)]))
    m = self._memory_tables(memory, None, f_nodes[-1], memory_updater, train)
    emb = None
    for layer in range(1, n_layers + 1):
      lvl = n_layers - layer
      nbr, eid, dt = samples[lvl]
      n = f_nodes[lvl].numel()
      if layer == 1:
        emb = self._attend(1, n, K, nbr, eid, dt, m, f_nodes[lvl].to(torch.int32), m, nbr)
      else:
        emb = self._attend(layer, n, K, nbr, eid, dt, dict(table=emb), None, dict(table=emb[n:]), None)
    return emb

  # ---- Path B: :154-182 --------------------------------------------------------------------------
  def compute_embedding_reuse(self, memory, source_nodes, timestamps, n_layers, n_neighbors, batch_id,
                              memory_updater=None, train=True):
    nodes = self._dev(source_nodes, torch.int64)
    ts = self._dev(timestamps, torch.float64)
    K, n = n_neighbors, nodes.numel()
    nbr, eid, dt = self._sample(nodes, ts, K)
    m = self._memory_tables(memory, nbr, nodes, memory_updater, train)
    emb = self._attend(1, n, K, nbr, eid, dt, m, nodes.to(torch.int32), m, nbr)
    for layer in range(2, n_layers + 1):
      h = self.histories[layer - 2]
      h.update_times[nodes.cpu()] = batch_id                       # :107
      if getattr(self.args, "gradient", False) and torch.is_grad_enabled() and emb.requires_grad:
        rows = h.push_and_pull(emb, nodes, nbr.reshape(-1))         # :112, rows of this batch keep their graph
        emb = self._attend(layer, n, K, nbr, eid, dt, dict(table=emb), None, dict(table=rows), None)
        continue
      h.push(emb, nodes)                                            # :109
      emb = self._attend(layer, n, K, nbr, eid, dt, dict(table=emb), None, dict(table=h.emb), nbr)   # :110 pull, fused
    return emb

  # ---- Path C: :186-285 --------------------------------------------------------------------------
  def find_uncached_out_of_batch_neighbors(self, source, neighbors, n_layers):
    """:200-212 (device tensors in, device tensor out)"""
    if n_layers == 1:
      return torch.zeros(0, dtype=torch.int64, device=self.device)
    h = self.histories[0]
    if getattr(self, "_bitmap", None) is None or self._bitmap.device != h.flag.device:
      self._bitmap = torch.zeros((h.flag.numel() + 31) // 32, dtype=torch.int32, device=h.flag.device)
      self._fu_slot = torch.zeros(h.flag.numel(), dtype=torch.int64, device=h.flag.device)
      self._fu_epoch = 1
    self._fu_epoch += 1
    # one device pass (orca_find_uncached): sorted unique, minus the batch, minus the cached ids, minus the first
    # id (:212 drops it, assuming it is the dummy node 0)
    return ops.find_uncached(self._dev(neighbors, torch.int32).flatten(), self._dev(source, torch.int64), h.flag,
                             self._fu_slot, self._fu_epoch, self._bitmap)

  def compute_embedding_budget_reuse(self, memory, source_nodes, timestamps, n_layers, n_neighbors, batch_id,
                                     cache_plan, memory_updater=None, train=True):
    nodes = self._dev(source_nodes, torch.int64)
    ts = self._dev(timestamps, torch.float64)
    if n_layers == 0:
      return self.compute_embedding(memory, nodes, ts, 0, n_neighbors, memory_updater, train)
    K, n = n_neighbors, nodes.numel()
    nbr, eid, dt = self._sample(nodes, ts, K)
    if n_layers == 1:                                               # :224-246
      m = self._memory_tables(memory, nbr, nodes, memory_updater, train)
      return self._attend(1, n, K, nbr, eid, dt, m, nodes.to(torch.int32), m, nbr)
    unc = self.find_uncached_out_of_batch_neighbors(nodes, nbr, n_layers)   # :251
    if unc.numel():                                                 # :253-260
      c_nodes = torch.cat([nodes, unc])
      c_ts = torch.cat([ts, ts.max().expand(unc.numel())])
    else:
      c_nodes, c_ts = nodes, ts
    comb = self.compute_embedding_budget_reuse(memory, c_nodes, c_ts, n_layers - 1, K, batch_id, cache_plan,
                                               memory_updater, train)       # :273
    h = self.histories[n_layers - 2]                                # :186-197
    h.update_times[c_nodes.cpu()] = batch_id
    h.push(comb, c_nodes)
    h.update_flag(cache_plan, c_nodes)
    return self._attend(n_layers, n, K, nbr, eid, dt, dict(table=comb), None, dict(table=h.emb), nbr)   # :284


def get_embedding_module(module_type, node_features, edge_features, memory, neighbor_finder, time_encoder, n_layers,
                         n_node_features, n_edge_features, n_time_features, embedding_dimension, device, n_heads=2,
                         dropout=0.1, n_neighbors=None, use_memory=True, reuse=False, history_budget=0, args=None,
                         num_nodes=-1):
  """:364-427.  Only "graph_attention" (the default, train.py:35) is on the hot path; the other module
  types are rejected the way the reference rejects an unknown one."""
  if module_type == "graph_attention":
    return GraphAttentionEmbedding(node_features=node_features, edge_features=edge_features, memory=memory,
                                   neighbor_finder=neighbor_finder, time_encoder=time_encoder, n_layers=n_layers,
                                   n_node_features=n_node_features, n_edge_features=n_edge_features,
                                   n_time_features=n_time_features, embedding_dimension=embedding_dimension,
                                   device=device, n_heads=n_heads, dropout=dropout, use_memory=use_memory, reuse=reuse,
                                   history_budget=history_budget, args=args, num_nodes=num_nodes)
  raise ValueError("Embedding Module {} not supported".format(module_type))
