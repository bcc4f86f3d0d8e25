"""model/tgn_model.py:13-195 -- the caller of the hot path, kept thin: it wires the mirrored modules in
the reference's order (train: embed -> persist memory -> store raw messages; eval: embed -> store ->
persist) so that a reference state_dict loads by key and the parity tests read like the reference's
own flow.  It trains: loss.backward() on compute_edge_probabilities' outputs reaches the attention layers, the GRU
and the link decoder exactly where the reference's autograd graph does (the state writes -- memory persist, raw
message store, history push -- are detached, as upstream)."""
import numpy as np
import torch

from .. import ops
from .embedding_module import get_embedding_module
from .memory import Memory
from .memory_updater import get_memory_updater
from .message_aggregator import get_message_aggregator
from .temporal_attention import MergeLayer
from .time_encoding import TimeEncode


class TGN(torch.nn.Module):
  def __init__(self, neighbor_finder, node_features, edge_features, device, n_layers=2, n_heads=2, dropout=0.1,
               use_memory=True, node_dimension=100, time_dimension=100, memory_dimension=100,
               embedding_module_type="graph_attention", message_function="identity", aggregator_type="last",
               memory_updater_type="gru", n_neighbors=None, args=None):
    super().__init__()
    if not use_memory or message_function != "identity":
      raise ValueError("TGN mirror: only use_memory=True with the identity message function is on the hot path")
    self.args = args
    self.n_layers = n_layers
    self.neighbor_finder = neighbor_finder
    self.device = device
    ef = edge_features if isinstance(edge_features, torch.Tensor) else torch.from_numpy(np.asarray(edge_features))
    self.edge_raw_features = ef.float().to(device).contiguous()           # tgn_model.py:25
    self.n_node_features = node_dimension
    self.n_nodes = args.n_nodes
    self.n_edge_features = self.edge_raw_features.shape[1]
    self.embedding_dimension = self.n_node_features
    self.n_neighbors = n_neighbors
    self.use_memory = use_memory
    self.time_encoder = TimeEncode(dimension=time_dimension)
    self.memory_dimension = memory_dimension
    raw_message_dimension = 2 * memory_dimension + self.n_edge_features + time_dimension   # :49-50
    self.memory = Memory(n_nodes=self.n_nodes, memory_dimension=memory_dimension, input_dimension=raw_message_dimension,
                         message_dimension=raw_message_dimension, device=device)
    self.message_aggregator = get_message_aggregator(aggregator_type=aggregator_type, device=device)   # :61 (unused upstream too)
    self.memory_updater = get_memory_updater(module_type=memory_updater_type, message_dimension=raw_message_dimension,
                                             memory_dimension=memory_dimension, device=device)
    self.embedding_module = get_embedding_module(
        module_type=embedding_module_type, node_features=None, edge_features=self.edge_raw_features, memory=self.memory,
        neighbor_finder=neighbor_finder, time_encoder=self.time_encoder, n_layers=n_layers,
        n_node_features=self.n_node_features, n_edge_features=self.n_edge_features, n_time_features=time_dimension,
        embedding_dimension=self.embedding_dimension, device=device, n_heads=n_heads, dropout=dropout,
        use_memory=use_memory, n_neighbors=n_neighbors, reuse=args.reuse, history_budget=args.budget, args=args,
        num_nodes=self.n_nodes)
    self.affinity_score = MergeLayer(self.n_node_features, self.n_node_features, self.n_node_features, 1)   # :92
    self.batch_counter = 0
    self.test_mode = False
    self._slot = torch.zeros(self.n_nodes, dtype=torch.int64, device=device)
    self._epoch = 1
    self.to(device)

  def compute_temporal_embeddings(self, source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                  n_neighbors, reuse, train, cache_plan):
    """tgn_model.py:95-147"""
    self.batch_counter += 1
    n_samples = len(source_nodes)
    nodes = np.concatenate([source_nodes, destination_nodes, negative_nodes])
    positives = np.concatenate([source_nodes, destination_nodes])
    timestamps = np.concatenate([edge_times, edge_times, edge_times])
    if train:
      memory = self.memory
      self.test_mode = False
    else:
      if self.test_mode is False:
        self.memory_updater.update_memory_in_test(self.memory)
        self.test_mode = True
      memory = self.memory.memory
    em = self.embedding_module
    if reuse:
      if self.args.budget != 0:
        emb = em.compute_embedding_budget_reuse(memory=memory, source_nodes=nodes, timestamps=timestamps,
                                                n_layers=self.n_layers, n_neighbors=n_neighbors,
                                                batch_id=self.batch_counter, cache_plan=cache_plan,
                                                memory_updater=self.memory_updater, train=train)
      else:
        emb = em.compute_embedding_reuse(memory=memory, source_nodes=nodes, timestamps=timestamps,
                                         n_layers=self.n_layers, n_neighbors=n_neighbors, batch_id=self.batch_counter,
                                         memory_updater=self.memory_updater, train=train)
    else:
      emb = em.compute_embedding(memory=memory, source_nodes=nodes, timestamps=timestamps, n_layers=self.n_layers,
                                 n_neighbors=n_neighbors, memory_updater=self.memory_updater, train=train)
    with torch.no_grad():
      if train:                                                            # :127-129
        self.memory_updater.update_memory(self.memory, positives)
        self.memory.clear_messages(positives)
      self.get_raw_messages_and_store(source_nodes, destination_nodes, edge_times, edge_idxs)   # :131-141
      if not train:                                                        # :143-145
        self.memory_updater.update_memory(self.memory, positives)
        self.memory.clear_messages(positives)
    return emb[:n_samples], emb[n_samples:2 * n_samples], emb[2 * n_samples:]

  def get_raw_messages_and_store(self, source_nodes, destination_nodes, edge_times, edge_idxs):
    """tgn_model.py:172-191 + memory.py:25-28 fused: last-occurrence dedupe of [src;dst], gather, time
    encoding of (t - last_update), concat and scatter into the pending-message slots."""
    dev = self.device
    f = lambda x, dt: torch.from_numpy(np.ascontiguousarray(x)).to(device=dev, dtype=dt)
    m, te = self.memory, self.time_encoder.w
    from .. import torch_ops  # noqa: F401  (registers the orca:: ops)
    torch.ops.orca.message_store(f(source_nodes, torch.int64), f(destination_nodes, torch.int64), f(edge_idxs, torch.int64),
                                 f(edge_times, torch.float64), m.memory, m.last_update, self.edge_raw_features,
                                 te.weight.detach().reshape(-1).contiguous(), te.bias.detach().contiguous(), self._slot,
                                 self._epoch, m.messages, m.timestamps, m.pending)
    self._epoch += 2 * len(source_nodes) + 1

  def compute_edge_probabilities(self, source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                 n_neighbors, reuse, train, cache_plan):
    """tgn_model.py:150-157"""
    n_samples = len(source_nodes)
    s, d, n = self.compute_temporal_embeddings(source_nodes, destination_nodes, negative_nodes, edge_times, edge_idxs,
                                               n_neighbors, reuse, train, cache_plan)
    score = self.affinity_score(torch.cat([s, s], dim=0), torch.cat([d, n])).squeeze(dim=0)
    return score[:n_samples].sigmoid(), score[n_samples:].sigmoid()

  def set_neighbor_finder(self, neighbor_finder):
    """tgn_model.py:193-195"""
    self.neighbor_finder = neighbor_finder
    self.embedding_module.neighbor_finder = neighbor_finder
