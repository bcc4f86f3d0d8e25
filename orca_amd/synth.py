"""Synthetic inputs of BASELINE.json's configs, generated on the device (there are no datasets in
the container and no network): a power-law temporal interaction graph in the reference's conventions
(utils/preprocess_data.py:55-57,78-79: ids start at 1, node 0 / edge 0 / feature row 0 are padding;
edge ids follow time order; timestamps float64) and a randomly initialised TGN state_dict with the
reference's parameter names and default initialisers."""
import math

import numpy as np
import torch

D = 100


def powerlaw_graph(n_nodes, n_edges, device, alpha=1.2, t_max=2.6e6, tie_frac=0.01, seed=2024):
  """src, dst, eidx int64 [E], ts float64 [E] (sorted, with ~tie_frac exact duplicates)."""
  g = torch.Generator(device=device)
  g.manual_seed(seed)
  w = torch.arange(1, n_nodes + 1, device=device, dtype=torch.float64).pow_(-alpha)
  cdf = torch.cumsum(w, 0)
  cdf /= cdf[-1].clone()
  perm = torch.randperm(n_nodes, generator=g, device=device) + 1
  ends = []
  for _ in range(2):
    u = torch.rand(n_edges, generator=g, device=device, dtype=torch.float64)
    ends.append(perm[torch.searchsorted(cdf, u).clamp_(max=n_nodes - 1)])
    del u
  ts = torch.rand(n_edges, generator=g, device=device, dtype=torch.float64).mul_(t_max)
  n_tie = int(tie_frac * n_edges)
  if n_tie:
    pos = torch.randint(1, n_edges, (n_tie,), generator=g, device=device)
    ts[pos] = ts[pos - 1]
  ts = torch.sort(ts).values
  eidx = torch.arange(1, n_edges + 1, device=device, dtype=torch.int64)
  return ends[0].contiguous(), ends[1].contiguous(), eidx, ts


def edge_features(n_edges, d_e, device, seed=11, zero=False):
  """[n_edges+1, d_e] fp32 N(0,1), row 0 zeros; zero=True gives the d_e=1 all-zero table of
  train.py:141 (datasets without edge features)."""
  if zero:
    return torch.zeros(n_edges + 1, 1, dtype=torch.float32, device=device)
  g = torch.Generator(device=device)
  g.manual_seed(seed)
  ef = torch.empty(n_edges + 1, d_e, dtype=torch.float32, device=device)
  step = 1 << 22
  for s in range(0, n_edges + 1, step):
    ef[s:s + step].normal_(generator=g)
  ef[0].zero_()
  return ef


def tgn_state_dict(d_e, n_layers, seed=0):
  """Random-init parameters of TGN(graph_attention, gru, identity, last) with the reference's key
  names, built from the same torch.nn modules the reference instantiates (so default inits match):
  model/tgn_model.py:54-92, model/temporal_attention.py:17-22, utils/util.py:12-24,
  modules/memory_updater.py:86-91, model/time_encoding.py:14-21."""
  torch.manual_seed(seed)
  sd = {}
  sd["time_encoder.w.weight"] = torch.from_numpy((1 / 10 ** np.linspace(0, 9, D, dtype=np.float32))).reshape(D, 1)
  sd["time_encoder.w.bias"] = torch.zeros(D)
  msg = 3 * D + d_e
  gru = torch.nn.GRUCell(msg, D)
  for k, v in gru.state_dict().items():
    sd["memory_updater.memory_updater." + k] = v.detach()
  sd["memory_updater.layer_norm.weight"] = torch.ones(D)
  sd["memory_updater.layer_norm.bias"] = torch.zeros(D)
  for i in range(n_layers):
    p = f"embedding_module.attention_models.{i}."
    mha = torch.nn.MultiheadAttention(embed_dim=2 * D, kdim=2 * D + d_e, vdim=2 * D + d_e, num_heads=2)
    for k, v in mha.state_dict().items():
      sd[p + "multi_head_target." + k] = v.detach()
    fc1, fc2 = torch.nn.Linear(3 * D, D), torch.nn.Linear(D, D)
    torch.nn.init.xavier_normal_(fc1.weight)
    torch.nn.init.xavier_normal_(fc2.weight)
    sd[p + "merger.fc1.weight"], sd[p + "merger.fc1.bias"] = fc1.weight.detach(), fc1.bias.detach()
    sd[p + "merger.fc2.weight"], sd[p + "merger.fc2.bias"] = fc2.weight.detach(), fc2.bias.detach()
    sd[p + "norm.weight"], sd[p + "norm.bias"] = torch.ones(D), torch.zeros(D)
  a1, a2 = torch.nn.Linear(2 * D, D), torch.nn.Linear(D, 1)
  torch.nn.init.xavier_normal_(a1.weight)
  torch.nn.init.xavier_normal_(a2.weight)
  sd["affinity_score.fc1.weight"], sd["affinity_score.fc1.bias"] = a1.weight.detach(), a1.bias.detach()
  sd["affinity_score.fc2.weight"], sd["affinity_score.fc2.bias"] = a2.weight.detach(), a2.bias.detach()
  return sd


def tgat_flops_per_query(d_e, K):
  """Multiply-add flops of one (query, layer) attention evaluation in the folded formulation the
  kernel computes: q-proj, Wk^T q and Wv x-bar per head, K scores and the K-row
  weighted sum, out-proj, merger fc1/fc2."""
  kd = 2 * D + d_e
  return 2 * (2 * D * 2 * D + 2 * D * kd + 2 * K * kd + 2 * K * kd + 2 * D * kd + 2 * D * 2 * D + 3 * D * D + D * D)


def tgat_flops_executed_per_query(d_e, K):
  """Multiply-add flops the kernel executes in gather mode since (a) the q projection is multiplied into the key
  map ([2 x (k_dim+1) x 100] over the source row) and (b) the linear tail (value map, out-proj, fc1) into one
  [100 x (2*k_dim + 1 + 100)] map, both at pack time: fewer than the yardstick above, which stays the figure
  roofline.achieved is priced with."""
  kd = 2 * D + d_e
  return 2 * (2 * D * (kd + 1) + 2 * K * kd + 2 * K * kd + D * (2 * kd + 1 + D) + D * D)


def tgat_bytes_per_query(d_e, K):
  """Algorithmic HBM bytes of one (query, layer) evaluation: K neighbour + edge rows, src row, out row."""
  return 4 * K * (d_e + D) + 4 * D + 4 * D
