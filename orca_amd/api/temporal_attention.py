"""model/temporal_attention.py:6-56 + MergeLayer utils/util.py:12-24 -- TemporalAttentionLayer."""
import torch
from torch import nn

from .. import ops


class MergeLayer(nn.Module):
  """utils/util.py:12-24 (parameter container; the fused kernel evaluates it)."""

  def __init__(self, dim1, dim2, dim3, dim4):
    super().__init__()
    self.fc1 = nn.Linear(dim1 + dim2, dim3)
    self.fc2 = nn.Linear(dim3, dim4)
    self.act = nn.ReLU()
    nn.init.xavier_normal_(self.fc1.weight)
    nn.init.xavier_normal_(self.fc2.weight)

  def forward(self, x1, x2):
    return self.fc2(self.act(self.fc1(torch.cat([x1, x2], dim=1))))


class TemporalAttentionLayer(nn.Module):
  """Same constructor, parameters (state_dict keys) and forward contract as the reference; forward runs the
  fused HIP kernel (the custom op orca::tgat_dense, orca_amd/torch_ops.py) and, under autograd, backward runs
  orca_tgat_backward (gradients of all twelve parameters and of the node / time feature inputs; edge features are
  data upstream and get none).  In train() mode the attention weights are dropped out with probability `dropout` like
  nn.MultiheadAttention's (counter RNG in the kernel: torch's own dropout stream is not reproduced)."""

  def __init__(self, n_node_features, n_neighbors_features, n_edge_features, time_dim, output_dimension, n_head=2,
               dropout=0.1):
    super().__init__()
    if (n_node_features, n_neighbors_features, time_dim, output_dimension, n_head) != (ops.D, ops.D, ops.D, ops.D, 2):
      raise ValueError("TemporalAttentionLayer: the HIP path is built for 100-d node/time features and 2 heads")
    self.n_head = n_head
    self.feat_dim = n_node_features
    self.time_dim = time_dim
    self.n_edge_features = n_edge_features
    self.query_dim = n_node_features + time_dim
    self.key_dim = n_neighbors_features + time_dim + n_edge_features
    self.merger = MergeLayer(self.query_dim, n_node_features, n_node_features, output_dimension)
    self.multi_head_target = nn.MultiheadAttention(embed_dim=self.query_dim, kdim=self.key_dim, vdim=self.key_dim,
                                                   num_heads=n_head, dropout=dropout)
    self.norm = nn.LayerNorm(output_dimension)
    self.dropout = float(dropout)
    self._blob = None
    self._blob_key = None

  def packed(self):
    """The kernel's parameter blob, re-packed whenever a parameter changed (in place or rebound)."""
    ps = list(self.parameters())
    key = tuple((p.data_ptr(), p._version) for p in ps)
    if self._blob is None or key != self._blob_key:
      dev = ps[0].device
      sd = {k: v for k, v in self.state_dict().items()}
      self._blob = ops.tgat_pack(sd, self.n_edge_features, dev, None)
      self._blob_key = key
    return self._blob

  def dropout_args(self):
    """(p, seed) of this call: attention-weight dropout is on in train() mode, as nn.MultiheadAttention's; the seed
    of the kernel's counter RNG is drawn from torch's CPU generator (torch.manual_seed makes a run repeatable)."""
    if not self.training or self.dropout <= 0.0:
      return 0.0, 0
    return self.dropout, int(torch.randint(0, 1 << 62, (1,)).item())

  def forward(self, src_node_features, src_time_features, neighbors_features, neighbors_time_features, edge_features,
              neighbors_padding_mask):
    """[B,100], [B,1,100], [B,K,100], [B,K,100], [B,K,d_e], bool [B,K] (mutated in place as
    temporal_attention.py:46-47 does) -> (out [B,100], weights [B,K])."""
    B, K = neighbors_padding_mask.shape
    mask = neighbors_padding_mask
    if mask.dtype != torch.bool or not mask.is_contiguous():
      raise TypeError("neighbors_padding_mask must be a contiguous torch.bool tensor")
    f = lambda x, w: x.float().contiguous().view(-1, w)
    src, st = f(src_node_features, self.feat_dim), f(src_time_features, self.time_dim)
    nbr, nt = f(neighbors_features, self.feat_dim), f(neighbors_time_features, self.time_dim)
    edge = f(edge_features, self.n_edge_features)
    named = dict(self.named_parameters())
    params = [named[k] for k in ops.TGAT_KEYS]
    from .. import torch_ops  # noqa: F401  (registers the orca:: ops)
    p, seed = self.dropout_args()
    out, wts, after = torch.ops.orca.tgat_dense(params, self.packed(), self.n_edge_features, src, st, nbr, nt, edge,
                                                mask.clone(), p, seed)     # (the op keeps the mask copy for its backward)
    mask.copy_(after.detach())          # temporal_attention.py:46-47 mutates the caller's mask
    return out, wts.detach()            # the weights are unused upstream (embedding_module.py:326): no gradient through them
