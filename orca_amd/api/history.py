"""modules/history.py:4-46 -- History, the per-layer embedding-reuse cache."""
import numpy as np
import torch

from .. import ops


class History(torch.nn.Module):
  def __init__(self, num_embeddings, embedding_dim, device=None, history_budget=0):
    super().__init__()
    self.num_embeddings = num_embeddings
    self.embedding_dim = embedding_dim
    self._device = device
    self.emb = torch.zeros(num_embeddings, embedding_dim, device=device)
    self.update_times = torch.zeros(num_embeddings)
    # the reference keeps a host float64 bitmap (history.py:16); here it lives in HBM as bytes
    self.flag = torch.zeros(num_embeddings, dtype=torch.uint8, device=device)
    self._slot = torch.zeros(num_embeddings, dtype=torch.int64, device=device)
    self._epoch = 1
    self.budget = history_budget if history_budget != 0 else num_embeddings

  @property
  def cache_flag(self):
    """host float64 view, as the reference exposes it"""
    return self.flag.cpu().numpy().astype(np.float64)

  def reset_parameters(self):
    self.emb.fill_(0)

  def detach_history(self):
    self.emb.detach_()

  def _ids(self, inds):
    if not isinstance(inds, torch.Tensor):
      inds = torch.from_numpy(np.ascontiguousarray(inds))
    return inds.to(device=self.emb.device, dtype=torch.int64).contiguous()

  def update_flag(self, cache_plan, source_node):
    """history.py:25-30"""
    if cache_plan is not None:
      ops.fill_u8(self.flag, 0)
      plan = self._ids(cache_plan)
      if plan.numel():
        ops.set_flags(self.flag, ids64=plan, v=1)
    else:
      ops.set_flags(self.flag, ids64=self._ids(source_node), v=1)

  @torch.no_grad()
  def pull(self, inds):
    """history.py:32-36"""
    from .. import torch_ops  # noqa: F401  (registers the orca:: ops)
    return torch.ops.orca.history_pull(self.emb, self._ids(inds))

  @torch.no_grad()
  def push(self, x, inds):
    """history.py:38-40; duplicate ids: the LAST occurrence wins."""
    ids = self._ids(inds)
    if self._slot.device != self.emb.device or self._slot.numel() != self.emb.shape[0]:   # emb was rebound
      self._slot = torch.zeros(self.emb.shape[0], dtype=torch.int64, device=self.emb.device)
      self._epoch = 1
    from .. import torch_ops  # noqa: F401
    torch.ops.orca.history_push(self.emb, x.detach().float().contiguous(), ids, self._slot, self._epoch)
    self._epoch += ids.numel() + 1

  def push_and_pull(self, x, push_inds, pull_inds):
    """history.py:42-46, the --gradient variant (train.py:43): the pulled rows that were pushed in this call stay
    connected to x in the autograd graph (orca::history_pull_pushed); the other rows are data.  Upstream rebinds the
    table to a copy that carries the graph until detach_history() (train.py:263); here the table itself never carries
    one -- the graph hangs on the returned rows -- so there is nothing to copy or detach."""
    ids = self._ids(push_inds)
    pull = self._ids(pull_inds).reshape(-1)
    epoch = self._epoch
    self.push(x, ids)
    if not (torch.is_grad_enabled() and x.requires_grad):
      return self.pull(pull)
    from .. import torch_ops  # noqa: F401
    with torch.no_grad():
      n = self.emb.shape[0]
      if getattr(self, "_pushed", None) is None or self._pushed.numel() != n or self._pushed.device != self.emb.device:
        self._pushed = torch.zeros(n, dtype=torch.uint8, device=self.emb.device)
        self._wpos = torch.zeros(n, dtype=torch.int32, device=self.emb.device)
      ops.fill_u8(self._pushed, 0)
      ops.set_flags(self._pushed, ids64=ids, v=1)
      # the push stamped slot[id] = epoch + (position of the last occurrence of id)
      self._wpos[ids] = (self._slot[ids] - epoch).to(torch.int32)
    return torch.ops.orca.history_pull_pushed(self.emb, x, pull, self._pushed.clone(), self._wpos.clone())
