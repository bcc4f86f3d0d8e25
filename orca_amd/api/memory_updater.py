"""modules/memory_updater.py:13-98 -- GRU memory updater on the MFMA GRU kernel."""
import numpy as np
import torch
from torch import nn

from .. import ops


class UpdatedMemoryView:
  """What get_updated_memory returns in place of the reference's full-table clone
  (memory_updater.py:72): the persistent table plus an overlay holding GRU(message, memory) for the
  rows that had a pending message.  Indexing gathers the effective rows."""

  def __init__(self, table, overlay, flags, upd_rows=None, pos=None):
    self.table, self.overlay, self.flags = table, overlay, flags
    # under autograd: the flagged rows again, compactly and attached to the graph (orca::gru_overlay), and
    # pos[id] = the row of id among them -- gradients reach the GRU parameters through these
    self.upd_rows, self.pos = upd_rows, pos
    self.shape = table.shape

  def __getitem__(self, idx):
    rows = idx[0] if isinstance(idx, tuple) else idx
    if not isinstance(rows, torch.Tensor):
      rows = torch.from_numpy(np.ascontiguousarray(rows))
    rows = rows.to(device=self.table.device, dtype=torch.int64).reshape(-1)
    base = ops.gather_rows(self.table, rows)
    if self.flags is None:
      return base
    sel = self.flags[rows].bool()
    if self.upd_rows is not None and self.upd_rows.requires_grad:     # differentiable gather of the GRU rows
      upd = self.upd_rows[torch.where(sel, self.pos[rows].long(), torch.zeros_like(rows))]
    else:
      upd = ops.gather_rows(self.overlay, rows)
    return torch.where(sel.unsqueeze(1), upd, base)

  def materialize(self):
    return torch.where(self.flags.bool().unsqueeze(1), self.overlay, self.table) if self.flags is not None else self.table


class GRUMemoryUpdater(nn.Module):
  def __init__(self, message_dimension, memory_dimension, device):
    super().__init__()
    if memory_dimension != ops.D:
      raise ValueError(f"GRUMemoryUpdater: the HIP path is built for memory dimension {ops.D}")
    self.layer_norm = nn.LayerNorm(memory_dimension)       # present (unused) upstream: memory_updater.py:20
    self.message_dimension = message_dimension
    self.device = device
    self.memory_updater = nn.GRUCell(input_size=message_dimension, hidden_size=memory_dimension)
    self._blob = None
    self._blob_key = None
    self._scratch = None

  def packed(self):
    ps = list(self.memory_updater.parameters())
    key = tuple((p.data_ptr(), p._version) for p in ps)
    if self._blob is None or key != self._blob_key:
      c = self.memory_updater
      self._blob = ops.gru_pack(c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh, ps[0].device)
      self._blob_key = key
    return self._blob

  def _work(self, memory):
    n = memory.memory.shape[0]
    dev = memory.memory.device
    s = self._scratch
    if s is None or s["slot"].numel() != n or s["slot"].device != dev:
      s = self._scratch = dict(slot=torch.zeros(n, dtype=torch.int64, device=dev), epoch=1,
                               worklist=torch.zeros(n, dtype=torch.int32, device=dev),
                               count=torch.zeros(1, dtype=torch.int32, device=dev),
                               overlay=torch.zeros(n, ops.D, device=dev),
                               pos=torch.zeros(n, dtype=torch.int32, device=dev))
    return s

  def _worklist(self, memory, ids64):
    s = self._work(memory)
    ops.worklist_build(None, ids64, memory.pending, s["slot"], s["epoch"], s["worklist"], s["count"])
    s["epoch"] += ids64.numel() + 1
    return s

  @staticmethod
  def _ids(memory, x):
    if not isinstance(x, torch.Tensor):
      x = np.ascontiguousarray(x)
      if x.size and (int(x.min()) < 0 or int(x.max()) >= memory.n_nodes):
        raise IndexError(f"node id out of range for a memory of {memory.n_nodes} nodes")
      x = torch.from_numpy(x)
    return x.to(device=memory.memory.device, dtype=torch.int64).contiguous()

  @torch.no_grad()
  def update_memory(self, memory, positives):
    """memory_updater.py:25-39: persist GRU(message, memory) for the positives holding a message."""
    ids = self._ids(memory, positives)
    if ids.numel() == 0:
      return
    s = self._worklist(memory, ids)
    ops.gru_update(self.packed(), self.message_dimension, s["worklist"], min(ids.numel(), memory.n_nodes), s["count"],
                   memory.messages, memory.memory, memory.memory, memory.last_update, memory.timestamps)

  @torch.no_grad()
  def update_memory_in_test(self, memory):
    """memory_updater.py:42-53: flush every pending message, then clear the flags."""
    ids = torch.arange(memory.n_nodes, dtype=torch.int64, device=memory.memory.device)
    s = self._worklist(memory, ids)
    ops.gru_update(self.packed(), self.message_dimension, s["worklist"], memory.n_nodes, s["count"], memory.messages,
                   memory.memory, memory.memory, memory.last_update, memory.timestamps)
    ops.fill_u8(memory.pending, 0)

  def get_updated_memory(self, memory, index=None):
    """memory_updater.py:56-78 -> (updated memory, updated last_update).  The memory is returned as an
    UpdatedMemoryView (table + overlay), not a clone of the full table.  With autograd on (the training loop,
    train.py:225-226) the GRU rows stay in the graph: gradients of whatever reads the view reach the GRU
    parameters, as upstream; the memory table and the stored messages are data (memory.py:45-47)."""
    n = memory.n_nodes
    c = self.memory_updater
    grad = torch.is_grad_enabled() and any(p.requires_grad for p in c.parameters())
    with torch.no_grad():
      ids = (torch.arange(n, dtype=torch.int64, device=memory.memory.device) if index is None
             else self._ids(memory, index))
      s = self._worklist(memory, ids)
      bound = min(max(ids.numel(), 1), n)
      sel = torch.zeros(n, dtype=torch.uint8, device=memory.memory.device)
      if ids.numel():
        sel[ids] = memory.pending[ids]
      last = torch.where(sel.bool(), memory.timestamps, memory.last_update)
    upd = None
    if grad:
      # the rows stay in the autograd graph: GRUCell as the custom op orca::gru_cell on the gathered rows of the
      # work-list (its length is read back: the one host sync of this path), then copied into the overlay table
      from .. import torch_ops  # noqa: F401  (registers the orca:: ops)
      wl = s["worklist"][:int(s["count"].item())].long()
      upd = torch.ops.orca.gru_cell(c.weight_ih, c.weight_hh, c.bias_ih, c.bias_hh, self.packed(),
                                    memory.messages.detach()[wl], memory.memory.detach()[wl])
      with torch.no_grad():
        s["overlay"][wl] = upd.detach()
        s["pos"][wl] = torch.arange(wl.numel(), dtype=torch.int32, device=wl.device)
    else:
      with torch.no_grad():
        ops.gru_update(self.packed(), self.message_dimension, s["worklist"], bound, s["count"], memory.messages,
                       memory.memory, s["overlay"])
    return UpdatedMemoryView(memory.memory, s["overlay"], sel, upd, s["pos"] if grad else None), last


def get_memory_updater(module_type, message_dimension, memory_dimension, device):
  """memory_updater.py:94-98; the RNN variant is non-default upstream and not built here -- like an
  unknown type upstream it yields None."""
  if module_type == "gru":
    return GRUMemoryUpdater(message_dimension, memory_dimension, device)
  return None
