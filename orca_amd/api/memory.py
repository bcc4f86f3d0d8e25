"""modules/memory.py:5-50 -- Memory: state tables + pending raw-message slots, all in HBM."""
import numpy as np
import torch
from torch import nn

from .. import ops


class Memory(nn.Module):
  def __init__(self, n_nodes, memory_dimension, input_dimension, message_dimension=None, device="cpu",
               combination_method="sum"):
    super().__init__()
    self.n_nodes = n_nodes
    self.memory_dimension = memory_dimension
    self.input_dimension = input_dimension
    self.message_dimension = message_dimension
    self.device = device
    self.combination_method = combination_method
    self.__init_memory__()

  def __init_memory__(self):
    dev = self.device
    self.memory = torch.zeros((self.n_nodes, self.memory_dimension), device=dev)
    self.last_update = torch.zeros(self.n_nodes, device=dev)
    self.pending = torch.zeros(self.n_nodes, dtype=torch.uint8, device=dev)   # upstream: host bool array `nodes`
    self.messages = torch.zeros((self.n_nodes, self.message_dimension), device=dev)
    self.timestamps = torch.zeros(self.n_nodes, device=dev)

  @property
  def nodes(self):
    """host bool view of the pending flags, as the reference exposes it (memory.py:21)"""
    return self.pending.cpu().numpy().astype(bool)

  def _ids(self, x):
    """node ids -> int64 device tensor.  Host ids (the NumPy arrays the reference's interface passes) are range-checked
    here and raise IndexError like the reference's NumPy indexing; ids that are already on the device are not read
    back: the kernels skip the ones outside the table (include/orca_hip.h)."""
    if not isinstance(x, torch.Tensor):
      x = np.ascontiguousarray(x)
      if x.size and (int(x.min()) < -self.n_nodes or int(x.max()) >= self.n_nodes):
        raise IndexError(f"node id out of range for a memory of {self.n_nodes} nodes")
      x = torch.from_numpy(np.where(x < 0, x + self.n_nodes, x)) if x.size and int(x.min()) < 0 else torch.from_numpy(x)
    return x.to(device=self.memory.device, dtype=torch.int64).contiguous()

  def store_raw_messages(self, nodes, messages, timestamps):
    """memory.py:25-28 (ids unique)"""
    ids = self._ids(nodes)
    ops.set_flags(self.pending, ids64=ids, v=1)
    ops.scatter_rows(self.messages, ids, messages.float().contiguous())
    self.timestamps[ids] = timestamps.to(self.timestamps.device).float()

  def get_memory(self, node_idxs):
    return ops.gather_rows(self.memory, self._ids(node_idxs))

  def set_memory(self, node_idxs, values):
    ops.scatter_rows(self.memory, self._ids(node_idxs), values.float().contiguous())

  def get_last_update(self, node_idxs):
    return self.last_update[self._ids(node_idxs)]

  def backup_memory(self):
    return (self.memory.clone(), self.last_update.clone(), self.messages.clone(), self.pending.clone(),
            self.timestamps.clone())

  def restore_memory(self, memory_backup):
    self.memory, self.last_update, self.messages, self.pending, self.timestamps = (x.clone() for x in memory_backup)

  def detach_memory(self):
    self.memory.detach_()
    self.messages.detach_()

  def clear_messages(self, positives):
    """memory.py:49-50"""
    ops.set_flags(self.pending, ids64=self._ids(positives), v=0)
