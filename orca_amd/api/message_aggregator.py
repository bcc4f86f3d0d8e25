"""modules/message_aggregator.py:6-90 -- last / mean message aggregation as a segmented reduce keyed by
node id (the reference loops over a dict of per-node message lists)."""
import numpy as np
import torch

from .. import ops


class MessageAggregator(torch.nn.Module):
  def __init__(self, device):
    super().__init__()
    self.device = device

  def group_by_id(self, node_ids, messages, timestamps):
    """message_aggregator.py:25-31"""
    from collections import defaultdict
    d = defaultdict(list)
    for i, node_id in enumerate(node_ids):
      d[node_id].append((messages[i], timestamps[i]))
    return d

  def _flatten(self, node_ids, messages):
    """dict id -> [(msg, ts), ...]  ->  flat (ids, msgs, ts) in per-node arrival order, on the device"""
    uniq = np.unique(node_ids)
    ids, ms, ts = [], [], []
    for n in uniq:
      for m, t in messages[n]:
        ids.append(int(n))
        ms.append(m)
        ts.append(t)
    if not ids:
      return None
    dev = torch.device(self.device)
    return (torch.tensor(ids, dtype=torch.int64, device=dev), torch.stack(ms).float().to(dev).contiguous(),
            torch.stack([torch.as_tensor(t) for t in ts]).float().to(dev).contiguous())

  def _reduce(self, node_ids, messages, mean):
    flat = self._flatten(node_ids, messages)
    if flat is None:
      return [], [], []
    ids, ms, ts = flat
    slot = torch.zeros(int(ids.max().item()) + 1, dtype=torch.int64, device=ids.device)
    if mean:
      win, vals = ops.segment_mean(ids, ms, ts, slot, 1)
    else:
      win, vals = ops.segment_last(ids, slot, 1), ms
    pos = torch.nonzero(win).reshape(-1)
    order = torch.argsort(ids[pos])                # ascending node id, as np.unique yields them
    pos = pos[order]
    return ids[pos].cpu().tolist(), vals[pos], ts[pos]


class LastMessageAggregator(MessageAggregator):
  def aggregate(self, node_ids, messages):
    """message_aggregator.py:38-54"""
    return self._reduce(node_ids, messages, mean=False)


class MeanMessageAggregator(MessageAggregator):
  def aggregate(self, node_ids, messages):
    """message_aggregator.py:61-81"""
    return self._reduce(node_ids, messages, mean=True)


def get_message_aggregator(aggregator_type, device):
  if aggregator_type == "last":
    return LastMessageAggregator(device=device)
  elif aggregator_type == "mean":
    return MeanMessageAggregator(device=device)
  else:
    raise ValueError("Message aggregator {} not implemented".format(aggregator_type))
