"""utils/util.py:91-151 -- get_neighbor_finder / NeighborFinder on the time-sorted CSR in HBM."""
import numpy as np
import torch

from .. import ops


def _device(device):
  return torch.device(device if device is not None else "cuda:0")


class NeighborFinder:
  """NeighborFinder(node_to_neighbors, node_to_edge_idxs, node_to_edge_timestamps) (utils/util.py:123-126):
  three ragged lists (one array per node, each sorted by time)."""

  def __init__(self, node_to_neighbors, node_to_edge_idxs, node_to_edge_timestamps, device=None, _csr=None):
    self.device = _device(device)
    if _csr is not None:
      self.csr = _csr
      return
    lens = np.fromiter((len(x) for x in node_to_neighbors), dtype=np.int64, count=len(node_to_neighbors))
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    cat = lambda xs, dt: (np.concatenate([np.asarray(x, dtype=dt) for x in xs]) if indptr[-1] else np.zeros(0, dt))
    self.csr = ops.CSR.from_arrays_host(indptr, cat(node_to_neighbors, np.int32), cat(node_to_edge_idxs, np.int32),
                                        cat(node_to_edge_timestamps, np.float64), self.device)

  # the reference's attributes, materialised on demand from the CSR
  def _lists(self):
    indptr, nbr, eid, ts = self.csr.arrays_host()
    sp = indptr[1:-1]
    return np.split(nbr, sp), np.split(eid, sp), np.split(ts, sp)

  @property
  def node_to_neighbors(self):
    return self._lists()[0]

  @property
  def node_to_edge_idxs(self):
    return self._lists()[1]

  @property
  def node_to_edge_timestamps(self):
    return self._lists()[2]

  def find_before(self, src_idx, cut_time):
    """utils/util.py:129-131: all interactions of src_idx strictly before cut_time."""
    src_idx = int(src_idx)
    if not 0 <= src_idx < self.csr.n_rows:
      raise IndexError(f"list index out of range: node {src_idx}")
    n = int(ops.find_before(self.csr, torch.tensor([src_idx], dtype=torch.int64, device=self.device),
                            torch.tensor([cut_time], dtype=torch.float64, device=self.device)).item())
    nbr, eid, ts = self._lists()
    return nbr[src_idx][:n], eid[src_idx][:n], ts[src_idx][:n]

  def get_temporal_neighbor_device(self, source_nodes, timestamps, n_neighbors):
    """Device-tensor form: (nbr int32, eidx int32, edge_times f32, Δt f32), all [Q, K] in HBM."""
    nodes = torch.as_tensor(np.asarray(source_nodes) if not isinstance(source_nodes, torch.Tensor) else source_nodes)
    ts = torch.as_tensor(np.asarray(timestamps) if not isinstance(timestamps, torch.Tensor) else timestamps)
    nodes = nodes.to(device=self.device, dtype=torch.int64)
    ts = ts.to(device=self.device, dtype=torch.float64)
    if nodes.numel() and (int(nodes.min()) < 0 or int(nodes.max()) >= self.csr.n_rows):
      raise IndexError("list index out of range: node id outside the neighbor finder")
    from .. import torch_ops  # noqa: F401  (registers the orca:: ops)
    return torch.ops.orca.sample_recent(self.csr._h.value, nodes, ts, int(n_neighbors))

  def get_temporal_neighbor(self, source_nodes, timestamps, n_neighbors):
    """utils/util.py:134-151: NumPy in, NumPy (int32, int32, float32) [Q, K] out, right-aligned, zero padded."""
    Q = len(source_nodes)
    if n_neighbors <= 0:                    # :136 quirk: width-1 all-zero arrays
      return (np.zeros((Q, 1), np.int32), np.zeros((Q, 1), np.int32), np.zeros((Q, 1), np.float32))
    nbr, eid, et, _ = self.get_temporal_neighbor_device(source_nodes, timestamps, n_neighbors)
    return nbr.cpu().numpy(), eid.cpu().numpy(), et.cpu().numpy()


def get_neighbor_finder(data, max_node_idx=None, device=None):
  """utils/util.py:91-105: undirected adjacency of `data` (.sources .destinations .edge_idxs .timestamps),
  each node's list stably sorted by time."""
  max_node_idx = max(data.sources.max(), data.destinations.max()) if max_node_idx is None else max_node_idx
  dev = _device(device)
  csr = ops.CSR.from_edges_host(data.sources, data.destinations, data.edge_idxs, data.timestamps,
                                int(max_node_idx) + 1, dev)
  return NeighborFinder(None, None, None, device=dev, _csr=csr)
