"""Loader for the reference's on-disk graph format: `ml_{name}.csv` with the
columns u, i, ts, label, idx and the `ml_{name}.npy` / `ml_{name}_node.npy` feature matrices, and the
chronological 70 / 15 / 15 split with held-out "new" nodes of utils/data_processing.py:31-120.  Same names,
arguments and return order as the reference; `data_dir` is the one addition (the reference hard-codes
'../data').  Host code: the arrays it returns feed `get_neighbor_finder` (the device CSR builder)."""
import os
import random

import numpy as np


class Data:
  """utils/data_processing.py:6-29"""

  def __init__(self, sources, destinations, timestamps, edge_idxs, labels):
    self.sources = sources
    self.destinations = destinations
    self.timestamps = timestamps
    self.edge_idxs = edge_idxs
    self.labels = labels
    self.n_interactions = len(sources)
    self.unique_nodes = set(sources) | set(destinations)
    self.n_unique_nodes = len(self.unique_nodes)
    self.tbatch = None
    self.n_batch = 0

  def _take(self, rows):
    return Data(self.sources[rows], self.destinations[rows], self.timestamps[rows], self.edge_idxs[rows],
                self.labels[rows])

  def sample(self, ratio):
    """a time-ordered random subset of the interactions (:19-29)"""
    n = int(ratio * self.n_interactions)
    rows = np.sort(random.sample(range(self.n_interactions), n))
    return self._take(rows)


def load_feat(d, data_dir="../data"):
  """(node_feats, edge_feats), None where the file is absent (:31-40)"""
  def maybe(path):
    return np.load(path) if os.path.exists(path) else None
  base = os.path.join(data_dir, d)
  return maybe(os.path.join(base, f"ml_{d}_node.npy")), maybe(os.path.join(base, f"ml_{d}.npy"))


def get_data(dataset_name, data_dir="../data", verbose=True):
  """full, train, val, test, new-node val, new-node test, n_total_unique_nodes, n_edges (:44-120).
  Split points are the 0.70 / 0.85 quantiles of the timestamps; 10 % of all nodes, drawn with
  random.seed(2020) from the nodes seen after the validation time, are removed from training so that
  the new-node splits are inductive."""
  import pandas as pd
  df = pd.read_csv(os.path.join(data_dir, dataset_name, f"ml_{dataset_name}.csv"))
  src, dst = df.u.values, df.i.values
  ts, eidx, labels = df.ts.values, df.idx.values, df.label.values
  val_time, test_time = (float(x) for x in np.quantile(ts, [0.70, 0.85]))
  full = Data(src, dst, ts, eidx, labels)
  n_nodes_total = len(set(src) | set(dst))

  random.seed(2020)                       # :59 "ensure we get the same graph"
  late = ts > val_time
  late_nodes = set(src[late]).union(set(dst[late]))
  # random.sample on a set draws from tuple(set) (CPython <= 3.10 converts it itself): keep that order
  held_out = set(random.sample(tuple(late_nodes), int(0.1 * n_nodes_total)))
  held = np.fromiter(held_out, dtype=np.int64, count=len(held_out))
  touches_held = np.isin(src, held) | np.isin(dst, held)
  train = full._take((ts <= val_time) & ~touches_held)
  assert not (set(train.sources) | set(train.destinations)) & held_out

  unseen = (set(src) | set(dst)) - (set(train.sources) | set(train.destinations))   # the val set may contain them
  unseen_a = np.fromiter(unseen, dtype=np.int64, count=len(unseen))
  has_unseen = np.isin(src, unseen_a) | np.isin(dst, unseen_a)
  in_val = (ts > val_time) & (ts <= test_time)
  in_test = ts > test_time
  val, test = full._take(in_val), full._take(in_test)
  nn_val, nn_test = full._take(in_val & has_unseen), full._take(in_test & has_unseen)
  if verbose:
    for name, d in (("dataset", full), ("training dataset", train), ("validation dataset", val),
                    ("test dataset", test), ("new node validation dataset", nn_val),
                    ("new node test dataset", nn_test)):
      print(f"The {name} has {d.n_interactions} interactions, involving {d.n_unique_nodes} different nodes")
    print(f"{len(held_out)} nodes were used for the inductive testing, i.e. are never seen during training")
  return full, train, val, test, nn_val, nn_test, n_nodes_total, len(src)
