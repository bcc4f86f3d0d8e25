"""Host-side callers either side of the path against vectors captured from
the reference (tests/golden/make_golden.py harness): the ml_{name}.csv loader and split, the seeded
negative sampler, and the evaluation loop's metrics with a deterministic stand-in model.  CPU only."""
import os

import numpy as np
import torch


def _write_dataset(g, root):
  import pandas as pd
  d = os.path.join(root, "toy")
  os.makedirs(d)
  pd.DataFrame({"u": g["data_u"], "i": g["data_i"], "ts": g["data_ts"], "label": g["data_label"],
                "idx": g["data_idx"]}).to_csv(os.path.join(d, "ml_toy.csv"))
  np.save(os.path.join(d, "ml_toy.npy"), g["data_feats"])


def test_get_data_split_matches_reference(golden, tmp_path):
  from orca_amd.api.data import get_data, load_feat
  g = golden("harness.npz")
  _write_dataset(g, str(tmp_path))
  full, train, val, test, nn_val, nn_test, n_nodes, n_edges = get_data("toy", data_dir=str(tmp_path), verbose=False)
  assert [n_nodes, n_edges] == list(g["data_counts"])
  for name, d in dict(full=full, train=train, val=val, test=test, nn_val=nn_val, nn_test=nn_test).items():
    assert np.array_equal(d.edge_idxs, g[f"data_{name}_edge_idxs"]), name
    assert np.array_equal(d.sources, g[f"data_{name}_sources"]), name
    assert d.n_unique_nodes == int(g[f"data_{name}_n_unique_nodes"][0]), name
  node_f, edge_f = load_feat("toy", data_dir=str(tmp_path))
  assert node_f is None and np.array_equal(edge_f, g["data_feats"])
  # the time-ordered random subset keeps rows aligned
  sub = train.sample(0.25)
  assert sub.n_interactions == int(0.25 * train.n_interactions) and np.all(np.diff(sub.timestamps) >= 0)


def test_rand_edge_sampler_matches_reference(golden):
  from orca_amd.api.evaluation import RandEdgeSampler
  g = golden("harness.npz")
  s = RandEdgeSampler(g["data_u"], g["data_i"], seed=3)
  a, b = s.sample(50)
  assert np.array_equal(a, g["neg_src_50"]) and np.array_equal(b, g["neg_dst_50"])
  a, b = s.sample(7)
  assert np.array_equal(a, g["neg_src_7"]) and np.array_equal(b, g["neg_dst_7"])
  s.reset_random_state()
  a, b = s.sample(50)
  assert np.array_equal(a, g["neg_src_50"]) and np.array_equal(b, g["neg_dst_50"])
  u = RandEdgeSampler(g["data_u"], g["data_i"])          # unseeded: draws from the global stream
  assert u.seed is None and len(u.sample(5)[0]) == 5


def test_eval_edge_prediction_matches_reference(golden, tmp_path):
  from orca_amd.api.data import get_data
  from orca_amd.api.evaluation import RandEdgeSampler, eval_edge_prediction
  g = golden("harness.npz")
  _write_dataset(g, str(tmp_path))
  full, _, val, *_ = get_data("toy", data_dir=str(tmp_path), verbose=False)

  class Model:
    def eval(self):
      return self

    def compute_edge_probabilities(self, s, d, n, t, e, n_neighbors, reuse, train, cache):
      assert train is False and cache is None
      pos = 1.0 / (1.0 + np.exp(-np.sin(0.37 * s + 0.11 * d + 1e-6 * t)))
      neg = 1.0 / (1.0 + np.exp(-np.sin(0.37 * s + 0.11 * n + 1e-6 * t) + 0.4))
      return torch.from_numpy(pos.reshape(-1, 1)), torch.from_numpy(neg.reshape(-1, 1))

  got = eval_edge_prediction(Model(), RandEdgeSampler(full.sources, full.destinations, seed=0), val, 10, 200)
  assert np.allclose(np.array(got), g["eval_metrics"], rtol=0, atol=1e-12)
