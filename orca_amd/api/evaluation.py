"""The step after the path: the seeded negative sampler of utils/util.py:68-87 and
the link-prediction evaluation loop of evaluation/evaluation.py:6-47, with the reference's names and
arguments.  Host code around `model.compute_edge_probabilities` (the HIP path)."""
import math

import numpy as np
import torch


class RandEdgeSampler:
  """uniform negatives from the distinct sources / destinations; a seed makes the draws repeatable and
  `reset_random_state` restarts the sequence (utils/util.py:68-87)"""

  def __init__(self, src_list, dst_list, seed=None):
    self.seed = None
    self.src_list = np.unique(src_list)
    self.dst_list = np.unique(dst_list)
    if seed is not None:
      self.seed = seed
      self.random_state = np.random.RandomState(seed)

  def sample(self, size):
    rng = np.random if self.seed is None else self.random_state
    si = rng.randint(0, len(self.src_list), size)     # two draws, sources first: the order fixes the stream
    di = rng.randint(0, len(self.dst_list), size)
    return self.src_list[si], self.dst_list[di]

  def reset_random_state(self):
    self.random_state = np.random.RandomState(self.seed)


def eval_edge_prediction(model, negative_edge_sampler, data, n_neighbors, batch_size, reuse=True, cache_plan=None):
  """mean AP / AUC / accuracy over consecutive batches of `data` in eval ordering
  (evaluation/evaluation.py:6-47).  Accuracy is "positive scored above its negative"."""
  from sklearn.metrics import accuracy_score, average_precision_score, roc_auc_score
  assert negative_edge_sampler.seed is not None
  negative_edge_sampler.reset_random_state()
  aps, aucs, accs = [], [], []
  with torch.no_grad():
    model = model.eval()
    n = data.n_interactions
    for b in range(math.ceil(n / batch_size)):
      rows = slice(b * batch_size, min(n, (b + 1) * batch_size))
      src, dst = data.sources[rows], data.destinations[rows]
      size = len(src)
      _, neg = negative_edge_sampler.sample(size)
      pos_p, neg_p = model.compute_edge_probabilities(src, dst, neg, data.timestamps[rows], data.edge_idxs[rows],
                                                      n_neighbors, reuse, False,
                                                      None if cache_plan is None else cache_plan[b])
      pos_p, neg_p = pos_p.cpu().numpy(), neg_p.cpu().numpy()
      score = np.concatenate([pos_p, neg_p])
      truth = np.concatenate([np.ones(size), np.zeros(size)])
      aps.append(average_precision_score(truth, score))
      aucs.append(roc_auc_score(truth, score))
      accs.append(accuracy_score(np.zeros(size), np.argmax(np.hstack([pos_p, neg_p]), axis=1)))
  return np.mean(aps), np.mean(aucs), np.mean(accs)
