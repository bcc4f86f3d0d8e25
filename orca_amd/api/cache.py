"""utils/cache.py:407-428 -- get_cache_plan: GPU sampling of every training query + the MRD planner, on the device
(csrc/plan_dev.hip; args.planner = "device", the default) or on the host (csrc/plan.hip; args.planner = "host", and
the fallback for batches with more than 16384 candidate ids).  The two produce the same plan entry for entry."""
import math
import sys
import time

import numpy as np
import torch

from .. import _lib, ops
from .._lib import check, ptr


def get_cache_plan(args, train_data, neighbor_finder, num_embeddings, strategy):
  """Returns the list (one int64 array per batch) of node ids to keep cached.  Only 'MRD' (the
  reference's default, train.py:51) is built; LRU / 2Q are RNG-dependent ablation baselines."""
  if strategy != "MRD":
    print(f"unsupported cache strategy {strategy}")
    sys.exit(1)                                                   # cache.py:422-424
  t0 = time.time()
  src = np.ascontiguousarray(train_data.sources, dtype=np.int64)
  dst = np.ascontiguousarray(train_data.destinations, dtype=np.int64)
  ts = np.ascontiguousarray(train_data.timestamps, dtype=np.float64)
  E, bs, K, budget = len(src), int(args.bs), int(args.n_degree), int(args.budget)
  dev = neighbor_finder.device
  planner = getattr(args, "planner", "device")
  num_batch = math.ceil(E / bs)
  stats = np.zeros(2, np.int64)
  avg = np.zeros(1, np.float64)
  if planner == "device":
    # everything stays in HBM: the sampled neighbours go straight from the sampler into the planner
    d_src, d_dst, d_ts = (torch.from_numpy(x).to(dev) for x in (src, dst, ts))
    nbr = [ops.sample_recent(neighbor_finder.csr, ids, d_ts, K, with_dt=False)[0] for ids in (d_src, d_dst)]
    torch.cuda.synchronize(dev)
    t_prepare = time.time() - t0
    t0 = time.time()
    d_plan = torch.zeros(max(num_batch * budget, 1), dtype=torch.int64, device=dev)
    d_len = torch.zeros(max(num_batch, 1), dtype=torch.int64, device=dev)
    try:
      with torch.cuda.device(dev):
        check(_lib.load().orca_mrd_plan_device(E, bs, K, ptr(d_src), ptr(d_dst), ptr(nbr[0]), ptr(nbr[1]), int(num_embeddings),
                                               budget, ptr(d_plan), ptr(d_len), ptr(stats), ptr(avg), _lib.stream()),
              "orca_mrd_plan_device")
      plan, plan_len = d_plan.cpu().numpy(), d_len.cpu().numpy()
    except ValueError as ex:                  # batches too large for the LDS sort: the host planner takes over, loudly
      print(f"device planner not applicable ({ex}); planning on the host")
      planner = "host"
      nbr = [n.cpu().numpy() for n in nbr]
  else:
    nbr = []
    for ids in (src, dst):                                          # cache.py:39-50, all batches at once
      out = np.empty((E, K), np.int32)
      step = 1 << 22
      for s in range(0, E, step):
        n, _, _, _ = ops.sample_recent(neighbor_finder.csr, torch.from_numpy(ids[s:s + step]).to(dev),
                                       torch.from_numpy(ts[s:s + step]).to(dev), K, with_dt=False)
        out[s:s + step] = n.cpu().numpy()
      nbr.append(out)
    t_prepare = time.time() - t0
    t0 = time.time()
  if planner != "device":
    plan = np.zeros(max(num_batch * budget, 1), np.int64)
    plan_len = np.zeros(max(num_batch, 1), np.int64)
    check(_lib.load().orca_mrd_plan_host(E, bs, K, ptr(src), ptr(dst), ptr(np.ascontiguousarray(nbr[0])),
                                         ptr(np.ascontiguousarray(nbr[1])), int(num_embeddings), budget, ptr(plan),
                                         ptr(plan_len), ptr(stats), ptr(avg)), "orca_mrd_plan_host")
  t_cache = time.time() - t0
  print(f"prepare {t_prepare}, cache {t_cache}, n_reuse {stats[0]}, n_recompute {stats[1]}, "
        f"average_reuse_distance {avg[0]}")                       # cache.py:427
  return [plan[b * budget:b * budget + plan_len[b]].copy() for b in range(num_batch)]
