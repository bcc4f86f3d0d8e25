#!/usr/bin/env python3
"""Times get_cache_plan (device sampling of every interaction's neighbours + the MRD planner, on the device and on
the host) on a trace of BASELINE config 3's size: Wiki-Talk-like 7.8 M interactions, 70 % of them training
interactions, bs = 200 -> 27 k batches, budget 1000."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orca_amd import api, synth   # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--nodes", type=int, default=1_140_000)
  ap.add_argument("--edges", type=int, default=7_800_000)
  ap.add_argument("--bs", type=int, default=200)
  ap.add_argument("--budget", type=int, default=1000)
  ap.add_argument("--k", type=int, default=10)
  a = ap.parse_args()
  dev = "cuda:0"
  src, dst, eidx, ts = (x.cpu().numpy() for x in synth.powerlaw_graph(a.nodes, a.edges, dev))
  n_train = int(0.7 * a.edges)
  data = api.Data(src[:n_train], dst[:n_train], ts[:n_train], eidx[:n_train], np.zeros(n_train))
  t0 = time.time()
  nf = api.get_neighbor_finder(data, max_node_idx=a.nodes, device=dev)
  torch.cuda.synchronize()
  t_csr = time.time() - t0
  plans = {}
  for planner in ("device", "host"):
    args = argparse.Namespace(bs=a.bs, n_degree=a.k, budget=a.budget, planner=planner)
    t0 = time.time()
    plan = plans[planner] = api.get_cache_plan(args, data, nf, a.nodes + 1, "MRD")
    t_plan = time.time() - t0
    print(f"[{planner} planner] interactions {n_train}, batches {len(plan)}, budget {a.budget}: CSR build {t_csr:.2f} s, "
          f"get_cache_plan {t_plan:.2f} s, mean plan size {np.mean([len(p) for p in plan]):.1f}", flush=True)
  same = all(np.array_equal(x, y) for x, y in zip(plans["device"], plans["host"]))
  print("device plan == host plan, entry for entry:", same)


if __name__ == "__main__":
  main()
