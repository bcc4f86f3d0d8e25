#!/usr/bin/env python3
"""Path C (Orca-R, --reuse --budget 1000) at BASELINE config 3's shape on one GPU: a Wiki-Talk-like graph (1.14 M nodes,
7.83 M interactions, zero-feature edges: d_e = 1), bs = 200, 2 layers, K = 10.  Plans the first `--batches` training
batches with the device MRD planner, then runs them through OrcaEngine in train order -- the budgeted-reuse step issued by
the native driver (orca_engine_step_budget) and, for comparison, Python-dispatched -- and reports temporal edges/s next to
Path B on the same batches."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orca_amd import api, ops, synth   # noqa: E402
from orca_amd.engine import OrcaEngine  # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--nodes", type=int, default=1_140_149)
  ap.add_argument("--edges", type=int, default=7_833_140)
  ap.add_argument("--bs", type=int, default=200)
  ap.add_argument("--budget", type=int, default=1000)
  ap.add_argument("--k", type=int, default=10)
  ap.add_argument("--batches", type=int, default=600)
  a = ap.parse_args()
  dev = torch.device("cuda:0")
  src, dst, eidx, ts = synth.powerlaw_graph(a.nodes, a.edges, dev)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, a.nodes + 1)
  ef = synth.edge_features(a.edges, 1, dev, zero=True)
  sd = synth.tgn_state_dict(1, 2)
  n = a.batches * a.bs
  start = int(0.5 * a.edges)
  h = lambda x: x[start:start + n].cpu().numpy()
  data = api.Data(h(src), h(dst), h(ts), h(eidx), np.zeros(n))
  nf = api.NeighborFinder(None, None, None, device=dev, _csr=csr)
  args = argparse.Namespace(bs=a.bs, n_degree=a.k, budget=a.budget, planner="device")
  plans = [torch.from_numpy(p).to(dev) for p in api.get_cache_plan(args, data, nf, a.nodes + 1, "MRD")]
  g = torch.Generator(device=dev)
  g.manual_seed(5)
  neg = torch.randint(1, a.nodes + 1, (n,), generator=g, device=dev)
  res = {}
  for name, budget, native in (("Path C (budget %d), native step driver" % a.budget, a.budget, True),
                               ("Path C (budget %d), Python-dispatched" % a.budget, a.budget, False),
                               ("Path B (budget 0)", 0, True)):
    eng = OrcaEngine(sd, ef, a.nodes + 1, 2, a.k, reuse=True, budget=budget, device=dev)
    eng.native_budget = native
    eng.set_neighbor_finder(csr)
    warm = a.batches // 6
    with torch.no_grad():
      for b in range(a.batches):
        if b == warm:
          torch.cuda.synchronize()
          t0 = time.perf_counter()
        s = start + b * a.bs
        eng.compute_temporal_embeddings(src[s:s + a.bs], dst[s:s + a.bs], neg[b * a.bs:(b + 1) * a.bs], ts[s:s + a.bs],
                                        eidx[s:s + a.bs], True, plans[b] if budget else None)
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
    res[name] = (a.batches - warm) * a.bs / dt
    if budget and native:
      assert not eng.budget_overflow(), "more uncached neighbours than budget_cap"
    print(f"{name}: {res[name] / 1e3:.1f} k temporal edges/s ({1e3 * dt / (a.batches - warm):.3f} ms per {a.bs}-interaction batch)", flush=True)


if __name__ == "__main__":
  main()
