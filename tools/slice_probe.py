#!/usr/bin/env python3
"""Compute-side cost of ONE rank's share of an N-rank step, measured on one GPU: rank 0 of a pretended world of N
runs the native sliced step on N x 1365 interactions with the exchanges stubbed out (the rows of the other slices are
stale, so the numbers are timings only).  What is missing from the picture is the all-gather itself."""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from orca_amd import ops, synth
from orca_amd.engine import OrcaEngine

dev = torch.device("cuda:0")
N, E, bs, K, d_e = 1_000_000, 50_000_000, 1365, 10, 172
import os
src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev, alpha=float(os.environ.get('SLICE_ALPHA', '1.2')))   # 0 = no hubs
csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
ef = synth.edge_features(E, d_e, dev)
sd = synth.tgn_state_dict(d_e, 2)
for world in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]):
  eng = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device=dev)
  eng.set_neighbor_finder(csr)
  eng.world, eng.rank = world, (int(__import__('os').environ.get('SLICE_RANK', '0')) % world)
  eng._gather_in_place = lambda *a, **k: None
  gbs = bs * world
  neg = torch.randint(1, N + 1, (80 * gbs,), device=dev)
  def batch(b):
    s = int(0.6 * E) + b * gbs
    return src[s:s + gbs], dst[s:s + gbs], neg[b * gbs:(b + 1) * gbs], ts[s:s + gbs], eidx[s:s + gbs]
  with torch.no_grad():
    for b in range(20):
      eng.compute_temporal_embeddings(*batch(b), train=True, gather_output=False)
    eng.profile_kernels(True, 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(20, 80):
      eng.compute_temporal_embeddings(*batch(b), train=True, gather_output=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 60
  prof = eng.profile_read()
  print(f"world {world}: {dt * 1e6:7.1f} us per step (compute only) -> {gbs / dt / 1e6:6.1f} M edges/s if the exchange were free; "
        + "  ".join(f"{k} {v[1] * 1e3:.1f} us" for k, v in prof.items()), flush=True)
