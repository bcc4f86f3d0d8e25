#!/usr/bin/env python3
"""Compute-side cost of ONE rank's share of an N-rank owner-computes step, measured on one GPU: rank r of a pretended
world of N runs the native step on N x 1365 interactions with the exchanges stubbed out (OrcaEngine(pretend=...): the
rows of the other ranks' chunks are stale, so the numbers are timings only).  What is missing from the picture is the
two all-gathers themselves.  usage: slice_probe.py [world ...]   env: SLICE_RANK, SLICE_ALPHA (0 = no hubs)"""
import os, sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from orca_amd import ops, synth
from orca_amd.engine import OrcaEngine

dev = torch.device("cuda:0")
N, E, bs, K, d_e = 1_000_000, 50_000_000, 1365, 10, 172
src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev, alpha=float(os.environ.get('SLICE_ALPHA', '1.2')))
csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
ef = synth.edge_features(E, d_e, dev)
sd = synth.tgn_state_dict(d_e, 2)
for world in ([int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]):
  rank = int(os.environ.get('SLICE_RANK', '0')) % world
  eng = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device=dev, pretend=(rank, world) if world > 1 else None)
  eng.set_neighbor_finder(csr)
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
  own = int(eng._w["xm_own"][0].item()) if world > 1 else -1
  print(f"world {world} rank {rank}: {dt * 1e6:7.1f} us per step (compute only) -> {gbs / dt / 1e6:6.1f} M edges/s if the exchange were free; "
        + "  ".join(f"{k} {v[1] * 1e3:.1f} us" for k, v in prof.items()) + (f"  owned rows last step {own}, exchanges sized for {eng._push_rows_last} (history push, of {3 * bs}) and {eng._xm_rows_last} (memory rows, cap {eng._xm_cap}) rows per rank" if own >= 0 else ""), flush=True)
