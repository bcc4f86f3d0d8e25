#!/usr/bin/env python3
"""Host-vs-device split of the native Path B step on the bench workload: time to ENQUEUE n steps (host side,
no synchronisation) against the time until the device has drained them, with and without the HIP-event
brackets bench.py keeps around the sampler / GRU / attention launches."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orca_amd import ops, synth   # noqa: E402
from orca_amd.engine import OrcaEngine   # noqa: E402


def main():
  dev = torch.device("cuda", 0)
  N, E, d_e, K, bs = 1_000_000, 100_000_000, 172, 10, 1365
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  ef = synth.edge_features(E, d_e, dev)
  eng = OrcaEngine(synth.tgn_state_dict(d_e, 2), ef, N + 1, 2, K, reuse=True, budget=0, device=dev)
  eng.set_neighbor_finder(csr)
  n = 300
  start = int(0.7 * E)
  g = torch.Generator(device=dev)
  g.manual_seed(7)
  neg = torch.randint(1, N + 1, (n * bs,), generator=g, device=dev)
  batches = [(src[start + b * bs:start + (b + 1) * bs], dst[start + b * bs:start + (b + 1) * bs], neg[b * bs:(b + 1) * bs],
              ts[start + b * bs:start + (b + 1) * bs], eidx[start + b * bs:start + (b + 1) * bs]) for b in range(n)]
  with torch.no_grad():
    for events in (False, True, False):
      eng.profile_kernels(events)
      for b in batches[:50]:
        eng.compute_temporal_embeddings(*b, train=True)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for b in batches[50:]:
        eng.compute_temporal_embeddings(*b, train=True)
      t1 = time.perf_counter()
      torch.cuda.synchronize()
      t2 = time.perf_counter()
      eng.profile_read()
      k = n - 50
      print(f"events={events}: enqueue {1e6 * (t1 - t0) / k:.1f} us/step, drained {1e6 * (t2 - t0) / k:.1f} us/step", flush=True)


if __name__ == "__main__":
  main()
