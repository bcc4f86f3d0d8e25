#!/usr/bin/env python3
"""Host-side profile (cProfile) of the training step at the bench workload: which Python / dispatch calls the CPU
spends a step's wall time in."""
import cProfile, pstats, sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from orca_amd import ops, synth
from orca_amd.engine import OrcaEngine

dev = torch.device("cuda:0")
N, E, bs, K, d_e = 1_000_000, 20_000_000, 1365, 10, 172
src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev)
csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
ef = synth.edge_features(E, d_e, dev)
eng = OrcaEngine(synth.tgn_state_dict(d_e, 2), ef, N + 1, 2, K, reuse=True, budget=0, device=dev)
eng.set_neighbor_finder(csr)
eng.optimizer()
neg = torch.randint(1, N + 1, (60 * bs,), device=dev)
def batch(b):
  s = int(0.7 * E) + b * bs
  return src[s:s + bs], dst[s:s + bs], neg[b * bs:(b + 1) * bs], ts[s:s + bs], eidx[s:s + bs]
for b in range(10):
  eng.step_optimizer(eng.train_step(*batch(b))[1])
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for b in range(10, 50):
  eng.step_optimizer(eng.train_step(*batch(b))[1])
torch.cuda.synchronize()
pr.disable()
print(f"{1e3 * (time.perf_counter() - t0) / 40:.3f} ms per step under cProfile")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
