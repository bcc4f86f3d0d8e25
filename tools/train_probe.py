#!/usr/bin/env python3
"""Where a training step's wall time goes: train_step vs optimizer + re-pack, at the bench workload."""
import sys, time
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
neg = torch.randint(1, N + 1, (40 * bs,), device=dev)
def batch(b):
  s = int(0.7 * E) + b * bs
  return src[s:s + bs], dst[s:s + bs], neg[b * bs:(b + 1) * bs], ts[s:s + bs], eidx[s:s + bs]
for b in range(10):
  eng.step_optimizer(eng.train_step(*batch(b))[1])
torch.cuda.synchronize()
t_step = t_opt = 0.0
for b in range(10, 30):
  t0 = time.perf_counter()
  loss, grads = eng.train_step(*batch(b))
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  eng.step_optimizer(grads)
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  t_step += t1 - t0
  t_opt += t2 - t1
print(f"train_step {1e3 * t_step / 20:.2f} ms   optimizer+refresh {1e3 * t_opt / 20:.2f} ms   loss {float(loss):.4f}")
# the whole iteration without a sync in between (what bench.py's train_step line times)
neg = torch.randint(1, N + 1, (140 * bs,), device=dev)
for rep in range(3):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for b in range(30 + rep * 30, 60 + rep * 30):
    eng.step_optimizer(eng.train_step(*batch(b))[1])
  torch.cuda.synchronize()
  print(f"pipelined: {1e3 * (time.perf_counter() - t0) / 30:.3f} ms per training iteration")
