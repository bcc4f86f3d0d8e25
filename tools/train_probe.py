#!/usr/bin/env python3
"""Is the training step bound by the GPU or by the host that dispatches it?  Runs bench.py's training leg and reports,
per step: wall time, the host time spent inside train_step + step_optimizer excluding the one wait on the work-list
count, and the GPU backlog left when the host has issued the last step.  usage: train_probe.py [steps]"""
import sys, time
import torch
sys.path.insert(0, __file__.rsplit("/", 2)[0])
from orca_amd import ops, synth
from orca_amd.engine import OrcaEngine

dev = torch.device("cuda:0")
N, E, bs, K, d_e = 1_000_000, 100_000_000, 1365, 10, 172
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev)
csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
ef = synth.edge_features(E, d_e, dev)
eng = OrcaEngine(synth.tgn_state_dict(d_e, 2), ef, N + 1, 2, K, reuse=True, budget=0, device=dev, dropout=0.4)
eng.set_neighbor_finder(csr)
neg = torch.randint(1, N + 1, ((steps + 30) * bs,), device=dev)
def batch(b):
  s = int(0.6 * E) + b * bs
  return src[s:s + bs], dst[s:s + bs], neg[b * bs:(b + 1) * bs], ts[s:s + bs], eidx[s:s + bs]
with torch.no_grad():
  for b in range(20):
    eng.compute_temporal_embeddings(*batch(b), train=True)
eng.optimizer(lr=1e-4)
for b in range(20, 30):
  eng.step_optimizer(eng.train_step(*batch(b))[1])
torch.cuda.synchronize()
waited = [0.0]
ev_sync = torch.cuda.Event.synchronize
def timed_sync(self):
  t = time.perf_counter()
  ev_sync(self)
  waited[0] += time.perf_counter() - t
torch.cuda.Event.synchronize = timed_sync
t0 = time.perf_counter()
for b in range(30, 30 + steps):
  loss, grads = eng.train_step(*batch(b))
  eng.step_optimizer(grads)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"wall {t_all / steps * 1e6:.1f} us per step; host issue {(t_issue - waited[0]) / steps * 1e6:.1f} us per step "
      f"(+ {waited[0] / steps * 1e6:.1f} us waiting on the count event); GPU backlog after the last issue {(t_all - t_issue) * 1e6:.1f} us")
if len(sys.argv) > 2:
  import cProfile, pstats
  pr = cProfile.Profile()
  pr.enable()
  for b in range(30, 30 + steps):
    loss, grads = eng.train_step(*batch(b % 60))
    eng.step_optimizer(grads)
  pr.disable()
  torch.cuda.synchronize()
  pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
