#!/usr/bin/env python3
"""bench.py -- temporal edges/sec through (sample + 2-layer embed + memory), K=10, on MI355X.

One "step" = one batch of the hot path in the reference's train ordering, forward only, no_grad,
dropout off (model/tgn_model.py:95-147 with --reuse, i.e. Path B modules/embedding_module.py:154-182):
GRU(message, memory) of the ids the previous batch stored -> sample 3*bs (node, t) root queries -> layer-1 TGAT
attention (+ persist memory update, history push) -> layer-2 attention over pulled history rows (+ raw-message
build and store): every operation of the reference's step, 3 kernel launches.

Workload (BASELINE.json configs[3], the one the metric's north_star is quoted on; it fits one GPU):
synthetic power-law temporal graph, 1M nodes / 100M edges, 172-d edge features, K=10, 2 layers,
4096-query batches (bs = 1365 interactions -> 4095 root queries).  Everything is generated on the
device and resident in HBM before the timed region.  N>1 (one process per GPU; `--gpus N` without a launcher in
front starts the N ranks itself): the global batch is N x 1365 interactions, each rank embeds the 4095 root queries
of its 1365 interactions (weak scaling) and does the batch's memory / message row writes for the node ids it owns
(id % N); two RCCL all-gathers per step carry the rows other ranks read (layer-1 rows for the history tables,
GRU(message, memory) of the stored ids); the final embeddings stay on the rank that computed them.

The timed region is R repeats of EXACTLY --steps consecutive steps (R chosen so that the region is >= 50 ms, at
least 5); ms_per_step / value are the median repeat, between barriers, max over ranks.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC = os.path.join(ROOT, "tools", "pmc_traffic.json")   # from separate rocprofv3 --pmc passes of this command


def pmc_traffic(args):
  """HBM bytes per k_tgat_forward launch from the committed PMC passes (2 x FETCH_SIZE + WRITE_SIZE, the gfx950
  correction of MI355X_MICROARCH.md), or None when the workload differs from the one the passes were taken on."""
  try:
    with open(PMC_TRAFFIC) as f:
      t = json.load(f)
  except (OSError, ValueError):
    return None
  w = t.get("workload", {})
  same = (w.get("nodes") == args.nodes and w.get("edges") == args.edges and w.get("d_e") == args.d_e and
          w.get("bs") == args.bs and w.get("k") == args.k and w.get("layers") == args.layers and
          w.get("reuse") == (not args.no_reuse))
  k = t.get("k_tgat_forward")
  if not same or not k:
    return None
  return (2.0 * k["fetch_size_kb"] + k["write_size_kb"]) * 1024.0


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=200)
  ap.add_argument("--warmup", type=int, default=50)
  ap.add_argument("--nodes", type=int, default=1_000_000)
  ap.add_argument("--edges", type=int, default=100_000_000)
  ap.add_argument("--d-e", type=int, default=172)
  ap.add_argument("--bs", type=int, default=1365, help="interactions per rank per batch (3*bs root queries)")
  ap.add_argument("--k", type=int, default=10)
  ap.add_argument("--layers", type=int, default=2)
  ap.add_argument("--sampling", choices=("recent", "uniform"), default="recent",
                  help="uniform = seeded-uniform neighbours (BASELINE config 5: --nodes 10000000 --edges 1000000000 --k 20 --d-e 1 "
                       "--sampling uniform); not in the reference")
  ap.add_argument("--no-reuse", action="store_true", help="Path A (exact TGN recursion) instead of Path B")
  ap.add_argument("--cpu-batches", type=int, default=200, help="batches of the CPU baseline sample (0 = skip)")
  ap.add_argument("--no-kernel-events", action="store_true")
  ap.add_argument("--event-stride", type=int, default=8,
                  help="attach HIP events to the kernels of every n-th timed step (the event plumbing costs ~20 us of "
                       "host time per step it is on)")
  ap.add_argument("--min-region-ms", type=float, default=50.0, help="repeat the --steps block until the timed region is this long")
  ap.add_argument("--path-a-steps", type=int, default=10,
                  help="N=1, Path B run: also time this many steps of Path A (exact recursion, no reuse; SURVEY section 8d), "
                       "reported as 'path_a' (0 = skip)")
  ap.add_argument("--train-steps", type=int, default=20,
                  help="N=1, Path B: also time this many full training steps (forward + backward + Adam + re-pack), "
                       "reported as the secondary figure 'train_step' (0 = skip)")
  ap.add_argument("--extra-configs", default="1,2,3,5",
                  help="N=1, default workload: BASELINE.json's other configurations timed as labelled secondary entries "
                       "('extra': config1 .. config5; '' = skip)")
  ap.add_argument("--train-dropout", type=float, default=0.4,
                  help="attention dropout of the 'train_step' figure (train.py:30 --drop_out, default 0.4; SURVEY section 8d)")
  return ap.parse_args()


# BASELINE.json configs other than the one the metric is quoted on (configs[3], the main run).  The real datasets are
# not in the image and cannot be fetched, so each is a synthetic power-law graph of the dataset's SHAPE (SURVEY
# section 8d); bs = 200 is the reference's batch size (train.py:21).
EXTRA_CONFIGS = {
    "1": dict(name="config1", what="Wikipedia shape: 9,227 nodes / 157,474 interactions, 172-d, K=10, n_layer=1, no reuse (the reference's "
                                   "default, Path A), bs=200", nodes=9227, edges=157474, d_e=172, k=10, layers=1, reuse=False,
              budget=0, bs=200, sampling="recent", warm=30, steps=200),
    "2": dict(name="config2", what="Reddit shape: 10,984 nodes / 672,447 interactions, 172-d, K=10, 2 layers, --reuse (Path B), bs=200",
              nodes=10984, edges=672447, d_e=172, k=10, layers=2, reuse=True, budget=0, bs=200, sampling="recent", warm=50, steps=400),
    "3": dict(name="config3", what="Wiki-Talk shape: 1,140,149 nodes / 7,833,140 interactions, zero-feature edges (d_e=1), K=10, 2 layers, "
                                   "--reuse --budget 1000 with the MRD plan of the device planner (Path C), bs=200",
              nodes=1140149, edges=7833140, d_e=1, k=10, layers=2, reuse=True, budget=1000, bs=200, sampling="recent", warm=100, steps=500),
    "5": dict(name="config5", what="10 M nodes / 1 B edges, zero-feature edges (d_e=1), K=20 seeded-uniform sampling (not in the reference), "
                                   "2 layers, --reuse (Path B), 4095-query batches; state replicated on the one GPU",
              nodes=10_000_000, edges=1_000_000_000, d_e=1, k=20, layers=2, reuse=True, budget=0, bs=1365, sampling="uniform",
              warm=20, steps=100),
}


def run_extra_config(c, dev):
  """one secondary configuration: forward train-order steps on consecutive time-ordered batches, drained wall time"""
  import argparse as _ap
  from orca_amd import api, ops, synth
  from orca_amd.engine import OrcaEngine
  torch.cuda.empty_cache()
  t_setup = time.perf_counter()
  src, dst, eidx, ts = synth.powerlaw_graph(c["nodes"], c["edges"], dev)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, c["nodes"] + 1)
  ef = synth.edge_features(c["edges"], c["d_e"], dev, zero=c["d_e"] == 1)
  sd = synth.tgn_state_dict(c["d_e"], c["layers"])
  n_b, bs = c["warm"] + c["steps"], c["bs"]
  start = int(0.7 * c["edges"]) if c["budget"] == 0 else int(0.5 * c["edges"])
  start = min(start, c["edges"] - n_b * bs)
  plans = None
  if c["budget"]:     # utils/cache.py:407-428 on the device, for exactly the batches that are run (train.py:157-162)
    n = n_b * bs
    h = lambda x: x[start:start + n].cpu().numpy()
    data = api.Data(h(src), h(dst), h(ts), h(eidx), np.zeros(n))
    nf = api.NeighborFinder(None, None, None, device=dev, _csr=csr)
    pa = _ap.Namespace(bs=bs, n_degree=c["k"], budget=c["budget"], planner="device")
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):      # (the planner prints the reference's statistics line, cache.py:427)
      plans = [torch.from_numpy(p).to(dev) for p in api.get_cache_plan(pa, data, nf, c["nodes"] + 1, "MRD")]
  eng = OrcaEngine(sd, ef, c["nodes"] + 1, c["layers"], c["k"], reuse=c["reuse"], budget=c["budget"], device=dev,
                   sampling=c["sampling"], seed=2024)
  eng.set_neighbor_finder(csr)
  g = torch.Generator(device=dev)
  g.manual_seed(7)
  neg = torch.randint(1, c["nodes"] + 1, (n_b * bs,), generator=g, device=dev)
  batches = [(src[start + b * bs:start + (b + 1) * bs].clone(), dst[start + b * bs:start + (b + 1) * bs].clone(),
              neg[b * bs:(b + 1) * bs], ts[start + b * bs:start + (b + 1) * bs].clone(),
              eidx[start + b * bs:start + (b + 1) * bs].clone()) for b in range(n_b)]
  del src, dst, eidx, ts
  torch.cuda.empty_cache()
  torch.cuda.synchronize()
  t_setup = time.perf_counter() - t_setup
  with torch.no_grad():
    for b in range(n_b):
      if b == c["warm"]:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
      eng.compute_temporal_embeddings(*batches[b], train=True, cache_plan=plans[b] if plans else None)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
  eng.check_errors()
  out = {"edges_per_s": c["steps"] * bs / dt, "ms_per_step": 1e3 * dt / c["steps"], "steps": c["steps"], "warmup": c["warm"],
         "queries_per_batch": 3 * bs, "setup_s": round(t_setup, 1), "data": "synthetic stand-in of the dataset's shape",
         "workload": c["what"] + "; forward train-order step (sample + embed + memory persist + message store)"}
  del eng, csr, ef, batches
  torch.cuda.empty_cache()
  return out


class DeviceBackedRows:
  """Host-side view of the 68.8 GB edge-feature table for the CPU baseline: rows are copied from HBM
  on demand (input staging only -- no arithmetic happens on the device)."""

  def __init__(self, table):
    self.table = table
    self.shape = table.shape

  def __getitem__(self, idx):
    idx = torch.as_tensor(idx)
    return self.table[idx.to(self.table.device)].cpu()


def cpu_sampler_rates(nf, batches, K, threads):
  """The CPU restatement of the reference's sampler loop (utils/util.py:141-150 is a serial Numba loop) on the
  batches' root queries: one core, and all cores (the same C loop over per-thread chunks of the queries; labelled,
  the reference itself has no parallel sampler)."""
  from concurrent.futures import ThreadPoolExecutor
  nodes = np.concatenate([np.concatenate([x.cpu().numpy() for x in (b[0], b[1], b[2])]) for b in batches])
  qts = np.concatenate([np.tile(b[3].cpu().numpy(), 3) for b in batches])
  t0 = time.perf_counter()
  nf.get_temporal_neighbor(nodes, qts, K)
  serial = len(nodes) / (time.perf_counter() - t0)
  cuts = np.linspace(0, len(nodes), threads + 1).astype(np.int64)
  t0 = time.perf_counter()
  with ThreadPoolExecutor(threads) as ex:      # ctypes releases the GIL around the C loop
    list(ex.map(lambda i: nf.get_temporal_neighbor(nodes[cuts[i]:cuts[i + 1]], qts[cuts[i]:cuts[i + 1]], K), range(threads)))
  par = len(nodes) / (time.perf_counter() - t0)
  return {"queries": int(len(nodes)), "serial_queries_per_s": serial, "all_cores_queries_per_s": par, "cores": threads,
          "what": "oracle C sampler (serial loop of utils/util.py:141-150); all-cores = the same loop on per-thread chunks"}


def cpu_baseline(args, csr, edge_feats, sd, batches, threads):
  """The oracle (oracle/oracle.py, the CPU restatement of the reference's PyTorch+Numba path; "port")
  timed on the host cores on the first batches of the timed sequence."""
  from oracle import oracle as O
  torch.set_num_threads(threads)
  nf = O.OracleNeighborFinder.from_arrays(*csr.arrays_host())
  ref = O.OracleTGN(sd, torch.zeros(1, args.d_e), args.nodes + 1, args.layers, args.k, not args.no_reuse, 0)
  ref.edge_feats = DeviceBackedRows(edge_feats)
  ref.neighbor_finder = nf
  if args.sampling == "uniform":
    ref.uniform_seed = 2024
  warm = 2
  t0 = None
  n_edges = n_timed = 0
  for i, b in enumerate(batches[:warm + args.cpu_batches]):
    if i == warm:
      t0 = time.perf_counter()
    s, d, n, t, e = (x.cpu().numpy() for x in b)
    ref.compute_temporal_embeddings(s, d, n, t, e, True)
    if i >= warm:
      n_edges += len(s)
      n_timed += 1
  dt = time.perf_counter() - t0
  return {"value": n_edges / dt, "unit": "temporal edges/s", "cores": threads, "kind": "port",
          "sample": f"{n_timed} consecutive {args.bs}-interaction batches of the same workload after {warm} "
                    f"warm-up batches, oracle.OracleTGN ({'Path A' if args.no_reuse else 'Path B'}, forward, torch CPU "
                    f"fp32 + serial C sampler), {dt:.1f} s of CPU work",
          "sampler": cpu_sampler_rates(nf, batches[:min(len(batches), 50)], args.k, threads)}


def launch_ranks(args):
  """`python bench.py --gpus N` with no launcher in front: start the N ranks as children (torch.distributed.run,
  one per GPU, rendezvous on 127.0.0.1) BEFORE this process touches the GPU, and exit with their status."""
  import socket
  import subprocess
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  raise SystemExit(subprocess.call(cmd))


def main():
  args = parse()
  if "WORLD_SIZE" not in os.environ and args.gpus > 1:
    launch_ranks(args)
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != args.gpus:
    raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
  rank = int(os.environ.get("RANK", "0"))
  local = int(os.environ.get("LOCAL_RANK", "0"))
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a HIP device (there is no CPU fallback for the product path)")
  if os.environ.get("ORCA_BENCH_BACKEND", "nccl") != "nccl":
    local %= torch.cuda.device_count()        # rehearsal: several ranks share the GPUs there are
  torch.cuda.set_device(local)
  dev = torch.device("cuda", local)
  group = None
  if world > 1:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # "nccl" (= RCCL) on a multi-GPU node; ORCA_BENCH_BACKEND=gloo rehearses the N-rank path with several ranks
    # on ONE GPU (RCCL refuses two ranks per device), rows staged through the host
    backend = os.environ.get("ORCA_BENCH_BACKEND", "nccl")
    if backend == "nccl":
      dist.init_process_group("nccl", device_id=dev)
    else:
      dist.init_process_group(backend)
    group = dist.group.WORLD

  from orca_amd import ops, synth
  from orca_amd.engine import OrcaEngine

  t_setup = time.perf_counter()
  src, dst, eidx, ts = synth.powerlaw_graph(args.nodes, args.edges, dev)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, args.nodes + 1)
  ef = synth.edge_features(args.edges, args.d_e, dev, zero=args.d_e == 1)
  sd = synth.tgn_state_dict(args.d_e, args.layers)
  eng = OrcaEngine(sd, ef, args.nodes + 1, args.layers, args.k, reuse=not args.no_reuse, budget=0, device=dev,
                   group=group, sampling=args.sampling, seed=2024)
  eng.set_neighbor_finder(csr)
  # consecutive time-ordered batches from the 70 % mark (train.py:131-137 chronological split)
  gbs = args.bs * world
  n_train = args.train_steps + 5 if (world == 1 and not args.no_reuse and args.train_steps > 0) else 0
  start = int(0.7 * args.edges)
  n_avail = (args.edges - start) // gbs - args.warmup - n_train          # batches left for the timed region
  max_repeats = max(1, min(64, n_avail // max(args.steps, 1)))
  n_batches = args.warmup + args.steps * max_repeats + n_train
  if start + n_batches * gbs > args.edges:
    start = max(0, args.edges - n_batches * gbs)
  g = torch.Generator(device=dev)
  g.manual_seed(7)
  neg_all = torch.randint(1, args.nodes + 1, (n_batches * gbs,), generator=g, device=dev)
  batches = []
  for b in range(n_batches):
    s, e = start + b * gbs, start + (b + 1) * gbs
    batches.append((src[s:e], dst[s:e], neg_all[b * gbs:(b + 1) * gbs], ts[s:e], eidx[s:e]))
  torch.cuda.synchronize()
  t_setup = time.perf_counter() - t_setup

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      import torch.distributed as dist
      dist.barrier()
    torch.cuda.synchronize()

  with torch.no_grad():
    # N > 1: every rank keeps the embeddings of its own queries (data-parallel consumers read them where they are);
    # the exchange that the path itself needs -- layer-1 rows for the replicated history -- stays
    gather = world == 1 or args.no_reuse
    for b in batches[:args.warmup]:
      eng.compute_temporal_embeddings(*b, train=True, gather_output=gather)
    if not args.no_kernel_events:
      eng.profile_kernels(True, args.event_stride)       # native driver (Path B)
      if args.no_reuse and args.layers > 1:
        eng.prof = {}                 # Python-dispatched path (Path A with recursion; one layer runs natively)
    # R repeats of exactly --steps steps, each between barriers (max over ranks); R is fixed after the first
    # repeat so that the whole timed region lasts >= --min-region-ms (at least 5 repeats, at most 64)
    def max_over_ranks(x):
      if world == 1:
        return x
      import torch.distributed as dist
      tt = torch.tensor([x], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
      dist.all_reduce(tt, op=dist.ReduceOp.MAX)
      return float(tt.item())
    reps, r, n_rep = [], 0, 5
    while r < n_rep:
      blk = batches[args.warmup + r * args.steps:args.warmup + (r + 1) * args.steps]
      barrier()
      t0 = time.perf_counter()
      for b in blk:
        eng.compute_temporal_embeddings(*b, train=True, gather_output=gather)
      barrier()
      reps.append(max_over_ranks(time.perf_counter() - t0))
      if r == 0:
        n_rep = int(max_over_ranks(float(min(max_repeats, max(5, int(np.ceil(1.25 * args.min_region_ms * 1e-3 / max(reps[0], 1e-9))))))))
      r += 1
    dt = float(np.median(reps))
    eng.check_errors()
  prof = eng.profile_read() or eng.prof_summary()   # HIP events recorded around the kernels inside the timed region
  eng.profile_kernels(False)
  eng.prof = None
  train = None
  if n_train:       # secondary figure: the full training step on the batches that follow
    eng.dropout = float(args.train_dropout)        # nn.MultiheadAttention(dropout=0.4) in train() mode, train.py:30
    eng.optimizer(lr=1e-4)
    tb = batches[args.warmup + args.steps * max_repeats:]
    for b in tb[:5]:
      eng.step_optimizer(eng.train_step(*b)[1])
    barrier()
    t1 = time.perf_counter()
    for b in tb[5:]:
      loss, grads = eng.train_step(*b)
      eng.step_optimizer(grads)
    barrier()
    t1 = time.perf_counter() - t1
    train = {"edges_per_s": args.train_steps * gbs / t1, "ms_per_step": 1e3 * t1 / args.train_steps,
             "steps": args.train_steps, "last_loss": float(loss), "dropout": float(args.train_dropout),
             "what": "train-ordered forward + BCE loss + backward (attention x layers, GRU) + Adam + blob re-pack; "
                     "backward per layer = fused row-chain kernels + single-pass per-query kernels + one weight-gradient GEMM; "
                     "fused Adam; forward issued by the C++ step driver (training capture), backward dispatched from Python"}

  path_a = None
  if world == 1 and not args.no_reuse and args.layers > 1 and args.path_a_steps > 0 and args.sampling == "recent":
    # secondary figure: the exact TGN recursion (Path A, modules/embedding_module.py:116-147) on the
    # same workload -- (1+K) times the attention evaluations of Path B's first layer, Python-dispatched
    ea = OrcaEngine(sd, ef, args.nodes + 1, args.layers, args.k, reuse=False, budget=0, device=dev)
    ea.set_neighbor_finder(csr)
    with torch.no_grad():
      for b in batches[:3]:
        ea.compute_temporal_embeddings(*b, train=True)
      barrier()
      t1 = time.perf_counter()
      for b in batches[3:3 + args.path_a_steps]:
        ea.compute_temporal_embeddings(*b, train=True)
      barrier()
      t1 = time.perf_counter() - t1
    path_a = {"edges_per_s": args.path_a_steps * gbs / t1, "ms_per_step": 1e3 * t1 / args.path_a_steps,
              "steps": args.path_a_steps, "what": "Path A (no reuse, exact recursion), forward train-order step, same batches"}
    del ea
  if rank == 0:
    Q = 3 * args.bs
    out = {
        "metric": "temporal edges/sec (sample+2-layer embed+memory), K=10",
        "value": args.steps * gbs / dt,
        "unit": "temporal edges/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "repeats": len(reps), "timed_region_s": float(sum(reps)),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"synthetic power-law temporal graph {args.nodes} nodes/{args.edges} edges, "
                               f"{args.d_e}-d edge feats, K={args.k}{' seeded-uniform' if args.sampling == 'uniform' else ''}, {args.layers} layers, "
                               f"{'Path A exact recursion' if args.no_reuse else 'Path B --reuse'}, "
                               f"{Q}-query batches ({args.bs} interactions) per GPU, forward train-order step",
                   "root_queries_per_s": args.steps * gbs * 3 / dt,
                   "outputs": "on the rank that computed them" if world > 1 and not args.no_reuse else "one tensor",
                   "parallelism": (f"{world} ranks: roots partitioned by interaction, state rows owned by id % {world}, "
                                   "2 all-gathers per step") if world > 1 else "1 rank",
                   "setup_s": round(t_setup, 1)},
    }
    if "tgat_forward" in prof:
      n, ms = prof["tgat_forward"]
      per_launch_q = Q * (1 + args.k) if args.no_reuse and args.layers > 1 else Q
      # Path A launches differ in size (layer 1: Q*(1+K) queries, layer 2: Q); use the mean per launch
      q_mean = (Q * (1 + args.k) + Q) / 2 if args.no_reuse and args.layers == 2 else per_launch_q
      flops = synth.tgat_flops_per_query(args.d_e, args.k) * q_mean
      ach = flops / (ms * 1e-3) / 1e12
      out["roofline"] = {"kernel": "k_tgat_forward", "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": pmc_traffic(args),
                         "traffic_unit": "HBM bytes per launch (profiles/r03_pmc_hbm_traffic.md)",
                         "launches": n, "avg_launch_ms": ms, "queries_per_launch": q_mean,
                         "flops_per_query": synth.tgat_flops_per_query(args.d_e, args.k),
                         "flops_executed_per_query": synth.tgat_flops_executed_per_query(args.d_e, args.k),
                         "bytes_per_query": synth.tgat_bytes_per_query(args.d_e, args.k),
                         "hbm_gbs_algorithmic": synth.tgat_bytes_per_query(args.d_e, args.k) * q_mean / (ms * 1e-3) / 1e9}
      out["kernels_ms"] = {k: {"launches": v[0], "avg_ms": v[1]} for k, v in prof.items()}
      if "sample_recent" in prof and not args.no_reuse:   # first kernel of the step: build + sample + elections
        deg = 2.0 * args.edges / max(args.nodes, 1)
        bpq = 32 * args.k + 16 + 8 * int(np.ceil(np.log2(deg + 1)))
        sms = prof["sample_recent"][1]
        out["sampler"] = {"bytes_per_query": bpq, "queries_per_launch": Q, "avg_launch_ms": sms,
                          "achieved_gbs": bpq * Q / (sms * 1e-3) / 1e9, "peak_gbs": PEAK_HBM_GBS,
                          "note": "latency-bound: ~5 dependent memory round trips per query, 1.6 MB per launch"}
      w = getattr(eng, "_w", None)
      if w is not None and "gru_update" in prof:      # rows of the last eager list = distinct ids a batch stores
        rows = int(w["eager_count"].max().item())
        if rows:
          gms = prof["gru_update"][1]
          out["gru"] = {"rows_per_launch": rows, "flops_per_row": 2 * (3 * 100 + args.d_e + 100) * 300,
                        "achieved_tflops": rows * 2 * (3 * 100 + args.d_e + 100) * 300 / (gms * 1e-3) / 1e12}
    if train:
      out["train_step"] = train
    if path_a:
      out["path_a"] = path_a
    if world == 1 and args.cpu_batches > 0:
      threads = min(16, os.cpu_count() or 1)
      out["cpu_baseline"] = cpu_baseline(args, csr, ef, sd, batches[args.warmup:], threads)
    default_workload = (world == 1 and not args.no_reuse and args.layers == 2 and args.d_e == 172 and args.k == 10 and
                        args.sampling == "recent" and args.nodes == 1_000_000 and args.edges == 100_000_000)
    if default_workload and args.extra_configs:
      # BASELINE.json's other configurations as labelled secondary entries (the bench line proper is configs[3])
      del eng, csr, ef, batches, neg_all, src, dst, eidx, ts
      torch.cuda.empty_cache()
      out["extra"] = {"config4": {"edges_per_s": out["value"], "ms_per_step": out["ms_per_step"],
                                  "workload": "this line's own workload (BASELINE.json configs[3])"}}
      for key in args.extra_configs.split(","):
        c = EXTRA_CONFIGS.get(key.strip())
        if c is None:
          continue
        try:
          out["extra"][c["name"]] = run_extra_config(c, dev)
        except Exception as ex:        # a secondary figure must not cost the line (e.g. less free HBM than config 5 needs)
          out["extra"][c["name"]] = {"error": f"{type(ex).__name__}: {ex}"}
    print(json.dumps(out), flush=True)
  if world > 1:
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
