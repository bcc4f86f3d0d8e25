"""N-rank == 1-rank equality of the partitioned step: several ranks share the one GPU of
the test box and exchange rows over gloo (RCCL refuses two ranks on one device; bench.py uses "nccl" on
the 8-GPU node).  Results must be bit-identical to the single-rank engine on the same batches.

Train-ordered Path B steps run the owner-computes route (state rows owned by id % world, two all-gathers per step);
eval-ordered steps, Paths A / C and train_step run on coherent replicas."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _state_equal(multi, single, reuse):
  multi.sync_replicas()     # owner rows -> every replica (collective: every rank calls it)
  ok = torch.equal(multi.memory, single.memory)
  ok = ok and torch.equal(multi.messages, single.messages) and torch.equal(multi.pending, single.pending)
  ok = ok and torch.equal(multi.last_update, single.last_update) and torch.equal(multi.msg_ts, single.msg_ts)
  if reuse:
    ok = ok and all(torch.equal(a, b) for a, b in zip(multi.hist, single.hist))
  return ok


def _worker(rank, world, port, q, full):
  try:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orca_amd import _lib, ops, synth
    from orca_amd.dist import interaction_slice, query_slice
    from orca_amd.engine import OrcaEngine
    dev = "cuda:0"
    N, E, bs, K, d_e = 3000, 60000, 301, 10, 172
    src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev, seed=5)
    ef = synth.edge_features(E, d_e, dev)
    sd = synth.tgn_state_dict(d_e, 2, seed=2)
    csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    ok, msg = True, ""

    def batch(b):
      s, e = 50000 + b * bs, 50000 + (b + 1) * bs
      return src[s:e], dst[s:e], torch.randint(1, N + 1, (bs,), generator=g, device=dev), ts[s:e], eidx[s:e]

    # (small_batch=False on the one-rank engines: bit equality is between EQUAL launch shapes -- a rank's attention launches
    # keep the one-wave-per-query form, which one rank alone would trade for the four-wave form at these batch sizes)
    for reuse in ((True, False) if full else (True,)):
      multi = OrcaEngine(sd, ef, N + 1, 2, K, reuse=reuse, device=dev, group=dist.group.WORLD)
      single = OrcaEngine(sd, ef, N + 1, 2, K, reuse=reuse, device=dev, group=None, small_batch=False)
      multi.set_neighbor_finder(csr)
      single.set_neighbor_finder(csr)
      for b in range(9):
        bt = batch(b)
        train = b not in (3, 4)
        if b == 6:      # a restored backup drops the overlay-complete state: the owner-computes step rebuilds the
          multi.restore(multi.backup())      # overlay rows of the pending ids first
          single.restore(single.backup())
        sharded = reuse and b in (2, 7)     # outputs left on the rank that computed them (bench.py, N > 1)
        a = multi.compute_temporal_embeddings(*bt, train, gather_output=not sharded)
        c = single.compute_temporal_embeddings(*bt, train)
        if sharded and train:   # the rank's own interactions: rows [src | dst | neg] of them
          lo, hi, _ = interaction_slice(bs, rank, world)
          c = torch.cat([c[k * bs + lo:k * bs + hi] for k in range(3)])
        elif sharded:
          lo, hi, _ = query_slice(3 * bs, rank, world)
          c = c[lo:hi]
        step_ok = torch.equal(a, c)
        if b in (1, 4, 8) or not reuse:       # also mid-stream: sync, compare, and carry on from coherent replicas
          step_ok = step_ok and _state_equal(multi, single, reuse)
        if not step_ok and not msg:
          msg = f"reuse={reuse} batch {b}: embeddings or state differ from the single-rank engine"
        ok = ok and step_ok
      multi.check_errors()
      if reuse:       # the updated-memory exchange of the last train step was sized by that step's own counts: a multiple
        # of 64 rows under the cap, the same on every rank, and enough for what this rank sent
        rows, sent = multi._xm_rows_last, int(multi._w["xm_own"][0].item())
        every = [None] * world
        dist.all_gather_object(every, rows)
        sized = rows % 64 == 0 and sent <= rows < multi._xm_cap and all(r == rows for r in every)
        if not sized and not msg:
          msg = f"exchange of {rows} rows per rank (ranks: {every}) for {sent} sent, cap {multi._xm_cap}"
        ok = ok and sized
        # the same for the history-push records: every rank's gathered chunk fits the step's size, which is under 3*per
        push = multi._push_rows_last
        words = int(_lib.load().orca_engine_xm_chunk_words(push))
        counts = [int(multi._w["xl"][c * words].item()) for c in range(world)]
        dist.all_gather_object(every, push)
        per = -(-bs // world)
        sized = push % 64 == 0 and max(counts) <= push <= 3 * per and push > max(counts) - 64 and all(r == push for r in every)
        if not sized and not msg:
          msg = f"history-push exchange of {push} rows per rank (ranks: {every}), chunk counts {counts}, 3*per {3 * per}"
        ok = ok and sized
    # three layers (two history exchanges per step), 20 uniformly sampled neighbours, zero-feature edges
    # (BASELINE config 5's sampler and width): owner-computes steps only
    ef1 = torch.zeros(E + 1, 1, device=dev)
    sd3 = synth.tgn_state_dict(1, 3, seed=4)
    multi = OrcaEngine(sd3, ef1, N + 1, 3, 20, reuse=True, device=dev, group=dist.group.WORLD, sampling="uniform", seed=77)
    single = OrcaEngine(sd3, ef1, N + 1, 3, 20, reuse=True, device=dev, group=None, small_batch=False, sampling="uniform", seed=77)
    multi.set_neighbor_finder(csr)
    single.set_neighbor_finder(csr)
    for b in range(4):
      bt = batch(b)
      a = multi.compute_temporal_embeddings(*bt, True)
      c = single.compute_temporal_embeddings(*bt, True)
      if not torch.equal(a, c) and not msg:
        msg = f"3 layers / uniform K=20: batch {b} differs"
      ok = ok and torch.equal(a, c)
    ok = ok and _state_equal(multi, single, True)
    # an owner that holds more distinct positives than exchange_cap must say so
    small = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, device=dev, group=dist.group.WORLD, exchange_cap=8)
    small.set_neighbor_finder(csr)
    small.compute_temporal_embeddings(*batch(0), True)
    try:
      small.check_errors()
      ok, msg = False, msg or "exchange overflow was not reported"
    except Exception as ex:
      if "exchange_cap" not in str(ex):
        raise
    if full:
      # Path C (budgeted reuse with a cache plan per batch): global set algebra on every rank, attention partitioned
      multi = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=40, device=dev, group=dist.group.WORLD)
      single = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=40, device=dev, group=None, small_batch=False, slot_history=False)   # (random plans)
      multi.set_neighbor_finder(csr)
      single.set_neighbor_finder(csr)
      for b in range(3):
        bt = batch(b)
        plan = torch.randint(1, N + 1, (40,), generator=g, device=dev)
        a = multi.compute_temporal_embeddings(*bt, True, plan)
        c = single.compute_temporal_embeddings(*bt, True, plan)
        ok = ok and torch.equal(a, c) and torch.equal(multi.memory, single.memory)
        ok = ok and torch.equal(multi.hist[0], single.hist[0]) and torch.equal(multi.cache_flag[0], single.cache_flag[0])
      # training step: every rank back-propagates its slice, parameter gradients are all-reduced; the sums associate
      # differently from the single-rank step, so gradients are compared at 1e-5 and the state bit for bit.  Forward
      # steps in between switch the state between the owner-computes and the coherent form.
      multi = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, device=dev, group=dist.group.WORLD)
      single = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, device=dev, group=None, small_batch=False)
      multi.set_neighbor_finder(csr)
      single.set_neighbor_finder(csr)
      for b in range(5):
        bt = batch(b)
        if b == 2:
          a = multi.compute_temporal_embeddings(*bt, True)
          c = single.compute_temporal_embeddings(*bt, True)
          ok = ok and torch.equal(a, c)
          continue
        la, ga = multi.train_step(*bt)
        lc, gc = single.train_step(*bt)
        ok = ok and abs(float(la) - float(lc)) <= 1e-6 and set(ga) == set(gc)
        for k in gc:
          scale = max(float(gc[k].abs().max()), 1e-12)
          err = float((ga[k] - gc[k]).abs().max()) / scale
          if err > 1e-5:
            ok, msg = False, f"train_step gradient {k} differs between {world} ranks and 1 rank: rel {err:.2e}"
        ok = ok and torch.equal(multi.memory, single.memory) and torch.equal(multi.messages, single.messages)
        ok = ok and torch.equal(multi.hist[0], single.hist[0])
      # the same for the budgeted reuse path (Path C): every rank back-propagates its slice of the roots
      multi = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=40, device=dev, group=dist.group.WORLD)
      single = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=40, device=dev, group=None, small_batch=False, slot_history=False)   # (random plans)
      multi.set_neighbor_finder(csr)
      single.set_neighbor_finder(csr)
      for b in range(3):
        bt = batch(b)
        plan = torch.randint(1, N + 1, (40,), generator=g, device=dev)
        la, ga = multi.train_step(*bt, cache_plan=plan)
        lc, gc = single.train_step(*bt, cache_plan=plan)
        ok = ok and abs(float(la) - float(lc)) <= 1e-6 and set(ga) == set(gc)
        for k in gc:
          scale = max(float(gc[k].abs().max()), 1e-12)
          err = float((ga[k] - gc[k]).abs().max()) / scale
          if err > 1e-5:
            ok, msg = False, f"Path C train_step gradient {k} differs between {world} ranks and 1 rank: rel {err:.2e}"
        same = torch.equal(multi.memory, single.memory) and torch.equal(multi.hist[0], single.hist[0])
        same = same and torch.equal(multi.cache_flag[0], single.cache_flag[0]) and torch.equal(multi.messages, single.messages)
        if not same and not msg:
          msg = f"Path C train_step batch {b}: state differs from the single-rank engine"
        ok = ok and same
    q.put((rank, bool(ok), msg))
    dist.barrier()
    dist.destroy_process_group()
  except Exception:   # surface the failure in the parent
    import traceback
    q.put((rank, False, traceback.format_exc()))


@pytest.mark.parametrize("world,full", [(2, True), (4, False)])
def test_ranks_equal_one_rank(world, full):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  ps = [ctx.Process(target=_worker, args=(r, world, port, q, full)) for r in range(world)]
  for p in ps:
    p.start()
  res = [q.get(timeout=300) for _ in range(world)]
  for p in ps:
    p.join(timeout=60)
  for r, ok, msg in res:
    assert ok, f"rank {r}: {msg or 'results differ from the single-rank engine'}"


def _rccl_worker(port, q):
  try:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    from orca_amd.dist import all_gather_chunks, all_gather_into, all_gather_rows, all_reduce_sum, sync_owned_rows
    dev = "cuda:0"
    buf = torch.arange(1000, dtype=torch.int32, device=dev)
    w = all_gather_chunks(buf, 640, dist.group.WORLD, async_op=True)     # in-place all_gather_into_tensor on RCCL
    if w is not None:
      w.wait()
    ok = torch.equal(buf, torch.arange(1000, dtype=torch.int32, device=dev))
    full = torch.zeros(1000, dtype=torch.int32, device=dev)             # out of place, a prefix of both buffers: the sized
    w = all_gather_into(full[:384], buf[:384], dist.group.WORLD, async_op=True)      # record exchanges of the N-rank step
    if w is not None:
      w.wait()
    ok = ok and torch.equal(full[:384], buf[:384]) and int(full[384:].abs().sum()) == 0
    t = torch.randn(50, 7, device=dev)
    c = t.clone()
    sync_owned_rows([t], dist.group.WORLD)
    ok = ok and torch.equal(t, c)
    g = all_reduce_sum(torch.ones(9, device=dev), dist.group.WORLD)
    ok = ok and torch.equal(g, torch.ones(9, device=dev))
    x = all_gather_rows(c, 50, 50, dist.group.WORLD)
    ok = ok and torch.equal(x, c)
    torch.cuda.synchronize()
    q.put((bool(ok), ""))
    dist.destroy_process_group()
  except Exception:
    import traceback
    q.put((False, traceback.format_exc()))


def test_exchange_helpers_on_rccl_single_rank():
  """The exchange helpers of the owner-computes step on the RCCL backend itself ("nccl"): one rank is all a
  one-GPU box allows (RCCL refuses two ranks per device), which still runs the in-place all-gather, the masked
  all-reduce and the async work handle through RCCL rather than through the gloo staging of the other tests."""
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
  p.start()
  ok, msg = q.get(timeout=240)
  p.join(timeout=60)
  assert ok, msg


def test_bench_launches_its_own_ranks():
  """`python bench.py --gpus 2` with no launcher in front -- the command form of the driver's scaling run -- starts its
  ranks itself (torch.distributed.run, 127.0.0.1) and rank 0 prints the one JSON line.  Rehearsed with two gloo ranks
  sharing this box's GPU (RCCL refuses two ranks per device) on a small graph; what is checked is the launcher path and
  the line's contract, not a rate."""
  import json
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, ORCA_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
  env.pop("WORLD_SIZE", None)
  env.pop("RANK", None)
  cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--nodes", "20000",
         "--edges", "400000", "--bs", "200", "--min-region-ms", "1", "--cpu-batches", "0", "--train-steps", "0",
         "--path-a-steps", "0", "--extra-configs", ""]
  r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
  assert r.returncode == 0, r.stderr[-3000:]
  lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
  assert len(lines) == 1, r.stdout[-2000:]
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2 and d["scaling"] == "weak"
  assert np.isfinite(d["value"]) and d["value"] > 0 and d["unit"] == "temporal edges/s"
  assert "2 ranks" in d["config"]["parallelism"]
