"""N-rank == 1-rank equality of the partitioned step: two ranks share the one GPU of
the test box and exchange rows over gloo (RCCL refuses two ranks on one device; bench.py uses "nccl" on
the 8-GPU node).  Results must be bit-identical to the single-rank engine on the same batches."""
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


def _worker(rank, world, port, q):
  try:
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from orca_amd import ops, synth
    from orca_amd.engine import OrcaEngine
    dev = "cuda:0"
    N, E, bs, K, d_e = 3000, 60000, 301, 10, 172
    src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev, seed=5)
    ef = synth.edge_features(E, d_e, dev)
    sd = synth.tgn_state_dict(d_e, 2, seed=2)
    csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    ok = True
    for reuse in (True, False):
      multi = OrcaEngine(sd, ef, N + 1, 2, K, reuse=reuse, device=dev, group=dist.group.WORLD)
      single = OrcaEngine(sd, ef, N + 1, 2, K, reuse=reuse, device=dev, group=None)
      multi.set_neighbor_finder(csr)
      single.set_neighbor_finder(csr)
      for b in range(7):
        s, e = 50000 + b * bs, 50000 + (b + 1) * bs
        neg = torch.randint(1, N + 1, (bs,), generator=g, device=dev)
        train = b != 3
        if b == 5:      # a restored backup drops the overlay-complete state: the sliced step then builds per-batch
          multi.restore(multi.backup())      # work-lists and exchanges the roots' updated rows
          single.restore(single.backup())
        sharded = reuse and b in (2, 6)     # outputs left on the rank that computed them (bench.py, N > 1)
        a = multi.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], train,
                                              gather_output=not sharded)
        c = single.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], train)
        if sharded:
          from orca_amd.dist import query_slice
          lo, hi, _ = query_slice(3 * bs, rank, world)
          c = c[lo:hi]
        ok = ok and torch.equal(a, c) and torch.equal(multi.memory, single.memory)
        ok = ok and torch.equal(multi.messages, single.messages) and torch.equal(multi.pending, single.pending)
        ok = ok and torch.equal(multi.last_update, single.last_update) and torch.equal(multi.msg_ts, single.msg_ts)
        if reuse:
          ok = ok and torch.equal(multi.hist[0], single.hist[0])
    # Path C (budgeted reuse with a cache plan per batch): global set algebra on every rank, attention partitioned
    multi = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=40, device=dev, group=dist.group.WORLD)
    single = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=40, device=dev, group=None)
    multi.set_neighbor_finder(csr)
    single.set_neighbor_finder(csr)
    for b in range(3):
      s, e = 50000 + b * bs, 50000 + (b + 1) * bs
      neg = torch.randint(1, N + 1, (bs,), generator=g, device=dev)
      plan = torch.randint(1, N + 1, (40,), generator=g, device=dev)
      a = multi.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True, plan)
      c = single.compute_temporal_embeddings(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e], True, plan)
      ok = ok and torch.equal(a, c) and torch.equal(multi.memory, single.memory)
      ok = ok and torch.equal(multi.hist[0], single.hist[0]) and torch.equal(multi.cache_flag[0], single.cache_flag[0])
    # training step: every rank back-propagates its slice, parameter gradients are all-reduced; the sums associate
    # differently from the single-rank step, so gradients are compared at 1e-5 and the state bit for bit
    multi = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, device=dev, group=dist.group.WORLD)
    single = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, device=dev, group=None)
    multi.set_neighbor_finder(csr)
    single.set_neighbor_finder(csr)
    msg = ""
    for b in range(4):
      s, e = 50000 + b * bs, 50000 + (b + 1) * bs
      neg = torch.randint(1, N + 1, (bs,), generator=g, device=dev)
      la, ga = multi.train_step(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e])
      lc, gc = single.train_step(src[s:e], dst[s:e], neg, ts[s:e], eidx[s:e])
      ok = ok and abs(float(la) - float(lc)) <= 1e-6 and set(ga) == set(gc)
      for k in gc:
        scale = max(float(gc[k].abs().max()), 1e-12)
        err = float((ga[k] - gc[k]).abs().max()) / scale
        if err > 1e-5:
          ok, msg = False, f"train_step gradient {k} differs between 2 ranks and 1 rank: rel {err:.2e}"
      ok = ok and torch.equal(multi.memory, single.memory) and torch.equal(multi.messages, single.messages)
      ok = ok and torch.equal(multi.hist[0], single.hist[0])
    q.put((rank, bool(ok), msg))
    dist.barrier()
    dist.destroy_process_group()
  except Exception as ex:   # surface the failure in the parent
    import traceback
    q.put((rank, False, traceback.format_exc()))


def test_two_ranks_equal_one_rank():
  world = 2
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in ps:
    p.start()
  res = [q.get(timeout=240) for _ in range(world)]
  for p in ps:
    p.join(timeout=60)
  for r, ok, msg in res:
    assert ok, f"rank {r}: {msg or 'results differ from the single-rank engine'}"
