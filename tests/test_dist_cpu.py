"""world_size-2 (and 3) gloo tests of the query partition / row exchange used by the multi-GPU step."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, q):
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from orca_amd.dist import all_gather_chunks, all_gather_rows, all_reduce_sum, query_slice, sync_owned_rows
  ok = True
  # chunk exchange of the owner-computes step: rank r fills chunk r, everybody ends with all chunks
  for chunk in (1, 64, 1000):
    buf = torch.full((world * chunk + 5,), -1, dtype=torch.int32)
    buf[rank * chunk:(rank + 1) * chunk] = torch.arange(chunk, dtype=torch.int32) + 1000 * rank
    w = all_gather_chunks(buf, chunk, dist.group.WORLD, async_op=True)
    if w is not None:
      w.wait()
    exp = torch.cat([torch.arange(chunk, dtype=torch.int32) + 1000 * r for r in range(world)])
    ok = ok and torch.equal(buf[:world * chunk], exp) and bool((buf[world * chunk:] == -1).all())
  # owner rows -> coherent replicas: rows current only on their owner (id % world), garbage elsewhere
  gen = torch.Generator().manual_seed(3)
  truth = [torch.randn(37, 5, generator=gen), torch.randn(37, generator=gen)]
  mine = []
  for t in truth:
    own = (torch.arange(37) % world) == rank
    junk = torch.full_like(t, 123.0 + rank)       # stale rows of a non-owner: anything, inf and nan included
    junk.view(-1)[::3] = float("inf")
    junk.view(-1)[1::3] = float("nan")
    m = torch.where(own.view(-1, *([1] * (t.dim() - 1))), t, junk)
    mine.append(m)
  sync_owned_rows(mine, dist.group.WORLD, chunk_bytes=160)      # several slabs
  ok = ok and all(torch.equal(a, b) for a, b in zip(mine, truth))
  g = all_reduce_sum(torch.full((7,), float(rank + 1)), dist.group.WORLD)       # the gradient all-reduce of train_step
  ok = ok and torch.equal(g, torch.full((7,), world * (world + 1) / 2.0))
  for Q in (0, 1, 2, 5, 64, 4095, 4096):
    full = torch.arange(Q * 3, dtype=torch.float32).reshape(Q, 3) * 0.5
    lo, hi, per = query_slice(Q, rank, world)
    got = all_gather_rows(full[lo:hi].clone(), Q, per, dist.group.WORLD)
    ok = ok and got.shape == full.shape and torch.equal(got, full)
    ids = torch.arange(Q, dtype=torch.int64).reshape(Q, 1)
    got = all_gather_rows(ids[lo:hi].clone(), Q, per, dist.group.WORLD)
    ok = ok and torch.equal(got, ids)
  q.put((rank, bool(ok)))
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partition_and_gather_gloo(world):
  ctx = mp.get_context("spawn")
  q = ctx.Queue()
  port = _free_port()
  ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
  for p in ps:
    p.start()
  res = [q.get(timeout=120) for _ in range(world)]
  for p in ps:
    p.join(timeout=60)
    assert p.exitcode == 0
  assert sorted(res) == [(r, True) for r in range(world)]


def test_query_slice_covers_everything():
  from orca_amd.dist import query_slice
  for Q in (0, 1, 7, 4095, 4096, 12285):
    for world in (1, 2, 4, 8):
      cuts = [query_slice(Q, r, world) for r in range(world)]
      assert cuts[0][0] == 0 and cuts[-1][1] == Q
      for a, b in zip(cuts, cuts[1:]):
        assert a[1] == b[0]
      assert all(hi - lo <= per for lo, hi, per in cuts)
