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
  from orca_amd.dist import all_gather_rows, all_reduce_sum, query_slice
  ok = True
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
