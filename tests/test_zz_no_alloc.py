"""The entry point that allocates nothing (include/orca_hip.h: orca_segment_last_static) against the existing
orca_segment_last, through the C ABI: bit-exact winner flags on two calls in a row, the second with a smaller input
(the static table keeps the first call's contents)."""
import numpy as np
import pytest
import torch

from orca_amd import _lib
from orca_amd._lib import check, ptr, stream

pytestmark = pytest.mark.gpu


def _both(L, ids, n_nodes, slot, epoch):
  dev = ids.device
  M = ids.numel()
  w_ref = torch.full((max(M, 1),), 7, dtype=torch.uint8, device=dev)
  w_new = torch.full((max(M, 1),), 9, dtype=torch.uint8, device=dev)
  check(L.orca_segment_last(ptr(ids), M, ptr(slot), epoch, n_nodes, ptr(w_ref), stream()), "orca_segment_last")
  check(L.orca_segment_last_static(ptr(ids), M, n_nodes, ptr(w_new), stream()), "orca_segment_last_static")
  torch.cuda.synchronize()
  return w_ref[:M].cpu().numpy(), w_new[:M].cpu().numpy()


def test_static_election_matches_segment_last():
  L = _lib.load()
  dev = torch.device("cuda:0")
  g = torch.Generator(device="cpu").manual_seed(11)
  n_nodes = 200_000
  slot = torch.zeros(n_nodes, dtype=torch.int64, device=dev)
  # power-law-ish ids with many repeats, including the padding id 0 and out-of-range ids
  big = (torch.rand(50_000, generator=g) ** 4 * (n_nodes - 1)).to(torch.int64)
  big[::997] = -1
  big[5::1013] = n_nodes + 3
  small = torch.randint(0, 300, (1_000,), generator=g, dtype=torch.int64)     # ids the first call touched too
  epoch = 1
  for ids in (big, small):
    ids = ids.to(dev)
    ref, new = _both(L, ids, n_nodes, slot, epoch)
    epoch += ids.numel() + 1
    assert np.array_equal(ref, new)
    # independent check of the reference semantics: winner = last occurrence of each valid id
    h = ids.cpu().numpy()
    valid = (h >= 0) & (h < n_nodes)
    last = {}
    for p, v in enumerate(h):
      if valid[p]:
        last[v] = p
    exp = np.zeros(len(h), np.uint8)
    exp[list(last.values())] = 1
    assert np.array_equal(new, exp)


def test_static_election_replays_from_a_graph():
  """The call keeps no state on the host and leaves its table zeroed, so a captured call can be replayed on new
  contents of the same buffers -- also when an id's last occurrence moves to an EARLIER position than in the previous
  replay, and with eager calls in between (a host-side epoch froze in the graph and broke exactly that)."""
  L = _lib.load()
  dev = torch.device("cuda:0")
  n_nodes, M = 5_000, 4_096
  g = torch.Generator(device="cpu").manual_seed(5)
  ids = torch.zeros(M, dtype=torch.int64, device=dev)
  win = torch.zeros(M, dtype=torch.uint8, device=dev)
  other = torch.randint(0, n_nodes, (777,), generator=g, dtype=torch.int64).to(dev)
  w_other = torch.zeros(777, dtype=torch.uint8, device=dev)
  check(L.orca_segment_last_static(ptr(ids), M, n_nodes, ptr(win), stream()), "warm-up (takes the table's address)")
  torch.cuda.synchronize()
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.graph(graph):
    check(L.orca_segment_last_static(ptr(ids), M, n_nodes, ptr(win), stream()), "captured call")

  def expected(h):
    last = {}
    for p, v in enumerate(h):
      if 0 <= v < n_nodes:
        last[v] = p
    e = np.zeros(len(h), np.uint8)
    e[list(last.values())] = 1
    return e

  base = torch.randint(0, 600, (M,), generator=g, dtype=torch.int64)
  for r in range(4):
    h = torch.flip(base, [0]) if r % 2 else base.clone()      # every id's last occurrence moves between replays
    if r == 2:
      h[::3] = -1
    ids.copy_(h.to(dev))
    win.fill_(5)
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(win.cpu().numpy(), expected(h.numpy())), f"replay {r}"
    check(L.orca_segment_last_static(ptr(other), 777, n_nodes, ptr(w_other), stream()), "eager call between replays")
    torch.cuda.synchronize()
    assert np.array_equal(w_other.cpu().numpy(), expected(other.cpu().numpy()))


def test_static_election_rejects_oversize():
  L = _lib.load()
  dev = torch.device("cuda:0")
  ids = torch.zeros(4, dtype=torch.int64, device=dev)
  w = torch.zeros(4, dtype=torch.uint8, device=dev)
  rc = L.orca_segment_last_static(ptr(ids), 4, (1 << 20) + 1, ptr(w), stream())
  assert rc == -1 and b"static table" in L.orca_last_error()
