"""Query partitioning and row exchange for the multi-GPU step (the reference is single
device, so this is defined here).  One process per GPU, torch.distributed over RCCL ("nccl") on the
node; the same code runs over gloo for the CPU / single-GPU tests.

State ownership: owner(id) = id % world (owner()).  In owner-computes steps a rank keeps the memory / message rows of
the ids it owns current and sends the rows other ranks read through fixed-size chunk buffers (all_gather_chunks);
sync_owned_rows makes full replicas coherent again.

Partition: the Q root (node, t) queries of a batch are cut into `world` contiguous slices of
per = ceil(Q / world) rows (the last slices may be short or empty).  Exchange: every rank contributes
its [n_local, W] rows padded to per rows; one all_gather_into_tensor returns the [world*per, W] buffer
whose first Q rows are the batch's rows in global query order -- so any later "last occurrence wins"
resolution sees the same order as a single-rank run."""
import torch
import torch.distributed as dist


def query_slice(Q, rank, world):
  """(lo, hi, per): rows [lo, hi) of Q belong to `rank`; per = rows per rank before clipping."""
  per = (Q + world - 1) // world if world > 0 else Q
  lo = min(rank * per, Q)
  return lo, min(lo + per, Q), per


def interaction_slice(bs, rank, world):
  """(lo, hi, per): interactions [lo, hi) of a batch of bs belong to `rank` in the owner-computes step; the rank
  embeds their src, dst and neg roots (local query order [src | dst | neg])."""
  return query_slice(bs, rank, world)


def interaction_root_index(bs, world, device=None):
  """index[i] for every root i = k*bs + e of [src;dst;neg]: its row in the rank-major buffer the owner-computes
  step all-gathers (3*per rows per rank)."""
  per = (bs + world - 1) // world if world > 0 else bs
  e = torch.arange(bs, device=device)
  r = e // max(per, 1)
  n_r = torch.clamp(bs - r * per, max=per)
  base = 3 * per * r + (e - r * per)
  return torch.cat([base + k * n_r for k in range(3)])


def all_gather_rows(x, Q, per, group=None):
  """x: this rank's [n_local, W] slice -> [Q, W] on every rank."""
  world = dist.get_world_size(group) if group is not None or dist.is_initialized() else 1
  if world == 1:
    return x
  W = x.shape[1]
  buf = torch.zeros(per, W, dtype=x.dtype, device=x.device)
  buf[:x.shape[0]] = x
  full = torch.empty(world * per, W, dtype=x.dtype, device=x.device)
  if x.is_cuda and dist.get_backend(group) == "gloo":
    # gloo has no device all_gather: stage through the host (tests only; RCCL is the product path)
    hb, hf = buf.cpu(), torch.empty(world * per, W, dtype=x.dtype)
    dist.all_gather_into_tensor(hf, hb, group=group)
    full.copy_(hf)
  else:
    dist.all_gather_into_tensor(full, buf, group=group)
  return full[:Q]


def all_reduce_sum(x, group=None):
  """In-place sum over the ranks of a flat device tensor (gradient all-reduce); staged through
  the host under gloo (the CPU rehearsal backend), RCCL otherwise."""
  if x.is_cuda and dist.get_backend(group) == "gloo":
    h = x.cpu()
    dist.all_reduce(h, group=group)
    x.copy_(h)
  else:
    dist.all_reduce(x, group=group)
  return x


def owner(ids, world):
  """rank that owns the state rows of each node id (rows sharded by node-id hash)"""
  return ids % world


def all_gather_chunks(buf, chunk, group=None, async_op=False):
  """In-place all-gather of the first world x chunk elements of the flat buffer buf: rank r contributes
  buf[r*chunk:(r+1)*chunk] and receives every other chunk.  Returns the work handle when async_op (None when the
  exchange already completed: gloo with device buffers stages through the host)."""
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  full = buf[:world * chunk]
  mine = full[rank * chunk:(rank + 1) * chunk]
  if buf.is_cuda and dist.get_backend(group) == "gloo":
    hf = torch.empty(full.shape, dtype=full.dtype)
    dist.all_gather_into_tensor(hf, mine.cpu(), group=group)
    full.copy_(hf)
    return None
  work = dist.all_gather_into_tensor(full, mine, group=group, async_op=async_op)
  return work if async_op else None


def all_gather_into(full, mine, group=None, async_op=False):
  """Out-of-place all-gather: rank r's `mine` becomes full[r*len(mine):(r+1)*len(mine)] on every rank.  Returns the work
  handle when async_op (None when the exchange already completed: gloo with device buffers stages through the host)."""
  if full.is_cuda and dist.get_backend(group) == "gloo":
    hf = torch.empty(full.shape, dtype=full.dtype)
    dist.all_gather_into_tensor(hf, mine.cpu(), group=group)
    full.copy_(hf)
    return None
  work = dist.all_gather_into_tensor(full, mine, group=group, async_op=async_op)
  return work if async_op else None


def sync_owned_rows(tables, group=None, chunk_bytes=64 << 20):
  """Every table [N, ...] of `tables` holds current rows only where owner(row) == this rank; afterwards every rank
  holds every owner's rows.  Masked all-reduce in slabs of about chunk_bytes: each row has exactly one non-zero
  contributor, so the sum is that contributor's row bit for bit (up to the sign of zero), whatever the other ranks' stale rows hold."""
  world, rank = dist.get_world_size(group), dist.get_rank(group)
  for t in tables:
    n = t.shape[0]
    row = max(1, t[0].numel() * t.element_size()) if n else 1
    step = max(1, chunk_bytes // row)
    for r0 in range(0, n, step):
      r1 = min(n, r0 + step)
      own = (torch.arange(r0, r1, device=t.device) % world) == rank
      # select, not multiply: a stale non-owner row may hold inf / nan, and 0 * inf would poison the owner's row in the sum
      mask = own if t.dim() == 1 else own.view(-1, *([1] * (t.dim() - 1)))
      part = torch.where(mask, t[r0:r1], torch.zeros_like(t[r0:r1]))
      all_reduce_sum(part, group)
      t[r0:r1] = part
