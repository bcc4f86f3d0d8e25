"""Query partitioning and row exchange for the multi-GPU step (the reference is single
device, so this is defined here).  One process per GPU, torch.distributed over RCCL ("nccl") on the
node; the same code runs over gloo for the CPU / single-GPU tests.

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
