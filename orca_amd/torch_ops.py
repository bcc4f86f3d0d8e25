"""PyTorch custom-op registration of the C-ABI entry points (torch.library, namespace `orca`), with autograd where
the reference's training loop differentiates (train.py:225-226 loss.backward()):

  orca::sample_recent / orca::sample_uniform   utils/util.py:134-151 (no gradient: integer / time outputs)
  orca::tgat_dense    model/temporal_attention.py:24-56 on the six dense tensors of the reference's forward
  orca::tgat_gather   the same layer with the gathers of modules/embedding_module.py:123,133,172-174 and the time
                      encodings of :126,132 fused in (rows fetched by index from the feature tables)
  orca::gru_cell      nn.GRUCell of modules/memory_updater.py:86 (get_updated_memory :56-78 applies it to the rows
                      of the pending ids a batch reads and keeps them in the graph)
  orca::message_store / orca::history_push / orca::history_pull     state writes and reads that are detached
                      upstream (tgn_model.py:172-191, history.py:32-40 run under no_grad)

The forward of every op is one call into liborca_hip.so (ops.py); the backward of the two attention ops is
orca_tgat_backward, of the GRU op orca_gru_backward.  Gradients flow exactly where autograd takes them upstream:
attention parameters, the source / neighbour rows that are layer outputs or updated-memory rows, and through the
latter into the GRU parameters; the time encoder is frozen (model/time_encoding.py:20-23), the memory table, the
stored messages, history rows and edge features are data.  There is no CPU implementation behind these ops: a CPU
tensor raises."""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from ._lib import ORCA_D as D

_lib = torch.library


def _i64(seed):
  """a 64-bit unsigned seed as the signed int the op schema carries"""
  seed &= 0xFFFFFFFFFFFFFFFF
  return seed - (1 << 64) if seed >= (1 << 63) else seed


class _Handle:
  """an ops.CSR stand-in built from the raw handle value an op receives"""

  def __init__(self, value):
    import ctypes
    self._h = ctypes.c_void_p(value)


# ---- sampler ------------------------------------------------------------------------------------------------
@_lib.custom_op("orca::sample_recent", mutates_args=())
def sample_recent(csr: int, nodes: Tensor, ts: Tensor, K: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
  return ops.sample_recent(_Handle(csr), nodes, ts, K)


@_lib.custom_op("orca::sample_uniform", mutates_args=())
def sample_uniform(csr: int, nodes: Tensor, ts: Tensor, K: int, seed: int, q0: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
  return ops.sample_uniform(_Handle(csr), nodes, ts, K, seed & 0xFFFFFFFFFFFFFFFF, q0=q0)


# ---- attention, dense boundary ------------------------------------------------------------------------------
@_lib.custom_op("orca::tgat_dense", mutates_args=())
def tgat_dense(params: List[Tensor], blob: Tensor, d_e: int, src: Tensor, src_time: Tensor, nbr: Tensor,
               nbr_time: Tensor, edge: Tensor, mask: Tensor, dropout_p: float, dropout_seed: int) -> Tuple[Tensor, Tensor, Tensor]:
  """-> (out [B,100], weights [B,K], mask_after): the reference un-masks slot 0 of all-padded rows in the caller's
  mask (temporal_attention.py:46-47); an op with an autograd formula has to be functional, so the mutated mask
  comes back as an output and the layer copies it into the caller's tensor."""
  B, K = mask.shape
  after = mask.clone()
  out, w = ops.tgat_forward(blob, d_e, B, K, want_weights=True, src_table=src, nbr_table=nbr, edge_table=edge,
                            mask=after.view(torch.uint8), src_time=src_time, nbr_time=nbr_time, dropout_p=dropout_p,
                            dropout_seed=dropout_seed)
  return out, w, after


def _tgat_dense_setup(ctx, inputs, output):
  params, blob, d_e, src, st, nbr, nt, edge, mask, dropout_p, dropout_seed = inputs
  ctx.d_e, ctx.drop = d_e, (dropout_p, dropout_seed)
  ctx.save_for_backward(src, st, nbr, nt, edge, mask, *params)     # the mask as it was BEFORE the forward


def _tgat_dense_backward(ctx, g_out, _g_w, _g_m):
  src, st, nbr, nt, edge, before, *params = ctx.saved_tensors
  B, K = before.shape
  grads, d_src, d_nbr, d_st, d_nt, _, _ = ops.tgat_backward(
      dict(zip(ops.TGAT_KEYS, params)), ctx.d_e, B, K, g_out.contiguous(), src_table=src, nbr_table=nbr, edge_table=edge,
      mask=before.view(torch.uint8), src_time=st, nbr_time=nt, dropout_p=ctx.drop[0], dropout_seed=ctx.drop[1])
  return [grads[k] for k in ops.TGAT_KEYS], None, None, d_src, d_st, d_nbr, d_nt, None, None, None, None


_lib.register_autograd("orca::tgat_dense", _tgat_dense_backward, setup_context=_tgat_dense_setup)


# ---- attention, fused gathers -------------------------------------------------------------------------------
@_lib.custom_op("orca::tgat_gather", mutates_args=())
def tgat_gather(params: List[Tensor], blob: Tensor, d_e: int, Q: int, K: int, src_table: Tensor,
                src_idx: Optional[Tensor], nbr_table: Tensor, nbr_idx: Optional[Tensor], nbr_ids: Tensor,
                edge_table: Tensor, edge_idx: Tensor, dt: Tensor, time_w: Tensor, time_b: Tensor,
                overlay: Optional[Tensor], flags: Optional[Tensor], upd_rows: Optional[Tensor],
                upd_pos: Optional[Tensor], q_folded: bool, dropout_p: float, dropout_seed: int) -> Tensor:
  """upd_rows / upd_pos carry no data the forward reads (it reads `overlay`, the table the GRU op filled): they tie
  the overlay rows into the autograd graph -- upd_rows[upd_pos[id]] is overlay[id] for the flagged ids."""
  kw = dict(src_table=src_table, src_idx=src_idx, nbr_table=nbr_table, nbr_idx=nbr_idx, nbr_ids=nbr_ids,
            edge_table=edge_table, edge_idx=edge_idx, dt=dt, time_w=time_w, time_b=time_b, q_folded=q_folded,
            dropout_p=dropout_p, dropout_seed=dropout_seed)
  if flags is not None:
    kw.update(src_overlay=overlay, src_flags=flags, nbr_overlay=overlay, nbr_flags=flags)
  return ops.tgat_forward(blob, d_e, Q, K, **kw)[0]


def _tgat_gather_setup(ctx, inputs, output):
  (params, blob, d_e, Q, K, src_table, src_idx, nbr_table, nbr_idx, nbr_ids, edge_table, edge_idx, dt, time_w, time_b,
   overlay, flags, upd_rows, upd_pos, q_folded, dropout_p, dropout_seed) = inputs
  ctx.d_e, ctx.Q, ctx.K = d_e, Q, K
  ctx.drop = (dropout_p, dropout_seed)
  ctx.src_rows, ctx.nbr_rows = src_table.shape[0], nbr_table.shape[0]
  ctx.upd_n = upd_rows.shape[0] if upd_rows is not None else 0
  ctx.opt = [t is not None for t in (src_idx, nbr_idx, overlay, flags, upd_pos)]
  keep = [t for t in (src_idx, nbr_idx, overlay, flags, upd_pos) if t is not None]
  ctx.save_for_backward(src_table, nbr_table, nbr_ids, edge_table, edge_idx, dt, time_w, time_b, *keep, *params)
  ctx.n_keep = len(keep)
  ctx.want_src = src_idx is None and src_table.requires_grad
  ctx.want_nbr = nbr_idx is None and nbr_table.requires_grad
  ctx.want_upd = upd_rows is not None and upd_rows.requires_grad and upd_pos is not None and flags is not None


def _tgat_gather_backward(ctx, g_out):
  src_table, nbr_table, nbr_ids, edge_table, edge_idx, dt, time_w, time_b, *rest = ctx.saved_tensors
  keep, params = rest[:ctx.n_keep], rest[ctx.n_keep:]
  it = iter(keep)
  src_idx, nbr_idx, overlay, flags, upd_pos = (next(it) if has else None for has in ctx.opt)
  Q, K = ctx.Q, ctx.K
  kw = dict(src_table=src_table, src_idx=src_idx, nbr_table=nbr_table, nbr_idx=nbr_idx, nbr_ids=nbr_ids,
            edge_table=edge_table, edge_idx=edge_idx, dt=dt, time_w=time_w, time_b=time_b, dropout_p=ctx.drop[0],
            dropout_seed=ctx.drop[1])
  if flags is not None:
    kw.update(src_overlay=overlay, src_flags=flags, nbr_overlay=overlay, nbr_flags=flags)
  grads, d_src, d_nbr, _, _, _, _ = ops.tgat_backward(dict(zip(ops.TGAT_KEYS, params)), ctx.d_e, Q, K, g_out.contiguous(), **kw)
  g_src_table = g_nbr_table = g_upd = None
  if ctx.want_src:                           # dense source rows: row q is query q
    g_src_table = d_src if ctx.src_rows == Q else torch.cat([d_src, d_src.new_zeros(ctx.src_rows - Q, D)])
  if ctx.want_nbr:                           # dense neighbour rows: row q*K+j is slot j of query q
    g_nbr_table = d_nbr if ctx.nbr_rows == Q * K else torch.cat([d_nbr, d_nbr.new_zeros(ctx.nbr_rows - Q * K, D)])
  if ctx.want_upd:                           # rows gathered from the updated-memory overlay
    g_upd = torch.zeros(ctx.upd_n, D, dtype=torch.float32, device=g_out.device)
    if src_idx is not None and ctx.upd_n:
      ops.scatter_add_flagged(d_src, src_idx, flags, upd_pos, g_upd)
    if nbr_idx is not None and ctx.upd_n:
      ops.scatter_add_flagged(d_nbr, nbr_idx.reshape(-1), flags, upd_pos, g_upd)
  return ([grads[k] for k in ops.TGAT_KEYS], None, None, None, None, g_src_table, None, g_nbr_table, None, None, None,
          None, None, None, None, None, None, g_upd, None, None, None, None)


_lib.register_autograd("orca::tgat_gather", _tgat_gather_backward, setup_context=_tgat_gather_setup)


# ---- GRU cell -------------------------------------------------------------------------------------------------
@_lib.custom_op("orca::gru_cell", mutates_args=())
def gru_cell(weight_ih: Tensor, weight_hh: Tensor, bias_ih: Tensor, bias_hh: Tensor, blob: Tensor, x: Tensor,
             h: Tensor) -> Tensor:
  """nn.GRUCell(x [n, msg], h [n, 100]) -> [n, 100] (modules/memory_updater.py:86) on the MFMA GRU kernel; blob =
  ops.gru_pack of the four parameters."""
  n = x.shape[0]
  out = torch.empty((n, D), dtype=torch.float32, device=x.device)
  if n:
    ops.gru_update(blob, x.shape[1], None, n, None, x, h, out)
  return out


def _gru_cell_setup(ctx, inputs, output):
  w_ih, w_hh, b_ih, b_hh, blob, x, h = inputs
  ctx.want_inputs = x.requires_grad or h.requires_grad
  ctx.save_for_backward(w_ih, w_hh, b_ih, b_hh, x, h)


def _gru_cell_backward(ctx, g_out):
  w_ih, w_hh, b_ih, b_hh, x, h = ctx.saved_tensors
  if x.shape[0] == 0:
    return (*[torch.zeros_like(p) for p in (w_ih, w_hh, b_ih, b_hh)], None, None, None)
  g = ops.gru_backward(w_ih, w_hh, b_ih, b_hh, None, x, h, g_out.contiguous(), want_inputs=ctx.want_inputs)
  return (*g[:4], None, g[4], g[5])


_lib.register_autograd("orca::gru_cell", _gru_cell_backward, setup_context=_gru_cell_setup)


# ---- detached state writes / reads ------------------------------------------------------------------------------
@_lib.custom_op("orca::message_store", mutates_args=("slot", "messages", "msg_ts", "pending"))
def message_store(src: Tensor, dst: Tensor, eidx: Tensor, t: Tensor, memory: Tensor, last_update: Tensor,
                  edge_feats: Tensor, time_w: Tensor, time_b: Tensor, slot: Tensor, epoch: int, messages: Tensor,
                  msg_ts: Tensor, pending: Tensor) -> None:
  ops.message_store(src, dst, eidx, t, memory, last_update, edge_feats, time_w, time_b, slot, epoch, messages, msg_ts,
                    pending)


@_lib.custom_op("orca::history_push", mutates_args=("emb", "slot"))
def history_push(emb: Tensor, x: Tensor, ids: Tensor, slot: Tensor, epoch: int) -> None:
  ops.history_push(emb, x, ids, slot, epoch)


@_lib.custom_op("orca::history_pull", mutates_args=())
def history_pull(emb: Tensor, ids: Tensor) -> Tensor:
  return ops.gather_rows(emb, ids)


# ---- history with gradients (train.py:43 --gradient; history.py:42-46 push_and_pull) ----------------------------
@_lib.custom_op("orca::history_pull_pushed", mutates_args=())
def history_pull_pushed(emb: Tensor, x: Tensor, pull_ids: Tensor, pushed: Tensor, winner_pos: Tensor) -> Tensor:
  """out[p] = emb[pull_ids[p]] AFTER x has been pushed into emb (the caller did that, under no_grad), as a node of
  the autograd graph: the rows that were pushed in this batch are x's rows (pushed[id] != 0, winner_pos[id] = the
  position in x that won the push), and their gradient goes back to x; the other rows are history: data."""
  return ops.gather_rows(emb, pull_ids)


def _hpp_setup(ctx, inputs, output):
  emb, x, pull_ids, pushed, winner_pos = inputs
  ctx.n = x.shape[0]
  ctx.save_for_backward(pull_ids, pushed, winner_pos)


def _hpp_backward(ctx, g_out):
  pull_ids, pushed, winner_pos = ctx.saved_tensors
  g_x = torch.zeros(ctx.n, D, dtype=torch.float32, device=g_out.device)
  if pull_ids.numel() and ctx.n:
    ops.scatter_add_flagged(g_out.contiguous(), pull_ids, pushed, winner_pos, g_x)
  return None, g_x, None, None, None


_lib.register_autograd("orca::history_pull_pushed", _hpp_backward, setup_context=_hpp_setup)


OPS = ("sample_recent", "sample_uniform", "tgat_dense", "tgat_gather", "gru_cell", "message_store", "history_push",
       "history_pull", "history_pull_pushed")
