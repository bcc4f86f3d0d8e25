"""Torch-tensor front end of the C ABI (include/orca_hip.h).  torch is used for device memory and
streams only; every function below is one call into liborca_hip.so on the current HIP stream.
There is no CPU fallback: tensors must live on a HIP device and the library must be built."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import ORCA_D, check, ptr, stream

D = ORCA_D


def _dev(t, dtype, what):
  if not (isinstance(t, torch.Tensor) and t.is_cuda):
    raise _lib.OrcaError(f"{what}: expected a tensor on the HIP device (no CPU fallback)")
  if t.dtype != dtype:
    raise TypeError(f"{what}: expected {dtype}, got {t.dtype}")
  return t.contiguous()


# ---------------------------------------------------------------------------------------------
# time-sorted CSR + sampler (utils/util.py:91-151)
# ---------------------------------------------------------------------------------------------
class CSR:
  """Opaque handle of the time-sorted CSR in HBM (orca_csr_t)."""

  def __init__(self, handle, device):
    self._h = handle
    self.device = torch.device(device)
    n_rows, nnz = _lib.i64(), _lib.i64()
    check(_lib.load().orca_csr_arrays(self._h, ctypes.byref(n_rows), ctypes.byref(nnz), None, None, None, None),
          "orca_csr_arrays")
    self.n_rows, self.nnz = n_rows.value, nnz.value

  @staticmethod
  def _index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()

  @classmethod
  def from_edges_host(cls, src, dst, eidx, ts, n_rows, device):
    """get_neighbor_finder (utils/util.py:91-105) from host edge arrays."""
    src = np.ascontiguousarray(src, dtype=np.int64)
    dst = np.ascontiguousarray(dst, dtype=np.int64)
    eidx = np.ascontiguousarray(eidx, dtype=np.int64)
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    h = _lib.vp()
    check(_lib.load().orca_csr_create_host(len(src), ptr(src), ptr(dst), ptr(eidx), ptr(ts), int(n_rows),
                                           cls._index(device), ctypes.byref(h)), "orca_csr_create_host")
    return cls(h, device)

  @classmethod
  def from_edges_device(cls, src, dst, eidx, ts, n_rows):
    """Same, from edge arrays already in HBM (radix-sort build on the device)."""
    src, dst, eidx = (_dev(x, torch.int64, "CSR edges") for x in (src, dst, eidx))
    ts = _dev(ts, torch.float64, "CSR timestamps")
    h = _lib.vp()
    with torch.cuda.device(src.device):
      check(_lib.load().orca_csr_create_device(src.numel(), ptr(src), ptr(dst), ptr(eidx), ptr(ts), int(n_rows),
                                               stream(), ctypes.byref(h)), "orca_csr_create_device")
    return cls(h, src.device)

  @classmethod
  def from_arrays_host(cls, indptr, nbr, eidx, ts, device):
    """NeighborFinder(node_to_neighbors, ...) (utils/util.py:123-126) from flattened ragged lists."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    nbr = np.ascontiguousarray(nbr, dtype=np.int32)
    eidx = np.ascontiguousarray(eidx, dtype=np.int32)
    ts = np.ascontiguousarray(ts, dtype=np.float64)
    h = _lib.vp()
    check(_lib.load().orca_csr_from_arrays_host(len(indptr) - 1, ptr(indptr), ptr(nbr), ptr(eidx), ptr(ts),
                                                cls._index(device), ctypes.byref(h)), "orca_csr_from_arrays_host")
    return cls(h, device)

  def arrays_host(self):
    """(indptr, nbr, eidx, ts) copied back to NumPy (tests / export / the cache planner)."""
    indptr = np.empty(self.n_rows + 1, np.int64)
    nbr = np.empty(self.nnz, np.int32)
    eidx = np.empty(self.nnz, np.int32)
    ts = np.empty(self.nnz, np.float64)
    with torch.cuda.device(self.device):
      check(_lib.load().orca_csr_copy_host(self._h, ptr(indptr), ptr(nbr), ptr(eidx), ptr(ts)), "orca_csr_copy_host")
    return indptr, nbr, eidx, ts

  def rows_host(self, nodes):
    """(indptr, nbr, eidx, ts) of a CSR with the same number of rows that holds the adjacency lists of `nodes` only
    (every other row empty): what a host-side check of a few queries needs from a graph too large to copy whole."""
    L = _lib.load()
    full = np.empty(self.n_rows + 1, np.int64)
    with torch.cuda.device(self.device):
      check(L.orca_csr_copy_host(self._h, ptr(full), None, None, None), "orca_csr_copy_host")
      nodes = np.unique(np.asarray(nodes, dtype=np.int64))
      deg = np.zeros(self.n_rows, np.int64)
      deg[nodes] = full[nodes + 1] - full[nodes]
      indptr = np.zeros(self.n_rows + 1, np.int64)
      np.cumsum(deg, out=indptr[1:])
      nnz = int(indptr[-1])
      nbr, eidx, ts = np.empty(nnz, np.int32), np.empty(nnz, np.int32), np.empty(nnz, np.float64)
      for v in nodes:
        a, b = int(indptr[v]), int(indptr[v + 1])
        if b > a:
          check(L.orca_csr_copy_range_host(self._h, int(full[v]), int(full[v + 1]), ptr(nbr[a:b]), ptr(eidx[a:b]),
                                           ptr(ts[a:b])), "orca_csr_copy_range_host")
    return indptr, nbr, eidx, ts

  def close(self):
    if self._h is not None and self._h.value:
      _lib.load().orca_csr_destroy(self._h)
    self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


def sample_recent(csr, nodes, ts, K, with_dt=True, err_flag=None):
  """orca_sample_recent: nodes int64 [Q], ts float64 [Q] on the device ->
  (nbr int32 [Q,K], eidx int32 [Q,K], edge_times float32 [Q,K], dt float32 [Q,K] or None)."""
  nodes = _dev(nodes, torch.int64, "sample_recent nodes")
  ts = _dev(ts, torch.float64, "sample_recent ts")
  Q = nodes.numel()
  dev = nodes.device
  nbr = torch.empty((Q, K), dtype=torch.int32, device=dev)
  eid = torch.empty((Q, K), dtype=torch.int32, device=dev)
  et = torch.empty((Q, K), dtype=torch.float32, device=dev)
  dt = torch.empty((Q, K), dtype=torch.float32, device=dev) if with_dt else None
  check(_lib.load().orca_sample_recent(csr._h, ptr(nodes), ptr(ts), Q, K, ptr(nbr), ptr(eid), ptr(et), ptr(dt),
                                       ptr(err_flag), stream()), "orca_sample_recent")
  return nbr, eid, et, dt


def sample_uniform(csr, nodes, ts, K, seed, err_flag=None, q0=0):
  """orca_sample_uniform (not in the reference): K seeded draws with replacement among the entries before t, in
  time order; q0 = index of nodes[0] in the caller's query numbering (the RNG key)."""
  nodes = _dev(nodes, torch.int64, "sample_uniform nodes")
  ts = _dev(ts, torch.float64, "sample_uniform ts")
  Q = nodes.numel()
  dev = nodes.device
  nbr = torch.empty((Q, K), dtype=torch.int32, device=dev)
  eid = torch.empty((Q, K), dtype=torch.int32, device=dev)
  et = torch.empty((Q, K), dtype=torch.float32, device=dev)
  dt = torch.empty((Q, K), dtype=torch.float32, device=dev)
  check(_lib.load().orca_sample_uniform(csr._h, ptr(nodes), ptr(ts), Q, K, int(seed), int(q0), ptr(nbr), ptr(eid),
                                        ptr(et), ptr(dt), ptr(err_flag), stream()), "orca_sample_uniform")
  return nbr, eid, et, dt


def find_before(csr, nodes, ts):
  nodes = _dev(nodes, torch.int64, "find_before nodes")
  ts = _dev(ts, torch.float64, "find_before ts")
  out = torch.empty(nodes.numel(), dtype=torch.int64, device=nodes.device)
  check(_lib.load().orca_find_before(csr._h, ptr(nodes), ptr(ts), nodes.numel(), ptr(out), stream()),
        "orca_find_before")
  return out


# ---------------------------------------------------------------------------------------------
def time_encode(t, w, b):
  """model/time_encoding.py:23-28: t float32 [...] -> [..., 100]."""
  t = _dev(t, torch.float32, "time_encode t")
  out = torch.empty(tuple(t.shape) + (D,), dtype=torch.float32, device=t.device)
  check(_lib.load().orca_time_encode(ptr(t), t.numel(), ptr(w), ptr(b), ptr(out), stream()), "orca_time_encode")
  return out


# ---------------------------------------------------------------------------------------------
# TGAT attention layer (model/temporal_attention.py)
# ---------------------------------------------------------------------------------------------
TGAT_KEYS = ("multi_head_target.q_proj_weight", "multi_head_target.k_proj_weight",
             "multi_head_target.v_proj_weight", "multi_head_target.in_proj_bias",
             "multi_head_target.out_proj.weight", "multi_head_target.out_proj.bias",
             "merger.fc1.weight", "merger.fc1.bias", "merger.fc2.weight", "merger.fc2.bias",
             "norm.weight", "norm.bias")


def tgat_pack(sd, d_e, device, blob=None, time_b=None):
  """Packs one TemporalAttentionLayer state_dict (keys TGAT_KEYS) into the kernel's blob.  With time_b (the
  time encoder's bias) the blob is also completed for gather mode (orca_tgat_pack_gather)."""
  n = _lib.load().orca_tgat_blob_floats(int(d_e))
  if n < 0:
    raise ValueError(f"tgat_pack: unsupported edge width {d_e}")
  if blob is None:
    blob = torch.empty(n, dtype=torch.float32, device=device)
  ts = [_f32(sd[k], device) for k in TGAT_KEYS]
  k_w = ts[1]
  if k_w.shape != (2 * D, 2 * D + d_e):
    raise ValueError(f"tgat_pack: k_proj_weight has shape {tuple(k_w.shape)}, expected {(2 * D, 2 * D + d_e)}")
  check(_lib.load().orca_tgat_pack(int(d_e), *(ptr(t) for t in ts), ptr(blob), stream()), "orca_tgat_pack")
  if time_b is not None:
    tb = _f32(time_b, device)
    if tb.numel() != D:
      raise ValueError(f"tgat_pack: time_b has {tb.numel()} values, expected {D}")
    check(_lib.load().orca_tgat_pack_gather(int(d_e), ptr(ts[0]), ptr(k_w), ptr(ts[3]), ptr(tb), ptr(blob), stream()),
          "orca_tgat_pack_gather")
  return blob


def tgat_pack_layers(sds, d_e, device, blobs=None, time_b=None):
  """tgat_pack for several layers at once (orca_tgat_pack_layers): the folds of all layers share launches.  sds: one
  TGAT_KEYS state_dict per layer; blobs: their blobs from an earlier pack (re-packed in place) or None."""
  lib = _lib.load()
  n = lib.orca_tgat_blob_floats(int(d_e))
  if n < 0:
    raise ValueError(f"tgat_pack_layers: unsupported edge width {d_e}")
  L = len(sds)
  before = blobs is not None and all(b is not None for b in blobs)
  if not before:
    blobs = [torch.empty(n, dtype=torch.float32, device=device) for _ in range(L)]
  P = (_lib.TgatParams * L)()
  keep = []
  for i, sd in enumerate(sds):
    ts = [_f32(sd[k], device) for k in TGAT_KEYS]
    if ts[1].shape != (2 * D, 2 * D + d_e):
      raise ValueError(f"tgat_pack_layers: k_proj_weight has shape {tuple(ts[1].shape)}, expected {(2 * D, 2 * D + d_e)}")
    keep.append(ts)
    for (name, _), t in zip(_lib.TgatParams._fields_, ts):
      setattr(P[i], name, ptr(t))
  tb = None
  if time_b is not None:
    tb = _f32(time_b, device)
    if tb.numel() != D:
      raise ValueError(f"tgat_pack_layers: time_b has {tb.numel()} values, expected {D}")
  B = (ctypes.c_void_p * L)(*[ptr(b) for b in blobs])
  check(lib.orca_tgat_pack_layers(L, int(d_e), ctypes.cast(P, ctypes.c_void_p), ptr(tb), ctypes.cast(B, ctypes.c_void_p),
                                  1 if before else 0, stream()), "orca_tgat_pack_layers")
  return list(blobs)


def tgat_forward(blob, d_e, Q, K, out=None, want_weights=False, save=None, **kw):
  """orca_tgat_forward.  kw: the fields of orca_tgat_inputs (tensors or None; q_folded: bool; dropout_p, dropout_seed,
  dropout_q0: attention-weight dropout, off by default; waves_per_query: the launch shape, 0 = the default).  save: a one-element list that receives the buffer of
  orca_tgat_forward_saving (what tgat_backward(saved=...) then does not recompute)."""
  inp, keep = _tgat_inputs(kw, "tgat_forward")
  dev = blob.device
  if out is None:
    out = torch.empty((Q, D), dtype=torch.float32, device=dev)
  w = torch.empty((Q, K), dtype=torch.float32, device=dev) if want_weights else None
  if save is not None and Q > 0:
    n = _lib.load().orca_tgat_saved_floats(int(d_e), int(Q), int(K))
    if n < 0:
      raise ValueError(f"tgat_forward: unsupported sizes d_e={d_e} K={K}")
    buf = torch.empty(n, dtype=torch.float32, device=dev)
    check(_lib.load().orca_tgat_forward_saving(ptr(blob), int(d_e), ctypes.byref(inp), int(Q), int(K), ptr(out), ptr(w),
                                               ptr(buf), stream()), "orca_tgat_forward_saving")
    save.append(buf)
    return out, w
  check(_lib.load().orca_tgat_forward(ptr(blob), int(d_e), ctypes.byref(inp), int(Q), int(K), ptr(out), ptr(w),
                                      stream()), "orca_tgat_forward")
  return out, w


def tgat_dropout_mask(seed, q0, Q, K, p, device):
  """the multipliers (0 or 1/(1-p)) [Q, 2, K] the attention kernels apply for (seed, q0): what a checker feeds its
  own softmax in place of torch's dropout"""
  m = torch.empty((Q, 2, K), dtype=torch.float32, device=device)
  check(_lib.load().orca_tgat_dropout_mask(int(seed) & 0xFFFFFFFFFFFFFFFF, int(q0), int(Q), int(K), float(p), ptr(m), stream()),
        "orca_tgat_dropout_mask")
  return m


def _tgat_inputs(kw, who):
  inp = _lib.TgatInputs()
  inp.q_folded = int(bool(kw.pop("q_folded", False))) | ((int(kw.pop("waves_per_query", 0) or 0) & 0xff) << 8)
  inp.dropout_p = float(kw.pop("dropout_p", 0.0) or 0.0)
  inp.dropout_seed = int(kw.pop("dropout_seed", 0) or 0) & 0xFFFFFFFFFFFFFFFF
  inp.dropout_q0 = int(kw.pop("dropout_q0", 0) or 0)
  keep = []
  for name, ty in _lib.TgatInputs._fields_:
    if ty is not _lib.vp:
      continue
    t = kw.pop(name, None)
    if t is not None:
      if not t.is_cuda:
        raise _lib.OrcaError(f"{who}: {name} is not on the HIP device")
      t = t.contiguous()
      keep.append(t)
    setattr(inp, name, ptr(t))
  if kw:
    raise TypeError(f"{who}: unknown inputs {sorted(kw)}")
  return inp, keep


class ZeroPool:
  """One zero-filled buffer handed out in pieces: the gradient buffers of a training step (decoder, every attention
  layer, GRU) start at zero, and one fill of their total is one launch instead of one per call.  take() falls back to
  a fill of its own when the pool is used up; `asked` is what a step needed (the next pool's size)."""

  def __init__(self, n, device):
    self.buf = torch.zeros(int(n), dtype=torch.float32, device=device) if n > 0 else None
    self.used = 0
    self.asked = 0
    self.device = device

  def take(self, n):
    n = int(n)
    self.asked += n
    if self.buf is not None and self.used + n <= self.buf.numel():
      v = self.buf[self.used:self.used + n]
      self.used += n
      return v
    return torch.zeros(n, dtype=torch.float32, device=self.device)


def _f32(t, device):
  """t as a contiguous fp32 tensor on `device`: t itself when it already is one (the common case on the training path,
  where three tensor operations per parameter and step are host time the step does not have)"""
  dev = torch.device(device)
  if t.dtype == torch.float32 and t.device.type == dev.type and (dev.index is None or t.device.index == dev.index) and t.is_contiguous():
    return t.detach() if t.requires_grad else t
  return t.detach().to(device=device, dtype=torch.float32).contiguous()


def _zeros(n, dev, pool):
  return pool.take(n) if pool is not None else torch.zeros(n, dtype=torch.float32, device=dev)


def tgat_backward(sd, d_e, Q, K, grad_out, want_time_param_grads=False, pool=None, saved=None, want_d_nbr=True,
                  deferred=None, **kw):
  """orca_tgat_backward.  sd: the layer's raw parameters (keys TGAT_KEYS, fp32 on the device); kw: the fields of
  orca_tgat_inputs of the forward call (mask = the mask BEFORE the forward).  Returns (grads: dict like sd,
  d_src [Q,D], d_nbr [Q*K,D], d_src_time, d_nbr_time, d_time_w, d_time_b) -- the time entries that do not apply
  to the mode are None.  deferred: a list -- only the data gradients are computed now (orca_tgat_backward_phase 1);
  a callable that issues the parameter gradients (phase 2: grads, d_time_w/b are valid after it) on the then-current
  stream is appended to the list and keeps the workspace alive; the caller orders that stream after this call and
  joins it before dropping the callable."""
  dev = grad_out.device
  ps = [_f32(sd[k], dev) for k in TGAT_KEYS]
  gather = kw.get("dt") is not None
  sizes = [p.numel() for p in ps] + ([D, D] if gather and want_time_param_grads else [])
  flat = _zeros(sum(-(-n // 64) * 64 for n in sizes), dev, pool)   # one fill for all of them (pool: for the whole step)
  views, o = [], 0
  for n in sizes:
    views.append(flat[o:o + n])
    o += -(-n // 64) * 64
  gs = [v.view(p.shape) for v, p in zip(views, ps)]
  P, G = _lib.TgatParams(), _lib.TgatParams()
  for (name, _), p, g in zip(_lib.TgatParams._fields_, ps, gs):
    setattr(P, name, ptr(p))
    setattr(G, name, ptr(g))
  inp, keep = _tgat_inputs(kw, "tgat_backward")
  n = _lib.load().orca_tgat_backward_workspace_floats(int(d_e), int(Q), int(K))
  if n < 0:
    raise ValueError(f"tgat_backward: unsupported sizes d_e={d_e} K={K}")
  wsp = torch.empty(n, dtype=torch.float32, device=dev)
  go = _f32(grad_out, grad_out.device)
  d_src = torch.empty((Q, D), dtype=torch.float32, device=dev)
  d_nbr = torch.empty((Q * K, D), dtype=torch.float32, device=dev) if want_d_nbr else None
  d_st = d_nt = d_tw = d_tb = None
  if gather:
    if want_time_param_grads:
      d_tw, d_tb = views[-2], views[-1]
  else:
    d_st = torch.empty((Q, D), dtype=torch.float32, device=dev)
    d_nt = torch.empty((Q * K, D), dtype=torch.float32, device=dev)
  if deferred is not None:
    def run(phases, _keep=(ps, gs, keep, wsp, go, d_src, d_nbr, d_st, d_nt, flat, saved, kw)):
      check(_lib.load().orca_tgat_backward_phase(ctypes.byref(P), int(d_e), ctypes.byref(inp), int(Q), int(K), ptr(go),
                                                 ctypes.byref(G), ptr(d_src), ptr(d_nbr), ptr(d_st), ptr(d_nt), ptr(d_tw),
                                                 ptr(d_tb), ptr(wsp), ptr(saved), phases, stream()), "orca_tgat_backward_phase")
    run(1)
    deferred.append(lambda: run(2))
  elif saved is not None:
    check(_lib.load().orca_tgat_backward_saved(ctypes.byref(P), int(d_e), ctypes.byref(inp), int(Q), int(K), ptr(go),
                                               ctypes.byref(G), ptr(d_src), ptr(d_nbr), ptr(d_st), ptr(d_nt), ptr(d_tw),
                                               ptr(d_tb), ptr(wsp), ptr(saved), stream()), "orca_tgat_backward_saved")
  else:
    check(_lib.load().orca_tgat_backward(ctypes.byref(P), int(d_e), ctypes.byref(inp), int(Q), int(K), ptr(go),
                                         ctypes.byref(G), ptr(d_src), ptr(d_nbr), ptr(d_st), ptr(d_nt), ptr(d_tw),
                                         ptr(d_tb), ptr(wsp), stream()), "orca_tgat_backward")
  return dict(zip(TGAT_KEYS, gs)), d_src, d_nbr, d_st, d_nt, d_tw, d_tb


# ---------------------------------------------------------------------------------------------
# GRU memory updater (modules/memory_updater.py)
# ---------------------------------------------------------------------------------------------
def gru_pack(weight_ih, weight_hh, bias_ih, bias_hh, device, blob=None):
  msg_dim = weight_ih.shape[1]
  n = _lib.load().orca_gru_blob_floats(int(msg_dim))
  if n < 0:
    raise ValueError(f"gru_pack: unsupported message width {msg_dim}")
  again = blob is not None      # (a blob handed back in was packed here before: orca_gru_repack skips the zero fill)
  if blob is None:
    blob = torch.empty(n, dtype=torch.float32, device=device)
  ts = [_f32(t, device) for t in (weight_ih, weight_hh, bias_ih, bias_hh)]
  fn = _lib.load().orca_gru_repack if again else _lib.load().orca_gru_pack
  check(fn(int(msg_dim), *(ptr(t) for t in ts), ptr(blob), stream()), "orca_gru_pack")
  return blob


def gru_backward(weight_ih, weight_hh, bias_ih, bias_hh, ids, x, h, d_out, want_inputs=False, pool=None, deferred=None,
                 gates=None):
  """orca_gru_backward.  ids int32 [n] or None (dense rows).  Returns (g_weight_ih, g_weight_hh, g_bias_ih,
  g_bias_hh, d_x, d_h) -- the last two None unless want_inputs.  deferred: as in tgat_backward (the parameter
  gradients are issued by the callable appended to the list).  gates: what gru_update(gates=) kept for the same id
  list (orca_gru_backward_saved: no recompute of the gate pre-activations)."""
  dev = d_out.device
  ps = [_f32(t, dev) for t in (weight_ih, weight_hh, bias_ih, bias_hh)]
  flat = _zeros(sum(-(-p.numel() // 64) * 64 for p in ps), dev, pool)    # one fill for the four
  gs, o = [], 0
  for p in ps:
    gs.append(flat[o:o + p.numel()].view(p.shape))
    o += -(-p.numel() // 64) * 64
  msg_dim = ps[0].shape[1]
  n = d_out.shape[0]
  nw = _lib.load().orca_gru_backward_workspace_floats(int(msg_dim), int(n))
  if nw < 0:
    raise ValueError(f"gru_backward: unsupported message width {msg_dim}")
  wsp = torch.empty(nw, dtype=torch.float32, device=dev)
  go = _f32(d_out, d_out.device)
  d_x = torch.empty((n, msg_dim), dtype=torch.float32, device=dev) if want_inputs else None
  d_h = torch.empty((n, D), dtype=torch.float32, device=dev) if want_inputs else None
  x, h = x.contiguous(), h.contiguous()
  if deferred is not None:
    def run(phases, _keep=(ps, gs, ids, x, h, go, d_x, d_h, wsp, flat)):
      check(_lib.load().orca_gru_backward_saved(int(msg_dim), *(ptr(p) for p in ps), ptr(ids), int(n), ptr(x), ptr(h), ptr(go),
                                                *(ptr(g) for g in gs), ptr(d_x), ptr(d_h), ptr(wsp), ptr(gates), phases, stream()),
            "orca_gru_backward_saved")
    run(1)
    deferred.append(lambda: run(2))
    return (*gs, d_x, d_h)
  if gates is not None:
    check(_lib.load().orca_gru_backward_saved(int(msg_dim), *(ptr(p) for p in ps), ptr(ids), int(n), ptr(x), ptr(h), ptr(go),
                                              *(ptr(g) for g in gs), ptr(d_x), ptr(d_h), ptr(wsp), ptr(gates), 3, stream()),
          "orca_gru_backward_saved")
    return (*gs, d_x, d_h)
  check(_lib.load().orca_gru_backward(int(msg_dim), *(ptr(p) for p in ps), ptr(ids), int(n), ptr(x), ptr(h), ptr(go),
                                      *(ptr(g) for g in gs), ptr(d_x), ptr(d_h), ptr(wsp), stream()), "orca_gru_backward")
  return (*gs, d_x, d_h)


def persist_overlay(ids64, pending, overlay, memory, last_update, msg_ts):
  """memory[id] = overlay[id], last_update[id] = msg_ts[id] for the pending ids of ids64 (memory_updater.py:25-39 when
  the overlay already holds their GRU rows)"""
  check(_lib.load().orca_persist_overlay(ptr(ids64), ids64.numel(), ptr(pending), ptr(overlay), ptr(memory), ptr(last_update),
                                         ptr(msg_ts), memory.shape[0], stream()), "orca_persist_overlay")


def commit_rows(ids32, n, n_dev, src_table, dst_table, last_update=None, msg_ts=None):
  """dst_table[id] = src_table[id] (+ last_update[id] = msg_ts[id]) for the ids of a device-counted work-list: the row
  commit of update_memory (memory_updater.py:36-39) from rows the GRU wrote to the overlay"""
  check(_lib.load().orca_commit_rows(ptr(ids32), int(n), ptr(n_dev), ptr(src_table), ptr(dst_table), ptr(last_update), ptr(msg_ts),
                                     dst_table.shape[0], stream()), "orca_commit_rows")


def gru_update(blob, msg_dim, ids, n, n_dev, x, h, out, last_update=None, msg_ts=None, gates=None):
  """gates: float32 [>= n * 4 * 100] -- orca_gru_update_saving (keeps r, z, n, h_n per row for gru_backward(gates=))"""
  if gates is not None:
    check(_lib.load().orca_gru_update_saving(ptr(blob), int(msg_dim), ptr(ids), int(n), ptr(n_dev), ptr(x), ptr(h), ptr(out),
                                             ptr(gates), stream()), "orca_gru_update_saving")
    return
  check(_lib.load().orca_gru_update(ptr(blob), int(msg_dim), ptr(ids), int(n), ptr(n_dev), ptr(x), ptr(h), ptr(out),
                                    ptr(last_update), ptr(msg_ts), stream()), "orca_gru_update")


# ---------------------------------------------------------------------------------------------
# node-id bookkeeping, messages, history
# ---------------------------------------------------------------------------------------------
def fill_i32(t, v):
  check(_lib.load().orca_fill_i32(ptr(t), t.numel(), int(v), stream()), "orca_fill_i32")


def fill_u8(t, v):
  check(_lib.load().orca_fill_u8(ptr(t), t.numel(), int(v), stream()), "orca_fill_u8")


def worklist_build(ids32, ids64, flags, slot, epoch, worklist, count):
  """distinct ids of ids32 ∪ ids64 with flags[id] != 0 -> worklist[:count]; uses n32+n64 epochs of `slot`"""
  n32 = ids32.numel() if ids32 is not None else 0
  n64 = ids64.numel() if ids64 is not None else 0
  check(_lib.load().orca_worklist_build(ptr(ids32), n32, ptr(ids64), n64, ptr(flags), ptr(slot), int(epoch),
                                        slot.numel(), ptr(worklist), ptr(count), stream()), "orca_worklist_build")


def set_flags(flags, ids64=None, ids32=None, n=None, n_dev=None, v=1):
  if n is None:
    n = (ids64 if ids64 is not None else ids32).numel()
  check(_lib.load().orca_set_flags(ptr(flags), ptr(ids64), ptr(ids32), int(n), ptr(n_dev), int(v), flags.numel(), stream()),
        "orca_set_flags")


def segment_last(ids, slot, epoch):
  ids = _dev(ids, torch.int64, "segment_last ids")
  winner = torch.empty(ids.numel(), dtype=torch.uint8, device=ids.device)
  check(_lib.load().orca_segment_last(ptr(ids), ids.numel(), ptr(slot), int(epoch), slot.numel(), ptr(winner),
                                      stream()), "orca_segment_last")
  return winner


def segment_mean(ids, msgs, ts, slot, epoch):
  ids = _dev(ids, torch.int64, "segment_mean ids")
  msgs = _dev(msgs, torch.float32, "segment_mean msgs")
  winner = torch.empty(ids.numel(), dtype=torch.uint8, device=ids.device)
  out = torch.zeros_like(msgs)
  check(_lib.load().orca_segment_mean(ptr(ids), ids.numel(), msgs.shape[1], ptr(msgs), ptr(ts), ptr(slot), int(epoch),
                                      slot.numel(), ptr(winner), ptr(out), stream()), "orca_segment_mean")
  return winner, out


def message_store(src, dst, eidx, t, memory, last_update, edge_feats, time_w, time_b, slot, epoch, messages, msg_ts,
                  pending):
  check(_lib.load().orca_message_store(ptr(src), ptr(dst), ptr(eidx), ptr(t), src.numel(), ptr(memory),
                                       ptr(last_update), ptr(edge_feats), edge_feats.shape[1], ptr(time_w),
                                       ptr(time_b), ptr(slot), int(epoch), slot.numel(), ptr(messages), ptr(msg_ts),
                                       ptr(pending), stream()), "orca_message_store")


def history_push(emb, x, ids, slot, epoch):
  check(_lib.load().orca_history_push(ptr(emb), ptr(x), ptr(ids), ids.numel(), ptr(slot), int(epoch), slot.numel(),
                                      stream()), "orca_history_push")


def index_positions(lst, n, n_dev, pos):
  check(_lib.load().orca_index_positions(ptr(lst), int(n), ptr(n_dev), ptr(pos), pos.numel(), stream()), "orca_index_positions")


def scatter_add_flagged(rows, idx, flags, pos, dst):
  i64 = idx if idx.dtype == torch.int64 else None
  i32 = idx if idx.dtype == torch.int32 else None
  if i64 is None and i32 is None:
    raise TypeError("scatter_add_flagged: idx must be int64 or int32")
  check(_lib.load().orca_scatter_add_flagged(ptr(rows.contiguous()), ptr(i64), ptr(i32), idx.numel(), ptr(flags), ptr(pos),
                                             flags.numel(), ptr(dst), stream()), "orca_scatter_add_flagged")


def scatter_add_flagged_pair(rows_a, idx_a, rows_b, idx_b, flags, pos, dst):
  """scatter_add_flagged of two lists into the same dst in one launch (orca_scatter_add_flagged_pair)"""
  def split(idx):
    if idx.dtype not in (torch.int64, torch.int32):
      raise TypeError("scatter_add_flagged_pair: idx must be int64 or int32")
    return (idx if idx.dtype == torch.int64 else None), (idx if idx.dtype == torch.int32 else None)
  a64, a32 = split(idx_a)
  b64, b32 = split(idx_b)
  rows_a, rows_b = rows_a.contiguous(), rows_b.contiguous()
  check(_lib.load().orca_scatter_add_flagged_pair(ptr(rows_a), ptr(a64), ptr(a32), idx_a.numel(), ptr(rows_b), ptr(b64), ptr(b32),
                                                  idx_b.numel(), ptr(flags), ptr(pos), flags.numel(), ptr(dst), stream()),
        "orca_scatter_add_flagged_pair")


def gather_rows(table, idx, out=None):
  n = idx.numel()
  width = table.shape[1]
  if out is None:
    out = torch.empty((n, width), dtype=table.dtype, device=table.device)
  i64 = idx if idx.dtype == torch.int64 else None
  i32 = idx if idx.dtype == torch.int32 else None
  if i64 is None and i32 is None:
    raise TypeError("gather_rows: idx must be int64 or int32")
  check(_lib.load().orca_gather_rows(ptr(table), table.shape[0], width, ptr(i64), ptr(i32), n, ptr(out), stream()), "orca_gather_rows")
  return out


def scatter_rows(table, ids, x):
  check(_lib.load().orca_scatter_rows(ptr(table), table.shape[0], table.shape[1], ptr(ids), ids.numel(), ptr(x), stream()),
        "orca_scatter_rows")


def decoder_train(emb, fc1_w, fc1_b, fc2_w, fc2_b, want_prob=False, pool=None):
  """orca_decoder_train: link decoder + BCE loss of a training step, forward and backward.  emb [3*bs, D] = embeddings
  of [src; dst; neg].  Returns (loss [1], d_emb [3*bs, D], grads {fc1.weight, fc1.bias, fc2.weight, fc2.bias}, prob
  [2*bs] or None)."""
  dev = emb.device
  bs = emb.shape[0] // 3
  ps = [_f32(t, dev) for t in (fc1_w, fc1_b, fc2_w, fc2_b)]
  flat = _zeros(sum(-(-p.numel() // 64) * 64 for p in ps) + 64, dev, pool)
  gs, o = [], 0
  for p in ps:
    gs.append(flat[o:o + p.numel()].view(p.shape))
    o += -(-p.numel() // 64) * 64
  loss = flat[o:o + 1]
  nw = _lib.load().orca_decoder_train_workspace_floats(int(bs))
  wsp = torch.empty(nw, dtype=torch.float32, device=dev)
  d_emb = torch.empty((3 * bs, D), dtype=torch.float32, device=dev)
  prob = torch.empty(2 * bs, dtype=torch.float32, device=dev) if want_prob else None
  e = emb.detach().contiguous()
  check(_lib.load().orca_decoder_train(ptr(e), int(bs), *(ptr(p) for p in ps), ptr(loss), ptr(prob), ptr(d_emb),
                                       *(ptr(g) for g in gs), ptr(wsp), stream()), "orca_decoder_train")
  return loss, d_emb, dict(zip(("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"), gs)), prob


class FusedAdam:
  """torch.optim.Adam(params, lr) (train.py:180) as one launch per step (orca_adam_step): params is a dict name ->
  fp32 device tensor, updated in place; step(grads) takes a dict with (a subset of) the same names -- a parameter
  without a gradient is skipped, as torch skips p.grad is None."""

  def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
    self.params = {k: v for k, v in params.items()}
    self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
    n = sum(-(-v.numel() // 64) * 64 for v in self.params.values())
    dev = next(iter(self.params.values())).device
    self._state = torch.zeros(2, max(n, 1), dtype=torch.float32, device=dev)     # exp_avg | exp_avg_sq, one fill
    self.exp_avg, self.exp_avg_sq, o = {}, {}, 0
    for k, v in self.params.items():
      if v.dtype != torch.float32 or not v.is_contiguous() or not v.is_cuda:
        raise _lib.OrcaError(f"FusedAdam: parameter {k} must be a contiguous fp32 tensor on the HIP device")
      self.exp_avg[k] = self._state[0, o:o + v.numel()]
      self.exp_avg_sq[k] = self._state[1, o:o + v.numel()]
      o += -(-v.numel() // 64) * 64
    self.steps = {k: 0 for k in self.params}       # torch counts steps per parameter (a skipped one does not advance)

  def step(self, grads):
    by_step = {}
    for k, p in self.params.items():
      g = grads.get(k)
      if g is None:
        continue
      if g.numel() != p.numel() or g.dtype != torch.float32 or not g.is_cuda:
        raise _lib.OrcaError(f"FusedAdam: gradient of {k} does not match its parameter")
      self.steps[k] += 1
      by_step.setdefault(self.steps[k], []).append((p, g.contiguous(), self.exp_avg[k], self.exp_avg_sq[k]))
    for t, lst in by_step.items():
      n = len(lst)
      arr = lambda j: (ctypes.c_void_p * n)(*[x[j].data_ptr() for x in lst])
      sizes = (ctypes.c_int64 * n)(*[x[0].numel() for x in lst])
      check(_lib.load().orca_adam_step(n, arr(0), arr(1), arr(2), arr(3), sizes, self.lr, self.betas[0], self.betas[1],
                                       self.eps, t, stream()), "orca_adam_step")


def find_uncached(nbr, batch_nodes, cache_flag, slot, epoch, bitmap):
  """orca_find_uncached (embedding_module.py:200-212): sorted distinct out-of-batch neighbours that are not cached,
  minus the first.  Returns an int64 tensor (one host read of the count: the caller concatenates it to the batch)."""
  n_nbr = nbr.numel()
  cap = min(n_nbr, slot.numel())
  out = torch.empty(max(cap, 1), dtype=torch.int64, device=nbr.device)
  count = torch.zeros(1, dtype=torch.int32, device=nbr.device)
  check(_lib.load().orca_find_uncached(ptr(nbr), n_nbr, ptr(batch_nodes), batch_nodes.numel(), ptr(cache_flag), ptr(slot),
                                       int(epoch), slot.numel(), ptr(bitmap), ptr(out), cap, ptr(count), stream()),
        "orca_find_uncached")
  return out[:int(count.item())]


def build_roots(src, dst, neg, ts):
  """orca_build_roots: ([src;dst;neg] int64, the same as int32, [t;t;t] float64) in one launch (tgn_model.py:99-102)."""
  bs = src.numel()
  dev = src.device
  nodes = torch.empty(3 * bs, dtype=torch.int64, device=dev)
  n32 = torch.empty(3 * bs, dtype=torch.int32, device=dev)
  t3 = torch.empty(3 * bs, dtype=torch.float64, device=dev)
  check(_lib.load().orca_build_roots(ptr(src), ptr(dst), ptr(neg), ptr(ts), bs, ptr(nodes), ptr(t3), ptr(n32), None, stream()),
        "orca_build_roots")
  return nodes, n32, t3
