"""Device-resident TGN step: the hot path of LuckyLYM/Orca with every table, id list and flag kept
in HBM between kernels (no host NumPy set algebra, no per-layer PCIe copies, no full-table clone).

Follows, stage by stage, model/tgn_model.py:95-147 (compute_temporal_embeddings) with
  Path A  modules/embedding_module.py:116-147   exact TGN recursion       (reuse=False)
  Path B  modules/embedding_module.py:154-182   Orca-P push-and-pull      (reuse=True, budget=0)
  Path C  modules/embedding_module.py:215-285   Orca-R budgeted reuse     (reuse=True, budget>0)
and the memory / message steps of modules/memory_updater.py:25-78, modules/memory.py:25-28,
model/tgn_model.py:172-191.  compute_temporal_embeddings / compute_edge_probabilities are the forward (the
reference's no_grad persist / eval path and the forward half of a train step); train_step adds the loss and the
backward pass of train.py:225-237 for Paths A and B.  Dropout is off (module.eval()).

Multi-GPU (not in the reference): each rank embeds a contiguous slice of the batch's (node, t)
queries.  The state rows are owned by node id (owner(id) = id % world): in a train-ordered Path B step only the
owner of an id runs its memory persist, raw-message store and GRU; the rows other ranks read -- layer outputs for
the history tables, GRU(message, memory) of the stored ids -- are exchanged with two RCCL all-gathers per step
into read replicas of the overlay / history tables.  Duplicate resolution (last occurrence wins) is by global
query position, so N-rank results equal 1-rank results bit for bit.  Eval-ordered steps and the Python-dispatched
paths (A, C, train_step) run on coherent replicas with the batch's row writes redone on every rank; sync_replicas()
is the switch from the owner-computes state to that one.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from . import dist as dist_utils
from . import ops
from ._lib import ORCA_D, OrcaError, check, ptr, stream

D = ORCA_D


class OrcaEngine:
  def __init__(self, state_dict, edge_feats, n_nodes, n_layers, n_neighbors, reuse=False, budget=0,
               device="cuda:0", group=None, sampling="recent", seed=0, exchange_cap=None, pretend=None, dropout=0.0,
               gradient=False, slot_history=None, small_batch=None):
    """state_dict: the TGN state_dict of the reference (key names of SURVEY §8b); edge_feats
    [n_edges+1, d_e] (row 0 = padding); n_nodes = number of table rows (args.n_nodes, train.py:129).
    sampling: "recent" (the reference's NeighborFinder) or "uniform" (seeded, not in the reference: BASELINE
    config 5) with `seed`.  exchange_cap: rows a rank can send per owner-computes step (default: 1.25 x its even
    share of the batch's positives + 256, at most all of them); a batch that gives one owner more raises at the
    next check_errors().  pretend=(rank, world): measurement aid (tools/slice_probe.py) -- behave as that rank of
    a world with the exchanges stubbed out.  small_batch: attention launches of at most 1024 queries take four waves per query (default: with one rank; equal to the default form to rounding, not to the bit).  slot_history: Path C keeps its layer-1 history in a budget-bounded slot
    cache (budget + one batch of rows and an id -> slot index) instead of the dense [n_nodes, 100] table of
    modules/history.py:11 -- None = wherever the native Path C step runs (budget > 0, two layers, one rank); it needs
    cache plans of the planner (include/orca_hip.h, orca_slot_cache) and falls back to the dense table, materialised from
    the slots, the moment a step without a plan (evaluation: flags accumulate) or a Python-dispatched step comes."""
    self.device = torch.device(device)
    if self.device.type != "cuda":
      raise OrcaError("OrcaEngine needs a HIP device (there is no CPU fallback)")
    self.N, self.L, self.K = int(n_nodes), int(n_layers), int(n_neighbors)
    self.reuse, self.budget = bool(reuse), int(budget)
    # Path C through the native driver (two layers, one rank): its launches are sized for `budget_cap` uncached
    # out-of-batch neighbours per batch (their actual number is a device-side count).  None = the safe bound
    # min(3*bs*K, n_nodes): every uncached neighbour is a distinct sampled id.  A smaller cap makes the launches
    # smaller; a batch that exceeds it evaluates only the first `cap` ids (embedding_module.py:251-280 evaluates all),
    # which the NEXT native step -- or check_errors() / budget_overflow() -- reports by raising OrcaError.
    self.native_budget, self.budget_cap = True, None
    self.native_train = os.environ.get("ORCA_NATIVE_TRAIN", "1") != "0"   # train_step's forward through the C++ driver
    # ORCA_OVERLAP=1: train_step / step_optimizer on two streams -- the parameter-gradient reductions and the re-pack after
    # the optimizer run beside the chain of data gradients and the next batch's first kernels.  Off by default: it wins
    # inside one synchronised step and loses in the pipelined loop ("a second stream")
    self.overlap = os.environ.get("ORCA_OVERLAP", "0") != "0"
    self.group = group
    if group is not None:
      import torch.distributed as dist
      self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
    else:
      self.rank, self.world = 0, 1
    self._pretend = pretend is not None
    if pretend is not None:
      self.rank, self.world = int(pretend[0]), int(pretend[1])
    if sampling not in ("recent", "uniform"):
      raise ValueError(f"OrcaEngine: unknown sampling mode {sampling!r}")
    self.sampling, self.seed = sampling, int(seed)
    # attention-weight dropout of train_step (train.py:30 --drop_out, nn.MultiheadAttention's); the forward-only
    # entry points never drop (module.eval()).  Counter RNG keyed (seed, batch, layer, query, head, slot).
    self.dropout = float(dropout)
    # train.py:43 --gradient (history.py:42-46 push_and_pull): in train_step the history rows pushed in the same batch
    # stay connected to the layer output that produced them (Path B); the forward values are the same either way
    self.gradient = bool(gradient)
    self.exchange_cap = exchange_cap
    self._sharded = False    # True: memory / last_update / messages / msg_ts rows are current only on their owner
    # the reference freezes its time encoder (model/time_encoding.py:20-23: requires_grad False, forward under no_grad):
    # train_step returns no gradient for it unless this is set (the kernels can produce one: tests)
    self.train_time_encoder = False
    dev = self.device
    ef = edge_feats if isinstance(edge_feats, torch.Tensor) else torch.as_tensor(np.asarray(edge_feats))
    self.edge_feats = ef.to(device=dev, dtype=torch.float32).contiguous()
    self.d_e = self.edge_feats.shape[1]
    self.msg_dim = 3 * D + self.d_e                                   # tgn_model.py:49-50
    self.load_state_dict(state_dict)
    f32 = dict(dtype=torch.float32, device=dev)
    # modules/memory.py:17-23 (+ the pending flag bitmap, a host bool array upstream)
    self.memory = torch.zeros(self.N, D, **f32)
    self.last_update = torch.zeros(self.N, **f32)
    self.messages = torch.zeros(self.N, self.msg_dim, **f32)
    self.msg_ts = torch.zeros(self.N, **f32)
    self.pending = torch.zeros(self.N, dtype=torch.uint8, device=dev)
    # replaces the per-batch full-table clone of memory_updater.py:72,76
    self.overlay = torch.zeros(self.N, D, **f32)
    # modules/history.py:11-20
    nh = self.L - 1 if self.reuse else 0
    # attention launches of <= 1024 queries in the four-waves-per-query form (orca_tgat_inputs.waves_per_query): on by default
    # with one rank; an N-rank run is held bit-equal to one rank in the default form
    self.small_batch = bool(self.world == 1 if small_batch is None else small_batch)
    if self.small_batch and self.world > 1:
      raise OrcaError("OrcaEngine: small_batch is a one-rank launch shape")
    self._slot_eligible = bool(self.reuse and self.budget > 0 and self.L == 2 and self.world == 1 and
                               (slot_history is None or slot_history))
    if slot_history and not self._slot_eligible:
      raise OrcaError("OrcaEngine: slot_history needs the native Path C step (reuse, budget > 0, two layers, one rank)")
    self._sc = None                  # slot-cache arrays while the history lives in slots (then self._hist_dense is None)
    self._hist_dense = None if self._slot_eligible else [torch.zeros(self.N, D, **f32) for _ in range(nh)]
    self.cache_flag = [torch.zeros(self.N, dtype=torch.uint8, device=dev) for _ in range(nh)]
    # scratch
    self.slot = torch.zeros(self.N, dtype=torch.int64, device=dev)
    self.slot_epoch = 1
    self.worklist = torch.zeros(self.N, dtype=torch.int32, device=dev)
    self.count = torch.zeros(1, dtype=torch.int32, device=dev)
    self.all_ids = None
    self.csr = None
    self.test_mode = False
    self.batch_counter = 0
    self._native = None      # orca_engine_t: the C++ step driver (created on first use, regrown with the batch)
    self._q_max = 0
    self._prof_native = 0
    # "overlay complete" (include/orca_hip.h, eager overlay): true for the zeroed state this object starts with;
    # mirrored into the native driver, which keeps it up to date while it owns the steps
    self._overlay_complete = True
    self.prof = None     # set to {} to bracket the kernels with HIP events (bench.py roofline)

  def _timed(self, name, fn, *a, **k):
    """Runs fn; when profiling is on, brackets it with HIP events on the launch stream."""
    if self.prof is None:
      return fn(*a, **k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*a, **k)
    e1.record()
    self.prof.setdefault(name, []).append((e0, e1))
    return r

  def prof_summary(self):
    """{kernel: (launches, mean ms)}; call after torch.cuda.synchronize()."""
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1)) for k, v in (self.prof or {}).items()}

  # -------------------------------------------------------------------------------------------
  def load_state_dict(self, sd, _time_unchanged=False):
    dev = self.device
    if getattr(self, "_packed_ready", None) is not None and torch.cuda.current_stream() != getattr(self, "_side", None):
      self._await_packed()               # a re-pack still running beside this stream writes the same blobs
    if sd is not getattr(self, "params", None):
      sd = {k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in sd.items()}
      # the raw tensors on the device: what the backward pass reads and an optimizer updates (train_step / refresh)
      self.params = {k: v.detach().to(device=dev, dtype=torch.float32).contiguous().clone() for k, v in sd.items()}
      sd = self.params
    first = not hasattr(self, "att_blobs")
    tw = sd["time_encoder.w.weight"].reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
    b = sd.get("time_encoder.w.bias")
    tb = b.reshape(-1).to(device=dev, dtype=torch.float32).contiguous() if b is not None else torch.zeros(D, device=dev)
    if first:
      self.time_w, self.time_b = tw.clone(), tb.clone()
      self.att_blobs, self.gru_blob = [None] * self.L, None
    elif not _time_unchanged:  # (step_optimizer with the time encoder frozen: nothing changed them)
      self.time_w.copy_(tw)    # re-pack in place: the native driver holds these addresses
      self.time_b.copy_(tb)
    sds = self._layer_params(sd)
    # in place when they exist (the native driver holds the addresses); all layers' folds in shared launches
    self.att_blobs = ops.tgat_pack_layers(sds, self.d_e, dev, blobs=None if first else self.att_blobs, time_b=self.time_b)
    p = "memory_updater.memory_updater."
    self.gru_blob = ops.gru_pack(sd[p + "weight_ih"], sd[p + "weight_hh"], sd[p + "bias_ih"], sd[p + "bias_hh"], dev,
                                 blob=self.gru_blob)
    self.aff = {k[len("affinity_score."):]: v for k, v in self.params.items() if k.startswith("affinity_score.")}
    if not first and hasattr(self, "_overlay_complete"):
      # overlay rows computed with the previous GRU weights are stale: the reference recomputes
      # get_updated_memory with the current weights on every batch (memory_updater.py:56-78)
      self._invalidate_overlay()

  def _layer_params(self, sd=None):
    """the attention layers' parameters as one dict per layer, keyed as ops.TGAT_KEYS; grouped once per parameter set
    (the optimizer updates self.params in place: same tensors every step)"""
    sd = self.params if sd is None else sd
    cached = getattr(self, "_layer_sds", None)
    if cached is not None and cached[0] is sd and cached[1] == len(sd):
      return cached[2]
    pre = "embedding_module.attention_models."
    sds = [{k[len(f"{pre}{i}."):]: v for k, v in sd.items() if k.startswith(f"{pre}{i}.")} for i in range(self.L)]
    self._layer_sds = (sd, len(sd), sds)
    return sds

  def refresh(self, beside=False, _time_unchanged=False):
    """Re-packs the kernels' parameter blobs from self.params (after an optimizer step changed them in place).
    beside: the re-pack (folds in fp64, fragment orders: seven launches that no kernel of the next batch needs before
    its GRU) is issued on the engine's second stream, ordered after everything enqueued so far; every method that reads
    the blobs waits for it (_await_packed), so the next batch's root, sampling and work-list kernels run beside it."""
    if beside and self.device.type == "cuda" and self.overlap:
      side = self._side_stream()
      side.wait_stream(torch.cuda.current_stream())
      with torch.cuda.stream(side):
        self.load_state_dict(self.params, _time_unchanged=_time_unchanged)
        ev = torch.cuda.Event()
        ev.record(side)
      self._packed_ready = ev
      return
    self.load_state_dict(self.params, _time_unchanged=_time_unchanged)     # in place: the native driver keeps its blob addresses

  def _side_stream(self):
    if getattr(self, "_side", None) is None:
      self._side = torch.cuda.Stream(device=self.device)
    return self._side

  def _await_packed(self):
    """orders the current stream after a re-pack issued beside it (refresh(beside=True)); free when there is none"""
    ev = getattr(self, "_packed_ready", None)
    if ev is not None:
      torch.cuda.current_stream().wait_event(ev)
      self._packed_ready = None

  # ---- history tables: dense (modules/history.py:11) or, for the native Path C step, a budget-bounded slot cache --------
  @property
  def hist(self):
    """the dense [n_nodes, 100] history tables of the reference (History.emb).  While the history lives in the slot cache
    this is a MATERIALISED VIEW built on every access (rows of the ids the cache plan keeps, zeros elsewhere): for tests
    and for the drop-in mirror, not for the hot path."""
    if self._hist_dense is not None:
      return self._hist_dense
    dense = torch.zeros(self.N, D, dtype=torch.float32, device=self.device)
    if self._sc is not None:
      check(_lib.load().orca_slot_cache_materialize(ctypes.byref(self._slot_desc()), self.N, ptr(dense), stream()),
            "orca_slot_cache_materialize")
    return [dense]

  def history_footprint_bytes(self):
    """device bytes that hold layer-1 history rows and their index right now"""
    if self._hist_dense is not None:
      return sum(t.numel() * 4 for t in self._hist_dense)
    if self._sc is None:
      return 0
    return sum(self._sc[k].numel() * self._sc[k].element_size() for k in ("rows", "stamp", "prev_plan"))

  def _slot_desc(self):
    """the slot cache as the native driver last left it (generation, plan region, cached ids: the steps advance them)"""
    sc = self._sc
    if self._native is not None and sc.get("bound"):
      check(_lib.load().orca_engine_get_slot_cache(self._native, ctypes.byref(sc["desc"])), "orca_engine_get_slot_cache")
    return sc["desc"]

  def _slots_for(self, need_rows, plan_n, n_pull):
    """the slot cache (include/orca_hip.h, orca_slot_cache), (re)allocated for `need_rows` evaluated nodes per batch and
    plans of plan_n ids, and bound to the native handle"""
    dev = self.device
    sc = self._sc
    plan_cap = max(int(self.budget), int(plan_n), 1)
    if sc is not None and sc["q_bound"] >= need_rows and sc["plan_cap"] >= plan_cap and sc["pull"].numel() >= n_pull:
      if not sc.get("bound"):
        check(_lib.load().orca_engine_set_slot_cache(self._native, ctypes.byref(sc["desc"])), "orca_engine_set_slot_cache")
        sc["bound"] = True
      return sc
    if need_rows >= (1 << 20):
      raise OrcaError(f"OrcaEngine: {need_rows} evaluated nodes per batch exceed the slot cache's 2^20 positions: construct "
                      "the engine with slot_history=False, or lower budget_cap")
    q_bound = int(need_rows) if sc is None else max(int(need_rows), sc["q_bound"])
    new = dict(plan_cap=plan_cap, q_bound=q_bound,
               rows=torch.zeros(1 + 2 * plan_cap + q_bound, D, dtype=torch.float32, device=dev),
               stamp=torch.zeros(self.N, dtype=torch.int64, device=dev) if sc is None else sc["stamp"],
               pull=torch.zeros(max(n_pull, 1), dtype=torch.int32, device=dev),
               prev_plan=torch.zeros(plan_cap, dtype=torch.int64, device=dev))
    d = _lib.SlotCache()
    d.gen, d.prev_n, d.cur = 2, 0, 0
    if sc is not None:      # grow: the cached rows move to region 0 of the new array, and their stamps follow
      old = self._slot_desc()
      n, base = int(old.prev_n), 1 + int(old.cur) * sc["plan_cap"]
      new["rows"][1:1 + n] = sc["rows"][base:base + n]
      new["prev_plan"][:n] = sc["prev_plan"][:n]
      d.gen, d.prev_n = int(old.gen), n
      self._restamp(new, n, d.gen)
    d.rows, d.stamp, d.prev_plan, d.pull_idx = ptr(new["rows"]), ptr(new["stamp"]), ptr(new["prev_plan"]), ptr(new["pull"])
    d.plan_cap, d.q_bound = plan_cap, q_bound
    new["desc"] = d
    check(_lib.load().orca_engine_set_slot_cache(self._native, ctypes.byref(d)), "orca_engine_set_slot_cache")
    new["bound"] = True
    self._sc = new
    return new

  def _restore_slot_backup(self, bk):
    """the slot cache holds exactly the rows of a backup taken in slot mode (None: an empty cache).  The generation moves
    past every stamp the table may still hold, so whatever was cached or evaluated since the backup is stale."""
    old, dev = self._sc, self.device
    if old is None and bk is None:
      return
    gen = int(self._slot_desc().gen) + 2 if old is not None else 2
    n = int(bk["ids"].numel()) if bk is not None else 0
    plan_cap = max(int(self.budget), n, 1, old["plan_cap"] if old is not None else 1)
    q_bound = old["q_bound"] if old is not None else 1
    new = dict(plan_cap=plan_cap, q_bound=q_bound,
               rows=torch.zeros(1 + 2 * plan_cap + q_bound, D, dtype=torch.float32, device=dev),
               stamp=old["stamp"] if old is not None else torch.zeros(self.N, dtype=torch.int64, device=dev),
               pull=old["pull"] if old is not None else torch.zeros(1, dtype=torch.int32, device=dev),
               prev_plan=torch.zeros(plan_cap, dtype=torch.int64, device=dev))
    if n:
      new["rows"][1:1 + n] = bk["rows"]
      new["prev_plan"][:n] = bk["ids"]
    self._restamp(new, n, gen)
    d = _lib.SlotCache()
    d.rows, d.stamp, d.prev_plan, d.pull_idx = ptr(new["rows"]), ptr(new["stamp"]), ptr(new["prev_plan"]), ptr(new["pull"])
    d.plan_cap, d.q_bound, d.gen, d.prev_n, d.cur = plan_cap, q_bound, gen, n, 0
    new["desc"], new["bound"] = d, False
    self._sc = new

  @staticmethod
  def _restamp(sc, n, gen):
    """stamps of the n cached ids of sc['prev_plan'] as the step of generation gen - 1 would have left them (kind 1, row i)"""
    if n:
      ids = sc["prev_plan"][:n]
      ok = (ids >= 0) & (ids < sc["stamp"].numel())
      val = ((gen - 1) << 21) | (1 << 20) | torch.arange(n, dtype=torch.int64, device=ids.device)
      sc["stamp"][ids[ok]] = val[ok]

  def _leave_slot_mode(self):
    """the history moves from the slot cache into the dense table (a step without a cache plan, a Python-dispatched
    step): from here on the engine behaves as the reference's dense History does, until reset_state() or a restore()
    of a backup taken in slot mode"""
    if self._hist_dense is not None:
      return
    self._hist_dense = self.hist            # the materialised view becomes the table
    self._drop_slots()
    if self._native is not None:
      self._rebind_native()

  def _drop_slots(self):
    self._sc = None
    if self._native is not None:
      check(_lib.load().orca_engine_set_slot_cache(self._native, None), "orca_engine_set_slot_cache")

  def _unbind_slots(self):
    """before the native handle goes away: take back the slot cache's generation / region / cached-id count"""
    if self._sc is not None and self._sc.get("bound"):
      self._slot_desc()
      self._sc["bound"] = False

  def _rebind_native(self):
    """the native handle caches table pointers: drop it so that the next step rebuilds it with the current tables"""
    L = _lib.load()
    self._unbind_slots()
    self.slot_epoch = int(L.orca_engine_take_epochs(self._native, 0))
    check(L.orca_engine_drain(self._native, stream()), "orca_engine_drain")
    self._overlay_complete = bool(L.orca_engine_get_overlay_complete(self._native))
    L.orca_engine_destroy(self._native)
    self._native = None
    self._q_max = 0

  def set_neighbor_finder(self, csr):
    """tgn_model.py:193-195; csr is an ops.CSR (or a shim NeighborFinder carrying .csr)."""
    self.csr = getattr(csr, "csr", csr)
    if self._native is not None:
      check(_lib.load().orca_engine_set_csr(self._native, self.csr._h), "orca_engine_set_csr")

  # modules/memory.py:17-23, modules/history.py:14-20
  def reset_state(self):
    for t in (self.memory, self.last_update, self.messages, self.msg_ts, self.pending, self.overlay):
      t.zero_()
    if self._slot_eligible and self._hist_dense is not None:     # back to the slot cache (empty)
      self._hist_dense = None
      if self._native is not None:
        self._rebind_native()
    self._drop_slots()
    for t in (self._hist_dense or []) + self.cache_flag:
      t.zero_()
    self.test_mode = False
    self._sharded = False
    self._set_overlay_complete(True)       # no pending ids

  def _invalidate_overlay(self):
    """the overlay rows no longer match GRU(messages, memory) of the pending ids (weights changed, tables restored,
    messages stored outside the native driver)"""
    self._set_overlay_complete(False)

  def _set_overlay_complete(self, flag):
    self._overlay_complete = bool(flag)
    if self._native is not None:
      check(_lib.load().orca_engine_set_overlay_complete(self._native, int(flag)), "orca_engine_set_overlay_complete")

  # modules/memory.py:39-43, embedding_module.py:83-91
  def backup(self):
    self.sync_replicas()
    if self._hist_dense is None:        # the history lives in the slot cache: the rows the plan keeps are the backup
      sc = self._sc
      if sc is None:
        hist = ("slots", None)
      else:
        d = self._slot_desc()
        n, base = int(d.prev_n), 1 + int(d.cur) * sc["plan_cap"]
        hist = ("slots", dict(rows=sc["rows"][base:base + n].clone(), ids=sc["prev_plan"][:n].clone()))
    else:
      hist = [h.clone() for h in self._hist_dense]
    return ([t.clone() for t in (self.memory, self.last_update, self.messages, self.msg_ts, self.pending)],
            hist, [c.clone() for c in self.cache_flag])

  def restore(self, b):
    for t, s in zip((self.memory, self.last_update, self.messages, self.msg_ts, self.pending), b[0]):
      t.copy_(s)
    if isinstance(b[1], tuple) and b[1][0] == "slots":      # taken in slot mode: back into slot mode
      if self._hist_dense is not None:
        self._hist_dense = None
        if self._native is not None:
          self._rebind_native()
      self._restore_slot_backup(b[1][1])
    else:
      self._leave_slot_mode()
      for t, s in zip(self._hist_dense, b[1]):
        t.copy_(s)
    for t, s in zip(self.cache_flag, b[2]):
      t.copy_(s)
    self._sharded = False                  # a backup is taken from coherent replicas
    self._set_overlay_complete(False)      # the overlay rows of the restored pending ids are not part of a backup

  # -------------------------------------------------------------------------------------------
  def _slot_epoch(self, n):
    """n consecutive epochs of the `slot` election table (shared with the native driver)"""
    if self._native is not None:
      return int(_lib.load().orca_engine_take_epochs(self._native, int(n)))
    e = self.slot_epoch
    self.slot_epoch += int(n) + 1
    return e

  # ---- native step driver (csrc/engine.hip) -------------------------------------------------------
  def _ensure_native(self, Q):
    """(re)creates the orca_engine_t with work buffers for Q root queries (padded to the rank grid)"""
    per = (Q + self.world - 1) // self.world
    need = max(per * self.world, 1)
    if self._native is not None and need <= self._q_max:
      return
    L = _lib.load()
    if self._native is not None:
      self._unbind_slots()
      self.slot_epoch = int(L.orca_engine_take_epochs(self._native, 0))
      check(L.orca_engine_drain(self._native, stream()), "orca_engine_drain")   # the ids the last step stored
      self._overlay_complete = bool(L.orca_engine_get_overlay_complete(self._native))
      L.orca_engine_destroy(self._native)
      self._native = None
    dev, K = self.device, self.K
    q_max = max(need, 2 * self._q_max)
    self._w = dict(nodes=torch.zeros(q_max, dtype=torch.int64, device=dev),
                   qts=torch.zeros(q_max, dtype=torch.float64, device=dev),
                   nodes32=torch.zeros(q_max, dtype=torch.int32, device=dev),
                   nbr=torch.zeros(q_max * K, dtype=torch.int32, device=dev),
                   eid=torch.zeros(q_max * K, dtype=torch.int32, device=dev),
                   et=torch.zeros(q_max * K, dtype=torch.float32, device=dev),
                   dt=torch.zeros(q_max * K, dtype=torch.float32, device=dev),
                   emb=[torch.zeros(q_max, D, dtype=torch.float32, device=dev) for _ in range(2)],
                   winners=torch.zeros(2 * q_max, dtype=torch.uint8, device=dev),
                   eager_ids=torch.zeros(2 * q_max, dtype=torch.int32, device=dev),
                   eager_count=torch.zeros(2, dtype=torch.int32, device=dev))
    if self.world > 1:      # exchange buffers of the owner-computes step (include/orca_hip.h)
      per_max = q_max // self.world
      worst = max(2 * q_max // 3, 1)                       # every positive of the global batch on one owner
      cap = self.exchange_cap if self.exchange_cap else min(worst, int(1.25 * 2 * per_max / 3) + 256)
      self._xm_cap = int(cap)
      self._xm_chunk = int(L.orca_engine_xm_chunk_words(self._xm_cap))
      xm = torch.zeros(self._xm_chunk * self.world, dtype=torch.int32, device=dev)
      self._xl_chunk = int(L.orca_engine_xm_chunk_words(per_max))       # history-push records: at most a rank's 3*per rows
      xl = torch.zeros(self._xl_chunk * self.world, dtype=torch.int32, device=dev)
      # (a pretended rank 0 -- tools/slice_probe.py -- packs where the all-gather would deliver its chunk: at word 0 whatever
      # the step's row count; a real rank's chunk is the all-gather's input)
      alias = self._pretend and self.rank == 0
      own = xm[:self._xm_chunk] if alias else torch.zeros(self._xm_chunk, dtype=torch.int32, device=dev)
      xl_own = xl[:self._xl_chunk] if alias else torch.zeros(self._xl_chunk, dtype=torch.int32, device=dev)
      self._w.update(xm=xm, xm_own=own, xl=xl, xl_own=xl_own,
                     own_ids=torch.zeros(q_max, dtype=torch.int32, device=dev),
                     own_pos=torch.zeros(q_max, dtype=torch.int32, device=dev),
                     own_count=torch.zeros(1, dtype=torch.int32, device=dev))
      if getattr(self, "_err", None) is None:
        self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.slot_m = torch.zeros(self.N, dtype=torch.int64, device=dev)
    d = _lib.EngineDesc()
    d.n_nodes, d.n_layers, d.K, d.d_e = self.N, self.L, K, self.d_e
    d.small_batch = int(self.small_batch)
    d.csr = self.csr._h if self.csr is not None else None
    d.edge_feats, d.time_w, d.time_b = ptr(self.edge_feats), ptr(self.time_w), ptr(self.time_b)
    for i, b in enumerate(self.att_blobs):
      d.att_blob[i] = ptr(b)
    d.gru_blob = ptr(self.gru_blob)
    d.memory, d.last_update, d.messages = ptr(self.memory), ptr(self.last_update), ptr(self.messages)
    d.msg_ts, d.pending, d.overlay = ptr(self.msg_ts), ptr(self.pending), ptr(self.overlay)
    for i, h in enumerate(self._hist_dense or []):
      d.hist[i] = ptr(h)
    d.slot, d.worklist, d.count = ptr(self.slot), ptr(self.worklist), ptr(self.count)
    d.q_max = q_max
    w = self._w
    d.nodes, d.qts, d.nodes32 = ptr(w["nodes"]), ptr(w["qts"]), ptr(w["nodes32"])
    d.nbr, d.eid, d.et, d.dt = ptr(w["nbr"]), ptr(w["eid"]), ptr(w["et"]), ptr(w["dt"])
    d.emb[0], d.emb[1] = ptr(w["emb"][0]), ptr(w["emb"][1])
    d.winners = ptr(w["winners"])
    if self.world > 1:
      d.world, d.rank = self.world, self.rank
      d.xm, d.xm_cap = ptr(w["xm"]), self._xm_cap
      d.xm_own = ptr(w["xm_own"])
      d.xl, d.xl_own = ptr(w["xl"]), ptr(w["xl_own"])
      d.own_ids, d.own_pos, d.own_count = ptr(w["own_ids"]), ptr(w["own_pos"]), ptr(w["own_count"])
      d.slot_m, d.err = ptr(self.slot_m), ptr(self._err)
    d.eager_ids, d.eager_count = ptr(w["eager_ids"]), ptr(w["eager_count"])
    d.att_q_folded = 1     # load_state_dict completes every blob with orca_tgat_pack_gather
    h = _lib.vp()
    check(L.orca_engine_create(ctypes.byref(d), ctypes.byref(h)), "orca_engine_create")
    L.orca_engine_take_epochs(h, self.slot_epoch)       # continue after the epochs already used on `slot`
    L.orca_engine_set_overlay_complete(h, int(self._overlay_complete))
    self._native, self._q_max = h, q_max
    if self._prof_native:
      L.orca_engine_profile(h, self._prof_native)

  def profile_kernels(self, on=True, every=1):
    """bench.py: attach HIP events to the sampler / GRU / attention launches of every `every`-th native step"""
    self._prof_native = int(every) if on else 0
    if self._native is not None:
      _lib.load().orca_engine_profile(self._native, self._prof_native)

  def profile_read(self):
    """{kernel: (launches, mean ms)} measured since profile_kernels(True); waits for the events"""
    out = {}
    if self._native is None:
      return out
    for name, cls in (("sample_recent", 0), ("gru_update", 1), ("tgat_forward", 2)):
      n, ms = _lib.i64(), ctypes.c_double()
      check(_lib.load().orca_engine_profile_read(self._native, cls, ctypes.byref(n), ctypes.byref(ms)), "profile_read")
      if n.value:
        out[name] = (n.value, ms.value)
    return out

  def __del__(self):
    try:
      if self._native is not None:
        _lib.load().orca_engine_destroy(self._native)
        self._native = None
    except Exception:
      pass

  def _step_seed(self):
    """uniform sampling: the seed of the current batch (same on every rank and in the oracle)"""
    return (self.seed + self.batch_counter * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF

  def _step_reuse_native(self, src, dst, neg, ts, eidx, train, gather_output=True):
    """Path B + memory/messages through the C++ driver: begin -> (all-gather) -> layer ... -> finish"""
    self._await_packed()
    L = _lib.load()
    bs = src.numel()
    Q = 3 * bs
    owner = self.world > 1 and train
    self._ensure_native(3 * self.world * (-(-bs // self.world)) if owner else Q)
    lo, hi, per = self._slice(Q)
    st = stream()
    if self.sampling == "uniform":
      check(L.orca_engine_set_sampling(self._native, 1, self._step_seed()), "orca_engine_set_sampling")
    if owner:
      return self._step_owner_computes(L, src, dst, neg, ts, eidx, bs, st, gather_output)
    if self.world > 1:                # eval order: replicated row writes on coherent tables
      self.sync_replicas()
    check(L.orca_engine_begin(self._native, ptr(src), ptr(dst), ptr(neg), ptr(eidx), ptr(ts), bs, lo, hi, int(train), st),
          "orca_engine_begin")
    emb = self._w["emb"]
    for layer in range(2, self.L + 1):
      self._gather_in_place(emb[layer & 1], per)
      check(L.orca_engine_layer(self._native, layer, lo, hi, st), "orca_engine_layer")
    out = emb[(self.L + 1) & 1]
    if gather_output:
      self._gather_in_place(out, per)
    check(L.orca_engine_finish(self._native, ptr(src), ptr(dst), ptr(eidx), ptr(ts), st), "orca_engine_finish")
    return out[:Q] if gather_output else out[lo:hi]

  def _step_budget_native(self, src, dst, neg, ts, eidx, train, plan):
    """Path C through the C++ driver (orca_engine_step_budget): the whole batch step on the stream, no host round trip --
    the size of the uncached set stays on the device (the launches are sized for `budget_cap` of them)."""
    self._await_packed()
    L = _lib.load()
    bs = src.numel()
    Q, K, dev = 3 * bs, self.K, self.device
    self._raise_budget_overflow(wait=False)        # the previous step's flag, copied out while it ran
    cap = int(self.budget_cap) if self.budget_cap else max(1, min(Q * K, self.N))
    use_slots = self._hist_dense is None
    if use_slots and plan is None:          # flags accumulate (history.py:25-30 with cache_plan None): the dense table's job
      self._leave_slot_mode()
      use_slots = False
    self._ensure_native(Q + cap)
    if use_slots:
      self._slots_for(Q + cap, plan.numel(), Q * K)
    b = getattr(self, "_bw", None)
    if b is None or b["cap"] != cap or b["Q"] < Q:
      b = self._bw = dict(cap=cap, Q=Q, unc=torch.zeros(cap, dtype=torch.int64, device=dev),
                          cnt=torch.zeros(2, dtype=torch.int32, device=dev), tmax=torch.zeros(1, dtype=torch.float64, device=dev),
                          nbr=torch.zeros(Q * K, dtype=torch.int32, device=dev), eid=torch.zeros(Q * K, dtype=torch.int32, device=dev),
                          et=torch.zeros(Q * K, dtype=torch.float32, device=dev), dt=torch.zeros(Q * K, dtype=torch.float32, device=dev),
                          err=torch.zeros(1, dtype=torch.int32, device=dev))
    if getattr(self, "_bitmap", None) is None:
      self._bitmap = torch.zeros((self.N + 31) // 32, dtype=torch.int32, device=dev)
    out = torch.empty(Q, D, dtype=torch.float32, device=dev)
    check(L.orca_engine_step_budget(self._native, ptr(src), ptr(dst), ptr(neg), ptr(ts), ptr(eidx), bs, int(train),
                                    ptr(plan) if plan is not None and plan.numel() else None,
                                    int(plan.numel()) if plan is not None else -1, ptr(self.cache_flag[0]), ptr(self._bitmap),
                                    ptr(b["unc"]), cap, ptr(b["cnt"]), ptr(b["tmax"]), ptr(b["nbr"]), ptr(b["eid"]), ptr(b["et"]),
                                    ptr(b["dt"]), ptr(out), ptr(b["err"]), stream()), "orca_engine_step_budget")
    self._overlay_complete = bool(L.orca_engine_get_overlay_complete(self._native))
    # the overflow word travels to pinned host memory behind the step (no synchronisation here): whoever looks next
    # -- the next native step, check_errors() -- raises if it was set
    if getattr(self, "_berr_host", None) is None:
      self._berr_host = torch.zeros(1, dtype=torch.int32).pin_memory()
    self._berr_host.copy_(b["err"], non_blocking=True)
    self._berr_ev = torch.cuda.Event()
    self._berr_ev.record()
    self._berr_cap = cap
    return out

  def _raise_budget_overflow(self, wait):
    ev = getattr(self, "_berr_ev", None)
    if ev is None or not (wait or ev.query()):
      return
    if wait:
      ev.synchronize()
    if int(self._berr_host[0]) & 2:
      self._berr_ev = None
      raise OrcaError(f"OrcaEngine: a Path C batch had more than budget_cap={self._berr_cap} uncached out-of-batch neighbours; "
                      "only the first budget_cap were evaluated, so embeddings and history rows from that batch on differ from "
                      "the reference (embedding_module.py:251-280) -- leave budget_cap at None (the safe bound) or raise it")

  def budget_overflow(self):
    """True if a native Path C step met more uncached neighbours than `budget_cap` (it then evaluated the first
    budget_cap of them); reads a device flag (one sync).  The same condition raises from the next native step and
    from check_errors()."""
    b = getattr(self, "_bw", None)
    return bool(b is not None and int(b["err"].item()) & 2)

  def _step_owner_computes(self, L, src, dst, neg, ts, eidx, bs, st, gather_output):
    """Train-ordered step of several ranks (include/orca_hip.h, orca_engine_desc): this rank embeds the src, dst
    and neg roots of its share of the batch's interactions and does the batch's row writes for the ids it owns;
    two all-gathers per step carry the rows other ranks read.  The owner's work and the second exchange (updated
    memory rows) hide behind the first exchange (layer-1 rows for the history tables) and the attention of the
    later layers.  gather_output=False returns this rank's rows [src | dst | neg] of ITS interactions
    (dist.interaction_slice) -- what a data-parallel link decoder consumes."""
    lo, hi, per = dist_utils.interaction_slice(bs, self.rank, self.world)
    if not L.orca_engine_get_overlay_complete(self._native):
      self.sync_replicas()
      self._rebuild_overlay()
    check(L.orca_engine_begin(self._native, ptr(src), ptr(dst), ptr(neg), ptr(eidx), ptr(ts), bs, lo, hi, 1, st),
          "orca_engine_begin")
    emb = self._w["emb"]
    rows, push = ctypes.c_int64(0), ctypes.c_int64(0)
    if self.L >= 2:
      # Both exchanges of the step carry as many rows per rank as the fullest rank has (in 64s): every rank derives the same
      # numbers from the same global id list (begin()'s second kernel counted; a host wait for begin()'s first two kernels).
      check(L.orca_engine_push_rows(self._native, ctypes.byref(push)), "orca_engine_push_rows")
      self._push_rows_last = push.value
      push_words = int(L.orca_engine_xm_chunk_words(push.value))
    def finish():
      check(L.orca_engine_finish(self._native, ptr(src), ptr(dst), ptr(eidx), ptr(ts), st), "orca_engine_finish")
      # The exchange carries as many rows per rank as the fullest owner has (in 64s), not exchange_cap: every rank derives
      # the same number from the same global id list (orca_engine_xm_rows: a host wait for begin()'s first two kernels).
      check(L.orca_engine_xm_rows(self._native, ctypes.byref(rows)), "orca_engine_xm_rows")
      self._xm_rows_last = rows.value
      return self._gather_own_chunk(int(L.orca_engine_xm_chunk_words(rows.value)), async_op=True)
    wm = None
    for layer in range(2, self.L + 1):
      wh = self._gather_own_chunk(push_words, async_op=True, which="xl")     # the previous layer's push winners
      if layer == 2:
        wm = finish()
      if wh is not None:
        wh.wait()
      check(L.orca_engine_layer(self._native, layer, lo, hi, st), "orca_engine_layer")
    if self.L == 1:
      wm = finish()
    if wm is not None:
      wm.wait()
    check(L.orca_engine_apply(self._native, rows.value, st), "orca_engine_apply")
    self._sharded = True
    out = self._w["emb"][(self.L + 1) & 1]
    if gather_output:       # every rank's rows, then back into the batch's [src;dst;neg] order
      self._gather_chunks(out.view(-1), 3 * per * D)
      return out[dist_utils.interaction_root_index(bs, self.world, self.device)]
    return out[3 * per * self.rank:3 * per * self.rank + 3 * (hi - lo)]

  def _gather_own_chunk(self, words, async_op=False, which="xm"):
    """all-gather of the first `words` words of every rank's xm_own into xm (rank-major chunks of `words`); which="xl":
    of xl_own into xl"""
    full, own = self._w[which], self._w[which + "_own"]
    if self._pretend:       # (a timing aid: one rank of a pretended world -- its own chunk arrives, the others are stale)
      if self.rank != 0:
        full[self.rank * words:(self.rank + 1) * words].copy_(own[:words])
      return None
    return dist_utils.all_gather_into(full[:self.world * words], own[:words], self.group, async_op)

  def _gather_chunks(self, buf, chunk, async_op=False):
    """in-place all-gather of the first world x chunk elements of buf: rank r contributes chunk r"""
    if self._pretend:
      return None
    return dist_utils.all_gather_chunks(buf, chunk, self.group, async_op)

  def sync_replicas(self):
    """Makes the state tables identical on every rank again after owner-computes steps, during which the memory,
    last_update, messages and msg_ts rows of an id are only kept current by its owner (pending flags, overlay and
    history rows are replicated all along).  A transition, not a per-step cost: it runs before an eval-ordered
    step, a Python-dispatched path, a backup; every rank must call it."""
    if self.world == 1 or not self._sharded or self._pretend:
      self._sharded = False
      return
    dist_utils.sync_owned_rows((self.memory, self.last_update, self.messages, self.msg_ts), self.group)
    self._sharded = False

  def _rebuild_overlay(self):
    """overlay[id] = GRU(messages[id], memory[id]) for every pending id, on coherent replicas: re-establishes the
    overlay-complete state the owner-computes step starts from (after a restore, a weight refresh or
    Python-dispatched steps)."""
    if self.all_ids is None:
      self.all_ids = torch.arange(self.N, dtype=torch.int64, device=self.device)
    bound = self._build_worklist(None, self.all_ids, self.N)
    self._await_packed()
    ops.gru_update(self.gru_blob, self.msg_dim, self.worklist, bound, self.count, self.messages, self.memory, self.overlay)
    self._set_overlay_complete(True)

  def check_errors(self):
    """raises if a device-side error word is set (synchronises): an owner-computes exchange that overflowed, or a
    native Path C step that met more uncached neighbours than budget_cap"""
    self._raise_budget_overflow(wait=True)
    err = getattr(self, "_err", None)
    if err is not None and int(err.item()) & 1:
      raise OrcaError(f"OrcaEngine: a rank owned more than exchange_cap={self._xm_cap} of a batch's distinct positive "
                      "ids; results after that step are invalid -- construct the engine with a larger exchange_cap")

  def _gather_in_place(self, buf, per, async_op=False):
    """every rank's rows [rank*per, rank*per+per) of buf -> all ranks (RCCL all-gather, in place).  async_op:
    returns the work handle (None when the exchange already completed) -- wait() before reading the rows."""
    if self.world == 1 or self._pretend:
      return None
    import torch.distributed as dist
    full = buf[:self.world * per]
    mine = full[self.rank * per:(self.rank + 1) * per]
    if dist.get_backend(self.group) == "gloo":      # tests: gloo has no device all_gather
      hf = torch.empty(full.shape, dtype=full.dtype)
      dist.all_gather_into_tensor(hf, mine.cpu(), group=self.group)
      full.copy_(hf)
      return None
    work = dist.all_gather_into_tensor(full, mine, group=self.group, async_op=async_op)
    return work if async_op else None

  def _build_worklist(self, ids32, ids64, bound):
    """distinct ids of (ids32 ∪ ids64) that hold a pending message -> self.worklist[:count]."""
    ops.worklist_build(ids32, ids64, self.pending, self.slot, self._slot_epoch(bound), self.worklist, self.count)
    return min(int(bound), self.N)

  def _overlay_updated_memory(self, ids32, ids64):
    """memory_updater.get_updated_memory (memory_updater.py:56-78) restricted to the rows this batch
    reads: GRU(messages, memory) of every pending id is written to the overlay table."""
    n32 = ids32.numel() if ids32 is not None else 0
    n64 = ids64.numel() if ids64 is not None else 0
    bound = self._build_worklist(ids32, ids64, n32 + n64)
    self._await_packed()
    self._timed("gru_update", ops.gru_update, self.gru_blob, self.msg_dim, self.worklist, bound, self.count,
                self.messages, self.memory, self.overlay)

  def _persist_memory(self, ids64, overlay_covers=False):
    """memory_updater.update_memory (:25-39) + memory.clear_messages (memory.py:49-50).  overlay_covers: the overlay
    already holds GRU(message, memory) of every pending id of ids64 (a train-ordered batch on one rank: its roots are
    part of the rows it read), so the rows are committed by a copy -- as the native driver does; otherwise the GRU runs
    out of place into the overlay (the split kernels) and the rows are committed from there."""
    if overlay_covers:
      ops.persist_overlay(ids64, self.pending, self.overlay, self.memory, self.last_update, self.msg_ts)
    else:
      bound = self._build_worklist(None, ids64, ids64.numel())
      self._await_packed()
      ops.gru_update(self.gru_blob, self.msg_dim, self.worklist, bound, self.count, self.messages, self.memory, self.overlay)
      ops.commit_rows(self.worklist, bound, self.count, self.overlay, self.memory, self.last_update, self.msg_ts)
    ops.set_flags(self.pending, ids64=ids64, v=0)

  def _flush_all_pending(self):
    """memory_updater.update_memory_in_test (:42-53)."""
    self.sync_replicas()
    if self.all_ids is None:
      self.all_ids = torch.arange(self.N, dtype=torch.int64, device=self.device)
    bound = self._build_worklist(None, self.all_ids, self.N)
    self._await_packed()
    ops.gru_update(self.gru_blob, self.msg_dim, self.worklist, bound, self.count, self.messages, self.memory,
                   self.memory, self.last_update, self.msg_ts)
    ops.fill_u8(self.pending, 0)
    self._set_overlay_complete(True)       # no pending ids left

  def _sample(self, nodes, ts, q0=0):
    if self.sampling == "uniform":
      nbr, eid, _, dt = self._timed("sample_recent", ops.sample_uniform, self.csr, nodes, ts, self.K, self._step_seed(),
                                    q0=q0)
    else:
      nbr, eid, _, dt = self._timed("sample_recent", ops.sample_recent, self.csr, nodes, ts, self.K)
    return nbr, eid, dt

  def _dropout_kw(self, layer, q0):
    """dropout inputs of this batch's attention call of `layer` (train_step), {} when off"""
    if self.dropout <= 0.0:
      return {}
    return dict(dropout_p=self.dropout, dropout_q0=int(q0),
                dropout_seed=(self.seed * 0x2545F4914F6CDD1D + self.batch_counter * 0x9E3779B97F4A7C15 + layer) & 0xFFFFFFFFFFFFFFFF)

  def _wpq(self, n):
    """orca_tgat_inputs.waves_per_query of an attention launch of n queries: the rule of the native driver (engine.hip,
    waves_per_query), so that the dispatched launches take the same form"""
    return 4 if (self.small_batch and self.world == 1 and n <= 1024) else 0

  def _attend(self, layer, Q, nbr, eid, dt, train, src_table, src_idx, nbr_table, nbr_idx, from_memory, drop=None, save=None,
              shape_n=None):
    """embedding_module.py:318-329 (aggregate) fused with the gathers of :123,133,172-174 and the
    time encodings of :126,132.  shape_n: the query count the launch shape is chosen for (default Q)."""
    kw = dict(src_table=src_table, src_idx=src_idx, nbr_table=nbr_table, nbr_idx=nbr_idx, nbr_ids=nbr,
              edge_table=self.edge_feats, edge_idx=eid, dt=dt, time_w=self.time_w, time_b=self.time_b,
              q_folded=True, waves_per_query=0 if save is not None else self._wpq(Q if shape_n is None else shape_n))
    if from_memory and train:
      kw.update(src_overlay=self.overlay, src_flags=self.pending, nbr_overlay=self.overlay, nbr_flags=self.pending)
    if drop:
      kw.update(drop)
    self._await_packed()
    out, _ = self._timed("tgat_forward", ops.tgat_forward, self.att_blobs[layer - 1], self.d_e, Q, self.K, save=save, **kw)
    return out

  # -------------------------------------------------------------------------------------------
  def _slice(self, Q):
    """contiguous query slice of this rank"""
    return dist_utils.query_slice(Q, self.rank, self.world)

  def _all_gather_rows(self, x, Q, per):
    """x: this rank's [n_local, D] rows of a [Q, D] result -> the full [Q, D] on every rank."""
    if self.world == 1:
      return x
    return self._timed("all_gather", dist_utils.all_gather_rows, x, Q, per, self.group)

  # Path A: embedding_module.py:116-147
  def _embed_exact(self, nodes, ts, train):
    Q = nodes.numel()
    lo, hi, per = self._slice(Q)
    f_nodes, f_ts = [nodes[lo:hi]], [ts[lo:hi]]
    samples = []
    for _ in range(self.L):                           # top-down frontier expansion (:127-139)
      nbr, eid, dt = self._sample(f_nodes[-1], f_ts[-1])
      samples.append((nbr, eid, dt))
      f_nodes.append(torch.cat([f_nodes[-1], nbr.flatten().to(torch.int64)]))
      f_ts.append(torch.cat([f_ts[-1], f_ts[-1].repeat_interleave(self.K)]))
    if train:                                         # base case (:119-124)
      self._overlay_updated_memory(None, f_nodes[-1])
    emb = None
    for layer in range(1, self.L + 1):                # bottom-up aggregation (:142-146)
      lvl = self.L - layer
      nbr, eid, dt = samples[lvl]
      n = f_nodes[lvl].numel()
      if layer == 1:
        emb = self._attend(1, n, nbr, eid, dt, train, self.memory, f_nodes[lvl].to(torch.int32), self.memory, nbr,
                           True)
      else:
        emb = self._attend(layer, n, nbr, eid, dt, train, emb, None, emb[n:], None, False)
    return self._all_gather_rows(emb, Q, per)

  # Path B: embedding_module.py:154-182
  def _embed_reuse(self, nodes, ts, train):
    Q = nodes.numel()
    lo, hi, per = self._slice(Q)
    my_nodes = nodes[lo:hi]
    n = my_nodes.numel()
    nbr, eid, dt = self._sample(my_nodes, ts[lo:hi], q0=lo)
    if train:                                         # :159-161
      self._overlay_updated_memory(nbr, my_nodes)
    emb = self._attend(1, n, nbr, eid, dt, train, self.memory, my_nodes.to(torch.int32), self.memory, nbr, True)
    for layer in range(2, self.L + 1):                # :176-181
      full = self._all_gather_rows(emb, Q, per)       # history rows pushed by every rank
      h = self.hist[layer - 2]
      ops.history_push(h, full, nodes, self.slot, self._slot_epoch(Q))     # :103-113, history.py:38-40
      emb = self._attend(layer, n, nbr, eid, dt, train, emb, None, h, nbr, False)
    return self._all_gather_rows(emb, Q, per)

  # embedding_module.py:200-212
  def _find_uncached(self, nodes, nbr):
    if getattr(self, "_bitmap", None) is None:
      self._bitmap = torch.zeros((self.N + 31) // 32, dtype=torch.int32, device=self.device)
    return ops.find_uncached(nbr, nodes, self.cache_flag[0], self.slot, self._slot_epoch(1), self._bitmap)

  def _attend_partitioned(self, layer, n, nbr, eid, dt, train, src_table, src_idx, nbr_table, from_memory, drop=False,
                          shape_n=None):
    """_attend over this rank's contiguous share of the n queries, rows all-gathered back to [n, D].  drop: apply
    train_step's attention-weight dropout (keyed by the row's index among the n queries, whatever the slicing)."""
    if self.world == 1:
      return self._attend(layer, n, nbr, eid, dt, train, src_table, src_idx, nbr_table, nbr, from_memory,
                          drop=self._dropout_kw(layer, 0) if drop else None, shape_n=shape_n)
    lo, hi, per = self._slice(n)
    if src_idx is None:                               # dense source rows: row q belongs to query q
      src_table = src_table[lo:hi]
    else:
      src_idx = src_idx[lo:hi]
    part = self._attend(layer, hi - lo, nbr[lo:hi], eid[lo:hi], dt[lo:hi], train, src_table, src_idx, nbr_table,
                        nbr[lo:hi], from_memory, drop=self._dropout_kw(layer, lo) if drop else None)
    return self._all_gather_rows(part, n, per)

  # Path C: embedding_module.py:215-285.  Its control flow depends on global sets (which neighbours are cached), so
  # every rank samples all the queries and derives the same sets; the attention evaluations -- the cost -- are cut
  # into per-rank query slices and their rows all-gathered.
  def _embed_budget(self, nodes, ts, n_layers, cache_plan, train, rec=None, shape_n=None):
    """rec (train_step): a list that receives, innermost layer first, what the backward re-reads of every attention
    evaluation -- (layer, samples, source ids or rows, history table) -- and switches train_step's dropout on."""
    n = nodes.numel()
    nbr, eid, dt = self._sample(nodes, ts)
    if n_layers == 1:                                 # :224-246
      n32 = nodes.to(torch.int32)
      if train:
        self._overlay_updated_memory(nbr, nodes)
      if rec is not None:
        self._cnt_ready = self._fetch_count()
        rec.append((1, (nbr, eid, dt), (nodes, n32), None))
      return self._attend_partitioned(1, n, nbr, eid, dt, train, self.memory, n32, self.memory, True, drop=rec is not None,
                                      shape_n=shape_n)
    unc = self._find_uncached(nodes, nbr)             # :251
    if unc.numel():                                   # :253-260
      c_nodes = torch.cat([nodes, unc])
      c_ts = torch.cat([ts, ts.max().expand(unc.numel())])
    else:
      c_nodes, c_ts = nodes, ts
    # (the inner launch takes the shape the native driver gives it: chosen for the roots plus a nominal 256 uncached nodes)
    comb = self._embed_budget(c_nodes, c_ts, n_layers - 1, cache_plan, train, rec, shape_n=n + 256)   # :273
    h = self.hist[n_layers - 2]
    ops.history_push(h, comb, c_nodes, self.slot, self._slot_epoch(c_nodes.numel()))  # :186-197
    flag = self.cache_flag[n_layers - 2]
    if cache_plan is not None:                        # history.py:25-28
      ops.fill_u8(flag, 0)
      if cache_plan.numel():
        ops.set_flags(flag, ids64=cache_plan, v=1)
    else:                                             # history.py:30
      ops.set_flags(flag, ids64=c_nodes, v=1)
    if rec is not None:
      rec.append((n_layers, (nbr, eid, dt), comb, h))
    return self._attend_partitioned(n_layers, n, nbr, eid, dt, train, comb, None, h, False, drop=rec is not None)

  # -------------------------------------------------------------------------------------------
  def compute_temporal_embeddings(self, src, dst, neg, ts, eidx, train=True, cache_plan=None, gather_output=True):
    """model/tgn_model.py:95-147.  src/dst/neg/eidx int64 [bs], ts float64 [bs], device tensors
    (NumPy arrays are uploaded).  Returns the [3*bs, 100] embeddings of [src; dst; neg].
    gather_output=False (several ranks, Path B): the final embeddings stay sharded by query -- each rank returns
    the rows of its own slice (`query_slice`) and the closing all-gather is skipped; the state tables are
    replicated and identical on every rank either way."""
    dev = self.device
    src, dst, neg, eidx = (self._dev(x, torch.int64) for x in (src, dst, neg, eidx))
    ts = self._dev(ts, torch.float64)
    if self.csr is None:
      raise OrcaError("OrcaEngine: set_neighbor_finder() first")
    self.batch_counter += 1
    if train:                                          # :105-112
      self.test_mode = False
    elif not self.test_mode:
      self._flush_all_pending()
      self.test_mode = True
    one_layer_exact = not self.reuse and self.L == 1    # Path A with one layer has no recursion: the same step as
    if ((self.reuse and self.budget == 0) or one_layer_exact) and self.prof is None:    # Path B's first layer
      # end to end in the native driver (embedding, memory persist, message store): Path B, and the reference's
      # default configuration (n_layer = 1, no reuse: BASELINE config 1)
      return self._step_reuse_native(src, dst, neg, ts, eidx, train, gather_output)
    if self.sampling == "uniform" and not (self.reuse and self.budget == 0):
      raise OrcaError("OrcaEngine: uniform sampling is wired into Path B (--reuse) and the one-layer exact path only")
    if (self.reuse and self.budget != 0 and self.L == 2 and self.world == 1 and self.prof is None and self.native_budget):
      # Path C end to end in the native driver (embedding_module.py:215-285 + tgn_model.py:127-145)
      self.sync_replicas()
      plan = self._dev(cache_plan, torch.int64) if cache_plan is not None else None
      return self._step_budget_native(src, dst, neg, ts, eidx, train, plan)
    self.sync_replicas()
    self._leave_slot_mode()                # (they work on the dense history tables)
    self._set_overlay_complete(False)      # the Python-dispatched paths store messages without overlay rows
    nodes = torch.cat([src, dst, neg])
    t3 = torch.cat([ts, ts, ts])
    positives = nodes[:2 * src.numel()]
    if self.reuse and self.budget != 0:                # :114-120
      plan = self._dev(cache_plan, torch.int64) if cache_plan is not None else None
      emb = self._embed_budget(nodes, t3, self.L, plan, train)
    elif self.reuse:
      emb = self._embed_reuse(nodes, t3, train)
    else:
      emb = self._embed_exact(nodes, t3, train)
    if train:                                          # :127-129
      self._persist_memory(positives, overlay_covers=self.world == 1)
    ops.message_store(src, dst, eidx, ts, self.memory, self.last_update, self.edge_feats, self.time_w, self.time_b,
                      self.slot, self._slot_epoch(2 * src.numel()), self.messages, self.msg_ts, self.pending)  # :131-141
    if not train:                                      # :143-145
      self._persist_memory(positives)
    return emb

  def compute_edge_probabilities(self, src, dst, neg, ts, eidx, train=True, cache_plan=None):
    """model/tgn_model.py:150-157; the link decoder (MergeLayer utils/util.py:12-24) is a plain
    library GEMM outside the hot path."""
    emb = self.compute_temporal_embeddings(src, dst, neg, ts, eidx, train, cache_plan)
    self.last_embeddings = emb
    n = emb.shape[0] // 3
    se, de, ne = emb[:n], emb[n:2 * n], emb[2 * n:]
    x = torch.cat([torch.cat([se, se]), torch.cat([de, ne])], dim=1)
    h = torch.relu(x @ self.aff["fc1.weight"].t() + self.aff["fc1.bias"])
    s = h @ self.aff["fc2.weight"].t() + self.aff["fc2.bias"]
    return s[:n].sigmoid(), s[n:].sigmoid()

  # -------------------------------------------------------------------------------------------
  def train_step(self, src, dst, neg, ts, eidx, cache_plan=None):
    """One training batch of train.py:200-246 up to (not including) optimizer.step(): the train-ordered forward
    of compute_edge_probabilities (tgn_model.py:95-157), the loss BCE(pos, 1) + BCE(neg, 0) (train.py:231-233) and
    its backward.  Path A (exact recursion), Path B (`reuse`) and Path C (`reuse` with a budget: pass the batch's
    cache_plan as to compute_temporal_embeddings); attention-weight dropout as the engine was built
    (dropout=..., train.py:30; counter RNG, see include/orca_hip.h).  Returns
    (loss, grads) with grads keyed like the state_dict (parameters the loss does not reach are absent); the state
    advances exactly as in compute_temporal_embeddings(train=True).  Gradients flow as upstream: through the
    attention layers (Path A: also through the inner-layer embeddings of the sampled neighbours) into the
    updated-memory rows of pending ids and from there into the GRU; history rows, stored messages and the memory
    table are detached (history.py:32-40: push / pull run under no_grad in the default configuration, without
    train.py:43's --gradient switch: OrcaEngine(gradient=True); memory.py:45-47).  An optimizer updates self.params in place; call
    refresh() afterwards.  With several ranks every rank back-propagates its query slice and the parameter
    gradients are all-reduced.  The link decoder (two small dense layers) and the loss run
    forward and backward in orca_decoder_train (csrc/decoder.hip)."""
    if self.reuse and self.budget != 0 and self.gradient:
      raise OrcaError("train_step: --gradient is wired into Path B (budget 0) only")
    if self.csr is None:
      raise OrcaError("OrcaEngine: set_neighbor_finder() first")
    dev = self.device
    src, dst, neg, eidx = (self._dev(x, torch.int64) for x in (src, dst, neg, eidx))
    ts = self._dev(ts, torch.float64)
    self.batch_counter += 1
    self.test_mode = False
    self.sync_replicas()
    self._leave_slot_mode()                # (train_step's forward reads the dense history tables)
    self._set_overlay_complete(False)
    bs = src.numel()
    src, dst, neg, ts, eidx = src.contiguous(), dst.contiguous(), neg.contiguous(), ts.contiguous(), eidx.contiguous()
    native_fwd = self.reuse and self._train_native_ok()      # (the driver's first kernel also builds the roots)
    if native_fwd:
      nodes = nodes32 = t3 = my_nodes = None
      Q = 3 * bs
    else:
      nodes, nodes32, t3 = ops.build_roots(src, dst, neg, ts)
      Q = nodes.numel()
    lo, hi, per = self._slice(Q)
    if nodes is not None:
      my_nodes = nodes[lo:hi]
    n = hi - lo
    K = self.K
    # ---- forward, keeping what the backward re-reads (the tables are not touched before it has run).  steps:
    # per attention call (layer, rows, sample, inputs of orca_tgat_inputs that differ between the paths) ----
    steps = []
    out = None
    if self.reuse and self.budget != 0:                  # embedding_module.py:215-285 (Path C)
      # The forward is _embed_budget as in compute_temporal_embeddings.  Every layer is evaluated for the batch's Q
      # roots followed by the uncached out-of-batch neighbours it recomputes; those extra rows only feed the history
      # tables, which are detached (history.py:32-40), so the loss reaches exactly the FIRST Q rows of every layer:
      # the backward runs over this rank's slice of them.
      if self.sampling == "uniform":
        raise OrcaError("OrcaEngine: uniform sampling is wired into Path B (--reuse) and the one-layer exact path only")
      rec = []
      plan = self._dev(cache_plan, torch.int64) if cache_plan is not None else None
      self._await_packed()
      out = self._embed_budget(nodes, t3, self.L, plan, True, rec=rec)
      cnt_ready = self._cnt_ready
      sl = slice(lo, hi)
      for layer, (nbr_l, eid_l, dt_l), src_l, h in rec:
        dk = self._dropout_kw(layer, lo)
        smp = (nbr_l[sl], eid_l[sl], dt_l[sl])
        if layer == 1:
          ids64, ids32 = src_l
          steps.append((1, n, smp, dict(src_table=self.memory, src_idx=ids32[sl], nbr_table=self.memory, nbr_idx=smp[0], **dk),
                        (ids64[sl], smp[0].reshape(-1))))
        else:
          steps.append((layer, n, smp, dict(src_table=src_l[sl], nbr_table=h, nbr_idx=smp[0], **dk), None))
    elif native_fwd:                                     # the same forward issued by the C++ driver (below)
      my32, nbr, eid, dt, embs, sv, cnt_ready = self._train_forward_native(src, dst, neg, ts, eidx)
      nodes = my_nodes = self._w["nodes"][:Q]
      steps.append((1, n, (nbr, eid, dt), dict(src_table=self.memory, src_idx=my32, nbr_table=self.memory, nbr_idx=nbr,
                                               saved=sv[0], **self._dropout_kw(1, 0)), (my_nodes, nbr.view(-1))))
      for layer in range(2, self.L + 1):
        steps.append((layer, n, (nbr, eid, dt), dict(src_table=embs[layer - 2], nbr_table=self.hist[layer - 2], nbr_idx=nbr,
                                                     saved=sv[layer - 1], **self._dropout_kw(layer, 0)), None))
      emb = embs[self.L - 1]
    elif self.reuse:                                     # embedding_module.py:154-182
      nbr, eid, dt = self._sample(my_nodes, t3[lo:hi], q0=lo)
      self._overlay_updated_memory(nbr, my_nodes)
      cnt_ready = self._fetch_count()
      my32 = nodes32[lo:hi]
      dk = self._dropout_kw(1, lo)
      sv = []       # (the forward keeps q~, the softmax weights and the key sums for the backward: ops.tgat_forward(save=))
      emb = self._attend(1, n, nbr, eid, dt, True, self.memory, my32, self.memory, nbr, True, drop=dk, save=sv)
      steps.append((1, n, (nbr, eid, dt), dict(src_table=self.memory, src_idx=my32, nbr_table=self.memory, nbr_idx=nbr,
                                               saved=sv[0] if sv else None, **dk),
                    (my_nodes, nbr.view(-1))))
      for layer in range(2, self.L + 1):
        full = self._all_gather_rows(emb, Q, per)
        ep = self._slot_epoch(Q)
        ops.history_push(self.hist[layer - 2], full, nodes, self.slot, ep)
        if self.gradient and layer == 2:      # which ids this batch pushed, and which root position won each (SURVEY H7:
          if getattr(self, "_pushed", None) is None:                       # the push stamped slot[id] = ep + position)
            self._pushed = torch.zeros(self.N, dtype=torch.uint8, device=dev)
            self._wpos = torch.zeros(self.N, dtype=torch.int32, device=dev)
          ops.fill_u8(self._pushed, 0)
          ops.set_flags(self._pushed, ids64=nodes, v=1)
          self._wpos[nodes] = (self.slot[nodes] - ep).to(torch.int32)
        prev = emb
        dk = self._dropout_kw(layer, lo)
        sv = []
        emb = self._attend(layer, n, nbr, eid, dt, True, prev, None, self.hist[layer - 2], nbr, False, drop=dk, save=sv)
        steps.append((layer, n, (nbr, eid, dt), dict(src_table=prev, nbr_table=self.hist[layer - 2], nbr_idx=nbr,
                                                     saved=sv[0] if sv else None, **dk), None))
    else:                                                # embedding_module.py:116-147
      f_nodes, f_ts, samples = [my_nodes], [t3[lo:hi]], []
      for _ in range(self.L):
        smp = self._sample(f_nodes[-1], f_ts[-1])
        samples.append(smp)
        f_nodes.append(torch.cat([f_nodes[-1], smp[0].flatten().to(torch.int64)]))
        f_ts.append(torch.cat([f_ts[-1], f_ts[-1].repeat_interleave(K)]))
      self._overlay_updated_memory(None, f_nodes[-1])
      cnt_ready = self._fetch_count()
      emb = None
      for layer in range(1, self.L + 1):
        lvl = self.L - layer
        nbr, eid, dt = samples[lvl]
        m = f_nodes[lvl].numel()
        if layer == 1:
          i32 = f_nodes[lvl].to(torch.int32)
          dk = self._dropout_kw(1, 0)      # (the exact recursion numbers its frontier per rank)
          sv = []
          emb = self._attend(1, m, nbr, eid, dt, True, self.memory, i32, self.memory, nbr, True, drop=dk, save=sv)
          steps.append((1, m, samples[lvl], dict(src_table=self.memory, src_idx=i32, nbr_table=self.memory, nbr_idx=nbr,
                                                 saved=sv[0] if sv else None, **dk),
                        (f_nodes[lvl], nbr.view(-1))))
        else:
          prev = emb
          dk = self._dropout_kw(layer, 0)
          sv = []
          emb = self._attend(layer, m, nbr, eid, dt, True, prev, None, prev[m:], None, False, drop=dk, save=sv)
          steps.append((layer, m, samples[lvl], dict(src_table=prev, nbr_table=prev[m:], saved=sv[0] if sv else None, **dk), "cat"))
    if out is None:
      out = self._all_gather_rows(emb, Q, per)
    # ---- link decoder + loss, forward and backward (tgn_model.py:150-157, train.py:231-233): orca_decoder_train ----
    # (every gradient buffer of the step out of one zero fill, sized by what the previous step asked for)
    pool = ops.ZeroPool(getattr(self, "_grad_pool_floats", 0), dev)
    loss, d_out, ga, _ = ops.decoder_train(out, self.aff["fc1.weight"], self.aff["fc1.bias"], self.aff["fc2.weight"],
                                           self.aff["fc2.bias"], pool=pool)
    loss = loss.reshape(())
    grads = {"affinity_score." + k: v for k, v in ga.items()}
    # ---- backward through the attention calls, last to first (this rank's slice) ----
    d = d_out[lo:hi].contiguous()
    want_t = self.train_time_encoder       # (frozen upstream, model/time_encoding.py:20-21: its gradient is not computed)
    d_time = torch.zeros(2, D, dtype=torch.float32, device=dev) if want_t else None
    into_memory = None
    # The parameter gradients (reductions over the batch's queries: nothing downstream in the backward reads them) are
    # issued on the second stream once the last attention layer's data gradients are enqueued, and run beside the
    # scatter into the memory rows, the GRU's backward and the state update; joined before the gradients are handed out.
    side = self._side_stream() if self.overlap else None
    deferred = [] if side is not None else None
    issued, time_grads = [], []

    def issue_deferred():
      ev = torch.cuda.Event()
      ev.record()
      with torch.cuda.stream(side):
        side.wait_event(ev)
        for f in deferred:
          f()
      issued.extend(deferred)
      deferred.clear()
    for layer, m, (nbr, eid, dt), tables, sink in reversed(steps):
      p = f"embedding_module.attention_models.{layer - 1}."
      sd = self._layer_params()[layer - 1]
      kw = dict(nbr_ids=nbr, edge_table=self.edge_feats, edge_idx=eid, dt=dt, time_w=self.time_w, time_b=self.time_b)
      kw.update(tables)
      saved = kw.pop("saved", None)
      if layer == 1:
        kw.update(src_overlay=self.overlay, src_flags=self.pending, nbr_overlay=self.overlay, nbr_flags=self.pending)
      # (sink None: the neighbour rows are detached history rows -- nobody reads their gradient unless --gradient)
      g, d_src, d_nbr, _, _, tw, tb = ops.tgat_backward(sd, self.d_e, m, K, d, want_time_param_grads=want_t, pool=pool, saved=saved,
                                                        want_d_nbr=sink is not None or self.gradient, deferred=deferred, **kw)
      for k, v in g.items():                           # Path A evaluates a layer once per batch, like Path B
        grads[p + k] = v
      if want_t:
        time_grads.append((tw, tb))
      if sink == "cat":                                # Path A: the neighbours' rows are embeddings of the level below
        d = torch.cat([d_src, d_nbr])
      elif sink is None:                               # Path B inner layer: history rows are detached (history.py:32-40),
        d = d_src                                      # except, with --gradient, those pushed in this batch (:42-46)
        if self.gradient:
          extra = torch.zeros(Q, D, dtype=torch.float32, device=dev)
          ops.scatter_add_flagged(d_nbr, nbr.view(-1), self._pushed, self._wpos, extra)
          if self.world > 1:
            dist_utils.all_reduce_sum(extra.view(-1), self.group)
          d = d_src + extra[lo:hi]
      else:
        into_memory = (sink, d_src, d_nbr)
    if deferred:
      issue_deferred()
    # ---- into the updated-memory rows of the pending ids this slice read, then through the GRU ----
    p = "memory_updater.memory_updater."
    cnt_ready.synchronize()                # distinct pending ids of the overlay work-list: the copy was enqueued right
    cnt = int(self._count_host[0])         # after the forward set it, so this wait does not drain the queue
    if cnt:
      if getattr(self, "_pos", None) is None:
        self._pos = torch.zeros(self.N, dtype=torch.int32, device=dev)
      wl = self.worklist[:cnt]
      ops.index_positions(wl, cnt, None, self._pos)
      d_upd = pool.take(cnt * D).view(cnt, D)
      (ids_src, ids_nbr), d_src, d_nbr = into_memory
      ops.scatter_add_flagged_pair(d_src, ids_src, d_nbr, ids_nbr, self.pending, self._pos, d_upd)
      gg = ops.gru_backward(self.params[p + "weight_ih"], self.params[p + "weight_hh"], self.params[p + "bias_ih"],
                            self.params[p + "bias_hh"], wl, self.messages, self.memory, d_upd, pool=pool, deferred=deferred,
                            gates=self._step_gates if native_fwd else None)
      for k, v in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), gg[:4]):
        grads[p + k] = v
      if deferred:
        issue_deferred()
    else:
      for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
        grads[p + k] = torch.zeros_like(self.params[p + k])
    # ---- the state moves on (tgn_model.py:127-141) ----
    positives = nodes[:2 * bs]
    if native_fwd:           # persist from the overlay + raw-message store with the winners begin() elected: two launches
      check(_lib.load().orca_engine_finish(self._native, ptr(src), ptr(dst), ptr(eidx), ptr(ts), stream()), "orca_engine_finish")
    else:
      if self.world > 1:     # this rank's forward put GRU(message, memory) of ITS pending roots into the overlay: the
        self._overlay_updated_memory(None, positives)  # other ranks' positives follow (same kernel: same bits as one rank)
      ops.persist_overlay(positives, self.pending, self.overlay, self.memory, self.last_update, self.msg_ts)
      ops.set_flags(self.pending, ids64=positives, v=0)
      ops.message_store(src, dst, eidx, ts, self.memory, self.last_update, self.edge_feats, self.time_w, self.time_b,
                        self.slot, self._slot_epoch(2 * bs), self.messages, self.msg_ts, self.pending)
    self._grad_pool_floats = pool.asked
    if side is not None:                   # the parameter gradients are complete from here on
      torch.cuda.current_stream().wait_stream(side)
      issued.clear()
    if want_t:
      for tw, tb in time_grads:
        d_time[0] += tw
        d_time[1] += tb
      grads["time_encoder.w.weight"] = d_time[0].view(D, 1)
      if "time_encoder.w.bias" in self.params:
        grads["time_encoder.w.bias"] = d_time[1]
    if self.world > 1:                     # every rank holds the decoder's full gradient already
      keys = sorted(k for k in grads if not k.startswith("affinity_score."))
      flat = torch.cat([grads[k].reshape(-1) for k in keys])
      dist_utils.all_reduce_sum(flat, self.group)
      o = 0
      for k in keys:
        m = grads[k].numel()
        grads[k] = flat[o:o + m].view_as(grads[k])
        o += m
    return loss, grads

  def _train_native_ok(self):
    """Path B's training forward through the C++ driver: one rank, at most two layers (the driver's two embedding
    buffers alternate: a third layer would overwrite the first layer's output, which the backward of layer 2 reads),
    no --gradient bookkeeping, no per-kernel host timing."""
    return (self.native_train and self.world == 1 and self.L <= 2 and not self.gradient and self.prof is None
            and self.budget == 0)

  def _train_forward_native(self, src, dst, neg, ts, eidx):
    """The forward of a Path B training batch as the native step issues it (orca_engine_begin / orca_engine_layer with
    orca_engine_train_capture): roots, sampling, the work-list claims and both last-occurrence elections in ONE launch
    (k_begin_queries) instead of six, the history push inside attention layer 1; the attention calls keep what the
    backward reads and apply dropout, and no table row is written -- train_step ends with orca_engine_finish.
    Returns (root ids int32, nbr, eid, dt, [layer outputs], [saved buffers], the count-copy event)."""
    L = _lib.load()
    bs = src.numel()
    Q, K = 3 * bs, self.K
    self._ensure_native(Q)
    self._await_packed()
    st = stream()
    if self.sampling == "uniform":
      check(L.orca_engine_set_sampling(self._native, 1, self._step_seed()), "orca_engine_set_sampling")
    n_sv = L.orca_tgat_saved_floats(self.d_e, Q, K)
    cache = getattr(self, "_cap_saved", None)
    if cache is None or cache[0] != (Q, self.L):
      cache = self._cap_saved = ((Q, self.L), [torch.empty(n_sv, dtype=torch.float32, device=self.device) for _ in range(self.L)])
    sv = cache[1]
    saved_p = (_lib.vp * self.L)(*[ptr(t) for t in sv])
    seeds = (ctypes.c_uint64 * self.L)(*[self._dropout_kw(l + 1, 0).get("dropout_seed", 0) for l in range(self.L)])
    gates = None
    if self.d_e == 172:      # the work-list's GRU keeps its gates for the backward (orca_gru_update_saving)
      need = min(Q * (K + 1), self.N) * 4 * D
      if getattr(self, "_gates", None) is None or self._gates.numel() < need:
        self._gates = torch.empty(need, dtype=torch.float32, device=self.device)
      gates = self._gates
    self._step_gates = gates
    check(L.orca_engine_train_capture(self._native, 1, saved_p, float(self.dropout), seeds, ptr(gates)), "orca_engine_train_capture")
    try:
      check(L.orca_engine_begin(self._native, ptr(src), ptr(dst), ptr(neg), ptr(eidx), ptr(ts), bs, 0, Q, 1, st), "orca_engine_begin")
      cnt_ready = self._fetch_count()
      for layer in range(2, self.L + 1):
        check(L.orca_engine_layer(self._native, layer, 0, Q, st), "orca_engine_layer")
    finally:
      L.orca_engine_train_capture(self._native, 0, None, 0.0, None, None)
    w = self._w
    view = lambda t: t[:Q * K].view(Q, K)
    embs = [w["emb"][l & 1][:Q] for l in range(self.L)]
    return w["nodes32"][:Q], view(w["nbr"]), view(w["eid"]), view(w["dt"]), embs, sv, cnt_ready

  def _fetch_count(self):
    """starts the copy of the work-list count to pinned host memory; the returned event completes with it"""
    if getattr(self, "_count_host", None) is None:
      self._count_host = torch.zeros(1, dtype=torch.int32).pin_memory()
    self._count_host.copy_(self.count, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    return ev

  def optimizer(self, lr=1e-4, fused=True):
    """Adam over the raw parameter tensors (train.py:180): updates self.params in place; step_optimizer() feeds it
    train_step's gradients and re-packs the kernels' blobs.  fused: one launch per step (ops.FusedAdam over
    orca_adam_step, torch.optim.Adam's arithmetic); fused=False: torch.optim.Adam itself (seven launches and 0.3 ms of
    host time per step)."""
    trainable = {k: v for k, v in self.params.items() if v.is_floating_point() and
                 (self.train_time_encoder or not k.startswith("time_encoder."))}
    if fused:
      self._opt_params = None
      self._opt = ops.FusedAdam(trainable, lr=lr)
    else:
      self._opt_params = {k: torch.nn.Parameter(v) for k, v in trainable.items()}
      self._opt = torch.optim.Adam(self._opt_params.values(), lr=lr)
    return self._opt

  def step_optimizer(self, grads):
    """optimizer.step() of train.py:238 with the gradients of train_step, then refresh()."""
    if self._opt_params is None:
      self._opt.step(grads)
    else:
      for k, p in self._opt_params.items():
        g = grads.get(k)
        p.grad = None if g is None else g.reshape(p.shape)
      self._opt.step()
    self.refresh(beside=True, _time_unchanged=not self.train_time_encoder)

  def _dev(self, x, dtype):
    if not isinstance(x, torch.Tensor):
      x = torch.from_numpy(np.ascontiguousarray(x))
    return x.to(device=self.device, dtype=dtype).contiguous()
