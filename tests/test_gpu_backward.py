"""Backward of the attention layer (orca_tgat_backward through the C ABI) against torch autograd of the oracle's
forward.  The reference holds no gradient fixtures and can no longer be run:
the oracle's forward is pinned to the reference's outputs, its autograd is the checker here -- gradient parity is
"unpinned" by a reference vector.  Bar: 1e-5 relative (normwise and against the largest element), fp32.
Needs a real MI355X."""
import numpy as np
import pytest
import torch

from conftest import assert_close_fp32
from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch.set_num_threads(1)
DEV = torch.device("cuda:0")
TOL = 1e-5


@pytest.fixture(scope="module")
def ops():
  from orca_amd import ops as _ops
  return _ops


def _sd(g, tag):
  return {k[len(tag + "sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "sd.")}


@pytest.mark.parametrize("tag", ["de172_K10_", "de1_K10_", "de4_K20_", "de172_K1_"])
def test_attention_layer_backward_dense(golden, tag):
  """loss.backward() through the drop-in TemporalAttentionLayer: gradients of the twelve parameters and of the
  four differentiable inputs, on the reference's own forward fixtures (which include all-padded rows)."""
  import orca_amd.api as api
  g = golden("attention.npz")
  sd = _sd(g, tag)
  d_e = g[tag + "edge"].shape[2]
  B, K = g[tag + "mask_in"].shape
  R = torch.from_numpy(np.random.RandomState(B + K).randn(B, 100).astype(np.float32))
  names = ["src", "src_t", "nbr", "nbr_t"]
  # oracle: torch autograd on the CPU
  P = {k: v.clone().requires_grad_() for k, v in sd.items()}
  xin = {k: torch.from_numpy(g[tag + k]).clone().requires_grad_() for k in names}
  out, _ = O.attention_forward(P, xin["src"], xin["src_t"], xin["nbr"], xin["nbr_t"], torch.from_numpy(g[tag + "edge"]),
                               torch.from_numpy(g[tag + "mask_in"].copy()))
  (out * R).sum().backward()
  # product
  layer = api.TemporalAttentionLayer(100, 100, d_e, 100, 100, n_head=2, dropout=0.1)
  layer.load_state_dict(sd, strict=True)
  layer = layer.to(DEV).eval()
  yin = {k: torch.from_numpy(g[tag + k]).to(DEV).requires_grad_() for k in names}
  mask = torch.from_numpy(g[tag + "mask_in"].copy()).to(DEV)
  got, w = layer(yin["src"], yin["src_t"], yin["nbr"], yin["nbr_t"], torch.from_numpy(g[tag + "edge"]).to(DEV), mask)
  assert_close_fp32(got.detach().cpu().numpy(), g[tag + "out"], 1e-5, tag + "out under autograd")
  assert np.array_equal(mask.cpu().numpy(), g[tag + "mask_out"])
  assert not w.requires_grad
  (got * R.to(DEV)).sum().backward()
  named = dict(layer.named_parameters())
  for k in sd:
    ref = P[k].grad.numpy()
    if k == "multi_head_target.in_proj_bias":
      # the key bias gets no gradient (softmax is shift-invariant): autograd leaves rounding noise there
      assert float(np.abs(ref[200:400]).max()) <= 1e-5 * float(np.abs(ref).max())
      ref = ref.copy()
      ref[200:400] = 0.0
    assert_close_fp32(named[k].grad.cpu().numpy(), ref, TOL, tag + "grad " + k)
  for k in names:
    assert_close_fp32(yin[k].grad.cpu().numpy(), xin[k].grad.numpy(), TOL, tag + "grad input " + k)


@pytest.mark.parametrize("d_e,K,Q", [(172, 10, 700), (1, 10, 333), (64, 7, 130), (172, 20, 45)])
def test_attention_backward_gather_mode(ops, golden, d_e, K, Q):
  """The engine's mode: rows gathered by index with overlay flags, time encodings from dt with trainable
  (w, b) -- per-slot row gradients and the time encoder's gradients."""
  g = golden("attention.npz")
  sd = _sd(g, "de172_K10_")
  torch.manual_seed(d_e + K)
  kd = 200 + d_e
  sd["multi_head_target.k_proj_weight"] = torch.randn(200, kd) * (1.0 / kd ** 0.5)
  sd["multi_head_target.v_proj_weight"] = torch.randn(200, kd) * (1.0 / kd ** 0.5)
  rs = np.random.RandomState(Q)
  N, E = 300, 2000
  table = rs.randn(N, 100).astype(np.float32)
  overlay = rs.randn(N, 100).astype(np.float32)
  flags = (rs.rand(N) < 0.3).astype(np.uint8)
  edge_feats = rs.randn(E, d_e).astype(np.float32)
  nodes = rs.randint(0, N, Q).astype(np.int32)
  nbr = rs.randint(1, N, (Q, K)).astype(np.int32)
  eid = rs.randint(1, E, (Q, K)).astype(np.int32)
  npad = rs.randint(0, K + 1, Q)
  npad[:3] = K
  for q in range(Q):
    nbr[q, :npad[q]] = 0
    eid[q, :npad[q]] = 0
  dt = rs.uniform(0, 5e4, (Q, K)).astype(np.float32)
  w_time = O.time_encode_weights(100)
  b_time = torch.from_numpy(rs.uniform(-3, 3, 100).astype(np.float32))
  R = torch.from_numpy(rs.randn(Q, 100).astype(np.float32))
  # oracle
  P = {k: v.clone().requires_grad_() for k, v in sd.items()}
  eff = torch.from_numpy(np.where(flags[:, None] != 0, overlay, table))
  nbr_t = torch.from_numpy(nbr.astype(np.int64))
  src_rows = eff[torch.from_numpy(nodes.astype(np.int64))].clone().requires_grad_()
  nbr_rows = eff[nbr_t.flatten()].view(Q, K, 100).clone().requires_grad_()
  wt, bt = w_time.clone().requires_grad_(), b_time.clone().requires_grad_()
  out, _ = O.attention_forward(P, src_rows, O.time_encode(torch.zeros(Q, 1), wt, bt), nbr_rows,
                               O.time_encode(torch.from_numpy(dt), wt, bt),
                               torch.from_numpy(edge_feats)[torch.from_numpy(eid.astype(np.int64))], nbr_t == 0)
  (out * R).sum().backward()
  # product
  _t = lambda a: torch.from_numpy(a).to(DEV)
  dsd = {k: v.to(DEV) for k, v in sd.items()}
  inputs = dict(src_table=_t(table), src_idx=_t(nodes), src_overlay=_t(overlay), src_flags=_t(flags), nbr_table=_t(table),
                nbr_idx=_t(nbr), nbr_overlay=_t(overlay), nbr_flags=_t(flags), nbr_ids=_t(nbr), edge_table=_t(edge_feats),
                edge_idx=_t(eid), dt=_t(dt), time_w=w_time.to(DEV), time_b=b_time.to(DEV))
  # twice: the backward recomputing the attention core, and reading what a forward call kept for it
  # (orca_tgat_forward_saving -> orca_tgat_backward_saved; with the q projection folded into the blob and without)
  for mode in ("recompute", "saved folded", "saved"):
    saved = None
    if mode != "recompute":
      folded = mode == "saved folded"
      blob = ops.tgat_pack(dsd, d_e, DEV, time_b=b_time.to(DEV) if folded else None)
      sv = []
      got, _ = ops.tgat_forward(blob, d_e, Q, K, save=sv, q_folded=folded, **inputs)
      assert_close_fp32(got.cpu().numpy(), out.detach().numpy(), TOL, mode + " forward")
      saved = sv[0]
    grads, d_src, d_nbr, d_st, d_nt, d_tw, d_tb = ops.tgat_backward(dsd, d_e, Q, K, R.to(DEV), want_time_param_grads=True,
                                                                    saved=saved, **inputs)
    assert d_st is None and d_nt is None
    for k in sd:
      ref = P[k].grad.numpy().copy()
      if k == "multi_head_target.in_proj_bias":
        ref[200:400] = 0.0
      assert_close_fp32(grads[k].cpu().numpy(), ref, TOL, f"gather grad {k} ({mode})")
    assert_close_fp32(d_src.cpu().numpy(), src_rows.grad.numpy(), TOL, f"gather d_src ({mode})")
    assert_close_fp32(d_nbr.view(Q, K, 100).cpu().numpy(), nbr_rows.grad.numpy(), TOL, f"gather d_nbr ({mode})")
    assert_close_fp32(d_tw.cpu().numpy(), wt.grad.numpy(), TOL, f"gather d_time_w ({mode})")
    assert_close_fp32(d_tb.cpu().numpy(), bt.grad.numpy(), TOL, f"gather d_time_b ({mode})")
    # padded slots get exactly zero
    assert float(d_nbr.view(Q, K, 100)[torch.from_numpy(nbr == 0).to(DEV)].abs().max()) == 0.0
    # the same call in two parts (orca_tgat_backward_phase): data gradients now, the parameter gradients later from a
    # second stream ordered after it -- the same bits
    later = []
    two = ops.tgat_backward(dsd, d_e, Q, K, R.to(DEV), want_time_param_grads=True, saved=saved, deferred=later, **inputs)
    assert len(later) == 1
    assert torch.equal(two[1], d_src) and torch.equal(two[2], d_nbr)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      later[0]()
    torch.cuda.current_stream().wait_stream(side)
    for k in sd:
      assert torch.equal(two[0][k], grads[k]), f"two-phase grad {k} ({mode})"
    assert torch.equal(two[5], d_tw) and torch.equal(two[6], d_tb)


def test_attention_backward_errors(ops, golden):
  g = golden("attention.npz")
  sd = {k: v.to(DEV) for k, v in _sd(g, "de1_K10_").items()}
  x = torch.zeros(4, 100, device=DEV)
  with pytest.raises(ValueError):        # K beyond ORCA_MAX_K
    ops.tgat_backward(sd, 1, 1, 64, x[:1], src_table=x, nbr_table=x, edge_table=x, mask=x, src_time=x, nbr_time=x)
  out = ops.tgat_backward(sd, 1, 0, 10, x[:0], src_table=x, nbr_table=x, edge_table=x, mask=x, src_time=x, nbr_time=x)
  assert out[1].shape == (0, 100) and float(out[0]["norm.weight"].abs().max()) == 0.0


@pytest.mark.parametrize("msg_dim,n,gather", [(472, 700, True), (301, 33, False), (472, 1, True), (472, 5000, True)])
def test_gru_backward(ops, msg_dim, n, gather):
  """orca_gru_backward against torch autograd of the oracle's GRUCell restatement (memory_updater.py:86)."""
  rs = np.random.RandomState(msg_dim + n)
  N = 3 * n + 5 if gather else n
  P = {"weight_ih": rs.randn(300, msg_dim) * 0.05, "weight_hh": rs.randn(300, 100) * 0.1,
       "bias_ih": rs.randn(300) * 0.1, "bias_hh": rs.randn(300) * 0.1}
  P = {k: torch.from_numpy(v.astype(np.float32)) for k, v in P.items()}
  x = torch.from_numpy(rs.randn(N, msg_dim).astype(np.float32))
  h = torch.from_numpy(rs.randn(N, 100).astype(np.float32))
  ids = torch.from_numpy(rs.permutation(N)[:n].astype(np.int32)) if gather else None
  R = torch.from_numpy(rs.randn(n, 100).astype(np.float32))
  Pg = {k: v.clone().requires_grad_() for k, v in P.items()}
  xr = (x[ids.long()] if gather else x).clone().requires_grad_()
  hr = (h[ids.long()] if gather else h).clone().requires_grad_()
  (O.gru_cell(Pg, xr, hr) * R).sum().backward()
  got = ops.gru_backward(P["weight_ih"].to(DEV), P["weight_hh"].to(DEV), P["bias_ih"].to(DEV), P["bias_hh"].to(DEV),
                         ids.to(DEV) if gather else None, x.to(DEV), h.to(DEV), R.to(DEV), want_inputs=True)
  for k, t in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), got[:4]):
    assert_close_fp32(t.cpu().numpy(), Pg[k].grad.numpy(), TOL, "gru grad " + k)
  assert_close_fp32(got[4].cpu().numpy(), xr.grad.numpy(), TOL, "gru d_x")
  assert_close_fp32(got[5].cpu().numpy(), hr.grad.numpy(), TOL, "gru d_h")
  if gather and msg_dim == 472:     # reading the gates the forward kept (orca_gru_update_saving -> orca_gru_backward_saved)
    blob = ops.gru_pack(*(P[k].to(DEV) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")), DEV)
    gates = torch.empty(n * 400, device=DEV)
    out = torch.zeros(N, 100, device=DEV)
    ops.gru_update(blob, msg_dim, ids.to(DEV), n, None, x.to(DEV), h.to(DEV), out, gates=gates)
    ref_out = O.gru_cell(P, x[ids.long()], h[ids.long()])
    assert_close_fp32(out[ids.long().to(DEV)].cpu().numpy(), ref_out.detach().numpy(), TOL, "gru forward (saving)")
    kept = ops.gru_backward(P["weight_ih"].to(DEV), P["weight_hh"].to(DEV), P["bias_ih"].to(DEV), P["bias_hh"].to(DEV),
                            ids.to(DEV), x.to(DEV), h.to(DEV), R.to(DEV), want_inputs=True, gates=gates)
    for k, t in zip(("weight_ih", "weight_hh", "bias_ih", "bias_hh"), kept[:4]):
      assert_close_fp32(t.cpu().numpy(), Pg[k].grad.numpy(), TOL, "gru grad (kept gates) " + k)
    assert_close_fp32(kept[4].cpu().numpy(), xr.grad.numpy(), TOL, "gru d_x (kept gates)")
    assert_close_fp32(kept[5].cpu().numpy(), hr.grad.numpy(), TOL, "gru d_h (kept gates)")
  later = []       # in two parts (orca_gru_backward_phase): the same bits
  two = ops.gru_backward(P["weight_ih"].to(DEV), P["weight_hh"].to(DEV), P["bias_ih"].to(DEV), P["bias_hh"].to(DEV),
                         ids.to(DEV) if gather else None, x.to(DEV), h.to(DEV), R.to(DEV), want_inputs=True, deferred=later)
  assert torch.equal(two[4], got[4]) and torch.equal(two[5], got[5])
  later[0]()
  for a, b in zip(two[:4], got[:4]):
    assert torch.equal(a, b)


@pytest.mark.parametrize("reuse,L,bs,drop,hgrad,budget", [
    (True, 2, 150, 0.0, False, 0), (False, 2, 150, 0.0, False, 0), (False, 1, 150, 0.0, False, 0), (True, 2, 1365, 0.0, False, 0),
    (True, 2, 150, 0.4, False, 0), (False, 2, 150, 0.4, False, 0), (True, 2, 150, 0.0, True, 0), (True, 2, 1365, 0.4, True, 0),
    (True, 2, 150, 0.0, False, 60), (True, 2, 150, 0.4, False, 60)])
def test_engine_train_step_vs_autograd(ops, reuse, L, bs, drop, hgrad, budget):
  """OrcaEngine.train_step (Path B with 2 layers; Path A, the exact recursion, with 2 layers and with 1 layer = the
  reference's default configuration): loss and every parameter gradient against torch autograd of the oracle's
  pieces composed as tgn_model.py:95-157 / embedding_module.py:116-182 compose them, on a warmed-up state (pending
  messages, pushed history); and the state after the step equals the forward-only step's.  bs = 1365 is the bench's
  batch shape (4095 roots: every split-K and tile variant of the backward GEMM is in play).  drop > 0: the
  attention-weight dropout of train.py:30 -- the checker multiplies its softmax by the keep masks the kernels used
  (orca_tgat_dropout_mask; RNG stream unpinned, masked arithmetic pinned).  hgrad: the --gradient history variant
  (train.py:43, history.py:42-46): the rows pushed in this batch keep their graph -- in the checker the winning
  (last) occurrence of every root id is written into the history table by a differentiable index_put.  budget > 0:
  Path C (embedding_module.py:215-285) with a cache plan per batch -- layer 1 also embeds the uncached out-of-batch
  neighbours (their rows only feed the detached history table)."""
  from orca_amd import synth
  from orca_amd.engine import OrcaEngine
  N, E, K, d_e = 1500, 30000, 10, 172
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, "cuda:0", seed=5)
  ef = synth.edge_features(E, d_e, "cuda:0")
  sd = synth.tgn_state_dict(d_e, L, seed=2)
  sd["time_encoder.w.bias"] = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, 100).astype(np.float32))
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  engs = []
  for _ in range(2):
    # (random cache plans below: the dense history table -- the slot cache follows plans of the planner, tests/test_gpu_engine.py)
    e = OrcaEngine(sd, ef, N + 1, L, K, reuse=reuse, budget=budget, device="cuda:0", dropout=drop, seed=11, gradient=hgrad,
                   slot_history=False if budget else None)
    e.set_neighbor_finder(csr)
    e.train_time_encoder = True     # the kernels' time-encoder gradient is checked too (frozen upstream: off by default)
    engs.append(e)
  A, B = engs
  rs = np.random.RandomState(0)
  batches = []
  for b in range(4):
    s, e = 20000 + b * bs, 20000 + (b + 1) * bs
    batches.append((src[s:e], dst[s:e], torch.from_numpy(rs.randint(1, N + 1, bs).astype(np.int64)).to(DEV), ts[s:e], eidx[s:e]))
  plans = [torch.from_numpy(rs.randint(1, N + 1, budget).astype(np.int64)).to(DEV) if budget else None for _ in range(4)]
  for bt, pl in zip(batches[:3], plans):
    A.compute_temporal_embeddings(*bt, True, pl)
    B.compute_temporal_embeddings(*bt, True, pl)
  mem, msg, pend = (t.cpu().clone() for t in (A.memory, A.messages, A.pending))
  hist = A.hist[0].cpu().clone() if reuse else None
  flag0 = A.cache_flag[0].clone() if budget else None
  bt = batches[3]
  loss, grads = A.train_step(*bt, cache_plan=plans[3]) if budget else A.train_step(*bt)
  B.compute_temporal_embeddings(*bt, True, plans[3])
  torch.cuda.synchronize()
  for name in ("memory", "last_update", "messages", "msg_ts", "pending"):
    assert_close_fp32(getattr(A, name).float().cpu().numpy(), getattr(B, name).float().cpu().numpy(), 1e-6, "state " + name)
  if reuse and not drop:
    assert_close_fp32(A.hist[0].cpu().numpy(), B.hist[0].cpu().numpy(), 1e-6, "state hist")
  if budget:
    assert torch.equal(A.cache_flag[0], B.cache_flag[0])
  # ---- the checker: torch autograd on the CPU ----
  nodes = torch.cat([bt[0], bt[1], bt[2]])
  t3 = torch.cat([bt[3]] * 3)
  def sample(nd, tt):
    nbr, eid, _, dt = ops.sample_recent(csr, nd, tt, K)
    return nbr.cpu().long(), eid.cpu().long(), dt.cpu()
  P = {k: v.clone().float().requires_grad_() for k, v in sd.items()}
  Q = nodes.numel()
  ids = torch.nonzero(pend).flatten()
  g = "memory_updater.memory_updater."
  upd = O.gru_cell({k: P[g + k] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}, msg[ids], mem[ids])
  mem2 = mem.clone().index_put((ids,), upd)
  wt, bt_ = P["time_encoder.w.weight"].reshape(-1), P["time_encoder.w.bias"]
  efc = ef.cpu()
  def layer(i, s_rows, n_rows, smp):
    nbr, eid, dt = smp
    n = s_rows.shape[0]
    p = f"embedding_module.attention_models.{i}."
    mult = None
    if drop:
      dk = A._dropout_kw(i + 1, 0)
      assert dk["dropout_p"] == drop
      mult = ops.tgat_dropout_mask(dk["dropout_seed"], 0, n, K, drop, DEV).cpu()
    return O.attention_forward({k[len(p):]: v for k, v in P.items() if k.startswith(p)}, s_rows,
                               O.time_encode(torch.zeros(n, 1), wt, bt_), n_rows, O.time_encode(dt, wt, bt_), efc[eid],
                               nbr == 0, drop_mult=mult)[0]
  s0 = sample(nodes, t3)
  nc = nodes.cpu()
  if budget:
    # embedding_module.py:251-260: the uncached out-of-batch neighbours join the layer-1 evaluation, at the batch's
    # latest timestamp; the first Q rows are the roots' own (the same queries as s0)
    unc = ops.find_uncached(s0[0].to(DEV).to(torch.int32).flatten(), nodes, flag0, torch.zeros(N + 1, dtype=torch.int64, device=DEV), 1,
                            torch.zeros((N + 32) // 32, dtype=torch.int32, device=DEV))
    c_nodes = torch.cat([nodes, unc])
    s1 = sample(c_nodes, torch.cat([t3, t3.max().expand(unc.numel())]))
    assert torch.equal(s1[0][:Q], s0[0]) and unc.numel() > 0
    cc = c_nodes.cpu()
    Qc = cc.numel()
    emb1 = layer(0, mem2[cc], mem2[s1[0].flatten()].view(Qc, K, 100), s1)
    hist2 = hist.clone()
    hist2[cc] = emb1.detach()
    top = layer(1, emb1[:Q], hist2[s0[0].flatten()].view(Q, K, 100), s0)
  elif reuse:
    emb1 = layer(0, mem2[nc], mem2[s0[0].flatten()].view(Q, K, 100), s0)
    if hgrad:
      last = {int(v): i for i, v in enumerate(nc.tolist())}           # id -> its last position
      widx = torch.tensor(sorted(last.values()), dtype=torch.int64)
      hist2 = hist.index_put((nc[widx],), emb1[widx])                 # differentiable: gradients reach emb1's winning rows
    else:
      hist2 = hist.clone()
      hist2[nc] = emb1.detach()           # single-threaded index_put: the last occurrence wins
    top = layer(1, emb1, hist2[s0[0].flatten()].view(Q, K, 100), s0)
  elif L == 1:
    top = layer(0, mem2[nc], mem2[s0[0].flatten()].view(Q, K, 100), s0)
  else:
    nodes1 = torch.cat([nodes, s0[0].flatten().to(DEV)])
    s1 = sample(nodes1, torch.cat([t3, t3.repeat_interleave(K)]))
    Q1 = nodes1.numel()
    emb1 = layer(0, mem2[nodes1.cpu()], mem2[s1[0].flatten()].view(Q1, K, 100), s1)
    top = layer(1, emb1[:Q], emb1[Q:].view(Q, K, 100), s0)
  se, de, ne = top[:bs], top[bs:2 * bs], top[2 * bs:]
  x = torch.cat([torch.cat([se, se]), torch.cat([de, ne])], dim=1)
  h = torch.relu(x @ P["affinity_score.fc1.weight"].t() + P["affinity_score.fc1.bias"])
  score = (h @ P["affinity_score.fc2.weight"].t() + P["affinity_score.fc2.bias"]).sigmoid().squeeze(1)
  bce = torch.nn.functional.binary_cross_entropy
  ref = bce(score[:bs], torch.ones(bs)) + bce(score[bs:], torch.zeros(bs))
  ref.backward()
  assert abs(float(loss) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
  # Bar: 2e-5 per parameter.  One exception, by construction of the comparison: the merger's ReLU gate (h1 > 0) is
  # decided by each side's own fp32 pre-activation, so a pre-activation within rounding of zero may gate differently in
  # the kernels and in torch on the CPU; ONE such flip moves everything upstream of that gate by the gradient of a single
  # (query, unit) -- a few 1e-5 of the norms here -- while leaving the layer's fc2 / LayerNorm and all later layers
  # exact.  Parameters that miss 2e-5 must therefore (a) hold 2e-4 and (b) all lie upstream of one layer's gate, whose
  # own fc2 / norm gradients and everything after them hold 2e-5.
  checked = 0
  loose = []
  for k, v in P.items():
    if v.grad is None:
      assert k not in grads or float(grads[k].abs().max()) == 0.0, k
      continue
    r = v.grad.numpy().copy()
    if k.endswith("in_proj_bias"):
      r[200:400] = 0.0
    assert k in grads, k
    got = grads[k].cpu().numpy().reshape(r.shape)
    try:
      assert_close_fp32(got, r, 2e-5, "train_step grad " + k)
    except AssertionError:
      assert_close_fp32(got, r, 2e-4, "train_step grad (upstream of a flipped ReLU gate?) " + k)
      loose.append(k)
    checked += 1
  if loose:
    pre = "embedding_module.attention_models."
    layers = [int(k[len(pre):].split(".")[0]) for k in loose if k.startswith(pre)]
    assert layers, loose                                  # the GRU / time encoder alone cannot miss
    top = max(layers)
    for k in loose:
      assert not k.startswith("affinity_score."), loose
      if k.startswith(pre) and int(k[len(pre):].split(".")[0]) == top:
        assert ".merger.fc2." not in k and ".norm." not in k, loose
  assert checked >= L * 12 + 4 + 2 + 4


def test_optimizer_step_and_refresh(ops):
  """One Adam step along train_step's gradients lowers the loss of the same batch from the same state, and the
  blobs re-packed in place by refresh() drive the native step exactly like blobs packed from scratch."""
  from orca_amd import synth
  from orca_amd.engine import OrcaEngine
  N, E, bs, K, d_e = 1500, 30000, 150, 10, 172
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, "cuda:0", seed=6)
  ef = synth.edge_features(E, d_e, "cuda:0")
  sd = synth.tgn_state_dict(d_e, 2, seed=3)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  eng = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device="cuda:0")
  eng.set_neighbor_finder(csr)
  g = torch.Generator(device="cuda:0")
  g.manual_seed(1)
  def batch(b):
    s = 20000 + b * bs
    return src[s:s + bs], dst[s:s + bs], torch.randint(1, N + 1, (bs,), generator=g, device="cuda:0"), ts[s:s + bs], eidx[s:s + bs]
  for b in range(3):
    eng.compute_temporal_embeddings(*batch(b), True)
  eng.optimizer(lr=1e-3)
  bt = batch(3)
  bk = eng.backup()
  l0, grads = eng.train_step(*bt)
  eng.restore(bk)
  before = {k: v.clone() for k, v in eng.params.items()}
  eng.step_optimizer(grads)
  assert any(not torch.equal(before[k], eng.params[k]) for k in grads)
  l1, _ = eng.train_step(*bt)
  assert float(l1) < float(l0), (float(l0), float(l1))
  # the native driver keeps the blob addresses: in-place re-pack == fresh pack
  fresh = OrcaEngine({k: v.clone() for k, v in eng.params.items()}, ef, N + 1, 2, K, reuse=True, budget=0, device="cuda:0")
  fresh.set_neighbor_finder(csr)
  fresh.restore(eng.backup())
  eng.restore(eng.backup())
  nb = batch(4)
  a = eng.compute_temporal_embeddings(*nb, True)
  c = fresh.compute_temporal_embeddings(*nb, True)
  assert torch.equal(a, c) and torch.equal(eng.memory, fresh.memory) and torch.equal(eng.messages, fresh.messages)


@pytest.mark.parametrize("drop,sampling", [(0.0, "recent"), (0.4, "recent"), (0.0, "uniform")])
def test_train_step_native_forward_equals_dispatched(ops, drop, sampling):
  """train_step's Path B forward through the C++ driver (orca_engine_train_capture + begin / layer / finish) against the
  Python-dispatched kernels: same loss and tables to the bit (row-wise kernels, same inputs), gradients to rounding
  (the work-list's order -- appended with atomics -- is the order the GRU's parameter gradients are summed in, and the
  native step's GRU backward reads the gates its forward kept instead of recomputing them)."""
  from orca_amd import synth
  from orca_amd.engine import OrcaEngine
  N, E, bs, K, d_e = 1500, 30000, 150, 10, 172
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, "cuda:0", seed=8)
  ef = synth.edge_features(E, d_e, "cuda:0")
  sd = synth.tgn_state_dict(d_e, 2, seed=5)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  engs = []
  for native in (True, False):
    e = OrcaEngine(sd, ef, N + 1, 2, K, reuse=True, budget=0, device="cuda:0", dropout=drop, sampling=sampling, seed=11)
    e.set_neighbor_finder(csr)
    e.native_train = native
    engs.append(e)
  assert engs[0]._train_native_ok() and not engs[1]._train_native_ok()
  g = torch.Generator(device="cuda:0")
  g.manual_seed(2)
  for b in range(4):
    s0 = 20000 + b * bs
    bt = (src[s0:s0 + bs], dst[s0:s0 + bs], torch.randint(1, N + 1, (bs,), generator=g, device="cuda:0"), ts[s0:s0 + bs],
          eidx[s0:s0 + bs])
    (la, ga), (lb, gb) = (e.train_step(*bt) for e in engs)
    assert float(la) == float(lb), (b, float(la), float(lb))
    assert set(ga) == set(gb)
    for k in ga:
      assert_close_fp32(ga[k].cpu().numpy(), gb[k].cpu().numpy(), TOL, f"batch {b} grad {k}")
    for name in ("memory", "last_update", "messages", "msg_ts", "pending"):
      assert torch.equal(getattr(engs[0], name), getattr(engs[1], name)), (b, name)
    assert torch.equal(engs[0].hist[0], engs[1].hist[0]), b
  # and an inference step after the training steps reads the same state
  nb = (src[25000:25000 + bs], dst[25000:25000 + bs], torch.randint(1, N + 1, (bs,), generator=g, device="cuda:0"),
        ts[25000:25000 + bs], eidx[25000:25000 + bs])
  assert torch.equal(engs[0].compute_temporal_embeddings(*nb, False), engs[1].compute_temporal_embeddings(*nb, False))


@pytest.mark.parametrize("tag", ["de172_K10_", "de4_K20_"])
def test_attention_dropout_forward_and_backward_vs_oracle(ops, golden, tag):
  """Attention-weight dropout (nn.MultiheadAttention(dropout=p), model/temporal_attention.py:17-21; train.py:30
  default 0.4).  torch's dropout RNG stream cannot be reproduced by another implementation -- "RNG stream unpinned":
  the kernel draws its keep decisions from a counter RNG and orca_tgat_dropout_mask writes them out; the MASKED
  ARITHMETIC is pinned: the oracle's softmax is multiplied by the same mask (F.dropout replaced by an explicit mask
  multiply) and output, parameter and input gradients must agree at 1e-5.  The inputs are the reference's golden
  attention cases (all-padded rows included)."""
  g = golden("attention.npz")
  sd = _sd(g, tag)
  d_e = g[tag + "edge"].shape[2]
  B, K = g[tag + "mask_in"].shape
  p, seed, q0 = 0.4, 0xC0FFEE1234, 77
  mult = ops.tgat_dropout_mask(seed, q0, B, K, p, DEV)
  mh = mult.cpu()
  assert set(np.unique(mh.numpy()).tolist()) <= {0.0, np.float32(1.0 / (1.0 - p))}
  assert 0.25 < float((mh == 0).float().mean()) < 0.55                      # about p of the weights are dropped
  R = torch.from_numpy(np.random.RandomState(B + K).randn(B, 100).astype(np.float32))
  names = ["src", "src_t", "nbr", "nbr_t"]
  P = {k: v.clone().requires_grad_() for k, v in sd.items()}
  xin = {k: torch.from_numpy(g[tag + k]).clone().requires_grad_() for k in names}
  out, _ = O.attention_forward(P, xin["src"], xin["src_t"], xin["nbr"], xin["nbr_t"], torch.from_numpy(g[tag + "edge"]),
                               torch.from_numpy(g[tag + "mask_in"].copy()), drop_mult=mh)
  (out * R).sum().backward()
  blob = ops.tgat_pack(sd, d_e, DEV)
  dsd = {k: v.to(DEV) for k, v in sd.items()}
  t = lambda k: torch.from_numpy(g[tag + k]).to(DEV).contiguous()
  f = lambda x, w: x.reshape(-1, w).contiguous()
  kw = dict(src_table=f(t("src"), 100), nbr_table=f(t("nbr"), 100), edge_table=f(t("edge"), d_e),
            src_time=f(t("src_t"), 100), nbr_time=f(t("nbr_t"), 100), dropout_p=p, dropout_seed=seed, dropout_q0=q0)
  mask = torch.from_numpy(g[tag + "mask_in"].copy()).to(DEV)
  got, _ = ops.tgat_forward(blob, d_e, B, K, mask=mask.clone().view(torch.uint8), **kw)
  assert_close_fp32(got.cpu().numpy(), out.detach().numpy(), TOL, tag + "dropout forward")
  plain, _ = ops.tgat_forward(blob, d_e, B, K, mask=mask.clone().view(torch.uint8), **{k: v for k, v in kw.items() if not k.startswith("dropout")})
  assert_close_fp32(plain.cpu().numpy(), g[tag + "out"], TOL, tag + "forward without dropout")
  assert float((got - plain).abs().max()) > 1e-3                             # and dropout does change the output
  grads, d_src, d_nbr, d_st, d_nt, _, _ = ops.tgat_backward(dsd, d_e, B, K, R.to(DEV), mask=mask.view(torch.uint8), **kw)
  for k, v in P.items():
    r = v.grad.numpy().copy()
    if k.endswith("in_proj_bias"):
      r[200:400] = 0.0           # softmax is shift-invariant: the key bias has no gradient (autograd leaves round-off)
    assert_close_fp32(grads[k].cpu().numpy().reshape(r.shape), r, TOL, tag + "dropout grad " + k)
  for nm, got_g in zip(names, (d_src, d_st, d_nbr, d_nt)):
    assert_close_fp32(got_g.cpu().numpy().reshape(-1), xin[nm].grad.numpy().reshape(-1), TOL, tag + "dropout d_" + nm)


def test_api_layer_train_mode_dropout(ops, golden):
  """The drop-in TemporalAttentionLayer in train() mode: the call draws its dropout seed from torch's CPU generator
  (torch.manual_seed makes a run repeatable, as it does upstream), eval() switches dropout off; output and gradients
  against the oracle under the same keep mask."""
  import orca_amd.api as api
  tag = "de172_K10_"
  g = golden("attention.npz")
  sd = _sd(g, tag)
  d_e = g[tag + "edge"].shape[2]
  B, K = g[tag + "mask_in"].shape
  p = 0.4
  layer = api.TemporalAttentionLayer(100, 100, d_e, 100, 100, n_head=2, dropout=p)
  layer.load_state_dict(sd, strict=True)
  layer = layer.to(DEV).train()
  names = ["src", "src_t", "nbr", "nbr_t"]
  yin = {k: torch.from_numpy(g[tag + k]).to(DEV).requires_grad_() for k in names}
  edge = torch.from_numpy(g[tag + "edge"]).to(DEV)
  call = lambda: layer(yin["src"], yin["src_t"], yin["nbr"], yin["nbr_t"], edge, torch.from_numpy(g[tag + "mask_in"].copy()).to(DEV))
  torch.manual_seed(123)
  seed = int(torch.randint(0, 1 << 62, (1,)).item())
  torch.manual_seed(123)
  got, w = call()
  torch.manual_seed(123)
  again, _ = call()
  assert torch.equal(got, again)
  other, _ = call()                                    # the generator moved on: another mask
  assert not torch.equal(got, other)
  mh = ops.tgat_dropout_mask(seed, 0, B, K, p, DEV).cpu()
  P = {k: v.clone().requires_grad_() for k, v in sd.items()}
  xin = {k: torch.from_numpy(g[tag + k]).clone().requires_grad_() for k in names}
  out, wr = O.attention_forward(P, xin["src"], xin["src_t"], xin["nbr"], xin["nbr_t"], torch.from_numpy(g[tag + "edge"]),
                                torch.from_numpy(g[tag + "mask_in"].copy()), drop_mult=mh)
  assert_close_fp32(got.detach().cpu().numpy(), out.detach().numpy(), TOL, "train-mode layer output")
  assert_close_fp32(w.cpu().numpy(), wr.detach().numpy(), TOL, "train-mode layer weights (after dropout, as nn.MultiheadAttention)")
  R = torch.from_numpy(np.random.RandomState(3).randn(B, 100).astype(np.float32))
  (out * R).sum().backward()
  (got * R.to(DEV)).sum().backward()
  named = dict(layer.named_parameters())
  for k in sd:
    r = P[k].grad.numpy().copy()
    if k.endswith("in_proj_bias"):
      r[200:400] = 0.0
    assert_close_fp32(named[k].grad.cpu().numpy(), r, TOL, "train-mode grad " + k)
  for k in names:
    assert_close_fp32(yin[k].grad.cpu().numpy(), xin[k].grad.numpy(), TOL, "train-mode grad input " + k)
  layer.eval()
  plain, _ = call()
  assert_close_fp32(plain.detach().cpu().numpy(), g[tag + "out"], TOL, "eval() switches dropout off")


def test_fused_adam_matches_torch_adam(ops):
  """orca_adam_step (ops.FusedAdam) against torch.optim.Adam (train.py:180,238) over several steps on tensors of
  ragged sizes, one of them without a gradient on the second step (torch skips it and does not advance its step)."""
  g = torch.Generator(device="cpu")
  g.manual_seed(4)
  shapes = [(200, 200), (300,), (100, 373), (1,), (7, 13), (300, 472)] + [(33,)] * 25      # more than one launch's worth
  P = {f"p{i}": torch.randn(*s, generator=g) for i, s in enumerate(shapes)}
  ref = {k: torch.nn.Parameter(v.clone()) for k, v in P.items()}
  opt = torch.optim.Adam(ref.values(), lr=1e-3)
  mine = {k: v.clone().to(DEV) for k, v in P.items()}
  fused = ops.FusedAdam(mine, lr=1e-3)
  for step in range(5):
    grads = {k: torch.randn(*v.shape, generator=g) * (10.0 ** (step - 2)) for k, v in P.items()}
    if step == 1:
      del grads["p2"]
    for k, p in ref.items():
      p.grad = grads.get(k)
    opt.step()
    fused.step({k: v.to(DEV) for k, v in grads.items()})
    for k in P:
      assert_close_fp32(mine[k].cpu().numpy(), ref[k].detach().numpy(), 1e-6, f"Adam step {step} {k}")
  assert fused.steps["p2"] == 4 and fused.steps["p0"] == 5


@pytest.mark.parametrize("bs", [1, 37, 1365])
def test_decoder_train_vs_autograd(ops, bs):
  """orca_decoder_train (link decoder + BCE of a training step, model/tgn_model.py:150-157, utils/util.py:12-24,
  train.py:231-233) against torch autograd of the same composition on the CPU."""
  rs = np.random.RandomState(bs)
  emb = torch.from_numpy(rs.randn(3 * bs, 100).astype(np.float32))
  P = {"fc1.weight": rs.randn(100, 200) * 0.1, "fc1.bias": rs.randn(100) * 0.1, "fc2.weight": rs.randn(1, 100) * 0.3,
       "fc2.bias": rs.randn(1) * 0.1}
  P = {k: torch.from_numpy(v.astype(np.float32)).requires_grad_() for k, v in P.items()}
  e = emb.clone().requires_grad_()
  se, de, ne = e[:bs], e[bs:2 * bs], e[2 * bs:]
  x = torch.cat([torch.cat([se, se]), torch.cat([de, ne])], dim=1)
  h = torch.relu(x @ P["fc1.weight"].t() + P["fc1.bias"])
  score = (h @ P["fc2.weight"].t() + P["fc2.bias"]).sigmoid().squeeze(1)
  bce = torch.nn.functional.binary_cross_entropy
  ref = bce(score[:bs], torch.ones(bs)) + bce(score[bs:], torch.zeros(bs))
  ref.backward()
  loss, d_emb, grads, prob = ops.decoder_train(emb.to(DEV), *(P[k].detach().to(DEV) for k in ("fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")),
                                               want_prob=True)
  assert abs(float(loss) - float(ref.detach())) <= 1e-5 * max(1.0, abs(float(ref.detach())))
  assert_close_fp32(prob.cpu().numpy(), score.detach().numpy(), TOL, "decoder probabilities")
  assert_close_fp32(d_emb.cpu().numpy(), e.grad.numpy(), TOL, "decoder d_emb")
  for k, v in P.items():
    assert_close_fp32(grads[k].cpu().numpy(), v.grad.numpy(), TOL, "decoder grad " + k)
