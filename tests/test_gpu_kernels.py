"""Kernel-level parity of liborca_hip.so (through the C ABI, orca_amd/ops.py) against the golden
vectors produced by the reference and against the CPU oracle.  Needs a real MI355X.

Bars (BASELINE.json north_star): sampling indices, edge ids, edge times and Δt bit-exact;
floating point within 1e-5 relative (fp32), via conftest.assert_close_fp32."""
import numpy as np
import pytest
import torch

from conftest import assert_close_fp32
from oracle import oracle as O

pytestmark = pytest.mark.gpu
torch.set_num_threads(1)
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
  from orca_amd import ops as _ops
  return _ops


def _t(a, dtype=None):
  t = torch.from_numpy(np.ascontiguousarray(a))
  if dtype is not None:
    t = t.to(dtype)
  return t.to(DEV)


def _u32(a):
  return np.ascontiguousarray(a).view(np.uint32)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("builder", ["host", "device"])
@pytest.mark.parametrize("tag", ["sorted", "shuf"])
def test_csr_and_sampler_bit_exact(ops, golden, tag, builder):
  g = golden("sampler.npz")
  src, dst, ts, eidx = g["src"], g["dst"], g["ts"], g["eidx"]
  if tag == "shuf":
    s = g["shuf"]
    src, dst, ts, eidx = src[s], dst[s], ts[s], eidx[s]
  if builder == "host":
    csr = ops.CSR.from_edges_host(src, dst, eidx, ts, 64, DEV)
  else:
    csr = ops.CSR.from_edges_device(_t(src), _t(dst), _t(eidx), _t(ts), 64)
  indptr, nbr, eid, t = csr.arrays_host()
  assert np.array_equal(indptr, g[f"{tag}_indptr"])
  assert np.array_equal(nbr, g[f"{tag}_nbr"])
  assert np.array_equal(eid, g[f"{tag}_eidx"])
  assert np.array_equal(t, g[f"{tag}_ts"])
  qn, qt = _t(g[f"{tag}_q_nodes"]), _t(g[f"{tag}_q_ts"])
  for K in (1, 5, 10, 20):
    o_n, o_e, o_t, o_d = (x.cpu().numpy() for x in ops.sample_recent(csr, qn, qt, K))
    assert np.array_equal(o_n, g[f"{tag}_K{K}_nbr"])
    assert np.array_equal(o_e, g[f"{tag}_K{K}_eidx"])
    assert np.array_equal(_u32(o_t), _u32(g[f"{tag}_K{K}_et"]))
    assert np.array_equal(_u32(o_d), _u32(g[f"{tag}_K{K}_dt"]))
  cnt = ops.find_before(csr, _t(g[f"{tag}_fb_nodes"]), _t(g[f"{tag}_fb_ts"])).cpu().numpy()
  assert np.array_equal(cnt, g[f"{tag}_fb_len"])
  csr.close()


def _powerlaw_graph(seed, n_nodes, n_edges, skew=1.1, tie_frac=0.02):
  rs = np.random.RandomState(seed)
  w = 1.0 / np.arange(1, n_nodes + 1) ** skew
  w /= w.sum()
  perm = rs.permutation(n_nodes) + 1
  src = perm[rs.choice(n_nodes, n_edges, p=w)].astype(np.int64)
  dst = perm[rs.choice(n_nodes, n_edges, p=w)].astype(np.int64)
  ts = np.sort(rs.uniform(0, 2.6e6, n_edges))
  pos = rs.choice(np.arange(1, n_edges), int(tie_frac * n_edges), replace=False)
  ts[pos] = ts[pos - 1]
  ts = np.sort(ts)
  return src, dst, ts, np.arange(1, n_edges + 1, dtype=np.int64)


def test_sampler_vs_oracle_large_degrees(ops):
  """Hub nodes with > 64^2 entries exercise every round of the 64-ary search; ragged degrees,
  empty rows, node 0, query times equal to edge times."""
  src, dst, ts, eidx = _powerlaw_graph(3, 3000, 200000, skew=1.3)
  nf = O.OracleNeighborFinder(src, dst, eidx, ts, max_node_idx=3100)
  csr = ops.CSR.from_edges_device(_t(src), _t(dst), _t(eidx), _t(ts), 3101)
  indptr, nbr, eid, t = csr.arrays_host()
  assert np.array_equal(indptr, nf.indptr) and np.array_equal(nbr, nf.nbr)
  assert np.array_equal(eid, nf.eid) and np.array_equal(t, nf.ts)
  assert int(np.diff(nf.indptr).max()) > 4096
  rs = np.random.RandomState(4)
  Q = 20000
  qn = np.concatenate([src[rs.randint(0, len(src), Q // 2)], rs.randint(0, 3101, Q // 2)]).astype(np.int64)
  qt = rs.uniform(0, 2.7e6, Q)
  ex = rs.randint(0, len(ts), Q // 4)
  qt[:Q // 4] = ts[ex]
  qn[:Q // 4] = dst[ex]
  for K in (10, 20):
    a = [x.cpu().numpy() for x in ops.sample_recent(csr, _t(qn), _t(qt), K)]
    b = nf.get_temporal_neighbor(qn, qt, K, with_deltas=True)
    for x, y in zip(a, b):
      assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
  cnt = ops.find_before(csr, _t(qn), _t(qt)).cpu().numpy()
  ref = np.array([len(nf.find_before(int(n), float(tt))[0]) for n, tt in zip(qn[:500], qt[:500])])
  assert np.array_equal(cnt[:500], ref)
  # uniform sampling: not in the reference ("parity unpinned"); GPU == own CPU twin, bit-exact
  for K in (10, 20):
    a = [x.cpu().numpy() for x in ops.sample_uniform(csr, _t(qn), _t(qt), K, seed=1234)]
    b = O.sample_uniform(nf, qn, qt, K, 1234)
    for x, y in zip(a, b):
      assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    # every draw is an entry strictly before t of that node, in time order
    assert np.all((a[2] < qt[:, None]) | (a[0] == 0))
    assert np.all(np.diff(a[2], axis=1)[a[0][:, :-1] != 0] >= 0)


def test_sampler_errors(ops):
  csr = ops.CSR.from_edges_host([1, 2], [2, 3], [1, 2], [1.0, 2.0], 4, DEV)
  err = torch.zeros(1, dtype=torch.int32, device=DEV)
  out = ops.sample_recent(csr, _t(np.array([9, 2])), _t(np.array([5.0, 5.0])), 3, err_flag=err)
  assert int(err.item()) == 1
  assert np.array_equal(out[0].cpu().numpy(), [[0, 0, 0], [0, 1, 3]])
  with pytest.raises(ValueError):
    ops.sample_recent(csr, _t(np.array([1])), _t(np.array([5.0])), 0)
  with pytest.raises(IndexError):
    ops.CSR.from_edges_host([1, 7], [2, 3], [1, 2], [1.0, 2.0], 4, DEV)
  e = ops.sample_recent(csr, _t(np.zeros(0, np.int64)), _t(np.zeros(0)), 5)
  assert e[0].shape == (0, 5)


# ---------------------------------------------------------------------------------------------
def test_time_encode(ops, golden):
  g = golden("time_encode.npz")
  out = ops.time_encode(_t(g["t"]), _t(g["w"]), _t(g["b"])).cpu().numpy()
  assert_close_fp32(out, g["out"], 1e-5, "time_encode")


def test_time_encode_bias(ops):
  """A trained time encoder has a non-zero bias; the reference's fixtures do not (parity unpinned by a reference
  vector, and torch's CPU linear rounds t*w + b host-dependently): checked against the two-rounding definition of
  oracle.time_encode."""
  rs = np.random.RandomState(11)
  t = torch.from_numpy(np.concatenate([rs.uniform(0, 2.6e6, 400), rs.uniform(0, 50, 100)]).astype(np.float32)).view(50, 10)
  w = O.time_encode_weights(100)
  b = torch.from_numpy(rs.uniform(-3, 3, 100).astype(np.float32))
  out = ops.time_encode(t.to(DEV), w.to(DEV), b.to(DEV)).cpu().numpy()
  assert_close_fp32(out, O.time_encode(t, w, b).numpy(), 1e-5, "time_encode with bias")


def test_time_encode_large_arguments(ops):
  """cos(t*w + b) for arguments from months to decades of seconds (the reference's datasets reach 2.6e6; Wiki-Talk's and
  GDELT's spans are far longer), also at the half-integers of x/pi where a rounded n = rint(x/pi) would be off by one:
  against cos in fp64 of the same fp32 argument (the argument's two roundings as oracle.time_encode defines them),
  absolute error <= 4e-7.  (orca_cosf reduces in fp64; an fp32 Cody-Waite reduction below |x| = 5e6 passed this test too
  and was no faster -- fp64 vector instructions run at the fp32 rate on this chip.)"""
  rs = np.random.RandomState(12)
  halves = ((np.rint(rs.uniform(0, 5e6 / np.pi, 2000)) + 0.5) * np.pi).astype(np.float32)
  t = np.concatenate([rs.uniform(0, 5e6, 4000), rs.uniform(4.9e6, 5.1e6, 2000), rs.uniform(5e6, 2e9, 2000), halves,
                      np.array([0.0, 4999999.5, 5e6, 5000000.5, 1e7, 2.0e9])]).astype(np.float32)
  t = t[: (t.size // 10) * 10]
  w = torch.ones(100, dtype=torch.float32)                    # omega = 1: the argument IS t (+ b)
  w[1::2] = 0.5
  b = torch.from_numpy(rs.uniform(-3, 3, 100).astype(np.float32))
  out = ops.time_encode(torch.from_numpy(t).view(-1, 10).to(DEV), w.to(DEV), b.to(DEV)).cpu().numpy().reshape(-1, 100)
  x = (t.reshape(-1, 1) * w.numpy().reshape(1, -1)).astype(np.float32)      # product rounded, then the sum rounded
  x = (x + b.numpy().reshape(1, -1)).astype(np.float32)
  ref = np.cos(x.astype(np.float64))
  err = np.abs(out.astype(np.float64) - ref)
  assert err.max() <= 4e-7, f"time_encode: max abs error {err.max():.3e} at argument {x.flat[err.argmax()]!r}"


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["de172_K10_", "de1_K10_", "de4_K20_", "de172_K1_"])
def test_attention_dense_boundary(ops, golden, tag):
  """TemporalAttentionLayer.forward on its six dense tensors (the drop-in boundary)."""
  g = golden("attention.npz")
  sd = {k[len(tag + "sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "sd.")}
  d_e = g[tag + "edge"].shape[2]
  B, K = g[tag + "mask_in"].shape
  blob = ops.tgat_pack(sd, d_e, DEV)
  mask = _t(g[tag + "mask_in"].astype(np.uint8))
  out, w = ops.tgat_forward(blob, d_e, B, K, want_weights=True,
                            src_table=_t(g[tag + "src"]), nbr_table=_t(g[tag + "nbr"]).view(B * K, 100),
                            edge_table=_t(g[tag + "edge"]).view(B * K, d_e), mask=mask,
                            src_time=_t(g[tag + "src_t"]).view(B, 100), nbr_time=_t(g[tag + "nbr_t"]).view(B * K, 100))
  assert_close_fp32(out.cpu().numpy(), g[tag + "out"], 1e-5, tag + "out")
  assert np.array_equal(mask.cpu().numpy().astype(bool), g[tag + "mask_out"])
  if tag + "weights" in g.files:
    assert_close_fp32(w.cpu().numpy(), g[tag + "weights"], 1e-5, tag + "weights")


@pytest.mark.parametrize("d_e,gather", [(172, True), (172, False), (64, True)])
def test_pack_layers_equals_pack(ops, d_e, gather):
  """orca_tgat_pack_layers (all layers' folds in shared launches) writes bit for bit the blobs that orca_tgat_pack +
  orca_tgat_pack_gather write layer by layer -- on fresh blobs and again as an in-place re-pack after the parameters
  changed (three layers: two launch groups)."""
  from orca_amd import synth
  L = 3
  pre = "embedding_module.attention_models."
  def layers(seed):
    sd = synth.tgn_state_dict(d_e, L, seed=seed)
    return [{k[len(f"{pre}{i}."):]: v.to(DEV) for k, v in sd.items() if k.startswith(f"{pre}{i}.")} for i in range(L)]
  tb = torch.from_numpy(np.random.RandomState(3).uniform(-1, 1, 100).astype(np.float32)).to(DEV) if gather else None
  def same_bits(a, b):     # (the blob's fp64 scratch read as fp32 holds NaN patterns: compare the bits)
    return torch.equal(a.view(torch.int32), b.view(torch.int32))
  sds = layers(1)
  blobs = ops.tgat_pack_layers(sds, d_e, DEV, time_b=tb)
  for i in range(L):
    assert same_bits(blobs[i], ops.tgat_pack(sds[i], d_e, DEV, time_b=tb)), f"layer {i}"
  sds = layers(2)
  again = ops.tgat_pack_layers(sds, d_e, DEV, blobs=blobs, time_b=tb)
  for i in range(L):
    assert again[i].data_ptr() == blobs[i].data_ptr()
    assert same_bits(again[i], ops.tgat_pack(sds[i], d_e, DEV, time_b=tb)), f"re-packed layer {i}"


@pytest.mark.parametrize("folded,waves", [(False, 0), (True, 0), (True, 4)])
@pytest.mark.parametrize("d_e,K,Q", [(172, 10, 1000), (1, 10, 333), (172, 20, 130), (64, 7, 17), (172, 10, 5), (1, 10, 2)])
def test_attention_gather_mode_vs_oracle(ops, golden, d_e, K, Q, folded, waves):
  """The fused path: rows gathered by index, time encodings computed in-kernel, overlay rows.  folded: the blob
  completed by orca_tgat_pack_gather (q projection multiplied into the key map), with a non-zero time bias so
  that the constant part of the fold is exercised.  waves = 4: the four-waves-per-query launch shape
  (orca_tgat_inputs.waves_per_query; the shapes it is not built for must fall back to the default form)."""
  g = golden("attention.npz")
  tag = "de172_K10_"
  sd = {k[len(tag + "sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "sd.")}
  torch.manual_seed(d_e + K)
  kd = 200 + d_e
  sd["multi_head_target.k_proj_weight"] = torch.randn(200, kd) * (1.0 / kd ** 0.5)
  sd["multi_head_target.v_proj_weight"] = torch.randn(200, kd) * (1.0 / kd ** 0.5)
  rs = np.random.RandomState(Q)
  N, E = 500, 4000
  table = rs.randn(N, 100).astype(np.float32)
  overlay = rs.randn(N, 100).astype(np.float32)
  flags = (rs.rand(N) < 0.3).astype(np.uint8)
  edge_feats = rs.randn(E, d_e).astype(np.float32)
  edge_feats[0] = 0
  nodes = rs.randint(0, N, Q).astype(np.int32)
  nbr = rs.randint(1, N, (Q, K)).astype(np.int32)
  eid = rs.randint(1, E, (Q, K)).astype(np.int32)
  npad = rs.randint(0, K + 1, Q)
  npad[:3] = K
  for q in range(Q):
    nbr[q, :npad[q]] = 0
    eid[q, :npad[q]] = 0
  dt = rs.uniform(0, 2.6e6, (Q, K)).astype(np.float32)
  w_time = O.time_encode_weights(100)
  b_time = torch.from_numpy(rs.uniform(-3, 3, 100).astype(np.float32)) if folded else torch.zeros(100)
  blob = ops.tgat_pack(sd, d_e, DEV, time_b=b_time if folded else None)
  out, w = ops.tgat_forward(blob, d_e, Q, K, want_weights=True, q_folded=folded, waves_per_query=waves,
                            src_table=_t(table), src_idx=_t(nodes), src_overlay=_t(overlay), src_flags=_t(flags),
                            nbr_table=_t(table), nbr_idx=_t(nbr), nbr_overlay=_t(overlay), nbr_flags=_t(flags),
                            nbr_ids=_t(nbr), edge_table=_t(edge_feats), edge_idx=_t(eid), dt=_t(dt),
                            time_w=w_time.to(DEV), time_b=b_time.to(DEV))
  eff = torch.from_numpy(np.where(flags[:, None] != 0, overlay, table))
  nbr_t = torch.from_numpy(nbr.astype(np.int64))
  mask = nbr_t == 0
  ref, ref_w = O.attention_forward(sd, eff[torch.from_numpy(nodes.astype(np.int64))],
                                   O.time_encode(torch.zeros(Q, 1), w_time, b_time),
                                   eff[nbr_t.flatten()].view(Q, K, 100),
                                   O.time_encode(torch.from_numpy(dt), w_time, b_time),
                                   torch.from_numpy(edge_feats)[torch.from_numpy(eid.astype(np.int64))], mask)
  assert_close_fp32(out.cpu().numpy(), ref.numpy(), 1e-5, "gather attention out")
  assert_close_fp32(w.cpu().numpy(), ref_w.numpy(), 1e-5, "gather attention weights")


def test_attention_errors(ops, golden):
  g = golden("attention.npz")
  tag = "de1_K10_"
  sd = {k[len(tag + "sd."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "sd.")}
  with pytest.raises(ValueError):
    ops.tgat_pack(sd, 172, DEV)          # wrong edge width for these weights
  with pytest.raises(ValueError):
    ops.tgat_pack(sd, 500, DEV)
  blob = ops.tgat_pack(sd, 1, DEV)
  x = torch.zeros(4, 100, device=DEV)
  with pytest.raises(ValueError):        # K beyond ORCA_MAX_K
    ops.tgat_forward(blob, 1, 1, 64, src_table=x, nbr_table=x, edge_table=x, mask=x, src_time=x, nbr_time=x)
  out, _ = ops.tgat_forward(blob, 1, 0, 10, src_table=x, nbr_table=x, edge_table=x, mask=x, src_time=x, nbr_time=x)
  assert out.shape == (0, 100)


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("msg_dim,n", [(472, 700), (301, 33), (472, 1), (472, 7000)])
def test_gru_vs_oracle(ops, msg_dim, n):
  """n = 7000 is past both size switches of the out-of-place path: more row tiles than the split kernel's grid
  covers in one pass (its workgroups loop) and, with an exact count, the one-wave-per-tile streaming kernel."""
  torch.manual_seed(msg_dim + n)
  cell = torch.nn.GRUCell(msg_dim, 100)
  P = {k: v.detach() for k, v in cell.state_dict().items()}
  N = max(2000, n + 1000)
  rs = np.random.RandomState(n)
  x = torch.from_numpy(rs.randn(N, msg_dim).astype(np.float32))
  h = torch.from_numpy(rs.randn(N, 100).astype(np.float32))
  ids = torch.from_numpy(rs.permutation(N)[:n].astype(np.int64))
  msg_ts = torch.from_numpy(rs.uniform(0, 1e6, N).astype(np.float32))
  ref = O.gru_cell(P, x[ids], h[ids])
  blob = ops.gru_pack(P["weight_ih"], P["weight_hh"], P["bias_ih"], P["bias_hh"], DEV)
  # in place (update_memory), with a device-side count larger launch bound
  hd = h.to(DEV)
  lu = torch.zeros(N, device=DEV)
  n_dev = torch.tensor([n], dtype=torch.int32, device=DEV)
  ids32 = torch.zeros(n + 50, dtype=torch.int32, device=DEV)
  ids32[:n] = ids.to(DEV).to(torch.int32)
  ops.gru_update(blob, msg_dim, ids32, n + 50, n_dev, x.to(DEV), hd, hd, lu, msg_ts.to(DEV))
  got = hd.cpu()
  assert_close_fp32(got[ids].numpy(), ref.numpy(), 1e-5, "gru in place")
  untouched = np.ones(N, bool)
  untouched[ids.numpy()] = False
  assert np.array_equal(got.numpy()[untouched], h.numpy()[untouched])
  assert np.array_equal(lu.cpu().numpy()[ids.numpy()], msg_ts.numpy()[ids.numpy()])
  assert np.all(lu.cpu().numpy()[untouched] == 0)
  # overlay (get_updated_memory without the table clone)
  ov = torch.zeros(N, 100, device=DEV)
  ops.gru_update(blob, msg_dim, ids32, n, None, x.to(DEV), h.to(DEV), ov)
  assert_close_fp32(ov.cpu()[ids].numpy(), ref.numpy(), 1e-5, "gru overlay")
  assert np.all(ov.cpu().numpy()[untouched] == 0)
  # overlay of a device-counted work-list (the engine's call: launch bound above the count)
  ov2 = torch.zeros(N, 100, device=DEV)
  ops.gru_update(blob, msg_dim, ids32, n + 50, n_dev, x.to(DEV), h.to(DEV), ov2)
  assert_close_fp32(ov2.cpu()[ids].numpy(), ref.numpy(), 1e-5, "gru overlay, device count")
  assert np.all(ov2.cpu().numpy()[untouched] == 0)


# ---------------------------------------------------------------------------------------------
def test_segment_last_and_mean(ops, golden):
  g = golden("aggregators.npz")
  ids, msgs, ts = g["ids"], g["msgs"], g["ts"]
  slot = torch.zeros(64, dtype=torch.int64, device=DEV)
  win = ops.segment_last(_t(ids), slot, 1).cpu().numpy().astype(bool)
  u, idx = O.segment_last(ids)
  assert np.array_equal(np.sort(np.nonzero(win)[0]), np.sort(idx))
  win2, mean = ops.segment_mean(_t(ids), _t(msgs), _t(ts), slot, 1000)
  win2 = win2.cpu().numpy().astype(bool)
  assert np.array_equal(win2, win)
  u2, m2, t2 = O.segment_mean(ids, msgs, ts)
  order = np.argsort(ids[win2])                 # winners by ascending id == oracle order
  pos = np.nonzero(win2)[0][order]
  assert np.array_equal(ids[pos], u2)
  assert_close_fp32(mean.cpu().numpy()[pos], m2, 1e-6, "segment mean")
  assert np.array_equal(ts[pos], t2)
  q = g["query_ids"]
  keep = np.isin(u2, q)
  assert np.array_equal(u2[keep], g["mean_ids"])
  assert_close_fp32(mean.cpu().numpy()[pos][keep], g["mean_msgs"], 1e-6, "segment mean vs reference")
  assert np.array_equal(msgs[pos][keep], g["last_msgs"])
  # larger random case
  rs = np.random.RandomState(1)
  big = rs.randint(0, 5000, 8192).astype(np.int64)
  slot = torch.zeros(5000, dtype=torch.int64, device=DEV)
  w = ops.segment_last(_t(big), slot, 7).cpu().numpy().astype(bool)
  uu, ii = O.segment_last(big)
  assert np.array_equal(np.nonzero(w)[0], np.sort(ii))


def test_history_push_pull(ops, golden):
  g = golden("history.npz")
  emb = torch.zeros(50, 100, device=DEV)
  slot = torch.zeros(50, dtype=torch.int64, device=DEV)
  ops.history_push(emb, _t(g["x"]), _t(g["ids"]), slot, 1)
  assert np.array_equal(emb.cpu().numpy(), g["emb"])
  pulled = ops.gather_rows(emb, _t(g["pull_ids"]))
  assert np.array_equal(pulled.cpu().numpy(), g["pulled"])
  pulled32 = ops.gather_rows(emb, _t(g["pull_ids"].astype(np.int32)))
  assert np.array_equal(pulled32.cpu().numpy(), g["pulled"])
  # a second push on the same scratch table with a later epoch
  ops.history_push(emb, _t(g["x"][::-1].copy()), _t(g["ids"][::-1].copy()), slot, 1 + 80)
  ref = np.zeros((50, 100), np.float32)
  for i, x in zip(g["ids"][::-1], g["x"][::-1]):
    ref[i] = x
  keep = np.isin(np.arange(50), g["ids"])
  assert np.array_equal(emb.cpu().numpy()[keep], ref[keep])
  t2 = torch.zeros(50, 100, device=DEV)
  uid = np.unique(g["ids"])
  ops.scatter_rows(t2, _t(uid), _t(g["emb"][uid]))
  assert np.array_equal(t2.cpu().numpy(), g["emb"])


def test_worklist_and_flags(ops):
  rs = np.random.RandomState(0)
  N = 3000
  ids32 = rs.randint(0, N, 5000).astype(np.int32)
  ids64 = rs.randint(0, N, 700).astype(np.int64)
  flags = (rs.rand(N) < 0.4).astype(np.uint8)
  stamp = torch.zeros(N, dtype=torch.int64, device=DEV)
  wl = torch.zeros(N, dtype=torch.int32, device=DEV)
  cnt = torch.zeros(1, dtype=torch.int32, device=DEV)
  for epoch in (1, 10001):
    ops.worklist_build(_t(ids32), _t(ids64), _t(flags), stamp, epoch, wl, cnt)
    n = int(cnt.item())
    u = np.unique(np.concatenate([ids32.astype(np.int64), ids64]))
    ref = u[flags[u] != 0]
    assert np.array_equal(np.sort(wl[:n].cpu().numpy()), ref)
  f = _t(flags.copy())
  ops.set_flags(f, ids32=wl, n=N, n_dev=cnt, v=0)
  exp = flags.copy()
  exp[ref] = 0
  assert np.array_equal(f.cpu().numpy(), exp)
  ops.set_flags(f, ids64=_t(ids64), v=1)
  exp[ids64] = 1
  assert np.array_equal(f.cpu().numpy(), exp)


@pytest.mark.parametrize("d_e", [172, 1])
def test_message_store_vs_oracle(ops, d_e):
  """tgn_model.py:172-191 + memory.py:25-28 against OracleTGN.store_raw_messages."""
  rs = np.random.RandomState(d_e)
  N, E, bs = 400, 3000, 300
  w_time = O.time_encode_weights(100)
  ef = rs.randn(E, d_e).astype(np.float32)
  sd = {"time_encoder.w.weight": w_time.view(100, 1)}
  tg = O.OracleTGN(sd, ef, N, 1, 10, False, 0)
  tg.memory = torch.from_numpy(rs.randn(N, 100).astype(np.float32))
  tg.last_update = torch.from_numpy(rs.uniform(0, 1e6, N).astype(np.float32))
  src = rs.randint(1, N, bs).astype(np.int64)
  dst = rs.randint(1, N, bs).astype(np.int64)
  ts = np.sort(rs.uniform(1e6, 2e6, bs))
  eidx = rs.randint(1, E, bs).astype(np.int64)
  msg_dim = 300 + d_e
  messages = torch.zeros(N, msg_dim, device=DEV)
  msg_ts = torch.zeros(N, device=DEV)
  pending = torch.zeros(N, dtype=torch.uint8, device=DEV)
  slot = torch.zeros(N, dtype=torch.int64, device=DEV)
  ops.message_store(_t(src), _t(dst), _t(eidx), _t(ts), tg.memory.to(DEV), tg.last_update.to(DEV), _t(ef),
                    w_time.to(DEV), torch.zeros(100, device=DEV), slot, 1, messages, msg_ts, pending)
  tg.store_raw_messages(src, dst, ts, eidx)
  assert np.array_equal(pending.cpu().numpy().astype(bool), tg.pending)
  assert np.array_equal(msg_ts.cpu().numpy(), tg.msg_ts.numpy())
  assert_close_fp32(messages.cpu().numpy(), tg.messages.numpy(), 1e-5, "messages")
  # exact copies outside the time-encoding columns
  assert np.array_equal(messages.cpu().numpy()[:, :200 + d_e], tg.messages.numpy()[:, :200 + d_e])


@pytest.mark.parametrize("n_a,n_b", [(300, 0), (4095, 40950), (70, 20000), (1, 1)])
def test_scatter_add_flagged_and_pair(ops, n_a, n_b):
  """dst[pos[id]] += rows[p] for the flagged ids (the gradient rows of attention layer 1 flowing into the updated-memory
  rows, embedding_module.py:159-175): one list, and two lists in one launch, against a float64 np.add.at (the order of
  the fp32 additions is the atomics': tolerance, not bits).  Hub ids repeat thousands of times."""
  rs = np.random.RandomState(n_a + n_b)
  N = 5000
  flags = (rs.rand(N) < 0.6).astype(np.uint8)
  wl = np.flatnonzero(flags)[rs.permutation(int(flags.sum()))].astype(np.int32)
  pos = np.zeros(N, np.int32)
  pos[wl] = np.arange(wl.size, dtype=np.int32)
  def ids(n):
    z = rs.zipf(1.3, n) % N            # power-law: a few ids take most positions
    z[rs.rand(n) < 0.05] = N + 7       # ids outside the table are skipped
    return z.astype(np.int64)
  ia, ib = ids(n_a), ids(n_b)
  ra, rb = rs.randn(n_a, 100).astype(np.float32), rs.randn(n_b, 100).astype(np.float32)
  want = np.zeros((wl.size, 100), np.float64)
  for i, r in ((ia, ra), (ib, rb)):
    ok = (i < N) & (flags[np.minimum(i, N - 1)] != 0)
    np.add.at(want, pos[i[ok]], r[ok].astype(np.float64))
  _t = lambda a: torch.from_numpy(a).to(DEV)
  d1 = torch.zeros(wl.size, 100, device=DEV)
  if n_a:
    ops.scatter_add_flagged(_t(ra), _t(ia), _t(flags), _t(pos), d1)
  if n_b:
    ops.scatter_add_flagged(_t(rb), _t(ib.astype(np.int32)), _t(flags), _t(pos), d1)
  d2 = torch.zeros(wl.size, 100, device=DEV)
  ops.scatter_add_flagged_pair(_t(ra), _t(ia), _t(rb), _t(ib.astype(np.int32)), _t(flags), _t(pos), d2)
  scale = max(1.0, float(np.abs(want).max()))
  assert float(np.abs(d1.cpu().numpy() - want).max()) <= 2e-5 * scale
  assert float(np.abs(d2.cpu().numpy() - want).max()) <= 2e-5 * scale
