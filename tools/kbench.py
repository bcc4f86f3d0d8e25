#!/usr/bin/env python3
"""Micro-benchmarks of the individual kernels at the bench.py shapes (HIP events, median of N launches).
Usage (on the GPU box): python tools/kbench.py [--nodes 1000000] [--edges 10000000]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orca_amd import ops, synth  # noqa: E402


def timeit(fn, n=30, warm=5):
  for _ in range(warm):
    fn()
  ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
  for a, b in ev:
    a.record()
    fn()
    b.record()
  torch.cuda.synchronize()
  ts = sorted(a.elapsed_time(b) for a, b in ev)
  return ts[len(ts) // 2] * 1e3   # us


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--nodes", type=int, default=1_000_000)
  ap.add_argument("--edges", type=int, default=10_000_000)
  ap.add_argument("--q", type=int, default=4095)
  ap.add_argument("--k", type=int, default=10)
  ap.add_argument("--d-e", type=int, default=172)
  ap.add_argument("--waves", type=int, default=0, help="orca_tgat_inputs.waves_per_query of the attention launches (0 or 4)")
  ap.add_argument("--attention-only", action="store_true")
  a = ap.parse_args()
  dev = "cuda:0"
  N, E, Q, K, d_e = a.nodes, a.edges, a.q, a.k, a.d_e
  src, dst, eidx, ts = synth.powerlaw_graph(N, E, dev)
  csr = ops.CSR.from_edges_device(src, dst, eidx, ts, N + 1)
  ef = synth.edge_features(E, d_e, dev)
  sd = synth.tgn_state_dict(d_e, 2)
  tb = torch.zeros(100, device=dev)
  p = "embedding_module.attention_models.0."
  blob = ops.tgat_pack({k[len(p):]: v for k, v in sd.items() if k.startswith(p)}, d_e, dev, time_b=tb)
  g = "memory_updater.memory_updater."
  gblob = ops.gru_pack(sd[g + "weight_ih"], sd[g + "weight_hh"], sd[g + "bias_ih"], sd[g + "bias_hh"], dev)
  tw = sd["time_encoder.w.weight"].reshape(-1).to(dev).contiguous()
  s0 = int(0.7 * E)
  nodes = torch.cat([src[s0:s0 + Q // 3], dst[s0:s0 + Q // 3], torch.randint(1, N + 1, (Q - 2 * (Q // 3),), device=dev)])
  t3 = ts[s0:s0 + Q // 3].repeat(3)[:Q].contiguous()
  if t3.numel() < Q:
    t3 = torch.cat([t3, t3[:Q - t3.numel()]])
  memory = torch.randn(N + 1, 100, device=dev)
  overlay = torch.randn(N + 1, 100, device=dev)
  pending = (torch.rand(N + 1, device=dev) < 0.5).to(torch.uint8)
  msg_dim = 300 + d_e
  messages = torch.randn(N + 1, msg_dim, device=dev)
  res = {}
  res["sample_recent"] = timeit(lambda: ops.sample_recent(csr, nodes, t3, K))
  nbr, eid, et, dt = ops.sample_recent(csr, nodes, t3, K)
  n32 = nodes.to(torch.int32)
  out = torch.empty(Q, 100, device=dev)
  res["tgat_layer1_gather_overlay"] = timeit(lambda: ops.tgat_forward(
      blob, d_e, Q, K, out=out, src_table=memory, src_idx=n32, src_overlay=overlay, src_flags=pending, nbr_table=memory,
      nbr_idx=nbr, nbr_overlay=overlay, nbr_flags=pending, nbr_ids=nbr, edge_table=ef, edge_idx=eid, dt=dt, time_w=tw,
      time_b=tb, q_folded=True, waves_per_query=a.waves))
  emb = torch.randn(Q, 100, device=dev)
  res["tgat_layer2_history"] = timeit(lambda: ops.tgat_forward(
      blob, d_e, Q, K, out=out, src_table=emb, nbr_table=memory, nbr_idx=nbr, nbr_ids=nbr, edge_table=ef, edge_idx=eid,
      dt=dt, time_w=tw, time_b=tb, q_folded=True, waves_per_query=a.waves))
  # phase breakdown of the attention kernel (100 MHz counter -> us), median over workgroups
  from orca_amd import _lib
  qpw = 4 if a.waves == 4 else 16
  nwg = (Q + qpw - 1) // qpw
  clk = torch.zeros(nwg * (32 + 16 * 8), dtype=torch.int64, device=dev)   # (+ the per-wave words of a -DORCA_WAVE_STAMPS build)
  # the clock the chip holds is read under load: two seconds of back-to-back launches first
  t_end = time.time() + 2.0
  while time.time() < t_end:
    for _ in range(200):
      ops.tgat_forward(blob, d_e, Q, K, out=out, src_table=memory, src_idx=n32, src_overlay=overlay, src_flags=pending,
                       nbr_table=memory, nbr_idx=nbr, nbr_overlay=overlay, nbr_flags=pending, nbr_ids=nbr, edge_table=ef,
                       edge_idx=eid, dt=dt, time_w=tw, time_b=tb, q_folded=True, waves_per_query=a.waves)
    torch.cuda.synchronize()
  _lib.load().orca_debug_tgat_phase_clock(clk.data_ptr())
  ops.tgat_forward(blob, d_e, Q, K, out=out, src_table=memory, src_idx=n32, src_overlay=overlay, src_flags=pending,
                   nbr_table=memory, nbr_idx=nbr, nbr_overlay=overlay, nbr_flags=pending, nbr_ids=nbr, edge_table=ef,
                   edge_idx=eid, dt=dt, time_w=tw, time_b=tb, q_folded=True, waves_per_query=a.waves)
  torch.cuda.synchronize()
  _lib.load().orca_debug_tgat_phase_clock(None)
  c32 = clk[:nwg * 32].view(nwg, 32).cpu().double()
  ws = clk[nwg * 32:].view(nwg, 16, 8).cpu().double()
  if float(ws.abs().sum()) > 0:     # the library was built with -DORCA_WAVE_STAMPS
    t0 = ws[:, :, 0].min(dim=1, keepdim=True).values
    t4 = ws[:, :, 4].min(dim=1, keepdim=True).values
    med = lambda x: [int(v) for v in x.median(dim=0).values]
    print("P2 per wave, cycles from the first wave's entry: entry", med(ws[:, :, 0] - t0))
    print("   after the chain and its epilogue", med(ws[:, :, 1] - t0))
    print("   after the row requests", med(ws[:, :, 2] - t0))
    print("   after the barrier", med(ws[:, :, 3] - t0))
    print("P4 per wave: entry", med(ws[:, :, 4] - t4))
    print("   after the chain and its epilogue", med(ws[:, :, 5] - t4))
    print("   at the barrier", med(ws[:, :, 6] - t4))
    print("   after the barrier", med(ws[:, :, 7] - t4))
  c, cyc = c32[:, :16], c32[:, 16:]
  d = (c[:, 1:9] - c[:, 0:8]) / 100.0
  dc = cyc[:, 1:9] - cyc[:, 0:8]
  names = ["T0 stage", "P1 q-proj", "P2 key fold", "P3 per-query", "P4 folded tail", "P4 reduce", "P7 fc2", "P8 LN"]
  print("tgat phases (median us over workgroups):", {n: round(float(d[:, i].median()), 2) for i, n in enumerate(names)})
  mhz = ((cyc[:, 8] - cyc[:, 0]) / ((c[:, 8] - c[:, 0]) / 100.0)).median()
  print("tgat in-kernel clock under load: %.0f MHz; shader cycles per phase (median):" % float(mhz),
        {n: int(dc[:, i].median()) for i, n in enumerate(names)})
  print("tgat workgroup span us: median", float(((c[:, 8] - c[:, 0]) / 100).median()), "chip span",
        float((c[:, 8].max() - c[:, 0].min()) / 100), " start skew", float((c[:, 0].max() - c[:, 0].min()) / 100))
  if a.attention_only:
    for k, v in res.items():
      print(f"{k:45s} {v:8.1f} us")
    return
  stamp = torch.zeros(N + 1, dtype=torch.int64, device=dev)
  wl = torch.zeros(N + 1, dtype=torch.int32, device=dev)
  cnt = torch.zeros(1, dtype=torch.int32, device=dev)
  ep = [1]

  def worklist():
    ops.worklist_build(nbr, nodes, pending, stamp, ep[0], wl, cnt)
    ep[0] += nbr.numel() + nodes.numel() + 1
  res["fill+worklist"] = timeit(worklist)
  n_rows = int(cnt.item())
  res["worklist_rows"] = n_rows
  bound = min(Q * (K + 1), N + 1)
  res[f"gru_overlay_{n_rows}rows_bound{bound}"] = timeit(lambda: ops.gru_update(gblob, msg_dim, wl, bound, cnt, messages,
                                                                                 memory, overlay))
  for n in (2730, 8192, 32768):
    ids = torch.randperm(N, device=dev)[:n].to(torch.int32)
    res[f"gru_{n}rows"] = timeit(lambda: ops.gru_update(gblob, msg_dim, ids, n, None, messages, memory, overlay))
  # phase breakdown of the streaming GRU kernel
  n = 2730
  ids = torch.randperm(N, device=dev)[:n].to(torch.int32)
  nw = ((n + 15) // 16) * 7 * 4   # (16-row tile) x (hidden tile) workgroups of 4 waves
  gclk = torch.zeros(nw * 4, dtype=torch.int64, device=dev)
  _lib.load().orca_debug_gru_phase_clock(gclk.data_ptr())
  ops.gru_update(gblob, msg_dim, ids, n, None, messages, memory, overlay)
  torch.cuda.synchronize()
  _lib.load().orca_debug_gru_phase_clock(None)
  g = gclk.view(nw, 4).cpu().double()
  g = g[g[:, 3] > 0]
  print("gru waves", g.shape[0], "phases us (median): locate rows", float(((g[:, 1] - g[:, 0]) / 100).median()), "chain",
        float(((g[:, 2] - g[:, 1]) / 100).median()), "epilogue", float(((g[:, 3] - g[:, 2]) / 100).median()), "| span",
        float((g[:, 3].max() - g[:, 0].min()) / 100), "start skew", float((g[:, 0].max() - g[:, 0].min()) / 100))
  slot = torch.zeros(N + 1, dtype=torch.int64, device=dev)
  se = [1]

  def push():
    ops.history_push(overlay, emb, nodes, slot, se[0])
    se[0] += Q + 1
  res["history_push"] = timeit(push)
  bs = Q // 3
  last_update = torch.zeros(N + 1, device=dev)
  msg_ts = torch.zeros(N + 1, device=dev)

  def store():
    ops.message_store(src[s0:s0 + bs], dst[s0:s0 + bs], eidx[s0:s0 + bs], ts[s0:s0 + bs], memory, last_update, ef, tw, tb,
                      slot, se[0], messages, msg_ts, pending)
    se[0] += 2 * bs + 1
  res["message_store"] = timeit(store)
  fl = synth.tgat_flops_per_query(d_e, K) * Q
  for k, v in res.items():
    extra = ""
    if k.startswith("tgat"):
      extra = f"  {fl / v / 1e6:.1f} TFLOP/s ({fl / v / 1e6 / 157.3 * 100:.1f}% of fp32 MFMA peak)"
    if k.startswith("gru_") and "rows" in k and not k.startswith("gru_overlay"):
      n = int(k.split("_")[1][:-4])
      extra = f"  {n * 2 * (msg_dim + 100) * 300 / v / 1e6:.1f} TFLOP/s"
    print(f"{k:45s} {v:10.1f} us{extra}" if isinstance(v, float) else f"{k:45s} {v}")


if __name__ == "__main__":
  main()
