#!/usr/bin/env python3
"""Launch-order trace of ONE training step from a rocprofv3 rocpd database (tools/train_prof.sh leaves one under
/tmp/prof/train): every dispatch between the Adam launches of two consecutive steps with its duration and grid.
Usage: tools/step_trace.py <results.db> [step index from the end, default 3]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
grid = "grid_x" if "grid_x" in cols else ("grid_size_x" if "grid_size_x" in cols else None)
sel = "name, start, end" + (f", {grid}" if grid else "")
rows = list(c.execute(f"select {sel} from kernels order by start"))
marks = [i for i, r in enumerate(rows) if "k_adam" in r[0] and (i + 1 == len(rows) or "k_adam" not in rows[i + 1][0])]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = marks[-k - 1], marks[-k]
t0 = rows[a][2]
tot = 0.0
for r in rows[a + 1:b + 1]:
  d = (r[2] - r[1]) / 1e3
  tot += d
  print(f"{(r[1] - t0) / 1e3:9.1f}  {d:7.1f} us  {('grid %d' % r[3]) if grid else '':>12}  {r[0][:90]}")
print(f"kernel time {tot:.1f} us, span {(rows[b][2] - t0) / 1e3:.1f} us, {b - a} launches")
