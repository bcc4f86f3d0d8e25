#!/usr/bin/env python3
"""min / median / max duration (us) per kernel name pattern from a rocprofv3 rocpd database.
Usage: tools/kernel_durations.py <results.db> <pattern> [<pattern> ...]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
  print("no kernels view; tables:", tabs[:40])
  sys.exit(0)
cols = [d[1] for d in c.execute(f"pragma table_info({view})")]
for pat in sys.argv[2:]:
  rows = sorted(r[0] / 1e3 for r in c.execute(f"select end - start from {view} where name like ?", (f"%{pat}%",)))
  if rows:
    print(f"{pat}: n={len(rows)} min={rows[0]:.1f} med={rows[len(rows) // 2]:.1f} p90={rows[int(len(rows) * 0.9)]:.1f} max={rows[-1]:.1f} us")
  else:
    print(f"{pat}: no rows (columns: {cols})")
