#!/usr/bin/env python3
"""Turns a rocprofv3 rocpd database (--kernel-trace --stats [-or- --pmc]) into a small markdown
summary for profiles/.  Usage: tools/rocprof_summary.py <results.db> <out.md> "<title / command>" """
import sqlite3
import sys


def main(db, out, title):
  c = sqlite3.connect(db)
  with open(out, "w") as f:
    f.write(f"# {title}\n\nSource: `top_kernels` view of the rocpd database written by rocprofv3 (durations in ns "
            "for totals, us for averages as reported by the view).\n\n")
    cols = [d[1] for d in c.execute("pragma table_info(top_kernels)")]
    f.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
    for r in c.execute("select * from top_kernels"):
      cells = []
      for x in r:
        if isinstance(x, float):
          cells.append(f"{x:.3f}")
        else:
          s = str(x).replace("|", "/")
          cells.append(s if len(s) <= 90 else s[:87] + "...")
      f.write("| " + " | ".join(cells) + " |\n")
    try:
      rows = c.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                       "where kernel_name like '%k_%' and kernel_name not like '%at::native%' and kernel_name not like '%rocprim%' "
                       "group by kernel_name, counter_name order by kernel_name").fetchall()
    except sqlite3.Error:
      rows = []
    if rows:
      f.write("\n## PMC counters (mean per dispatch)\n\n| kernel | counter | mean | dispatches |\n|---|---|---|---|\n")
      for n, cn, v, k in rows:
        n = n if len(n) <= 60 else n[:57] + "..."
        f.write(f"| {n} | {cn} | {v:.1f} | {k} |\n")


if __name__ == "__main__":
  main(*sys.argv[1:4])
