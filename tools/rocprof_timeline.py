#!/usr/bin/env python3
"""One window of a rocprofv3 --kernel-trace database as a timeline: every dispatch between the n-th and the (n+1)-th
launch of an anchor kernel, with its start relative to the window, its duration, the gap to the end of the previous
dispatch on any queue, and its queue.  Usage: tools/rocprof_timeline.py <results.db> <anchor substring> <n> <out.txt>"""
import sqlite3
import sys


def main(db, anchor, n, out):
  c = sqlite3.connect(db)
  views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
  with open(out, "w") as f:
    if "kernels" not in views:
      f.write("views/tables: " + ", ".join(views) + "\n")
      return
    cols = [d[1] for d in c.execute("pragma table_info(kernels)")]
    f.write("# columns of `kernels`: " + ", ".join(cols) + "\n")
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"select name, start, end, {qcol or '0'} from kernels order by start"
    rows = c.execute(sel).fetchall()
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    n = int(n)
    if len(marks) < n + 2:
      f.write(f"only {len(marks)} launches of {anchor}\n")
      return
    lo, hi = marks[n], marks[n + 1]
    t0 = rows[lo][1]
    last_end = t0
    busy = 0
    f.write(f"# window: launch {n} of '{anchor}' to the next one: {(rows[hi][1] - t0) / 1e3:.1f} us, {hi - lo} dispatches\n")
    f.write(f"{'start us':>9} {'dur us':>8} {'gap us':>7} {'queue':>6}  kernel\n")
    for name, s, e, q in rows[lo:hi]:
      f.write(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - last_end) / 1e3:7.1f} {str(q):>6}  {name[:110]}\n")
      busy += e - s
      last_end = max(last_end, e)
    f.write(f"# sum of kernel durations {busy / 1e3:.1f} us\n")


if __name__ == "__main__":
  main(*sys.argv[1:5])
