#!/bin/bash
# Copies what tools/profile_round.sh left under gpurun_out/prof/ into profiles/ under this round's names and carries the
# two PMC figures per kernel into tools/pmc_traffic.json (bench.py's roofline.traffic reads them).  usage: collect_profiles.sh r03
set -e
R=${1:-r03}
P=gpurun_out/prof
cp $P/bench_final.json profiles/${R}_bench_final.json
cp $P/bench_driver_args.json profiles/${R}_bench_driver_args.json
cp $P/kernel_stats.md profiles/${R}_bench_kernel_stats.md
cp $P/kernel_stats_unfused.md profiles/${R}_bench_kernel_stats_unfused.md
cp $P/kbench.txt profiles/${R}_kbench.txt
cp $P/path_c_config3_shape.txt profiles/${R}_path_c_config3_shape.txt
cp $P/path_c_kernel_stats.md profiles/${R}_path_c_kernel_stats.md
python3 - "$R" <<'PY'
import json, re, sys
R = sys.argv[1]
md = f"profiles/{R}_pmc_hbm_traffic.md"
head = open(md).read().split("## pass 1")[0]
fetch, write = open("gpurun_out/prof/pmc_fetch.md").read(), open("gpurun_out/prof/pmc_write.md").read()
open(md, "w").write(head + "## pass 1 (FETCH_SIZE)\n\n" + fetch + "\n## pass 2 (WRITE_SIZE)\n\n" + write)
j = json.load(open("tools/pmc_traffic.json"))
def val(text, kernel, counter):     # mean per dispatch over every instantiation whose name starts like `kernel`
  rows = re.findall(r"\| [^|]*" + re.escape(kernel) + r"[^|]*\| " + counter + r" \| ([0-9.]+) \| (\d+) \|", text)
  n = sum(int(c) for _, c in rows)
  return round(sum(float(v) * int(c) for v, c in rows) / n, 1), n
for key, name in (("k_tgat_forward", "k_tgat_forward<10, 172, true"), ("k_gru_split", "k_gru_split<472"), ("k_begin_queries", "k_begin_queries<0, 1>")):
  f, n = val(fetch, name, "FETCH_SIZE")
  w, _ = val(write, name, "WRITE_SIZE")
  j[key].update(fetch_size_kb=f, write_size_kb=w, dispatches=n)
json.dump(j, open("tools/pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: v for k, v in j.items() if k.startswith("k_")}))
PY
