#!/bin/bash
# k_wgrad's duration in the training step against the number of k pieces per tile (ORCA_WGRAD_SPLITS)
export TMPDIR=/tmp
R=$PWD
cd /tmp
for sp in 0 5 6 7 10 12 16; do
  ORCA_WGRAD_SPLITS=$sp timeout -k 10 120 rocprofv3 --kernel-trace --stats -d /tmp/wg$sp -o stats -- python3 $R/bench.py --cpu-batches 0 --steps 10 --warmup 2 --min-region-ms 0 --train-steps 40 --path-a-steps 0 --extra-configs "" --no-kernel-events > /tmp/wg$sp.out 2>&1 || exit 1
  python3 $R/tools/rocprof_summary.py "$(ls /tmp/wg$sp/*/*.db /tmp/wg$sp/*.db 2>/dev/null | head -1)" /tmp/wg$sp.md "splits $sp"
  echo "splits $sp: $(grep -E 'k_wgrad|k_splitk_reduce' /tmp/wg$sp.md | awk -F'|' '{printf "%s calls %s avg %s us; ", substr($2,1,40), $3, $5}')"
done
