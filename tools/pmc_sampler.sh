#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the step kernels with and without the sampler's fence index (ORCA_NO_FENCES=1), first
# kernel and GRU as separate dispatches (ORCA_FUSE_GRU=0).  Run on the GPU box through gpurun from the repo root.
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof /tmp/prof
cd /tmp
for v in fences nofences; do
  if [ $v = nofences ]; then export ORCA_NO_FENCES=1; else unset ORCA_NO_FENCES; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/prof/$v$c
    ORCA_FUSE_GRU=0 timeout -s KILL 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/prof/$v$c -o p -- python3 $R/bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --path-a-steps 0 --no-kernel-events > $R/gpurun_out/prof/pmc_$v$c.out 2>&1 || exit 1
    python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/$v$c/*/*.db /tmp/prof/$v$c/*.db 2>/dev/null | head -n 1)" $R/gpurun_out/prof/pmc_${v}_$c.md "ORCA_FUSE_GRU=0 ${ORCA_NO_FENCES:+ORCA_NO_FENCES=1 }rocprofv3 --pmc $c --kernel-trace -- python3 bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --path-a-steps 0 --no-kernel-events" || exit 1
  done
done
echo pmc_sampler done
