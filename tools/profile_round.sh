#!/bin/bash
# Refreshes the numbers under profiles/: run on the GPU box through gpurun from the repo root.
# Databases go to /tmp (gpurun_out/ is size-limited); only the summaries are written under gpurun_out/prof/.
set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof /tmp/prof
timeout -k 10 300 python3 bench.py > gpurun_out/prof/bench_final.json 2> gpurun_out/prof/bench_final.err
timeout -k 10 120 python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof/bench_driver_args.json 2> gpurun_out/prof/bench_driver_args.err
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof/stats -o stats -- python3 $R/bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs '' > $R/gpurun_out/prof/stats.out 2>&1
python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/stats/*/*.db /tmp/prof/stats/*.db 2>/dev/null | head -1)" $R/gpurun_out/prof/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs ''"
# unfused first kernel (ORCA_FUSE_GRU=0): the GRU and the sampler as launches of their own, for per-kernel figures
ORCA_FUSE_GRU=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof/stats_unfused -o stats -- python3 $R/bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs '' > $R/gpurun_out/prof/stats_unfused.out 2>&1
python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/stats_unfused/*/*.db /tmp/prof/stats_unfused/*.db 2>/dev/null | head -1)" $R/gpurun_out/prof/kernel_stats_unfused.md "ORCA_FUSE_GRU=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs ''"
ORCA_FUSE_GRU=0 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof/fetch -o fetch -- python3 $R/bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --path-a-steps 0 --extra-configs '' --no-kernel-events > $R/gpurun_out/prof/fetch.out 2>&1
python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/fetch/*/*.db /tmp/prof/fetch/*.db 2>/dev/null | head -1)" $R/gpurun_out/prof/pmc_fetch.md "ORCA_FUSE_GRU=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --path-a-steps 0 --extra-configs '' --no-kernel-events"
ORCA_FUSE_GRU=0 timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof/write -o write -- python3 $R/bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --path-a-steps 0 --extra-configs '' --no-kernel-events > $R/gpurun_out/prof/write.out 2>&1
python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/write/*/*.db /tmp/prof/write/*.db 2>/dev/null | head -1)" $R/gpurun_out/prof/pmc_write.md "ORCA_FUSE_GRU=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --path-a-steps 0 --extra-configs '' --no-kernel-events"
cd $R
timeout -k 10 120 python3 tools/kbench.py > gpurun_out/prof/kbench.txt 2>&1 || true
# Path C at config 3's shape: rates, then the per-kernel table of the same command
timeout -k 10 200 python3 tools/path_c_bench.py > gpurun_out/prof/path_c_config3_shape.txt 2>&1
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof/path_c -o stats -- python3 $R/tools/path_c_bench.py --batches 300 > $R/gpurun_out/prof/path_c.out 2>&1
python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/path_c/*/*.db /tmp/prof/path_c/*.db 2>/dev/null | head -1)" $R/gpurun_out/prof/path_c_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/path_c_bench.py --batches 300"
cd $R
echo profile_round done
