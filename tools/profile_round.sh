#!/bin/bash
# Refreshes the numbers under profiles/: run on the GPU box through gpurun from the repo root.
# Databases go to /tmp (gpurun_out/ is size-limited); only the summaries are written under gpurun_out/prof/.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out/prof /tmp/prof
timeout -k 10 240 python3 bench.py > gpurun_out/prof/bench_final.json 2> gpurun_out/prof/bench_final.err
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof/stats -o stats -- python3 bench.py --cpu-batches 0 --train-steps 0 > gpurun_out/prof/stats.out 2>&1
python3 tools/rocprof_summary.py "$(ls /tmp/prof/stats/*/*.db /tmp/prof/stats/*.db 2>/dev/null | head -1)" gpurun_out/prof/kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 bench.py --cpu-batches 0 --train-steps 0"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/prof/fetch -o fetch -- python3 bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --no-kernel-events > gpurun_out/prof/fetch.out 2>&1
python3 tools/rocprof_summary.py "$(ls /tmp/prof/fetch/*/*.db /tmp/prof/fetch/*.db 2>/dev/null | head -1)" gpurun_out/prof/pmc_fetch.md "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --no-kernel-events"
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/prof/write -o write -- python3 bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --no-kernel-events > gpurun_out/prof/write.out 2>&1
python3 tools/rocprof_summary.py "$(ls /tmp/prof/write/*/*.db /tmp/prof/write/*.db 2>/dev/null | head -1)" gpurun_out/prof/pmc_write.md "rocprofv3 --pmc WRITE_SIZE --kernel-trace -- python3 bench.py --cpu-batches 0 --steps 20 --warmup 5 --train-steps 0 --no-kernel-events"
timeout -k 10 120 python3 tools/kbench.py > gpurun_out/prof/kbench.txt 2>&1
echo profile_round done
