#!/bin/bash
# backward parity tests, then wall time and rocprofv3 kernel statistics of the training step (tools/train_probe.py);
# run on the GPU box through gpurun from the repo root
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof /tmp/prof
timeout -k 10 400 python -u -m pytest tests/test_gpu_backward.py tests/test_gpu_api.py -x -q --timeout 300 --timeout-method thread > gpurun_out/bwd.log 2>&1
tail -n 3 gpurun_out/bwd.log
timeout -k 10 120 python3 tools/train_probe.py > gpurun_out/prof/train_probe.txt 2>&1 || exit 1
cd /tmp
rm -rf /tmp/prof/train
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof/train -o train -- python3 $R/tools/train_probe.py > $R/gpurun_out/prof/train_prof.out 2>&1 || exit 1
DB="$(ls /tmp/prof/train/*/*.db /tmp/prof/train/*.db 2>/dev/null | head -n 1)"
python3 $R/tools/step_trace.py "$DB" > $R/gpurun_out/prof/train_step_trace.txt 2>&1
python3 $R/tools/rocprof_summary.py "$DB" $R/gpurun_out/prof/train_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/train_probe.py"
cat $R/gpurun_out/prof/train_probe.txt
