#!/bin/bash
# SQ counters of the training step's kernels (one pass): where the wave cycles of each kernel go.
# usage (through gpurun, from the repo root): bash tools/pmc_train.sh
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/prof /tmp/prof
cd /tmp
rm -rf /tmp/prof/pmc_train
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/prof/pmc_train -o pmc -- python3 $R/tools/train_probe.py > $R/gpurun_out/prof/pmc_train.out 2>&1 || exit 1
python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/pmc_train/*/*.db /tmp/prof/pmc_train/*.db 2>/dev/null | head -1)" $R/gpurun_out/prof/pmc_train.md "rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -- python3 tools/train_probe.py"
