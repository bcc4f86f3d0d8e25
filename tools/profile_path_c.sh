set -e
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/r3k /tmp/prof
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof/path_c -o stats -- python3 $R/tools/path_c_bench.py --batches 300 > $R/gpurun_out/r3k/path_c.out 2>&1
python3 $R/tools/rocprof_summary.py "$(ls /tmp/prof/path_c/*/*.db /tmp/prof/path_c/*.db 2>/dev/null | head -1)" $R/gpurun_out/r3k/path_c_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python3 tools/path_c_bench.py --batches 300"
