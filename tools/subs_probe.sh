set -e
mkdir -p gpurun_out/r3q
B="--cpu-batches 0 --train-steps 0 --path-a-steps 0"
for S in 4 2 1; do
  ORCA_GRU_SUBS=$S timeout -k 10 300 python3 bench.py $B --extra-configs "" > gpurun_out/r3q/bench_s$S.json 2> gpurun_out/r3q/bench_s$S.err
done
timeout -k 10 400 python3 bench.py $B --extra-configs "1,2,3" > gpurun_out/r3q/bench_auto.json 2> gpurun_out/r3q/bench_auto.err
timeout -k 10 200 python -u tools/path_c_bench.py > gpurun_out/r3q/pathc.txt 2>&1
