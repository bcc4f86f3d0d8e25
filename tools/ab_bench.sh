#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a few per cent): tools/ab_bench.sh libA.so libB.so [rounds]
# prints us per step and us per attention launch of bench.py's headline leg, alternating A, B, A, B ...
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    ORCA_HIP_LIB=$PWD/$L timeout -k 10 200 python3 bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs "" > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$L', round(d['ms_per_step']*1e3,2), 'us/step', round(d['roofline']['avg_launch_ms']*1e3,2), 'us/attention launch', round(d['kernels_ms']['sample_recent']['avg_ms']*1e3+d['kernels_ms']['gru_update']['avg_ms']*1e3,2), 'first-kernel roles')"
  done
done
