#!/bin/bash
# A/B of one environment switch on ONE box: tools/ab_env.sh VAR valueA valueB [rounds] -- bench.py's headline leg, alternating
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do
  for X in $A $B; do
    env $V=$X timeout -k 10 200 python3 bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs "" > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
    python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$V=$X', round(d['ms_per_step']*1e3,2), 'us/step', round(d['roofline']['avg_launch_ms']*1e3,2), 'us/attention launch', {k: round(v['avg_ms']*1e3,2) for k,v in d['kernels_ms'].items()})"
  done
done
