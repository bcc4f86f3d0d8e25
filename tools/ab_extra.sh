#!/bin/bash
# A/B of two builds of the library on ONE box, the bench line's `extra` configurations included (BASELINE configs 1, 2, 3, 5
# at their own shapes): tools/ab_extra.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do for L in $A $B; do
ORCA_HIP_LIB=$PWD/$L timeout -k 10 300 python3 bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs "1,2,3,5" > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$L', round(d['ms_per_step']*1e3,2), [(k, round(v.get('ms_per_step',0)*1e3,2)) for k,v in d.get('extra',{}).items()])"
done; done
