V=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do for X in $A $B; do
env $V=$X timeout -k 10 300 python3 bench.py --cpu-batches 0 --train-steps 0 --path-a-steps 0 --extra-configs "1,2,3,5" > /tmp/ab.json 2>/tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
python3 -c "
import json; d=json.load(open('/tmp/ab.json')); print('$V=$X', round(d['ms_per_step']*1e3,2), [(k, round(v.get('ms_per_step',0)*1e3,2)) for k,v in d.get('extra',{}).items()])"
done; done
