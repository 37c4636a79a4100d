#!/bin/bash
# interleaved runs of bench.py under different values of one environment knob:  tools/knob_bench.sh VAR "v1 v2 v3" [rounds] [bench args]
VAR=$1; VALS=$2; R=${3:-2}; shift 3
for r in $(seq 1 $R); do
  for v in $VALS; do
    env $VAR=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_breakdown_us']
print('$VAR=$v', round(d['value'],1), 'Mps/s', {n: round(k[n]['avg_us'],1) for n in list(k)[:5]})"
  done
done
