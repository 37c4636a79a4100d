#!/bin/bash
# A/B one development override on the same box, interleaved:  tools/ab_knob.sh SPH_FIN_RIDE 0 [rounds] [extra bench flags]
#   prints the headline and a few kernels for the default (knob unset) and for KNOB=VALUE, alternating
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1
K=$1; V=$2; R=${3:-2}; shift 3
for r in $(seq 1 $R); do
  for mode in default knob; do
    if [ $mode = knob ]; then export $K=$V; else unset $K; fi
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-base "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_breakdown_us']; rx=d.get('relaxed') or {}
print('$mode $K=${!K:-unset}', round(d['value'],1), 'Mps/s exact,', round(rx.get('value',0),1), 'relaxed', {n: round(k[n]['avg_us'],1) for n in ('dfsph_div_residual','dfsph_div_correct','dfsph_dens_correct','finalize') if n in k}, {n: round(k[n]['launches_per_step'],1) for n in ('finalize',) if n in k})"
  done
done
