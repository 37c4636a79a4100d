#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# A/B two builds of libsph_mi355x.so on the same box, interleaved:  tools/ab_bench.sh libA.so libB.so [rounds]
A=$1; B=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for L in $A $B; do
    SPH_LIB=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_breakdown_us']
print('$L'.split('/')[-1], round(d['value'],1), 'Mps/s', {n: round(k[n]['avg_us'],1) for n in ('dfsph_div_residual','dfsph_div_correct','build_nl','dfsph_ext_force') if n in k})"
  done
done
