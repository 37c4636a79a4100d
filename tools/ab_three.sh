#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# Interleaved whole-step A/B of several library builds on the three bench scenes:  tools/ab_three.sh ab/libA.so ab/libB.so ...
export SPH_BENCH_PREROLL=0
for r in 1 2; do
for L in "$@"; do
  for w in "breaking_dam_30k_wcsph --steps 1000 --warmup 200" "wcsph_250k --steps 300 --warmup 100" "dfsph_1m --steps 60 --warmup 20 --no-scaling-base"; do
    SPH_LIB=$PWD/$L python bench.py --workload $w --no-cpu-baseline 2>>gpurun_out/bench_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d.get('kernel_breakdown_us',{})
print('$L'.split('/')[-1].ljust(18), d['config']['workload'].ljust(24), round(d['value'],1), round(d['ms_per_step'],4), 'build_nl', round(kb['build_nl']['avg_us'],1))"
  done
done
done
