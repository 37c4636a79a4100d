#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
export SPH_BENCH_PREROLL=0
for w in breaking_dam_30k_wcsph wcsph_250k; do python bench.py --workload $w --steps 1000 --warmup 200 --no-cpu-baseline 2>>gpurun_out/bench_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d.get('kernel_breakdown_us',{}); print('$w', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in kb.items()})"; done
python bench.py --workload dfsph_1m --steps 100 --warmup 20 --no-cpu-baseline --no-scaling-base 2>>gpurun_out/bench_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d.get('kernel_breakdown_us',{}); print('dfsph_1m', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in kb.items()})"
