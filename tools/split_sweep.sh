#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# list build for small scenes: classic / three waves per 64 particles / nine, at several scene sizes:  tools/split_sweep.sh "scene bench-args" ...
export SPH_BENCH_PREROLL=0
for w in "$@"; do
for v in 0 3 9; do set -- $v
  SPH_BNL_SPLIT=$1 python bench.py --workload $w --no-cpu-baseline 2>>gpurun_out/bench_stderr.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); kb=d.get('kernel_breakdown_us',{})
print('split=$1', d['config']['workload'].ljust(24), d['config']['particles'], round(d['value'],1), round(d['ms_per_step'],4), 'build_nl', round(kb['build_nl']['avg_us'],1))"
done; done
