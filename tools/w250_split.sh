#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# null-result experiment of round 3 (see tools/README.md); output under gpurun_out/r03/
set -e
mkdir -p gpurun_out/r03/w250
run() { name=$1; shift
  env "$@" timeout -k 10 240 python bench.py --workload wcsph_250k --no-cpu-baseline --no-scaling-base --profile-steps 1 > gpurun_out/r03/w250/$name.json 2> gpurun_out/r03/w250/$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r03/w250/$name.json').read().strip().splitlines()[-1])
kb=d.get('kernel_breakdown_us') or {}
print('$name', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in kb.items()})
PY
}
run base SPH_DUMMY=1
run split3 SPH_BNL_SPLIT=3
run split9 SPH_BNL_SPLIT=9
run base2 SPH_DUMMY=1
