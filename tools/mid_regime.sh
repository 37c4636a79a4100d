#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# null-result experiment of round 3 (see tools/README.md); output under gpurun_out/r03/
set -e
mkdir -p gpurun_out/r03/mid
run() { # name workload envs...
  name=$1; wl=$2; shift 2
  env "$@" timeout -k 10 240 python bench.py --workload $wl --no-cpu-baseline --no-scaling-base --no-relaxed --profile-steps 0 > gpurun_out/r03/mid/$name.json 2> gpurun_out/r03/mid/$name.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/r03/mid/$name.json').read().strip().splitlines()[-1])
print('$name', round(d['value'],1), round(d['ms_per_step'],4), d['config'].get('n_dens_mean'))
PY
}
for wl in breaking_dam_demo default; do
run ${wl}_base $wl SPH_DUMMY=1
run ${wl}_lin_quad $wl SPH_CELL_ORDER=linear SPH_QUAD_BELOW=200000
run ${wl}_lin_quad_split3 $wl SPH_CELL_ORDER=linear SPH_QUAD_BELOW=200000 SPH_BNL_SPLIT=3
run ${wl}_mor_quad $wl SPH_CELL_ORDER=morton SPH_STAGE=0 SPH_QUAD_BELOW=200000
run ${wl}_lin_plain $wl SPH_CELL_ORDER=linear
done
