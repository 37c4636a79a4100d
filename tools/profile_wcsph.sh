#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# rocprofv3 kernel stats of the two WCSPH configs (BASELINE configs 1 and 2):  bash tools/profile_wcsph.sh <tag>   -> gpurun_out/<tag>_stats_<workload>/
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp SPH_BENCH_PREROLL=0
for w in "breaking_dam_30k_wcsph --steps 1000 --warmup 200" "wcsph_250k --steps 200 --warmup 50"; do
  set -- $w
  python3 $R/bench.py --workload $w --no-cpu-baseline > $R/gpurun_out/${tag}_bench_$1.json 2> /dev/null || exit 1
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats_$1 -o stats -- python3 $R/bench.py --workload $w --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/${tag}_bench_$1_under_rocprof.json 2> $R/gpurun_out/${tag}_stats_$1.err || exit 1
  find $R/gpurun_out/${tag}_stats_$1 -name "*.db" -delete
done
