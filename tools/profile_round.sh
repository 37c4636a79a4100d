#!/bin/bash
# The measurements profiles/ holds for one round, on a gpurun box:  bash tools/profile_round.sh r01g
#   1. default bench, un-profiled (with cpu_baseline)         -> gpurun_out/<tag>_bench_dfsph_1m.json
#   2. the same command under rocprofv3 --kernel-trace --stats -> gpurun_out/<tag>_stats/ + <tag>_bench_dfsph_1m_under_rocprof.json
#   3. three separate --pmc passes on a short run              -> gpurun_out/<tag>_fetch, _write, _sq  (tools/pmc_traffic.py)
#   0. first of all the CPU suite (pytest -m "not gpu"), logged next to the GPU suite's log: a round never closes on a red CPU test again
#      (tools/removal_build.py's patches had drifted from the kernels and only the CPU suite notices)
set -o pipefail
tag=${1:-r01x}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R && python3 -m pytest tests -q -m "not gpu" > gpurun_out/${tag}_cpu_suite.log 2>&1 || { tail -20 gpurun_out/${tag}_cpu_suite.log; exit 1; }
tail -2 gpurun_out/${tag}_cpu_suite.log
cd $R && python3 bench.py > gpurun_out/${tag}_bench_dfsph_1m.json 2> gpurun_out/${tag}_bench.err || exit 1
tail -c 1500 gpurun_out/${tag}_bench_dfsph_1m.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o stats -- python3 $R/bench.py --no-cpu-baseline --no-scaling-base > $R/gpurun_out/${tag}_bench_dfsph_1m_under_rocprof.json 2> $R/gpurun_out/${tag}_stats.err || exit 1
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/${tag}_$name -o pmc -- python3 $R/bench.py --preroll 30 --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-scaling-base > /dev/null 2> $R/gpurun_out/${tag}_$name.err || exit 1
done
cd $R && python3 tools/pmc_traffic.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write 1000000 gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_sq
