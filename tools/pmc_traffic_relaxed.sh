#!/bin/bash
# The two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) behind profiles/pmc_traffic_relaxed.json: the headline handle itself under the tolerance-grade
# arithmetic (development override SPH_ARITH=relaxed), short run.   bash tools/pmc_traffic_relaxed.sh <tag>   -> gpurun_out/<tag>_pmc_traffic_relaxed.json
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
export SPH_DEV=1 SPH_ARITH=relaxed SPH_BENCH_ALLOW_OVERRIDES=1
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/${tag}_rx_$name -o pmc -- python3 $R/bench.py --preroll 30 --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-scaling-base --no-relaxed > /dev/null 2> /tmp/${tag}_rx_$name.err || { tail -3 /tmp/${tag}_rx_$name.err; exit 1; }
done
cd $R && python3 tools/pmc_traffic.py /tmp/${tag}_rx_fetch /tmp/${tag}_rx_write 1000000 gpurun_out/${tag}_pmc_traffic_relaxed.json > /dev/null && python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_pmc_traffic_relaxed.json"))["kernels"]
for k in sorted(d):
    if "rx" in k: print("%-28s %.1f MB per launch" % (k, d[k]["hbm_bytes_per_launch"] / 1e6))
PY
