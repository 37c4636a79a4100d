#!/bin/bash
# The two separate --pmc passes (FETCH_SIZE, WRITE_SIZE) behind profiles/pmc_traffic.json, on a short run:  bash tools/pmc_traffic_only.sh <tag>
tag=${1:-x}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/${tag}_$name -o pmc -- python3 $R/bench.py --preroll 30 --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-scaling-base > /dev/null 2> /tmp/${tag}_$name.err || { tail -3 /tmp/${tag}_$name.err; exit 1; }
done
cd $R && python3 tools/pmc_traffic.py /tmp/${tag}_fetch /tmp/${tag}_write 1000000 gpurun_out/${tag}_pmc_traffic.json > /dev/null && python3 - <<PY
import json
d = json.load(open("gpurun_out/${tag}_pmc_traffic.json"))["kernels"]
for k in ("dfsph_div_residual", "dfsph_div_correct", "dfsph_dens_residual", "dfsph_dens_correct", "dfsph_density_alpha", "dfsph_ext_force", "build_nl"):
    if k in d: print("%-22s %.1f MB per launch (fetch raw %.1f x 2 + write %.1f)" % (k, d[k]["hbm_bytes_per_launch"] / 1e6, d[k]["fetch_bytes_raw_per_launch"] / 1e6, d[k]["write_bytes_per_launch"] / 1e6))
PY
