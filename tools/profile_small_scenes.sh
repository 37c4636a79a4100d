#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# rocprofv3 kernel stats of the small / coupled scenes: is any small kernel unexpectedly heavy?  (round 3: the single-workgroup rigid reductions were)
R=$PWD; mkdir -p $R/gpurun_out/r03; cd /tmp; export TMPDIR=/tmp SPH_BENCH_PREROLL=0
for wl in breaking_dam_30k_dfsph coupling_demo breaking_dam_30k_pcisph; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/ps_$wl -o s -- python3 $R/bench.py --workload $wl --no-cpu-baseline --no-scaling-base --steps 200 --warmup 50 --profile-steps 0 > /dev/null 2> $R/gpurun_out/r03/ps_$wl.err
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/r03/ps_$wl/**/*kernel_stats.csv",recursive=True)[0]
print("== $wl")
for r in list(csv.DictReader(open(f)))[:14]:
    print("  ", r["Name"][:64].ljust(64), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
  rm -rf $R/gpurun_out/r03/ps_$wl
done
