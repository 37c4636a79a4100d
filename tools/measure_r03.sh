#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# Round-3 measurement set at the final kernels: every BASELINE config that fits one GPU (tools/measure_configs.sh) plus the round's extra scenes
# (both arithmetics where it applies).  Output: gpurun_out/measure_*.json
set -o pipefail
bash tools/measure_configs.sh || exit 1
unset SPH_BENCH_PREROLL
for wl in breaking_dam_demo default dfsph_rigid_2m_clear; do
  python bench.py --workload $wl --no-cpu-baseline --no-scaling-base > gpurun_out/measure_r03_$wl.json 2> gpurun_out/measure_r03_$wl.err || exit 1
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/measure_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" in d:
        r = d.get("relaxed") or {}
        c = d.get("config", {})
        print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), c.get("timed_steps"), c.get("n_dens_mean", c.get("pressure_iterations_mean")), "relaxed", r.get("value"))
PY
