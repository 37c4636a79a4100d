#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# Round-4 measurement set at the final kernels (gpurun box): every BASELINE config that fits one GPU (tools/measure_configs.sh) + the round's extra scenes,
# both arithmetics where a relaxed path exists (dfsph staged scenes, wcsph, dfsph next to a rigid body), and the rehearsal of the sharded path
# (config 4 on 2 and 4 ranks sharing the one GPU over gloo: message counts, not speed).  Output: gpurun_out/measure_*.json
set -o pipefail
[ "$SKIP_CONFIGS" = 1 ] || bash tools/measure_configs.sh || exit 1      # (SKIP_CONFIGS=1: the configs were measured in an earlier gpurun call)
mkdir -p gpurun_out
unset SPH_BENCH_PREROLL
for wl in breaking_dam_demo default dfsph_rigid_2m_clear; do
  python bench.py --workload $wl --no-cpu-baseline --no-scaling-base > gpurun_out/measure_r04_$wl.json 2> gpurun_out/measure_r04_$wl.err || exit 1
done
# the pressure solvers and the small dfsph scene under the relaxed arithmetic, same windows as measure_configs.sh
for wl in pcisph_1m iisph_1m; do
  SPH_BENCH_PREROLL=0 SPH_ARITH=relaxed python bench.py --workload $wl --steps 100 --warmup 20 --no-cpu-baseline --no-scaling-base > gpurun_out/measure_r04_${wl}_relaxed.json 2> gpurun_out/measure_r04_${wl}_relaxed.err || exit 1
done
SPH_ARITH=relaxed python bench.py --workload default --no-cpu-baseline --no-scaling-base > gpurun_out/measure_r04_default_relaxed.json 2> gpurun_out/measure_r04_default_relaxed.err || exit 1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-scaling-base > gpurun_out/measure_r04_dfsph_1m_driver_flags.json 2> gpurun_out/measure_r04_driver_flags.err || exit 1
# relaxed arithmetic next to a body (the bench's relaxed leg does not build bodies: the headline handle itself runs relaxed, config.arith says so)
for wl in dfsph_rigid_2m_clear dfsph_rigid_2m; do
  SPH_ARITH=relaxed python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-scaling-base > gpurun_out/measure_r04_${wl}_relaxed.json 2> gpurun_out/measure_r04_${wl}_relaxed.err || exit 1
done
# the sharded path, rehearsed: all ranks on GPU 0 over gloo
for n in 2 4; do
  SPH_BENCH_REHEARSAL=1 SPH_BENCH_PREROLL=20 python bench.py --gpus $n --steps 10 --warmup 5 > gpurun_out/measure_r04_rehearsal_${n}ranks.json 2> gpurun_out/measure_r04_rehearsal_${n}ranks.err || exit 1
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/measure_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    if "value" in d:
        r = d.get("relaxed") or {}
        c = d.get("config", {})
        print(f.split("/")[-1], round(d["value"], 1), round(d["ms_per_step"], 4), c.get("arith"), c.get("timed_steps"), c.get("n_dens_mean", c.get("pressure_iterations_mean")), "relaxed", r.get("value"),
              (c.get("rank0_comm") or {}).get("per_step"))
PY
