#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# Runs the BASELINE.json configs that fit one GPU and the CPU baselines of BASELINE.md section 3 on a gpurun box.
# Output: gpurun_out/measure_<name>.json
set -o pipefail
mkdir -p gpurun_out
export SPH_BENCH_PREROLL=0      # the windows below are SURVEY.md 8d's: warm-up + timed steps, no extra pre-roll
python bench.py --workload breaking_dam_30k_wcsph --steps 1000 --warmup 200 --no-cpu-baseline > gpurun_out/measure_c1_wcsph.json 2> gpurun_out/measure_c1_wcsph.err
python bench.py --workload breaking_dam_30k_dfsph --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/measure_c1_dfsph.json 2> gpurun_out/measure_c1_dfsph.err
python bench.py --workload wcsph_250k --steps 200 --warmup 50 > gpurun_out/measure_c2_wcsph_250k.json 2> gpurun_out/measure_c2.err
python bench.py --workload dfsph_1m --steps 200 --warmup 50 > gpurun_out/measure_c3_dfsph_1m.json 2> gpurun_out/measure_c3.err
python bench.py --workload dfsph_rigid_2m --steps 50 --warmup 10 --no-scaling-base > gpurun_out/measure_c5_dfsph_rigid_2m.json 2> gpurun_out/measure_c5.err
python bench.py --workload dfsph_10m --steps 50 --warmup 50 --no-cpu-baseline > gpurun_out/measure_c4_dfsph_10m_1gpu.json 2> gpurun_out/measure_c4.err
python bench.py --workload breaking_dam_30k_iisph --steps 300 --warmup 100 > gpurun_out/measure_c1_iisph_as_shipped.json 2> gpurun_out/measure_c1i.err
python bench.py --workload breaking_dam_30k_pcisph --steps 300 --warmup 100 > gpurun_out/measure_c1_pcisph.json 2> gpurun_out/measure_c1p.err
python bench.py --workload pcisph_1m --steps 100 --warmup 20 > gpurun_out/measure_c3_pcisph_1m.json 2> gpurun_out/measure_c3p.err
python bench.py --workload iisph_1m --steps 100 --warmup 20 > gpurun_out/measure_c3_iisph_1m.json 2> gpurun_out/measure_c3i.err
python bench.py --workload breaking_dam_30k_pbf --steps 1000 --warmup 200 --no-cpu-baseline --profile-steps 0 > gpurun_out/measure_c1_pbf.json 2> gpurun_out/measure_c1b.err
python bench.py --workload coupling_demo --steps 200 --warmup 50 > gpurun_out/measure_coupling_demo_pcisph_rigid.json 2> gpurun_out/measure_cd.err
python - <<'PY' > gpurun_out/measure_cpu_config1.json
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from oracle import oracle as orc
from cfd_taichi_amd import scenes
import bench
res = {"cores_available": bench.host_cores()}
for solver, scene, steps in (("wcsph", "breaking_dam_30k_wcsph", 300), ("dfsph", "breaking_dam_30k_dfsph", 60)):
    for threads in (1, bench.host_cores()):
        o = orc.Oracle(scenes.get(scene), num_threads=threads)
        warm = 20
        (o.step_wcsph(warm) if solver == "wcsph" else [o.step_dfsph(1, 100) for _ in range(warm)])
        t0 = time.perf_counter()
        if solver == "wcsph":
            o.step_wcsph(steps)
        else:
            for _ in range(steps):
                o.step_dfsph(1, 100)
        dt = time.perf_counter() - t0
        res["%s_%dthreads" % (solver, threads)] = {"Mparticle_steps_per_s": o.N * steps / dt / 1e6, "steps": steps, "seconds": dt, "N": o.N}
        o.close()
print(json.dumps(res))
PY
for f in gpurun_out/measure_*.json; do echo "== $f"; python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1])
if 'value' in d:
    print(round(d['value'],2), d['unit'], round(d['ms_per_step'],3),'ms/step', {k:d['config'].get(k) for k in ('workload','n_div_mean','n_dens_mean')}, 'roofline', {k: (round(v,4) if isinstance(v,float) else v) for k,v in d.get('roofline',{}).items() if k in ('kernel','achieved','frac','avg_launch_us')}, 'cpu', d.get('cpu_baseline',{}).get('value'))
else: print(d)
"; done
