#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# slabs with the large-scene path forced (Morton curve, LDS staging): small leaky scenes for thousands of steps, and the 1 M scene
set -o pipefail
run() { # scene world steps rebalance
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 300)) tests/slab_worker.py --scene $1 --steps $3 --backend gloo --rebalance $4 --out gpurun_out/soaks_$1_$2.json > gpurun_out/soaks_$1_$2.log 2>&1
  echo "rc=$? $1 world=$2 steps=$3"; python - <<PY
import json
r=json.load(open("gpurun_out/soaks_$1_$2.json"))
print({k:r[k] for k in ("pos_equal","vel_equal","rho_equal","stats_equal","pos_rel_err")}, [(s["owned"],s["x_lo"],s["x_hi"],s["recuts"]) for s in r["slabs"]])
PY
}
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 SPH_DEV=1 SPH_SLAB_CHECK=1     # (the host's edge-column bookkeeping is checked against the sorted arrays every step)
export SPH_CELL_ORDER=morton
run dfsph_dam_x 3 2500 7
run dfsph_tiny_wall_iisph 3 1500 9
unset SPH_CELL_ORDER
run dfsph_1m 3 40 10
run iisph_1m 2 25 10
