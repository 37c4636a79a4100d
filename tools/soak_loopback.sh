#!/bin/bash
# Long sharded runs over the NATIVE transport (tests/loopback_rccl.hip stands in for librccl: the ranks are handles of one process) against one GPU:
# cuts that follow the flow, walls that leak, the protocol switched every few steps, the Morton curve forced onto the small scenes, a rigid body.
#   bash tools/soak_loopback.sh > gpurun_out/soak_loopback.log
set -o pipefail
run() { # scene world steps rebalance extra-args env
  env SPH_DEV=1 SPH_SLAB_CHECK=1 $6 timeout -k 10 900 python3 tests/loopback_worker.py --scene $1 --world $2 --steps $3 --rebalance $4 $5 --out gpurun_out/soakl_$1_$2.json > gpurun_out/soakl_$1_$2.log 2>&1
  echo "rc=$? $1 world=$2 steps=$3 rebalance=$4 $5 $6"; python3 - <<PY
import json
r=json.load(open("gpurun_out/soakl_$1_$2.json"))
print({k:r[k] for k in ("pos_equal","vel_equal","rho_equal","stats_equal","body_equal","pos_rel_err")}, [(s["owned"],s["x_lo"],s["x_hi"],s["recuts"]) for s in r["slabs"]])
assert r["pos_equal"] and r["vel_equal"] and r["rho_equal"] and r["stats_equal"] and r["body_equal"] in (None, True)
PY
}
run dfsph_dam_x 3 2500 7 "" "" || exit 1
run dfsph_dam_x 4 1500 5 "--toggle-overlap 9" "SPH_CELL_ORDER=morton" || exit 1
run breaking_dam_30k_dfsph 4 500 6 "--toggle-overlap 13" "SPH_CELL_ORDER=morton" || exit 1
run breaking_dam_30k_dfsph 3 400 10 "--overlap 1" "SPH_CELL_ORDER=morton" || exit 1
run dfsph_rigid_tilted 3 300 9 "" "SPH_CELL_ORDER=morton" || exit 1
run wcsph_dam_x 3 6000 11 "" "" || exit 1
run dfsph_tiny_wall_iisph 3 1500 9 "" "" || exit 1
run dfsph_1m 8 60 10 "" "" || exit 1
