#!/bin/bash
# The native transport (the one a multi-GPU node runs: stream-ordered, halo and all-reduce on their own streams) rehearsed on ONE GPU:
# tests/loopback_rccl.hip stands in for librccl, the ranks are handles of one process (tests/loopback_worker.py).  All ranks share the GPU, so the
# time per step is the SUM of the ranks' GPU work: against the one-GPU step of the same scene it prices what sharding adds on the device --
# ghosts' sweeps, packing, split launches, the reductions -- with no host waits in the way (the gloo rehearsal measures mostly those).
#   bash tools/loopback_rehearsal.sh [tag]   ->  gpurun_out/<tag>_loopback_*.json
set -o pipefail
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
export SPH_DEV=1
# bit-identity first, at a size where sweeps take tens of microseconds and three streams really overlap
SPH_SLAB_CHECK=1 python3 tests/loopback_worker.py --scene dfsph_1m --world 2 --steps 6 --out gpurun_out/${tag}_loopback_dfsph_1m_2.json || exit 1
SPH_SLAB_CHECK=1 python3 tests/loopback_worker.py --scene dfsph_1m --world 4 --steps 4 --rebalance 2 --out gpurun_out/${tag}_loopback_dfsph_1m_4.json || exit 1
python3 - <<PY
import json
for w in (2, 4):
    r = json.load(open("gpurun_out/${tag}_loopback_dfsph_1m_%d.json" % w))
    print("dfsph_1m on %d loopback ranks: equal to one GPU:" % w, r["pos_equal"], r["vel_equal"], r["rho_equal"], r["stats_equal"])
    assert r["pos_equal"] and r["vel_equal"] and r["rho_equal"] and r["stats_equal"]
PY
# the timing: config 4's scene, steps 51-70 (the window of the one-GPU line measure_c4_dfsph_10m_1gpu.json)
for w in 2 4 8; do
  SPH_SLAB_CHECK=0 python3 tests/loopback_worker.py --scene dfsph_10m --world $w --steps 50 --time 20 --no-compare --out gpurun_out/${tag}_loopback_dfsph_10m_$w.json || exit 1
  python3 -c "import json; r = json.load(open('gpurun_out/${tag}_loopback_dfsph_10m_$w.json')); print('dfsph_10m, %d ranks sharing one GPU: %.2f ms per step' % ($w, r['timing']['ms_per_step']), r['lib_comm'], [s['owned'] for s in r['slabs']])"
done
