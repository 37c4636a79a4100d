#!/bin/bash
# ONE rank of a sharded run alone on the GPU (tests/loopback_rccl.hip record / replay): its device time per step with the link time set to zero,
# and (7th argument "profile") its kernel profile.  LOOPBACK_REBALANCE=M: the cuts re-chosen every M steps (SphConfig.slab_rebalance_every; bench.py's default is 50).  LOOPBACK_OVERLAP=2 in the environment: the overlapped protocol (SphConfig.slab_overlap; 0 = the native transport's default, in order).   bash tools/loopback_replay.sh <tag> <scene> <world> <rank> <preroll> <timed> [profile]
# Phase 1: all ranks in one process (threads), what rank <rank> receives goes to a log file.  Phase 2: a fresh process replays that rank
# against the log -- one thread, one handle -- and must end in the same state (digest of ids, positions, velocities, densities).
set -o pipefail
tag=$1; scene=$2; world=$3; rank=$4; pre=$5; timed=$6; prof=$7
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
export SPH_DEV=1 SPH_SLAB_CHECK=0
out=gpurun_out/${tag}_replay_${scene}_${world}_rank${rank}
log=${TMPDIR:-/tmp}/loopback_${scene}_${world}_${rank}.log
python3 tests/loopback_worker.py --scene $scene --world $world --steps $pre --time $timed --no-compare --overlap ${LOOPBACK_OVERLAP:-0} --rebalance ${LOOPBACK_REBALANCE:-0} --replay-rank $rank --save-log $log --out ${out}_recorded.json || exit 1
if [ -n "$prof" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $R/${out}_trace -o trace -- python3 $R/tests/loopback_worker.py --scene $scene --world $world --steps $pre --time $timed --no-compare --overlap ${LOOPBACK_OVERLAP:-0} --rebalance ${LOOPBACK_REBALANCE:-0} --replay-rank $rank --load-log $log --one-gpu --out $R/$out.json || exit 1
  cd $R && python3 tools/replay_trace.py ${out}_trace 0 3000 > $out.kernels.txt && python3 tools/replay_trace.py ${out}_trace 0 @k_dfsph_integrate:10:1800 | sed -n "/^timeline slice/,\$p" > $out.exchange_slice.txt && python3 tools/replay_trace.py ${out}_trace 1 > $out.one_gpu_kernels.txt && rm -rf ${out}_trace && tail -80 $out.kernels.txt && grep -A12 "^kernel " $out.one_gpu_kernels.txt
else
  python3 tests/loopback_worker.py --scene $scene --world $world --steps $pre --time $timed --no-compare --overlap ${LOOPBACK_OVERLAP:-0} --rebalance ${LOOPBACK_REBALANCE:-0} --replay-rank $rank --load-log $log --one-gpu --out $out.json || exit 1
fi
rm -f $log
python3 - <<PY
import json
a, b = json.load(open("${out}_recorded.json")), json.load(open("$out.json"))
same = a["recorded"]["digest"] == b["replay"]["digest"]
print("$scene, rank $rank of $world: all ranks on one GPU %.2f ms per step; this rank alone %.3f ms per step; the whole scene on one handle %.3f ms per step (= %.2fx); same final state: %s" % (a["timing"]["ms_per_step"], b["replay"]["ms_per_step"], b["one_gpu"]["ms_per_step"], b["one_gpu"]["ms_per_step"] / b["replay"]["ms_per_step"], same), b["replay"])
assert same
PY
