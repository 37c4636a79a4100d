#!/bin/bash
# One recorded run, several replays of the same rank under different development knobs (they must not change the bits: the final digest is checked).
#   bash tools/replay_variants.sh <tag> <scene> <world> <rank> <preroll> <timed> "name1:ENV=1 ENV2=x" "name2:..." ...
set -o pipefail
tag=$1; scene=$2; world=$3; rank=$4; pre=$5; timed=$6; shift 6
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
export SPH_DEV=1 SPH_SLAB_CHECK=0
log=${TMPDIR:-/tmp}/loopback_${scene}_${world}_${rank}.log
out=gpurun_out/${tag}_variants_${scene}_${world}_rank${rank}
python3 tests/loopback_worker.py --scene $scene --world $world --steps $pre --time $timed --no-compare --overlap ${LOOPBACK_OVERLAP:-0} --replay-rank $rank --save-log $log --out ${out}_recorded.json || exit 1
for v in "base:" "$@"; do
  name=${v%%:*}; envs=${v#*:}
  env $envs python3 tests/loopback_worker.py --scene $scene --world $world --steps $pre --time $timed --no-compare --overlap ${LOOPBACK_OVERLAP:-0} --replay-rank $rank --load-log $log --out ${out}_$name.json || exit 1
  python3 - <<PY
import json
a, b = json.load(open("${out}_recorded.json")), json.load(open("${out}_$name.json"))
print("%-14s %-40s %.3f ms per step   same state: %s" % ("$name", "$envs", b["replay"]["ms_per_step"], a["recorded"]["digest"] == b["replay"]["digest"]))
PY
done
rm -f $log
