#!/bin/bash
# The two slab protocols against a link that is not free: ONE rank replayed alone (tools/loopback_replay.sh), with every group of point-to-point
# transfers and every all-reduce of the replay preceded by a kernel that waits LOOPBACK_LATENCY_US microseconds on its stream.
#   bash tools/link_latency_sweep.sh <tag> <scene> <world> <rank> <preroll> <timed> "0 10 20 40"
set -o pipefail
tag=$1; scene=$2; world=$3; rank=$4; pre=$5; timed=$6; lats=${7:-"0 10 20 40"}
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
export SPH_DEV=1 SPH_SLAB_CHECK=0
for ov in 2 0; do
  log=${TMPDIR:-/tmp}/lat_${scene}_${world}_${rank}_$ov.log
  python3 tests/loopback_worker.py --scene $scene --world $world --steps $pre --time $timed --no-compare --overlap $ov --replay-rank $rank --save-log $log --out gpurun_out/${tag}_lat_rec_$ov.json || exit 1
  for us in $lats; do
    LOOPBACK_LATENCY_US=$us python3 tests/loopback_worker.py --scene $scene --world $world --steps $pre --time $timed --no-compare --overlap $ov --replay-rank $rank --load-log $log --out gpurun_out/${tag}_lat_${ov}_$us.json || exit 1
    python3 -c "
import json; b=json.load(open('gpurun_out/${tag}_lat_${ov}_$us.json')); a=json.load(open('gpurun_out/${tag}_lat_rec_$ov.json'))
print('%s  latency %3d us per transfer group / all-reduce:  rank $rank of $world alone %.3f ms per step   (same state: %s)' % ('overlapped' if $ov == 2 else 'in order  ', $us, b['replay']['ms_per_step'], a['recorded']['digest']==b['replay']['digest']))"
  done
  rm -f $log
done
