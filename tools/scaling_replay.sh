#!/bin/bash
# The strong-scaling yardstick a one-GPU box can run: every rank of a sharded run of <scene> replayed ALONE on the GPU (tests/loopback_rccl.hip
# record / replay, link time zero), next to the whole scene on one handle in the same process.  The step of a sharded run is its slowest rank's.
#   bash tools/scaling_replay.sh <tag> [scene] [preroll] [timed] ["2 4 8"]      -> gpurun_out/<tag>_scaling.txt
# One recorded run per slab count would log only one rank; each rank is therefore recorded and replayed on its own (tools/loopback_replay.sh).
tag=${1:-r05}; scene=${2:-dfsph_10m}; pre=${3:-50}; timed=${4:-20}; worlds=${5:-"2 4 8"}
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
out=gpurun_out/${tag}_scaling.txt
: > $out
for w in $worlds; do
  for r in $(seq 0 $((w - 1))); do
    LOOPBACK_OVERLAP=${LOOPBACK_OVERLAP:-0} bash tools/loopback_replay.sh ${tag}s $scene $w $r $pre $timed > gpurun_out/${tag}s_${w}_${r}.log 2>&1 || { tail -5 gpurun_out/${tag}s_${w}_${r}.log; exit 1; }
    python3 - <<PY >> $out
import json
b = json.load(open("gpurun_out/${tag}s_replay_${scene}_${w}_rank${r}.json"))
print("%5d %5d %9d %9d %14.3f %14.3f" % ($w, $r, b["replay"]["owned"], b["replay"]["ghosts"], b["replay"]["ms_per_step"], b["one_gpu"]["ms_per_step"]))
PY
    tail -1 $out
  done
done
python3 - <<PY
import collections
rows = [l.split() for l in open("$out") if l.strip()]
by = collections.defaultdict(list)
for w, r, own, gh, ms, one in rows:
    by[int(w)].append((float(ms), float(one)))
with open("$out", "a") as f:
    for w, v in sorted(by.items()):
        slow = max(m for m, _ in v); one = sum(o for _, o in v) / len(v)
        line = "%d slabs: slowest rank %.3f ms per step, one GPU %.3f ms -> %.2fx (%.0f %% of %d); link time zero" % (w, slow, one, one / slow, 100 * one / slow / w, w)
        print(line); f.write(line + "\n")
PY
