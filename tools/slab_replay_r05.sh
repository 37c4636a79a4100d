#!/bin/bash
# Round 5: rank <rank> of 8 of config 4 alone on the GPU (one recorded run, replays under development knobs; the final digest is checked)
#   bash tools/slab_replay_r05.sh <tag> [rank] ["name:ENV=.." ...]
tag=${1:-r05}; rank=${2:-2}; shift 2
bash tools/replay_variants.sh $tag dfsph_10m 8 $rank 50 20 "$@"
