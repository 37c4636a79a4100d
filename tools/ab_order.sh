#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# A/B of the cell storage order (Morton curve with tiles of 4/8/16 cells per axis, the reference's linear stride): whole steps, interleaved in one process.
set -e
L=cfd_taichi_amd/libsph_mi355x.so
for sc in ${SCENES:-dfsph_1m pcisph_1m iisph_1m wcsph_250k}; do
  echo "== $sc" | tee -a gpurun_out/ab_order.txt
  AB_CHUNK=${AB_CHUNK:-20} python tools/ab_steps.py $sc 40 m4=$L:SPH_CELL_TILE=4 m8=$L:SPH_CELL_TILE=8 m16=$L:SPH_CELL_TILE=16 linear=$L:SPH_CELL_ORDER=linear 2>&1 | tee -a gpurun_out/ab_order.txt
done
