#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# PMC passes over tools/placement_pmc.py: which counter separates a fast handle from a slow one?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SPH_ARENA_MODE=1
pass() {  # name counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pp_$name -o pp -- python3 $R/tools/placement_pmc.py dfsph_1m 30 6 > $R/gpurun_out/pp_$name.log 2>&1 || return 1
  tail -2 $R/gpurun_out/pp_$name.log
}
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum &&
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum &&
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum &&
pass dram TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_GMI_32B_sum TCC_EA0_RDREQ_LEVEL_sum
