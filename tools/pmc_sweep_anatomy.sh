#!/bin/bash
# TCP / SQ / TCC counters of the sweeps (a pass with the TA_* counters hung rocprofv3 on this pool: left out) (separate --pmc passes on a short run) -> gpurun_out/anat_*/ ; tools/pmc_sweep_report.py prints per-kernel means
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
pass() { name=$1; shift
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/anat_$name -o a -- python3 $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/anat_$name.err || return 1; }
pass tcp TCP_TOTAL_ACCESSES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_FLAT_READ_WAVEFRONTS_sum &&
pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE &&
pass l2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
