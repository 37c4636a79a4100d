#!/bin/bash
export SPH_DEV=1 SPH_BENCH_ALLOW_OVERRIDES=1     # the SPH_* knobs below are development overrides (include/sph_mi355x.h: sph_overrides)
# SQ / LDS counters of the sweeps, separate --pmc passes over a short run; prints per-kernel means (tools/pmc_sweep_report.py) into
# gpurun_out/anat_report[_TAG].txt and removes the (large) rocpd databases.  (A pass with the TA_* counters hung rocprofv3 on this pool: left out.)
# Usage: [SPH_LIB=alt.so] [TAG=name] tools/pmc_sweep_anatomy.sh [bench args]       values are means per (launch, XCD x SE instance): x 32 = per launch
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS=${@:---preroll 30 --steps 3 --warmup 1}
OUT=$R/gpurun_out/anat_report${TAG:+_$TAG}.txt
: > $OUT
pass() { name=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/anat_$name -o a -- python3 $R/bench.py $ARGS --profile-steps 0 --no-cpu-baseline --no-scaling-base > /dev/null 2> /tmp/anat_$name.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "pass $name failed rc=$rc" >> $OUT; tail -5 /tmp/anat_$name.err >> $OUT; return 0; fi
  python3 $R/tools/pmc_sweep_report.py $(find /tmp/anat_$name -name "*.db") >> $OUT 2>&1; rm -rf /tmp/anat_$name; }
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
pass sq2 SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
pass sq3 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_LEVEL_WAVES
[ -n "$QUIET" ] || cut -c1-1200 $OUT
