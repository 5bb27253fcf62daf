#!/bin/bash
# PMC passes over one short run of the merge kernel (GPU box).  Usage: tools/pmc_run.sh <tag> [phase_profile args]
# Each pass is its own rocprofv3 run (counter slots: SQ 8, TCC 4 — FETCH_SIZE and WRITE_SIZE do not fit together).
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/$name -- python $ROOT/tools/phase_profile.py --no-phases --iters 2 "${EXTRA[@]}" > $OUT/$name.log 2>&1
  local db=$(find $OUT/$name -name '*.db' | head -1)
  [ -n "$db" ] && python $ROOT/tools/prof_summary.py $db --pmc > $OUT/$name.txt 2>&1
  rm -rf $OUT/$name
}
EXTRA=("$@")
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES
run waits SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT
run lds SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|name)|SQ_|TCC_|TCP_" | head -400 > $OUT/counters_available.txt
