#!/bin/bash
# Instruction counts per phase: the kernel truncated after phase k (a -DPTX_DIAG build, --stop-after k), one rocprofv3 PMC run each;
# differences between consecutive k are the phase's own instructions.  Usage: tools/pmc_phases.sh <tag> [phase_profile args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcph_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for k in 1 11 2 12 3 4 14 5 6 13 7 8 0; do
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $OUT/k$k -- python $ROOT/tools/phase_profile.py --lib peritext_amd/lib/exp_diag.so --stop-after $k --no-phases --no-check --iters 2 "$@" > $OUT/k$k.log 2>&1
  db=$(find $OUT/k$k -name '*.db' | head -1)
  [ -n "$db" ] && python $ROOT/tools/prof_summary.py $db --pmc | grep per_dispatch | sed "s/^/stop_after=$k /" >> $OUT/summary.txt
  rm -rf $OUT/k$k
done
