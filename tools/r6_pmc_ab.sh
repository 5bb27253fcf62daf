#!/bin/bash
# instruction / activity counters of the bench's own kernel under two (or more) builds: tools/r6_pmc_ab.sh <tag> <lib> [<lib> ...]  (GPU box)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for LIB in "$@"; do
  B=$(basename $LIB .so)
  bash tools/pmc_one.sh ${TAG}_${B}_a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" --lib $LIB --docs 65536
  bash tools/pmc_one.sh ${TAG}_${B}_b "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" --lib $LIB --docs 65536
  echo "== $B"; cat gpurun_out/pmc1_${TAG}_${B}_a/summary.txt gpurun_out/pmc1_${TAG}_${B}_b/summary.txt | grep lean | awk '{print $1, $2, $NF}'
  grep '"ms"' gpurun_out/pmc1_${TAG}_${B}_a/run.log | cut -c1-200
done
