#!/bin/bash
# tools/phase_insts.sh <tag> <config> <docs>: one rocprofv3 PMC pass over tools/phase_insts.py -> gpurun_out/phinsts_<tag>.txt
set -u
TAG=$1; CFG=$2; DOCS=$3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/phinsts_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$OUT" -- python "$ROOT/tools/phase_insts.py" --config "$CFG" --docs "$DOCS" > "$OUT.log" 2>&1
db=$(find "$OUT" -name '*.db' | head -1)
LOGS=$(grep '^logs' "$OUT.log" | awk '{print $2}')
python "$ROOT/tools/phase_insts.py" --table "$db" --logs "${LOGS:-$DOCS}" > "$ROOT/gpurun_out/phinsts_$TAG.txt" 2>&1
rm -rf "$OUT"
cat "$ROOT/gpurun_out/phinsts_$TAG.txt"
