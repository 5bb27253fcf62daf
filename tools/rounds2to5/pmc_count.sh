#!/bin/bash
# Instruction totals of the merge kernel for one or more builds: tools/pmc_count.sh <tag> lib1.so [lib2.so ...]  (8 192 documents of config 4)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmccount_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename "$lib" .so)
  timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d "$OUT/$name" -- python "$ROOT/tools/phase_profile.py" --lib "$lib" --no-phases --iters 2 --docs 8192 > "$OUT/$name.log" 2>&1
  db=$(find "$OUT/$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc | grep per_dispatch | sed "s/^/$name /" | tee -a "$OUT/summary.txt"
  rm -rf "$OUT/$name"
done
