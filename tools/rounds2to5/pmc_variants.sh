#!/bin/bash
# L2 -> fabric read requests (all / 128-byte / DRAM-bound) of ptx_merge_kernel for several builds of the library on the 8 192-document
# shard: one rocprofv3 PMC pass (--kernel-trace only) per build.  Usage: tools/pmc_variants.sh <tag> <flags> <lib|default> ...
set -u
TAG=$1; FLAGS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcvar_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  L=""; name=default
  if [ "$lib" != default ]; then L="--lib $ROOT/$lib"; name=$(basename "$lib" .so); fi
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace -d "$OUT/$name" -- python "$ROOT/tools/phase_profile.py" --no-phases --no-check --iters 3 --flags "$FLAGS" $L > "$OUT/$name.log" 2>&1
  db=$(find "$OUT/$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc | grep "ptx_merge_kernel " | sed "s/^/$name /" >> "$OUT/summary.txt"
  rm -rf "$OUT/$name"
done
cat "$OUT/summary.txt"
