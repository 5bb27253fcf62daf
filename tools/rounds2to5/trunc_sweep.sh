#!/bin/bash
# Throughput time of the merge kernel truncated after a phase (a -DPTX_DIAG build, --stop-after = stamp index: 2 after P1, 3 P3a, 4 P3b, 5 P3d, 6 P4, 7 P5a,
# 8 P5c, 0 whole kernel), for one or more builds of the library.  Usage: tools/trunc_sweep.sh <out.jsonl> <flags> <lib|default> ...
OUT=$1; FLAGS=$2; shift 2
for lib in "$@"; do
  for k in 2 3 5 7 0; do
    L=""; [ "$lib" != default ] && L="--lib $lib"
    timeout 120 python tools/phase_profile.py --lib peritext_amd/lib/exp_diag.so --stop-after $k --no-phases --no-check --iters 10 --flags $FLAGS $L 2>&1 | sed "s/^{/{\"stop_after\": $k, /" >> $OUT
  done
done
