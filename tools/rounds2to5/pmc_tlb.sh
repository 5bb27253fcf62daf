#!/bin/bash
# Address translation and memory latency of ptx_merge_kernel next to the calibration stream (TCP counters), separate passes.
# Usage: tools/pmc_tlb.sh <tag> [traffic_run.py args]   ->  gpurun_out/tlb_<tag>/*.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/tlb_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
EXTRA=("$@")
pass() {
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -- python "$ROOT/tools/traffic_run.py" "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  local db
  db=$(find "$OUT/$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc --all > "$OUT/$name.txt" 2>&1
  find "$OUT" -name '*.db' -delete
}
pass utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum
pass lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
grep -h "ptx_merge_kernel  .*TCP\|ptx_calib.*TCP" "$OUT/utcl1.txt" "$OUT/lat.txt"
tail -3 "$OUT/utcl1.log" "$OUT/lat.log" | cut -c1-300
