#!/bin/bash
# HBM traffic of the merge kernel on the bench workload, from rocprofv3 PMC counters (separate passes: FETCH_SIZE and
# WRITE_SIZE do not fit one pass; no trace domains besides --kernel-trace).  Usage: tools/pmc_traffic.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/traffic_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -- python "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  local db
  db=$(find "$OUT/$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc | grep -E "per_dispatch|^ptx_merge" > "$OUT/$name.txt" 2>&1
  find "$OUT" -name '*.db' -delete
}
EXTRA=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum
pass wrreq TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
grep -h '"metric"' "$OUT"/fetch.log | tail -1 > "$OUT/bench_line.json"
