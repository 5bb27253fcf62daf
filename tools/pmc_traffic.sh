#!/bin/bash
# HBM traffic of ptx_merge_kernel on the bench workload from rocprofv3 PMC counters (MI355X_MICROARCH.md "HBM"): separate passes,
# --kernel-trace only.  Reads: FETCH_SIZE calibrated on a known byte count (gfx950 tallies 128-byte requests at 64) AND, exactly,
# the L2's read requests by size (32 / 64 / 128 bytes); writes: WRITE_SIZE.
# Usage: tools/pmc_traffic.sh <tag> [traffic_run.py args]   ->  gpurun_out/traffic_<tag>/{fetch,write,size}.txt + hbm_traffic.json
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/traffic_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -- python "$ROOT/tools/traffic_run.py" "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  local db
  db=$(find "$OUT/$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc --all > "$OUT/$name.txt" 2>&1
  find "$OUT" -name '*.db' -delete
}
EXTRA=("$@")
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass size TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
python "$ROOT/tools/traffic_json.py" "$OUT" > "$OUT/hbm_traffic.json"
cat "$OUT/hbm_traffic.json"
