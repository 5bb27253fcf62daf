#!/bin/bash
# Read requests of the L2 to the fabric by size (gfx950 has 32 / 64 / 128-byte request counters) and the DRAM-bound part,
# for ptx_merge_kernel and the calibration stream on the bench workload: the exact read bytes = 32 a + 64 b + 128 c,
# without the single calibration factor of tools/pmc_traffic.sh.  Separate passes, --kernel-trace only.
# Usage: tools/pmc_reqs.sh <tag> [traffic_run.py args]   ->  gpurun_out/reqs_<tag>/{size,dram}.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/reqs_$TAG
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
pass size TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass dram TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
grep -h "ptx_merge_kernel \|ptx_calib" "$OUT/size.txt" "$OUT/dram.txt"
