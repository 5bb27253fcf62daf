#!/bin/bash
# Memory-path counters of the merge kernel (TA/TCP/TCC/LDS) + the full counter list.
# Usage: tools/pmc_mem.sh <tag> [phase_profile args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcmem_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "Counter_Name\s*:\s*\S+" | sed 's/Counter_Name\s*:\s*//' | sort -u > "$OUT/counters.txt"
EXTRA=("$@")
run() {
  local name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -- python "$ROOT/tools/phase_profile.py" --no-phases --iters 2 "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  local db
  db=$(find "$OUT/$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc | grep per_dispatch > "$OUT/$name.txt" 2>&1
  find "$OUT" -name '*.db' -delete   # the raw traces are large; only the summaries travel back
}
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_TOTAL_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
run tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
run lds SQ_INSTS_LDS_ATOMIC SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
run busy GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
