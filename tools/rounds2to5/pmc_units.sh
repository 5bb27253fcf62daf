#!/bin/bash
# Which unit of the CU is the merge kernel's limit?  Issue / busy counters of the sequencer (VALU, scalar, LDS, memory instructions), the instruction and scalar
# caches, the texture addresser / data return path, and why the dispatcher could not place the next workgroup.  Separate PMC passes, --kernel-trace only.
# Usage: tools/pmc_units.sh <tag> [phase_profile args]   ->  gpurun_out/pmcunits_<tag>/*.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcunits_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
EXTRA=("$@")
run() {
  local name=$1; shift
  timeout 240 rocprofv3 --pmc "$@" --kernel-trace -d "$OUT/$name" -- python "$ROOT/tools/phase_profile.py" --no-phases --iters 2 "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
  local db
  db=$(find "$OUT/$name" -name '*.db' | head -1)
  [ -n "$db" ] && python "$ROOT/tools/prof_summary.py" "$db" --pmc | grep per_dispatch > "$OUT/$name.txt" 2>&1
  find "$OUT" -name '*.db' -delete
  echo "== pass $name"; cat "$OUT/$name.txt" 2>/dev/null
}
run issue GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VALU_INT64 SQ_IFETCH
run caches SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQ_INST_CYCLES_SMEM
run ta TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
run td TD_TD_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run spi SPI_RA_LDS_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_SGPR_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN SPI_RA_REQ_NO_ALLOC_CSN SPI_CSN_BUSY
