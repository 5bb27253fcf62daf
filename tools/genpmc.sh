#!/bin/bash
# instruction counters of the generator kernel: bash tools/genpmc.sh  -> gpurun_out/genpmc/gen.txt
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/genpmc
mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d $OUT/p -- python $GRAFT_REPO_ROOT/tools/gen_bench.py --docs 12288 --list-cap 1536 > $OUT/log.txt 2>&1
db=$(find $OUT/p -name '*.db' | head -1)
python $GRAFT_REPO_ROOT/tools/prof_summary.py $db --pmc --all > $OUT/all.txt
grep -E "ptx_gen_kernel" $OUT/all.txt | cut -c1-200
rm -rf $OUT/p
grep '^{' $OUT/log.txt | cut -c1-300
