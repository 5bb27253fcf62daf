cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01_i
for t in 128 192 256 320 384 512; do
  python tools/phase_profile.py --threads $t --no-phases --docs 8192 --flags 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['threads'], d['us_per_log_per_cu'], d['Gops_s'])"
done
python bench.py --steps 10 --warmup 2 > gpurun_out/r01_i/bench.log 2>&1; grep '"metric"' gpurun_out/r01_i/bench.log | tail -1 > gpurun_out/r01_i/bench.json; cat gpurun_out/r01_i/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r01_i/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/r01_i/kt.log 2>&1
cd $GRAFT_REPO_ROOT
db=$(find gpurun_out/r01_i/kt -name '*.db' | head -1)
python tools/prof_summary.py $db > gpurun_out/r01_i/kernel_stats.txt 2>&1
find gpurun_out/r01_i -name '*.db' -delete
cat gpurun_out/r01_i/kernel_stats.txt | head -20
bash tools/pmc_run.sh i --threads 256 --docs 8192 --flags 3 > /dev/null 2>&1
cat gpurun_out/pmc_i/lds.txt | head -30
