for v in "" 8 16 32 64 128 256 512; do
  if [ -z "$v" ]; then l=""; else l="--lib peritext_amd/lib/exp_rp_x$v.so"; fi
  timeout 100 python tools/replay_bench.py --docs 4096 $l 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['lib'], round(d['kernel_ms'],2), d['patches'], d['launches'])"
done
