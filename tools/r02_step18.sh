cd /root/repo
for v in "" big; do
  echo "== c4 PFGRAD_VARIANT=$v"
  PFGRAD_VARIANT=$v timeout -k 10 200 python bench.py --config c4 --steps 4 --warmup 1 --no-cpu-baseline --no-single-chain 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['config']['kernel_variant'], 'value', round(l['value']), 'kernel_ms', round(l['roofline']['kernel_ms'],3))"
done
bash tools/r02_profiles.sh c5 > /dev/null 2>&1
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
