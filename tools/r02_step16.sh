cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_device_replay.py tests/test_gpu_kalman.py -x -q 2>&1 | tail -6
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
for lib in libpfgrad_unsorted.so libpfgrad.so; do
  for c in c5 c4; do
    echo "== $lib $c"
    PFGRAD_LIB=$CS/$lib timeout -k 10 200 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline --no-single-chain 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['config']['kernel_variant'], 'value', round(l['value']), 'kernel_ms', round(l['roofline']['kernel_ms'],3))"
    PFGRAD_LIB=$CS/$lib PFGRAD_VARIANT=mem1024 true
  done
done
# config 4 forced onto the large-N kernel (its default is the LDS-resident wg1024x4s): tools/config_perf has a c4 leg with mem1024; use a direct ensemble instead
python - <<'PY'
import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd')
PY
