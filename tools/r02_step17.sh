cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_device_replay.py tests/test_gpu_kalman.py tests/test_gpu_pf_parity.py -x -q 2>&1 | tail -6
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
for lib in libpfgrad_unsorted.so libpfgrad.so; do
  echo "== $lib c5"
  PFGRAD_LIB=$CS/$lib timeout -k 10 200 python bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-single-chain 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(l['config']['kernel_variant'], 'value', round(l['value']), 'kernel_ms', round(l['roofline']['kernel_ms'],3))"
done
cd /tmp && export TMPDIR=/tmp
mkdir -p /root/repo/gpurun_out/r02_c5sorted
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /root/repo/gpurun_out/r02_c5sorted/fetch -o pmc -- python3 /root/repo/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-single-chain > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /root/repo/gpurun_out/r02_c5sorted/write -o pmc -- python3 /root/repo/bench.py --config c5 --steps 5 --warmup 1 --no-cpu-baseline --no-single-chain > /dev/null 2>&1
python3 /root/repo/tools/pmc_table.py /root/repo/gpurun_out/r02_c5sorted/fetch pf_big | tail -2
python3 /root/repo/tools/pmc_table.py /root/repo/gpurun_out/r02_c5sorted/write pf_big | tail -2
