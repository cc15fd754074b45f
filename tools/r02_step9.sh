cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_kalman.py tests/test_gpu_sampler.py tests/test_gpu_ensemble.py -x -q 2>&1 | tail -8
mkdir -p gpurun_out/r02_bench
for c in c2 c1 c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 1 --cpu-budget 6 > gpurun_out/r02_bench/bench_$c.json 2> gpurun_out/r02_bench/bench_$c.err || echo "bench $c failed"
  tail -c 1500 gpurun_out/r02_bench/bench_$c.json; echo
done
