cd /root/repo
mkdir -p gpurun_out/r02_final
python bench.py > gpurun_out/r02_final/bench_default.json 2> gpurun_out/r02_final/bench_default.err
for c in c1 c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 1 --cpu-budget 5 > gpurun_out/r02_final/bench_$c.json 2> gpurun_out/r02_final/bench_$c.err || echo "bench $c failed"
done
python tools/config_perf.py dropin 2>&1 | grep -v amdgpu.ids > gpurun_out/r02_final/dropin.txt
cat gpurun_out/r02_final/dropin.txt
head -c 600 gpurun_out/r02_final/bench_default.json
