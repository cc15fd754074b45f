#!/bin/bash
# Round-4 final evidence run (via gpurun, after tools/r04_profiles.sh): the default bench line and every other config
# un-profiled, drop-in timings (Sampler API: one chain, host in the loop; fit resident; configs 4 / 5 seed-compatible),
# the giant-N drop-in call, whole-GPU-window timings per N, phase stamps.  Outputs under gpurun_out/r04_final/.
cd /root/repo
OUT=gpurun_out/r04_final
rm -rf $OUT; mkdir -p $OUT
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "default done"
for c in c1 c3 c4 c5 g1; do
  timeout -k 10 300 python bench.py --config $c --cpu-budget 5 > $OUT/bench_$c.json 2> $OUT/bench_$c.err || echo "bench $c failed"
  echo "$c done"
done
for m in dropin dropin_fit dropin_large; do
  timeout -k 10 300 python tools/config_perf.py $m 2>&1 | grep -v amdgpu.ids >> $OUT/dropin.txt
done
echo "dropin done"
timeout -k 10 200 python tools/grid_time.py 100000 1000000 2>&1 | grep -v amdgpu.ids > $OUT/grid_time.txt
[ -f $CS/libpfgrad_gstamps.so ] && PFGRAD_LIB=$CS/libpfgrad_gstamps.so timeout -k 10 100 python tools/grid_phases.py 100000 1000000 2>&1 | grep -v amdgpu.ids > $OUT/grid_phases.txt
[ -f $CS/libpfgrad_stamps.so ] && PFGRAD_LIB=$CS/libpfgrad_stamps.so timeout -k 10 100 python tools/phase_profile.py svm 12288 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_svm.txt
# kernel averages of the seed-compatible giant-N call (CDF kernel + step kernel per timestep)
(cd /tmp && export TMPDIR=/tmp && GRID_TIME_DEVICE=0 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$OUT/giant_replay_trace -o trace -- python3 /root/repo/tools/grid_time.py 1000000 > /root/repo/$OUT/giant_replay_trace.out 2> /root/repo/$OUT/giant_replay_trace.err)
cat $OUT/dropin.txt
head -c 600 $OUT/bench_default.json
