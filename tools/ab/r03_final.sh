#!/bin/bash
# Round-3 final evidence run (via gpurun): calibration, phase stamps, default bench line + the other configs, drop-in
# timings, PCIe-inclusive rate.  Outputs under gpurun_out/r03_final/.
cd /root/repo
OUT=gpurun_out/r03_final
rm -rf $OUT; mkdir -p $OUT
tools/calib/valu_calib 100000 > $OUT/valu_calibration.txt 2>&1
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
PFGRAD_LIB=$CS/libpfgrad_stamps.so python tools/phase_profile.py svm 12288 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_svm.txt
PFGRAD_LIB=$CS/libpfgrad_stamps.so python tools/phase_profile.py svm 256 2>&1 | grep -v amdgpu.ids > $OUT/phase_stamps_svm_lone_workgroup.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
for c in c1 c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 --cpu-budget 5 > $OUT/bench_$c.json 2> $OUT/bench_$c.err || echo "bench $c failed"
done
python tools/config_perf.py dropin 2>&1 | grep -v amdgpu.ids > $OUT/dropin.txt
python tools/pcie_rate.py 2>&1 | grep -v amdgpu.ids > $OUT/pcie_inclusive.txt
cat $OUT/phase_stamps_svm.txt $OUT/dropin.txt $OUT/pcie_inclusive.txt
head -c 1500 $OUT/bench_default.json
