#!/bin/bash
# A/B of csrc/libpfgrad_old.so against the current build on the bench workload, then the recorded-draw parity tests
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
OUT=gpurun_out/ab_quick.txt; : > $OUT
for L in libpfgrad_old.so libpfgrad.so; do
  for m in svm garch; do
    PFGRAD_LIB=$CS/$L timeout -k 10 120 python tools/kernel_time.py $m 3072 >> $OUT 2>&1 || echo FAILED $L >> $OUT
  done
done
grep -v amdgpu.ids $OUT
