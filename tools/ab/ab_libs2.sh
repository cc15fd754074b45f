#!/bin/bash
# A/B kernel timing of library builds: tools/ab_libs2.sh <model> <chains> <tag> [tag...]  ("default" = csrc/libpfgrad.so)
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
M=$1; C=$2; shift 2
OUT=gpurun_out/ab_libs2_$M.txt; : > $OUT
for t in "$@"; do
  if [ "$t" == "default" ]; then L=$CS/libpfgrad.so; else L=$CS/libpfgrad_$t.so; fi
  PFGRAD_LIB=$L timeout -k 10 120 python tools/kernel_time.py $M $C 2>/dev/null >> $OUT || echo "FAILED $t" >> $OUT
done
cat $OUT
