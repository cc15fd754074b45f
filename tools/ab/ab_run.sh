#!/bin/bash
# A/B kernel timing: one process per (library tag, kernel variant) pair on one BASELINE config.
# usage: tools/ab_run.sh <outfile-tag> <config> <chains|0> <lib[:variant]> ...     (lib "default" = csrc/libpfgrad.so)
TAG=$1; CFG=$2; CH=$3; shift 3
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
OUT=/root/repo/gpurun_out/ab_${TAG}.txt
mkdir -p /root/repo/gpurun_out
for spec in "$@"; do
  t=${spec%%:*}; v=""; [[ "$spec" == *:* ]] && v=${spec#*:}
  if [ "$t" == "default" ]; then L=$CS/libpfgrad.so; else L=$CS/libpfgrad_$t.so; fi
  PFGRAD_LIB=$L PFGRAD_VARIANT=$v timeout -k 10 150 python /root/repo/tools/cfg_time.py $CFG $CH >> $OUT 2>&1 || echo "FAILED $spec" >> $OUT
done
cat $OUT
