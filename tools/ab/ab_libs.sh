#!/bin/bash
# A/B kernel timing over every csrc/libpfgrad*.so (or the tags given), one process per library.
# usage: tools/ab_libs.sh <outfile-tag> [model] [chains] [tags...]
TAG=$1; MODEL=${2:-svm}; CH=${3:-3072}; shift 3 || true
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
OUT=/root/repo/gpurun_out/ab_${TAG}.txt
mkdir -p /root/repo/gpurun_out
: > $OUT
LIBS="$@"
if [ -z "$LIBS" ]; then LIBS=$(cd $CS && ls libpfgrad*.so | grep -v stamps | sed 's/libpfgrad_\?//; s/\.so//'); fi
for t in $LIBS; do
  if [ "$t" == "" ] || [ "$t" == "default" ]; then L=$CS/libpfgrad.so; else L=$CS/libpfgrad_$t.so; fi
  PFGRAD_LIB=$L timeout -k 10 120 python /root/repo/tools/kernel_time.py $MODEL $CH >> $OUT 2>&1 || echo "FAILED $t" >> $OUT
done
cat $OUT
