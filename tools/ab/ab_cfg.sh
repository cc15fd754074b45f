#!/bin/bash
# A/B of library builds on one bench config: tools/ab_cfg.sh <config> <tag> [tag...]   ("default" = csrc/libpfgrad.so)
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
CFG=$1; shift
OUT=gpurun_out/ab_cfg_$CFG.txt; : > $OUT
for t in "$@"; do
  if [ "$t" == "default" ]; then L=$CS/libpfgrad.so; else L=$CS/libpfgrad_$t.so; fi
  echo "== $t" >> $OUT
  PFGRAD_LIB=$L timeout -k 10 200 python bench.py --config $CFG --steps 10 --warmup 3 --no-cpu-baseline --no-single-chain 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))" >> $OUT || echo FAILED >> $OUT
done
cat $OUT
