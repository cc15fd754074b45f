#!/bin/bash
# rocprofv3 kernel trace of one bench.py invocation with extra args; summary -> profiles/<tag>_kernel_stats.csv
set -e
TAG=$1; shift
OUT=/root/repo/gpurun_out/prof_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 /root/repo/bench.py --no-cpu-baseline --no-single-chain "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
head -3 $OUT/trace/trace_kernel_stats.csv | cut -c1-220
cat $OUT/bench_trace.json | cut -c1-300
