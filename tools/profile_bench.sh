#!/bin/bash
# rocprofv3 passes for bench.py on the GPU box (run via gpurun).  Kernel trace + stats first,
# then the HBM counters in their own passes (MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE do
# not fit in one pass).  Outputs under gpurun_out/prof_<tag>/ ; copy summaries to profiles/.
set -e
TAG=${1:-r01}
shift || true
ARGS="${@:---steps 5 --warmup 1 --no-cpu-baseline --no-single-chain}"
OUT=/root/repo/gpurun_out/prof_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 /root/repo/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 /root/repo/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o write -- python3 /root/repo/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
