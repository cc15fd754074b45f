#!/bin/bash
# Round-2 profile passes for every BASELINE config (run via gpurun): kernel trace + stats, per-class VALU
# instruction counters (two passes), FETCH_SIZE and WRITE_SIZE (separate passes, as the guide prescribes).
# Outputs under gpurun_out/r02_prof/<config>/ ; tools/make_profiles.py condenses them into profiles/.
CFGS="${@:-c2 c1 c3 c4 c5}"
cd /tmp && export TMPDIR=/tmp
for c in $CFGS; do
  OUT=/root/repo/gpurun_out/r02_prof/$c
  mkdir -p $OUT
  ARGS="--config $c --steps 5 --warmup 1 --no-cpu-baseline --no-single-chain"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 /root/repo/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
      --output-format csv -d $OUT/pmc_cls1 -o pmc -- python3 /root/repo/bench.py $ARGS > $OUT/bench_cls1.json 2> $OUT/cls1.err
  rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES \
      --output-format csv -d $OUT/pmc_cls2 -o pmc -- python3 /root/repo/bench.py $ARGS > $OUT/bench_cls2.json 2> $OUT/cls2.err
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 /root/repo/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 /root/repo/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
  echo "$c done: $(head -c 300 $OUT/bench_trace.json)"
done
