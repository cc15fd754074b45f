#!/bin/bash
# Round-2 GPU pass 2: deterministic replay tests of the device-generator kernels, per-phase stamps,
# calibration with in-kernel clock, per-class VALU instruction counts of the bench kernel.
OUT=/root/repo/gpurun_out/r02_step2
mkdir -p $OUT
cd /root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_device_replay.py -x -q > $OUT/pytest_replay.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_replay.log
tail -15 $OUT/pytest_replay.log
CS=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
PFGRAD_LIB=$CS/libpfgrad_stamps.so timeout -k 10 300 python tools/phase_profile.py svm 3072 > $OUT/phase_svm.txt 2>&1
cat $OUT/phase_svm.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 /root/repo/tools/calib/valu_calib 100000 > $OUT/calib_clock.txt 2>&1
cat $OUT/calib_clock.txt
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
    --output-format csv -d $OUT/pmc_cls1 -o pmc -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-chain > $OUT/bench_cls1.json 2> $OUT/bench_cls1.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH \
    --output-format csv -d $OUT/pmc_cls2 -o pmc -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-chain > $OUT/bench_cls2.json 2> $OUT/bench_cls2.err
python3 /root/repo/tools/pmc_table.py $OUT/pmc_cls1 pf_reg_kernel > $OUT/cls1_table.txt
python3 /root/repo/tools/pmc_table.py $OUT/pmc_cls2 pf_reg_kernel > $OUT/cls2_table.txt
cat $OUT/cls1_table.txt $OUT/cls2_table.txt | cut -c1-300
