#!/bin/bash
# Round-2 calibration pass (run via gpurun): what do SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES report for
# instruction streams of known length, and what do they report for the bench kernel.
set -e
OUT=/root/repo/gpurun_out/r02_calib
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CAL=/root/repo/tools/calib/valu_calib
$CAL 100000 > $OUT/calib_plain.txt 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_calib -o pmc -- $CAL 100000 > $OUT/calib_pmc.txt 2> $OUT/calib_pmc.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $OUT/pmc_bench -o pmc -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-chain > $OUT/bench_pmc.json 2> $OUT/bench_pmc.err
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM \
    --output-format csv -d $OUT/pmc_bench2 -o pmc -- python3 /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-single-chain > $OUT/bench_pmc2.json 2> $OUT/bench_pmc2.err || true
python3 /root/repo/tools/pmc_table.py $OUT/pmc_calib > $OUT/calib_table.txt
python3 /root/repo/tools/pmc_table.py $OUT/pmc_bench pf_reg_kernel > $OUT/bench_table.txt
python3 /root/repo/tools/pmc_table.py $OUT/pmc_bench2 pf_reg_kernel > $OUT/bench2_table.txt || true
cat $OUT/calib_plain.txt $OUT/calib_table.txt $OUT/bench_table.txt $OUT/bench2_table.txt
