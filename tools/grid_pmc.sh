#!/bin/bash
# rocprofv3 passes over the whole-GPU window (run via gpurun): kernel trace + stats, issue counters, HBM traffic.
# usage: tools/grid_pmc.sh <tag> MODEL N B [REPS]
TAG=$1; MODEL=$2; N=$3; B=$4; REPS=${5:-3}
OUT=/root/repo/gpurun_out/${TAG}
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RUN="python3 /root/repo/tools/grid_run.py $MODEL $N $B $REPS"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $RUN > $OUT/trace.out 2> $OUT/trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -o pmc -- $RUN > $OUT/pmc1.out 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc2 -o pmc -- $RUN > $OUT/pmc2.out 2> $OUT/pmc2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- $RUN > $OUT/fetch.out 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- $RUN > $OUT/write.out 2> $OUT/write.err
python3 - <<PY
import csv, collections, glob
def load(p):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % p, recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
for p in ("pmc1", "pmc2", "fetch", "write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in load(p):
        k = r["Kernel_Name"].split("<")[0].replace("void pfg::", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(p, k, {c: (sum(v) / len(v), len(v)) for c, v in d.items()})
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        print("stats", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
