#!/bin/bash
# usage: tools/pmc_probe.sh <tag> "<counters>" [bench args]
TAG=$1; CTRS=$2; shift 2
ARGS="${@:---steps 3 --warmup 1 --no-cpu-baseline --no-single-chain}"
OUT=/root/repo/gpurun_out/pmc_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 /root/repo/bench.py $ARGS > $OUT/bench.json 2> $OUT/err.log
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$OUT/pmc_counter_collection.csv")))
acc=collections.defaultdict(list)
for r in rows:
    if 'pf_reg_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
