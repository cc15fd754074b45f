#!/bin/bash
# usage: tools/pmc_cfg.sh <tag> "<counters>" <config_perf.py argument> [kernel-name substring]
TAG=$1; CTRS=$2; CFG=$3; KN=${4:-pf_mem_kernel}
OUT=/root/repo/gpurun_out/pmc_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT -o pmc -- python3 /root/repo/tools/config_perf.py $CFG > $OUT/run.log 2> $OUT/err.log
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$OUT/pmc_counter_collection.csv")))
acc=collections.defaultdict(list)
for r in rows:
    if '$KN' in r['Kernel_Name']:
        acc[(r['Counter_Name'], r.get('Grid_Size','?'))].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
