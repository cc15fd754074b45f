#!/bin/bash
# Round 4: where the LDS bank-conflict cycles of the headline kernel come from.  Three builds of libpfgrad --
# production, -DPFG_EXP_OWNGATHER (real search, conflict-free gathers), -DPFG_EXP_EVENWORDS (conflict-free search AND
# gathers; round 3) -- one PMC pass each over bench.py's c2 launch (12288 chains) + HIP-event kernel times.
# usage (gpurun): tools/lds_split.sh   -> gpurun_out/r04_lds_split/summary.txt
OUT=/root/repo/gpurun_out/r04_lds_split
rm -rf $OUT; mkdir -p $OUT
C=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
cd /tmp && export TMPDIR=/tmp
export PFG_BENCH_KNOCKOUT=1
ARGS="--config c2 --steps 4 --warmup 1 --no-cpu-baseline --no-single-chain"
for tag in prod owngather evenwords; do
  LIB=$C/libpfgrad_$tag.so; [ $tag = prod ] && LIB=$C/libpfgrad.so
  export PFGRAD_LIB=$LIB
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/$tag -o pmc -- python3 /root/repo/bench.py $ARGS > $OUT/$tag.json 2> $OUT/$tag.err
done
unset PFGRAD_LIB
python3 - <<PY > $OUT/summary.txt
import csv, glob, json, collections
out = {}
for tag in ("prod", "owngather", "evenwords"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "pf_reg_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    line = json.loads(open("$OUT/%s.json" % tag).read().strip().splitlines()[-1])
    out[tag] = dict(kernel_ms=line["roofline"]["kernel_ms"], **{k: sum(v) / len(v) for k, v in acc.items()})
print(json.dumps(out, indent=1))
p, o, e = out["prod"], out["owngather"], out["evenwords"]
print("bank-conflict cycles per launch: production %.3e | real search + own-slot gathers %.3e | even words (search and gathers conflict-free) %.3e" % (p["SQ_LDS_BANK_CONFLICT"], o["SQ_LDS_BANK_CONFLICT"], e["SQ_LDS_BANK_CONFLICT"]))
print("=> search share %.0f %%, gather share %.0f %% of the production conflict cycles (rest: tables, scans)" % (100 * (o["SQ_LDS_BANK_CONFLICT"] - e["SQ_LDS_BANK_CONFLICT"]) / p["SQ_LDS_BANK_CONFLICT"], 100 * (p["SQ_LDS_BANK_CONFLICT"] - o["SQ_LDS_BANK_CONFLICT"]) / p["SQ_LDS_BANK_CONFLICT"]))
print("kernel ms: production %.2f, own-slot gathers %.2f, even words %.2f" % (p["kernel_ms"], o["kernel_ms"], e["kernel_ms"]))
PY
cat $OUT/summary.txt
