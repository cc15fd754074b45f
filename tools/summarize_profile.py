"""Copy the rocprofv3 summaries judged from gpurun_out/prof_<tag>/ into profiles/ and record
the HBM traffic per particle-filter launch for bench.py's roofline.traffic.

HBM bytes (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports 1/2 of the bytes of wide coalesced streaming reads, so the read side is
doubled (an upper bound for this kernel, whose few global reads are mostly scalar loads)."""
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
key = sys.argv[2] if len(sys.argv) > 2 else "svm_f64_C512"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
for f in ("bench_trace.json",):
    shutil.copy(os.path.join(src, f), os.path.join(dst, tag + "_" + f))


def counter(path, name):
    rows = list(csv.DictReader(open(path)))
    vals = [float(r["Counter_Value"]) for r in rows
            if ("pf_reg_kernel" in r["Kernel_Name"] or "pf_mem_kernel" in r["Kernel_Name"])
            and r["Counter_Name"] == name]
    return vals


fetch = counter(os.path.join(src, "pmc_fetch", "fetch_counter_collection.csv"), "FETCH_SIZE")
write = counter(os.path.join(src, "pmc_write", "write_counter_collection.csv"), "WRITE_SIZE")
stats = list(csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))))
pf = [r for r in stats if "pf_reg_kernel" in r["Name"] or "pf_mem_kernel" in r["Name"]][0]
fk, wk = sum(fetch) / len(fetch), sum(write) / len(write)
summary = dict(tag=tag, key=key, kernel=pf["Name"], calls=int(pf["Calls"]),
               average_ns=float(pf["AverageNs"]), min_ns=float(pf["MinNs"]), max_ns=float(pf["MaxNs"]),
               FETCH_SIZE_KiB_per_launch=fk, WRITE_SIZE_KiB_per_launch=wk,
               hbm_bytes_per_launch_raw=(fk + wk) * 1024.0,
               hbm_bytes_per_launch_corrected=(2.0 * fk + wk) * 1024.0,
               correction="read side x2 (gfx950 FETCH_SIZE under-count for wide reads); upper bound here")
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
tpath = os.path.join(dst, "hbm_traffic.json")
table = json.load(open(tpath)) if os.path.exists(tpath) else {}
table[key] = dict(bytes_per_launch=summary["hbm_bytes_per_launch_corrected"], source=tag + "_pmc_summary.json")
json.dump(table, open(tpath, "w"), indent=1)
print(json.dumps(summary, indent=1))
