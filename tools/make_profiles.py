"""Condense the rocprofv3 passes of tools/r04_profiles.sh (gpurun_out/<tag>_prof/<config>/) into the
tracked files bench.py and the judge read:
  profiles/<tag>_<config>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<tag>_<config>_bench.json         the bench line of the traced run
  profiles/valu_issue.json                   per-class VALU instruction counts per launch of the PF kernel
  profiles/lds_activity.json                 SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT / wait counters per launch of the PF kernel
  profiles/hbm_traffic.json                  FETCH_SIZE (x2: gfx950 reports half of wide reads) + WRITE_SIZE per launch
  profiles/profile_meta.json                 the hash of the kernel sources these counters were taken on (bench.py: roofline.stale)
usage: python tools/make_profiles.py <tag> [configs...]"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
cfgs = sys.argv[2:] or ["c1", "c2", "c3", "c4", "c5", "g1"]
PROF = os.path.join(ROOT, "profiles")


def pmc_mean(d, kernel_sub):
    acc = collections.defaultdict(list)
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if kernel_sub in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, (max(len(v) for v in acc.values()) if acc else 0)


def load(name):
    p = os.path.join(PROF, name)
    return json.load(open(p)) if os.path.exists(p) else {}


valu, traffic, ldsact = load("valu_issue.json"), load("hbm_traffic.json"), load("lds_activity.json")
valu = {k: v for k, v in valu.items() if isinstance(v, dict) and "classes" in v}        # drop round-1 records
traffic = {k: v for k, v in traffic.items() if isinstance(v, dict) and "chains" in v}
for c in cfgs:
    src = os.path.join(ROOT, "gpurun_out", tag + "_prof", c)
    line = json.loads([l for l in open(os.path.join(src, "bench_trace.json")) if l.startswith("{")][-1])
    variant, chains, dtype = line["config"]["kernel_variant"], line["config"]["chains_per_gpu"], line["dtype"]
    ksub = ("pf_big_kernel" if variant.startswith("big") else "pf_mem_kernel" if variant.startswith("mem")
            else "pfg_grid_step_dev_kernel" if variant.startswith("grid") else "pf_reg_kernel")
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(PROF, "{0}_{1}_kernel_stats.csv".format(tag, c)))
    json.dump(line, open(os.path.join(PROF, "{0}_{1}_bench.json".format(tag, c)), "w"), indent=1)
    key = "{0}_{1}_{2}".format(c, dtype, variant)
    c1, n1 = pmc_mean(os.path.join(src, "pmc_cls1"), ksub)
    c2, n2 = pmc_mean(os.path.join(src, "pmc_cls2"), ksub)
    if c1 and c2:
        cls = {k.replace("SQ_INSTS_VALU_", ""): v for k, v in c1.items()}
        cls.update({k.replace("SQ_INSTS_VALU_", ""): v for k, v in c2.items() if k.startswith("SQ_INSTS_VALU_")})
        cls["VALU"] = c2["SQ_INSTS_VALU"]
        valu[key] = {"chains": chains, "launches_averaged": n1, "classes": cls,
                     "other": {k: c2[k] for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES") if k in c2},
                     "workload": line["config"]["workload"], "source": "rocprofv3 --pmc, tools/{0}_profiles.sh, ".format(tag) + tag}
    # one record per config: a re-profiled config replaces the records of its older kernel variants
    for k in [k for k in list(valu) + list(traffic) + list(ldsact) if k.startswith(c + "_") and k != key]:
        valu.pop(k, None); traffic.pop(k, None); ldsact.pop(k, None)
    l, nl = pmc_mean(os.path.join(src, "pmc_lds"), ksub)
    if l:
        ldsact[key] = dict(l, chains=chains, launches_averaged=nl, kernel_ms_of_the_traced_run=line["roofline"]["kernel_ms"],
                           source="rocprofv3 --pmc, tools/{0}_profiles.sh".format(tag))
    f, nf = pmc_mean(os.path.join(src, "pmc_fetch"), ksub)
    w, nw = pmc_mean(os.path.join(src, "pmc_write"), ksub)
    if f and w:
        # FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of 1 KB?  rocprofv3 documents them in KB.
        fetch_b, write_b = f["FETCH_SIZE"] * 1024.0, w["WRITE_SIZE"] * 1024.0
        traffic[key] = {"chains": chains, "fetch_bytes_reported": fetch_b, "fetch_bytes_corrected_x2": 2 * fetch_b,
                        "write_bytes": write_b, "bytes_per_launch": 2 * fetch_b + write_b,
                        "note": "separate --pmc passes; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 wide reads",
                        "source": "tools/{0}_profiles.sh, ".format(tag) + tag}
    print(key, "kernel avg ns:", [r for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv"))) if ksub in r["Name"]][0]["AverageNs"],
          "bench kernel_ms:", line["roofline"]["kernel_ms"], "traffic MB:", traffic.get(key, {}).get("bytes_per_launch", 0) / 1e6)
json.dump(valu, open(os.path.join(PROF, "valu_issue.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(PROF, "hbm_traffic.json"), "w"), indent=1)
json.dump(ldsact, open(os.path.join(PROF, "lds_activity.json"), "w"), indent=1)

# The traced run computed its roofline from the counters committed BEFORE it; restate it with the counters of
# these very passes so that profiles/<tag>_<config>_bench.json agrees with valu_issue.json / hbm_traffic.json.
sys.path.insert(0, ROOT)
import bench
shas = set()
for c in cfgs:
    p = os.path.join(ROOT, "gpurun_out", tag + "_prof", c, "source_sha.txt")
    shas.add(open(p).read().strip() if os.path.exists(p) else "unrecorded")
if len(shas) != 1:
    raise SystemExit("the configs were profiled on different kernel sources: {0}".format(sorted(shas)))
json.dump({"kernel_source_sha": shas.pop(), "tag": tag, "configs": cfgs,
           "note": "sha256[:16] over csrc/*.hpp, csrc/*.hip, include/pfgrad.h and _build.py, computed on the GPU box by the profile script "
                   "(bench.kernel_source_sha); bench.py sets roofline.stale when the sources have changed since"},
          open(os.path.join(PROF, "profile_meta.json"), "w"), indent=1)
for c in cfgs:
    p = os.path.join(PROF, "{0}_{1}_bench.json".format(tag, c))
    line = json.load(open(p))
    variant, chains, dtype = line["config"]["kernel_variant"], line["config"]["chains_per_gpu"], line["dtype"]
    key = "{0}_{1}_{2}".format(c, dtype, variant)
    roof = line["roofline"]
    if key in traffic:
        roof["traffic"] = traffic[key]["bytes_per_launch"] * chains / traffic[key]["chains"]
    if roof.get("bound") in ("valu", "lds"):
        vr = bench.valu_roofline(key, chains, roof["kernel_ms"], roof["in_kernel_clock_ghz"])
        lr = bench.lds_roofline(key, chains, roof["kernel_ms"], roof["in_kernel_clock_ghz"])
        if vr:
            # bench.py's rule: the pipe with the highest USEFUL utilisation (LDS: conflict-free cycles), the busier pipe beside it
            useful = lr["frac_conflict_free"] if lr else 0.0
            lds_binds = bool(lr and useful > vr["frac"])
            roof.update(bound="lds" if lds_binds else "valu",
                        achieved=(lr["achieved"] * (1.0 - lr["bank_conflict_share"])) if lds_binds else vr["achieved"],
                        peak=(lr if lds_binds else vr)["peak"], frac=useful if lds_binds else vr["frac"], valu=vr, lds=lr,
                        lds_frac_conflict_free=useful if lr else None,
                        busiest_pipe=("lds" if (lr and lr["frac"] > vr["frac"]) else "valu"),
                        busiest_pipe_busy_frac=max(lr["frac"], vr["frac"]) if lr else vr["frac"],
                        unit="G conflict-free LDS-array cycles/s" if lds_binds else "G VALU issue-cycles/s")
    roof["stale"] = False
    roof.pop("stale_note", None)
    roof["counters"] = "restated by tools/make_profiles.py with the PMC passes of this same profile run"
    json.dump(line, open(p, "w"), indent=1)
    print(c, "roofline", roof["bound"], round(roof["frac"], 3), "valu", (round(roof["valu"]["frac"], 3) if roof.get("valu") else ""),
          "lds", (round(roof["lds"]["frac"], 3) if roof.get("lds") else ""))
