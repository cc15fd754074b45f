"""The committed profile files bench.py reads are consistent with the bench lines committed beside them:
every config's roofline inputs exist under the key bench.py looks up, the VALU-issue and LDS-array fractions
are fractions, the rocprofv3 kernel average agrees with the HIP-event time of the same run -- and the counters
were taken on the kernel sources that are committed (a kernel edit without a re-profile fails here)."""
import csv
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


@pytest.mark.parametrize("cfg", ["c1", "c2", "c3", "c4", "c5", "g1"])
def test_profiles_match_bench_line(cfg):
    line = json.load(open(os.path.join(PROF, "r04_{0}_bench.json".format(cfg))))
    variant, chains, dtype = line["config"]["kernel_variant"], line["config"]["chains_per_gpu"], line["dtype"]
    assert chains == bench.CONFIGS[cfg][5]                    # profiled at the bench's default batch size
    key = "{0}_{1}_{2}".format(cfg, dtype, variant)
    kern_ms = line["roofline"]["kernel_ms"]
    rows = list(csv.DictReader(open(os.path.join(PROF, "r04_{0}_kernel_stats.csv".format(cfg)))))
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    assert ("pf_big_kernel" if variant.startswith("big") else "pfg_grid_step_dev_kernel" if variant.startswith("grid") else "pf_reg_kernel") in top["Name"]
    assert abs(float(top["AverageNs"]) * 1e-6 - kern_ms) < 0.05 * kern_ms      # rocprofv3 vs HIP events
    traffic = json.load(open(os.path.join(PROF, "hbm_traffic.json")))
    assert key in traffic and traffic[key]["chains"] == chains
    if variant.startswith(("big", "grid")):
        assert line["roofline"]["bound"] == "hbm" and 0.3 < line["roofline"]["frac"] < 1.0
        alg = line["roofline"]["achieved"] * 1e9 * kern_ms * 1e-3
        # traffic ~ algorithmic bytes (the whole-GPU window reads a second parent tile's scan per child tile and its tile
        # partials: up to 1.5 x)
        assert 0.9 < traffic[key]["bytes_per_launch"] / alg < (1.5 if variant.startswith("grid") else 1.3)
        return
    issue = json.load(open(os.path.join(PROF, "valu_issue.json")))
    assert key in issue and issue[key]["chains"] == chains
    lds = json.load(open(os.path.join(PROF, "lds_activity.json")))
    assert key in lds and lds[key]["chains"] == chains
    clock = line["roofline"]["in_kernel_clock_ghz"]
    r = bench.valu_roofline(key, chains, kern_ms, clock)
    l = bench.lds_roofline(key, chains, kern_ms, clock)
    assert r is not None and 0.25 < r["frac"] < 1.0 and r["frac_vs_measured_streams"] > r["frac"]
    assert l is not None and 0.25 < l["frac"] < 1.0 and 0.0 < l["bank_conflict_share"] < 0.7
    # `bound` = the pipe with the highest USEFUL utilisation (LDS: conflict-free cycles), the busier pipe beside it
    useful_lds = l["frac_conflict_free"]
    name, frac = ("lds", useful_lds) if useful_lds > r["frac"] else ("valu", r["frac"])
    assert line["roofline"]["bound"] == name and abs(line["roofline"]["frac"] - frac) < 2e-3
    assert line["roofline"]["busiest_pipe"] == ("lds" if l["frac"] > r["frac"] else "valu")
    assert abs(line["roofline"]["busiest_pipe_busy_frac"] - max(l["frac"], r["frac"])) < 2e-3
    assert traffic[key]["bytes_per_launch"] < 0.01 * line["roofline"]["hbm_model"]["algorithmic_bytes_per_launch"]


def test_counters_belong_to_the_committed_kernel_sources():
    """profiles/profile_meta.json carries the hash of the kernel sources the PMC passes ran on; bench.py flags a
    roofline from other sources as stale.  Committed state: not stale."""
    meta = json.load(open(os.path.join(PROF, "profile_meta.json")))
    assert meta["kernel_source_sha"] == bench.kernel_source_sha(), \
        "kernel sources changed since the last profile run: re-run tools/r04_profiles.sh (gpurun) + tools/make_profiles.py r04"
    assert not bench.profile_is_stale()
