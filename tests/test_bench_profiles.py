"""The committed profile files bench.py reads are consistent with the bench lines committed beside them:
every config's roofline inputs exist under the key bench.py looks up, the VALU-issue fraction is a
fraction, and the rocprofv3 kernel average agrees with the HIP-event time of the same run."""
import csv
import json
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


@pytest.mark.parametrize("cfg", ["c1", "c2", "c3", "c4", "c5"])
def test_profiles_match_bench_line(cfg):
    line = json.load(open(os.path.join(PROF, "r02_{0}_bench.json".format(cfg))))
    variant, chains, dtype = line["config"]["kernel_variant"], line["config"]["chains_per_gpu"], line["dtype"]
    assert chains == bench.CONFIGS[cfg][5]                    # profiled at the bench's default batch size
    key = "{0}_{1}_{2}".format(cfg, dtype, variant)
    kern_ms = line["roofline"]["kernel_ms"]
    rows = list(csv.DictReader(open(os.path.join(PROF, "r02_{0}_kernel_stats.csv".format(cfg)))))
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    assert ("pf_big_kernel" if variant.startswith("big") else "pf_reg_kernel") in top["Name"]
    assert abs(float(top["AverageNs"]) * 1e-6 - kern_ms) < 0.05 * kern_ms      # rocprofv3 vs HIP events
    traffic = json.load(open(os.path.join(PROF, "hbm_traffic.json")))
    assert key in traffic and traffic[key]["chains"] == chains
    if variant.startswith("big"):
        assert line["roofline"]["bound"] == "hbm" and 0.3 < line["roofline"]["frac"] < 1.0
        alg = line["roofline"]["achieved"] * 1e9 * kern_ms * 1e-3
        assert 0.9 < traffic[key]["bytes_per_launch"] / alg < 1.3              # traffic ~ algorithmic bytes
        return
    issue = json.load(open(os.path.join(PROF, "valu_issue.json")))
    assert key in issue and issue[key]["chains"] == chains
    r = bench.valu_roofline(key, chains, kern_ms, line["roofline"]["in_kernel_clock_ghz"])
    assert r is not None and 0.25 < r["frac"] < 1.0
    assert line["roofline"]["bound"] == "valu" and abs(line["roofline"]["frac"] - r["frac"]) < 2e-3
    assert traffic[key]["bytes_per_launch"] < 0.01 * line["roofline"]["hbm_model"]["algorithmic_bytes_per_launch"]
