#!/usr/bin/env python
"""Register / LDS / occupancy record of every particle-filter kernel instantiation (no GPU needed).

Compiles each instantiation unit with the flags _build.py uses plus -Rpass-analysis=kernel-resource-usage
and prints one line per kernel: VGPRs, AGPRs, SGPRs, spilled VGPRs / SGPRs, scratch bytes per lane, the
occupancy the register allocation allows.  `--write` stores the table as profiles/r04_resource_usage.txt
(LDS is dynamic: the column is what the host asks for at launch, from pfg_variant LDS formulas).

    python tools/resource_usage.py [--write] [--units svm_prior_device ...]
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
from sgmcmc_ssm_amd import _build  # noqa: E402

CXXFILT = "c++filt"
# the instantiations bench.py times (config -> demangled template arguments)
BENCH = {
    "pf_reg_kernel<2, 1, double, 64, 2, 1, false, 0, false, true>": "c1 wg64x2s_score1 (LGSSM optimal)",
    "pf_reg_kernel<0, 0, double, 256, 4, 1, false, 0, false, false>": "c2 wg256x4s (SVM, headline)",
    "pf_reg_kernel<1, 1, double, 512, 2, 1, false, 0, false, false>": "c3 wg512x2s (GARCH optimal)",
    "pf_reg_kernel<0, 0, double, 1024, 4, 1, false, 0, false, true>": "c4 wg1024x4s_score1 (SVM N=4000)",
    "pf_big_kernel<0, 0, double, 16384>": "c5 big16384 (SVM N=10000)",
    "pfg_grid_step_dev_kernel<0, 0, double, 256, 8, 2>": "g1 grid2048 (SVM N=10^6)",
}


def unit_records(unit):
    src = os.path.join(_build.CSRC, "pfg_inst_{0}.hip".format(unit))
    cmd = [_build._hipcc()] + _build.FLAGS + _build._contract(src) + [
        "-I", _build.INCLUDE, "-I", _build.CSRC, "--cuda-device-only", "-c", src, "-o", os.devnull,
        "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise SystemExit(res.stdout)
    recs, cur = [], None
    for line in res.stdout.splitlines():
        m = re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        key, _, val = m.group(1).partition(": ")
        if key == "Function Name":
            cur = {"mangled": val}
            recs.append(cur)
        elif cur is not None:
            cur[key.strip()] = val.strip()
    names = subprocess.run([CXXFILT] + [r["mangled"] for r in recs], stdout=subprocess.PIPE, text=True).stdout.splitlines()
    for r, n in zip(recs, names):
        r["name"] = n.replace("void pfg::", "").replace("(pfg_dev_problem const*, int)", "").replace("(pfg_dev_problem const*)", "")
        r["unit"] = unit
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--units", nargs="*", default=["{0}_{1}".format(u, r) for u in _build._UNITS for r in ("device", "replay")])
    args = ap.parse_args()
    lines = ["# kernel-resource-usage of every particle-filter instantiation, device-generator and REPLAY units (hipcc -Rpass-analysis=kernel-resource-usage,",
             "# flags of sgmcmc_ssm_amd/_build.py); '*' = an instantiation bench.py times",
             "# {0:<58} {1:>5} {2:>5} {3:>5} {4:>7} {5:>7} {6:>8} {7:>4}  {8}".format(
                 "kernel", "VGPR", "AGPR", "SGPR", "spillV", "spillS", "scratchB", "occ", "unit / bench config")]
    for unit in args.units:
        for r in unit_records(unit):
            tag = BENCH.get(r["name"])
            lines.append("{0} {1:<58} {2:>5} {3:>5} {4:>5} {5:>7} {6:>7} {7:>8} {8:>4}  {9}{10}".format(
                "*" if tag else " ", r["name"], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
                r.get("VGPRs Spill", "?"), r.get("SGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"),
                r.get("Occupancy [waves/SIMD]", "?"), unit, (" -- " + tag) if tag else ""))
    out = "\n".join(lines) + "\n"
    sys.stdout.write(out)
    if args.write:
        with open(os.path.join(ROOT, "profiles", "r04_resource_usage.txt"), "w") as f:
            f.write(out)


if __name__ == "__main__":
    main()
