#!/usr/bin/env python
"""ISA histogram of the T-loop of one pf_reg_kernel instantiation (no GPU needed).

Compiles an instantiation unit with the flags _build.py uses plus -S -DPFG_ISA_MARKERS (the kernel then carries its
phase boundaries "; PFG_PHASE i" and its rarely executed regions "; PFG_MARK cold ..." / "w<weight> ..." as comments in
the listing), cuts the loop into basic blocks, weights every block with how often a wave executes it per timestep
(1, or 0 for a `cold` region, or the stated fraction), and counts instructions by class and phase.

A cold region = the blocks between the conditional branch that skips the region and that branch's target label.
Output: per phase and in total, instructions per lane-timestep by class; VALU per particle-step = VALU / PPT.

    python tools/isa_histogram.py [--unit svm_prior_device] [--kernel 'pf_reg_kernel<0, 0, double, 256, 4, 1, false, 0>']
                                  [--flags=-DPFG_EXP_NOTRACE=1 ...] [--write profiles/r03_isa_histogram_c2.txt]
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
from sgmcmc_ssm_amd import _build  # noqa: E402

# PFG_PH(i) closes phase i (the stamps build adds the cycles since the previous marker to slot i): instructions that
# follow marker i belong to phase i + 1, those behind marker 9 and before marker 0 to phase 0 of the next timestep
PHASES = {0: "A  block max: cvt, thread max, v_max_f32 DPP", 1: "-  barrier 1 (wait + s_barrier)",
          2: "BC max exchange, exp(lw-m), thread sums, wave scan", 3: "-  barrier 2",
          4: "D  wave offsets, 1/W, log-lik park, y_t / weight, CDF -> LDS", 5: "-  barrier 3",
          6: "E  generator words + binary search", 7: "F  gather parents", 8: "-  barrier 4",
          9: "GH normals, propose, weight, score, publish"}

F64_ARITH = ("v_add_f64", "v_mul_f64", "v_fma_f64", "v_fmac_f64")


def classify(m):
    """instruction mnemonic -> (unit, class)."""
    if m.startswith("v_"):
        if m.endswith("_dpp") or "_dpp" in m:
            return "VALU", "dpp (cross-lane)"
        if m in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
            return "VALU", "lane <-> scalar (readlane / readfirstlane / writelane)"
        base = m.replace("_e32", "").replace("_e64", "")
        if base in F64_ARITH:
            return "VALU", "fp64 add / mul / fma (counted by SQ_INSTS_VALU_{ADD,MUL,FMA}_F64)"
        if base in ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64"):
            return "VALU", "fp64 transcendental"
        if "f64" in base:
            return "VALU", "fp64 other (cvt, rndne, ldexp, max/min, cmp, cndmask pairs)"
        if base in ("v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_sqrt_f32", "v_rcp_f32", "v_rsq_f32"):
            return "VALU", "f32 transcendental"
        if "f32" in base or "f16" in base:
            return "VALU", "f32 arithmetic / convert"
        if base.startswith("v_cmp") or base.startswith("v_cndmask"):
            return "VALU", "integer compare / select"
        if base.startswith("v_mov") or base.startswith("v_accvgpr"):
            return "VALU", "moves"
        return "VALU", "integer / logic (add, xor, rotate, shift, and/or, mad)"
    if m.startswith("ds_"):
        return "LDS", "ds_" + ("read" if "read" in m else "write" if "write" in m else "other")
    if m.startswith(("flat_", "global_", "scratch_", "buffer_")):
        return "VMEM", "scratch (spill)" if m.startswith("scratch_") else "global / flat"
    if m == "s_barrier":
        return "SALU", "s_barrier"
    if m == "s_waitcnt":
        return "SALU", "s_waitcnt"
    if m == "s_nop":
        return "SALU", "s_nop"
    if m.startswith("s_load") or m.startswith("s_buffer_load"):
        return "SMEM", "s_load"
    if m.startswith(("s_cbranch", "s_branch")):
        return "SALU", "branch"
    return "SALU", "scalar ALU"


def listing(unit, flags):
    src = os.path.join(_build.CSRC, "pfg_inst_{0}.hip".format(unit))
    cmd = [_build._hipcc()] + _build.FLAGS + _build._contract(src) + ["-DPFG_ISA_MARKERS"] + flags + [
        "-I", _build.INCLUDE, "-I", _build.CSRC, "--cuda-device-only", "-S", src, "-o", "-"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if res.returncode != 0:
        raise SystemExit(res.stderr)
    return res.stdout.split("\n")


def loop_blocks(lines, kernel):
    """Basic blocks of the (outermost, longest) loop of `kernel`: list of dict(label, insts, marks, phase markers)."""
    names = {}
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            names[m.group(1)] = None
    dem = subprocess.run(["c++filt"] + list(names), stdout=subprocess.PIPE, text=True).stdout.splitlines()
    want = None
    for mangled, d in zip(names, dem):
        if d.replace("void pfg::", "").startswith(kernel):
            want = mangled
    if want is None:
        raise SystemExit("kernel not found: " + kernel)
    start = next(i for i, l in enumerate(lines) if l.startswith(want + ":"))
    end = next(i for i in range(start, len(lines)) if ".Lfunc_end" in lines[i])
    blocks, cur = [], None
    for l in lines[start:end]:
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):(.*)$", s) or re.match(r"^; %bb\.(\d+):(.*)$", s)
        if m:
            cur = dict(label=m.group(1), is_label=s.startswith(".LBB"), in_loop="Loop" in m.group(2), items=[])
            blocks.append(cur)
            continue
        if cur is None or not s:
            continue
        if s.startswith("; PFG_PHASE") or s.startswith("; PFG_MARK"):
            cur["items"].append(("marker", s[2:]))
        elif s.startswith(";") or s.startswith("."):
            continue
        else:
            cur["items"].append(("inst", s))
    return [b for b in blocks if b["in_loop"]]


def weights(blocks):
    """Execution weight of every loop block per timestep (see the module docstring)."""
    w = [1.0] * len(blocks)
    why = [""] * len(blocks)
    index = {b["label"]: i for i, b in enumerate(blocks) if b["is_label"]}
    for i, b in enumerate(blocks):
        for kind, text in b["items"]:
            if kind != "marker" or not text.startswith("PFG_MARK"):
                continue
            tag = text[len("PFG_MARK "):]
            weight = 0.0 if tag.startswith("cold") else float(tag.split()[0][1:])
            # the conditional branch that skips this region: the nearest one before the marker whose target lies after it
            lo, hi = None, None
            for j in range(i, -1, -1):
                insts = [t for k, t in blocks[j]["items"] if k == "inst"]
                if j == i:      # only branches before the marker inside its own block
                    pos = [n for n, (k, t) in enumerate(b["items"]) if k == "marker" and t == text][0]
                    insts = [t for k, t in b["items"][:pos] if k == "inst"]
                br = [t for t in insts if t.startswith("s_cbranch")]
                tgt = [index.get(t.split()[-1]) for t in br]
                tgt = [t for t in tgt if t is not None and t > i]
                if tgt:
                    lo, hi = j + 1, min(tgt)
                    break
            if lo is None:
                continue
            for j in range(max(lo, i if lo <= i else lo), hi):
                if weight < w[j] or w[j] == 1.0:
                    w[j] = min(w[j], weight) if w[j] != 1.0 else weight
                    why[j] = tag
            if lo <= i:          # the marker's own block starts the region only if the branch ended the previous block
                pass
    return w, why


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="svm_prior_device")
    ap.add_argument("--kernel", default="pf_reg_kernel<0, 0, double, 256, 4, 1, false, 0, false, false>")
    ap.add_argument("--ppt", type=int, default=4)
    ap.add_argument("--flags", default="", help="extra compiler flags, space separated in ONE argument")
    ap.add_argument("--write", default=None)
    ap.add_argument("--blocks", action="store_true", help="also list the blocks with their weights")
    ap.add_argument("--pmc", default=None, help="key,waves_per_workgroup,timesteps+1: print the measured SQ_INSTS_VALU per "
                    "wave-timestep of profiles/valu_issue.json beside the static count (cross-check)")
    args = ap.parse_args()
    flags = args.flags.split()
    blocks = loop_blocks(listing(args.unit, flags), args.kernel)
    w, why = weights(blocks)
    per_phase = collections.defaultdict(lambda: collections.Counter())
    mnem = collections.defaultdict(lambda: collections.Counter())
    phase = 0
    for b, wt in zip(blocks, w):
        for kind, text in b["items"]:
            if kind == "marker":
                if text.startswith("PFG_PHASE"):
                    phase = (int(text.split()[1]) + 1) % 10
                continue
            m = text.split()[0]
            if m.startswith(";;"):
                continue
            unit, cls = classify(m)
            per_phase[phase][(unit, cls)] += wt
            mnem[(unit, cls)][m.replace("_e32", "").replace("_e64", "")] += wt
    out = []
    out.append("# ISA histogram of the T-loop of {0} ({1}.hip{2})".format(args.kernel, "pfg_inst_" + args.unit,
               ", extra flags: " + " ".join(flags) if flags else ""))
    out.append("# instructions one wave executes per timestep (= per lane-timestep; a lane carries {0} particles), blocks weighted by".format(args.ppt))
    out.append("# how often they run: 0 for regions the bench workload never enters (filter / lambda != 1 / tracing / final-step sums), the")
    out.append("# stated fraction for wave-0-only work.  Phases = the kernel's PFG_PH markers; a marker pins no instruction, so the")
    out.append("# scheduler moves some work across a boundary: read the per-phase split as +-10 instructions.")
    tot = collections.Counter()
    for ph in sorted(per_phase):
        c = per_phase[ph]
        v = sum(n for (u, _), n in c.items() if u == "VALU")
        out.append("")
        out.append("phase {0:>2}  {1:<48} VALU {2:7.1f}  LDS {3:5.1f}  other {4:6.1f}".format(
            ph, PHASES.get(ph, ""), v, sum(n for (u, _), n in c.items() if u == "LDS"),
            sum(n for (u, _), n in c.items() if u not in ("VALU", "LDS"))))
        for (u, cls), n in sorted(c.items(), key=lambda kv: (kv[0][0] != "VALU", -kv[1])):
            if n > 0:
                out.append("    {0:<5} {1:<72} {2:7.1f}".format(u, cls, n))
            tot[(u, cls)] += n
    valu = sum(n for (u, _), n in tot.items() if u == "VALU")
    out.append("")
    out.append("TOTAL per lane-timestep: VALU {0:.1f} (= {1:.1f} per particle-step), LDS {2:.1f}, all {3:.1f}".format(
        valu, valu / args.ppt, sum(n for (u, _), n in tot.items() if u == "LDS"), sum(tot.values())))
    for (u, cls), n in sorted(tot.items(), key=lambda kv: (kv[0][0] != "VALU", -kv[1])):
        if n <= 0:
            continue
        top = ", ".join("{0} {1:.0f}".format(k, v) for k, v in mnem[(u, cls)].most_common(6) if v > 0)
        out.append("    {0:<5} {1:<72} {2:7.1f}  {3:5.1f} %{4}   [{5}]".format(u, cls, n, 100.0 * n / valu if u == "VALU" else 0.0,
                   " of VALU" if u == "VALU" else "        ", top))
    if args.pmc:
        import json
        key, nw, steps = args.pmc.split(",")
        rec = json.load(open(os.path.join(ROOT, "profiles", "valu_issue.json")))[key]
        meas = rec["classes"]["VALU"] / (rec["chains"] * int(nw) * int(steps))
        f64 = (rec["classes"]["ADD_F64"] + rec["classes"]["MUL_F64"] + rec["classes"]["FMA_F64"]) / (rec["chains"] * int(nw) * int(steps))
        out.append("")
        out.append("cross-check, rocprofv3 PMC of the production kernel (profiles/valu_issue.json:{0}): SQ_INSTS_VALU {1:.1f} per wave-timestep, "
                   "fp64 add/mul/fma {2:.1f} -- static count above: {3:.1f} / {4:.1f}".format(
                       key, meas, f64, valu, sum(n for (u, c), n in tot.items() if c.startswith("fp64 add"))))
    if args.blocks:
        out.append("")
        out.append("blocks (label, instructions, weight, why):")
        for b, wt, y in zip(blocks, w, why):
            out.append("    {0:<12} {1:5d}  {2:4.2f}  {3}".format(b["label"], sum(1 for k, _ in b["items"] if k == "inst"), wt, y))
    text = "\n".join(out) + "\n"
    sys.stdout.write(text)
    if args.write:
        with open(os.path.join(ROOT, args.write), "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
