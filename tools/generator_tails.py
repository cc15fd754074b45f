#!/usr/bin/env python3
"""Large-sample check of the device generator's inputs to the filter (`rng_precision` of the bench line):
the Box-Muller normals (f32 transcendental units, |z| <= 6.7, 24-bit angle) and the 32-bit resampling words, as the
bench instantiation draws them.  Every chain of a ChainEnsemble launch (the trace-honouring twin of the production
kernel, bitwise the same results: tests/test_gpu_launch_path.py) records its draws into device buffers; the statistics
are reduced on the device.  K launches x C chains x T x N draws, default 2.0e9.

    python tools/generator_tails.py [--chains 256] [--launches 8] [--out profiles/r03_generator_tails.txt]

Reported: moments, two-sided tail counts against N(0,1) with their Poisson z-scores, chi-square over 512 equiprobable
bins (normals), over the top 12 and the low 8 bits of the words, the correlation of neighbouring chains' draws."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
sys.path.insert(0, ROOT)

TAILS = (3.0, 3.5, 4.0, 4.5, 5.0, 5.5, 6.0)


def collect(model="svm", T=1000, N=1000, chains=256, launches=8, seed=2025, bins=512):
    """Returns a dict of statistics accumulated over `launches` launches of `chains` chains."""
    import torch
    from scipy import stats
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd.models.svm import SVMParameters, generate_svm_data
    assert model == "svm"
    p = SVMParameters(A=np.eye(1) * .95, Q=np.eye(1) * .5, R=np.eye(1) * .5)
    np.random.seed(seed)
    y = generate_svm_data(T=T, parameters=p)["observations"]
    ens = ChainEnsemble(model, y, p, num_chains=chains, N=N, epsilon=0.1, seed=seed)
    dev = ens.device
    rec_z = torch.zeros((chains, T, N), dtype=torch.float64, device=dev)
    rec_u = torch.zeros((chains, T, N), dtype=torch.int32, device=dev)
    dummy_x = torch.zeros(((T + 1) * N,), dtype=torch.float64, device=dev)       # every chain's trace goes here: unused
    dummy_lw = torch.zeros(((T + 1) * N,), dtype=torch.float64, device=dev)
    for c in range(chains):
        ens._desc["trace_x"][c] = dummy_x.data_ptr()
        ens._desc["trace_logw"][c] = dummy_lw.data_ptr()
        ens._desc["rec_z"][c] = rec_z[c].data_ptr()
        ens._desc["rec_u"][c] = rec_u[c].data_ptr()
    ens.desc_dev.copy_(torch.from_numpy(ens._desc.view(np.uint8).reshape(ens.C, -1)))
    edges = torch.tensor(stats.norm.ppf(np.arange(1, bins) / bins), dtype=torch.float64, device=dev)
    acc = dict(n=0, s1=0.0, s2=0.0, s3=0.0, s4=0.0, zmax=0.0, tails=np.zeros(len(TAILS), dtype=np.int64),
               hist=np.zeros(bins, dtype=np.int64), top12=np.zeros(4096, dtype=np.int64), low8=np.zeros(256, dtype=np.int64),
               cross=0.0, cross_n=0, uz=0.0)
    variant = None
    for k in range(launches):
        ens.launch_pf(traced=True)
        ens.synchronize()
        variant = ens.ctx.last_variant()
        z = rec_z.reshape(-1)
        acc["n"] += z.numel()
        z2 = z * z
        acc["s1"] += float(z.sum()); acc["s2"] += float(z2.sum())
        acc["s3"] += float((z2 * z).sum()); acc["s4"] += float((z2 * z2).sum())
        a = z.abs()
        acc["zmax"] = max(acc["zmax"], float(a.max()))
        acc["tails"] += np.array([int((a > t).sum()) for t in TAILS])
        for c0 in range(0, chains, 32):                       # bucketize in slices: int64 indices are 8 B per draw
            zc = rec_z[c0:c0 + 32].reshape(-1)
            acc["hist"] += torch.bincount(torch.bucketize(zc, edges), minlength=bins).cpu().numpy()
            w = rec_u[c0:c0 + 32].reshape(-1).to(torch.int64) & 0xFFFFFFFF
            acc["top12"] += torch.bincount(w >> 20, minlength=4096).cpu().numpy()
            acc["low8"] += torch.bincount(w & 255, minlength=256).cpu().numpy()
            uc = (w.to(torch.float64) + 0.5) / 4294967296.0 - 0.5
            acc["uz"] += float((uc * zc).sum())
        acc["cross"] += float((rec_z[:-1] * rec_z[1:]).sum())
        acc["cross_n"] += rec_z[:-1].numel()
        ens.step_ctr.add_(1)                                  # the next launch draws with the next step's key
    n = acc["n"]
    out = dict(model=model, T=T, N=N, chains=chains, launches=launches, variant=variant, draws=n)
    m1, m2, m3, m4 = acc["s1"] / n, acc["s2"] / n, acc["s3"] / n, acc["s4"] / n
    out["moments"] = dict(mean=m1, var=m2 - m1 * m1, third=m3, fourth=m4,
                          z_mean=m1 * np.sqrt(n), z_var=(m2 - 1.0) / np.sqrt(2.0 / n), z_third=m3 / np.sqrt(15.0 / n),
                          z_fourth=(m4 - 3.0) / np.sqrt(96.0 / n))
    out["max_abs_z"] = acc["zmax"]
    out["tails"] = []
    for t, cnt in zip(TAILS, acc["tails"]):
        e = 2.0 * stats.norm.sf(t) * n
        out["tails"].append(dict(threshold=t, count=int(cnt), expected=e, z=(cnt - e) / np.sqrt(e)))

    def chi2(h):
        e = h.sum() / h.shape[0]
        x = float(np.sum((h - e) ** 2) / e)
        return dict(bins=int(h.shape[0]), chi2=x, dof=int(h.shape[0] - 1), p=float(stats.chi2.sf(x, h.shape[0] - 1)))
    out["chi2_normal_equiprobable"] = chi2(acc["hist"])
    out["chi2_words_top12"] = chi2(acc["top12"])
    out["chi2_words_low8"] = chi2(acc["low8"])
    out["corr_neighbouring_chains"] = dict(r=acc["cross"] / acc["cross_n"], z=acc["cross"] / np.sqrt(acc["cross_n"]))
    out["corr_word_normal"] = dict(r=acc["uz"] / n * np.sqrt(12.0), z=acc["uz"] * np.sqrt(12.0) / np.sqrt(n))
    return out


def render(o):
    lines = ["# device generator, large sample: %d launches x %d chains x T = %d x N = %d = %.3g draws of each kind,"
             % (o["launches"], o["chains"], o["T"], o["N"], o["draws"]),
             "# recorded by the trace-honouring twin of %s (tools/generator_tails.py); z = deviation in standard errors" % o["variant"],
             "normals  mean %+.3e (z %+.2f)  var %.8f (z %+.2f)  E z^3 %+.3e (z %+.2f)  E z^4 %.6f (z %+.2f)  max |z| %.4f"
             % (o["moments"]["mean"], o["moments"]["z_mean"], o["moments"]["var"], o["moments"]["z_var"],
                o["moments"]["third"], o["moments"]["z_third"], o["moments"]["fourth"], o["moments"]["z_fourth"], o["max_abs_z"])]
    for t in o["tails"]:
        lines.append("  P(|z| > %.1f): %12d observed  %14.1f expected  z %+.2f" % (t["threshold"], t["count"], t["expected"], t["z"]))
    for k in ("chi2_normal_equiprobable", "chi2_words_top12", "chi2_words_low8"):
        c = o[k]
        lines.append("%-28s %5d bins  chi2 %.1f  (dof %d)  p %.3f" % (k, c["bins"], c["chi2"], c["dof"], c["p"]))
    lines.append("corr(z of chain c, chain c+1)  r %+.3e  z %+.2f" % (o["corr_neighbouring_chains"]["r"], o["corr_neighbouring_chains"]["z"]))
    lines.append("corr(word, normal of a child)   r %+.3e  z %+.2f" % (o["corr_word_normal"]["r"], o["corr_word_normal"]["z"]))
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=256)
    ap.add_argument("--launches", type=int, default=8)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--N", type=int, default=1000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    o = collect(T=a.T, N=a.N, chains=a.chains, launches=a.launches)
    txt = render(o)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
        with open(os.path.splitext(a.out)[0] + ".json", "w") as f:
            json.dump(o, f, indent=1, default=float)
