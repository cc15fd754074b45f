"""GPU tests of the WHOLE-GPU window (csrc/pfg_grid_kernel.hpp, pfg_grid_cdf.hpp): N above the one-workgroup kernels'
16384, up to the N = 10^6 the reference's bias experiments call pf_gradient_estimate with
(gradient_error_fig_scripts/svm_grad_compare.py:68-82).

What is asserted.
* REPLAY (the reference's np.random stream): the kernel builds the reference's resampling CDF -- NumPy's chunked pairwise
  np.sum, its sequential cumsum -- BIT FOR BIT, so the ancestors are the reference's exactly (compared as integers against
  the oracle at every step), and trajectories, statistics, log-likelihood and gradient agree to rtol 1e-9 (what is left:
  exp / log rounding).  A parallel tree scan could not do that at these sizes; see pfg_grid_cdf.hpp.
* every reference fixture of the one-workgroup kernels again, forced through the grid path (PFGRAD_VARIANT=grid).
* DEVICE generator (sorted uniforms): the launch records its draws and the oracle replays that very launch
  (tile-wise CDF as the kernel lays it out): zero ancestor flips, rtol 1e-8; order-statistic properties of the uniforms.
* size-independent properties at N = 10^6: resampling counts against N softmax(logw), Kalman exact gradient (LGSSM).
"""
import os
import sys

import numpy as np
import pytest

from oracle import pf_oracle as po

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "helpers"))
from grid_layout import grid_layout  # noqa: E402

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-9, 1e-9


@pytest.fixture(scope="module")
def ctx():
    from sgmcmc_ssm_amd import _capi
    return _capi.default_context(0)


# ---------------------------------------------------------------------------------------------------------------------
# the reference fixtures through the grid path
# ---------------------------------------------------------------------------------------------------------------------
def test_reference_fixtures_through_the_grid_path(ctx, golden_trace, golden_window, monkeypatch):
    import test_gpu_pf_parity as tp
    monkeypatch.setenv("PFGRAD_VARIANT", "grid")
    tp.test_trace_cases_f64(ctx, golden_trace)
    assert ctx.last_variant().startswith("grid"), ctx.last_variant()
    tp.test_window_cases_f64(ctx, golden_window)
    assert ctx.last_variant().startswith("grid"), ctx.last_variant()


def test_theta_grid_through_the_grid_path(ctx, monkeypatch):
    import test_gpu_pf_parity as tp
    if not hasattr(tp, "test_theta_grid_replay_f64"):
        pytest.skip("no theta-grid test to reuse")
    monkeypatch.setenv("PFGRAD_VARIANT", "grid")
    tp.test_theta_grid_replay_f64(ctx)
    assert ctx.last_variant().startswith("grid"), ctx.last_variant()


# ---------------------------------------------------------------------------------------------------------------------
# REPLAY at giant N against the oracle: ancestors exact at every step
# ---------------------------------------------------------------------------------------------------------------------
GIANT = [
    # model, kernel, theta, N, T, pf, lambduh
    ("svm", "prior", [0.95, 1.4, 1.4], 20000, 10, "poyiadjis_N", 1.0),
    ("garch", "optimal", [0.0, 2.0, 2.0, 1.8], 70001, 6, "nemeth", 0.9),
    ("lgssm", "optimal", [0.9, 1.0, 1.2, 1.0], 131072, 5, "nemeth", 0.95),
    ("garch", "prior", [0.0, 2.0, 2.0, 1.8], 300000, 3, "poyiadjis_N", 1.0),
    ("lgssm", "prior", [0.9, 0.7, 1.2, 1.0], 262145, 3, "filter", 1.0),
    ("svm", "prior", [0.9, 1.2, 1.1], 1100003, 3, "poyiadjis_N", 1.0),            # 2048-particle tiles (N > 2^19)
    ("svm", "prior", [0.9, 1.2, 1.1], 50000, 6, "filter", 1.0),
    ("garch", "prior", [0.0, 2.0, 2.0, 1.8], 4194304, 2, "poyiadjis_N", 1.0),     # the maximum: 2048 tiles, 1024 CDF blocks, 512 np.sum chunks
]


@pytest.mark.parametrize("case", GIANT, ids=lambda c: "{0}-{1}-N{3}-{5}".format(*c))
def test_giant_replay_ancestors_are_the_references(ctx, case):
    model, kernel, theta, N, T, pf, lam = case
    rs = np.random.RandomState(N % 9973)
    y = rs.normal(size=T)
    t1, tL = (1, T - 1) if T > 3 else (0, T)
    w = rs.uniform(1.0, 40.0, size=tL - t1)
    z0, u, z = po.draw_streams(rs, N, T)
    smoother = "filter" if pf == "filter" else "nemeth"
    q = dict(model=model, kernel=kernel, smoother=smoother, stat="score", dtype="f64", rng="replay", N=N,
             t1=t1, tL=tL, lambduh=lam, prior_mean=0.0, prior_var=1.5, y=y, weights=w, theta=theta, z0=z0, u=u, z=z)
    o = ctx.run_batch([q], want_final=True, want_trace=True)[0]
    assert ctx.last_variant() == ("grid2048" if N > (1 << 19) else "grid1024")
    r = po.pf_window(model, theta, y, N, z0, u, z, kernel=kernel, pf=pf, lambduh=lam, t1=t1, tL=tL, weights=w,
                     prior_mean=0.0, prior_var=1.5, save_all=True)
    flips = int(np.sum(o["all_ancestors"] != r["all_ancestors"]))
    assert flips == 0, (case, flips)
    np.testing.assert_allclose(o["all_x_t"], r["all_x_t"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_log_weights"], r["all_log_weights"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_loglikelihood_estimate"], r["all_loglikelihood_estimate"], rtol=RTOL, atol=ATOL)
    if pf != "filter":
        np.testing.assert_allclose(o["all_statistics"], r["all_statistics"], rtol=RTOL, atol=1e-8)
        ref = r["mean_statistic"]
    else:
        ref = r["statistics"]
    l2 = np.linalg.norm(o["mean_stat"] - ref)
    assert l2 <= 1e-8 * max(1.0, np.linalg.norm(ref)), (case, o["mean_stat"], ref)
    assert abs(o["loglik"] - r["loglikelihood_estimate"]) <= ATOL + RTOL * abs(r["loglikelihood_estimate"])
    np.testing.assert_allclose(o["x_t"], r["x_t"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["log_weights"], r["log_weights"], rtol=RTOL, atol=ATOL)


def test_giant_replay_batch_of_windows(ctx):
    """Several whole-GPU windows in one call (blockIdx.y = window), different N, T and parameters."""
    rs = np.random.RandomState(5)
    qs, refs = [], []
    for N, T, A in ((17000, 5, 0.9), (30011, 3, 0.5), (16385, 7, 0.97)):
        theta = [A, 1.3, 1.1]
        y = rs.normal(size=T)
        z0, u, z = po.draw_streams(rs, N, T)
        qs.append(dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="replay", N=N,
                       t1=0, tL=T, lambduh=1.0, prior_mean=0.0, prior_var=2.0, y=y, theta=theta, z0=z0, u=u, z=z))
        refs.append(po.pf_window("svm", theta, y, N, z0, u, z, pf="poyiadjis_N", prior_mean=0.0, prior_var=2.0))
    outs = ctx.run_batch(qs, want_final=True)
    for o, r in zip(outs, refs):
        np.testing.assert_allclose(o["mean_stat"], r["mean_statistic"], rtol=1e-9, atol=1e-8)
        np.testing.assert_allclose(o["x_t"], r["x_t"], rtol=RTOL, atol=ATOL)
        assert abs(o["loglik"] - r["loglikelihood_estimate"]) <= 1e-9 * abs(r["loglikelihood_estimate"])


# ---------------------------------------------------------------------------------------------------------------------
# the CDF kernel against NumPy, bit for bit, on adversarial weights (read back from the window's scratch)
# ---------------------------------------------------------------------------------------------------------------------
def _logweights(kind, N, rs):
    if kind == "normal":
        return rs.randn(N) * 2.0
    if kind == "wide":
        return rs.randn(N) * 40.0
    if kind == "very_wide":
        return rs.randn(N) * 300.0          # most weights underflow, subnormals on the way
    if kind == "equal":
        return np.zeros(N)
    if kind == "one_hot":
        lw = np.full(N, -1e300)
        lw[N // 3] = 0.0
        return lw
    if kind == "decreasing":
        return np.sort(rs.randn(N) * 5.0)[::-1].copy()
    if kind == "increasing":
        return np.sort(rs.randn(N) * 5.0)
    if kind == "dyadic":
        return np.log(2.0) * rs.randint(-30, 1, N).astype(float)     # powers of two: exact ties in the running sum
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["normal", "wide", "very_wide", "equal", "one_hot", "decreasing", "increasing", "dyadic"])
@pytest.mark.parametrize("N", [16385, 65536, 100003, 1000000])
def test_cdf_kernel_is_numpy_bit_for_bit(ctx, kind, N, monkeypatch):
    """cdf = cumsum(p) / cumsum(p)[-1] with p = exp(lw - max); p /= np.sum(p), as np.random.choice builds it (pf.py:26-30),
    read back from the window's scratch after one REPLAY step with the given log-weights (warm start).  Everything
    downstream of exp is NumPy's arithmetic operation for operation -- the chunked pairwise np.sum, the division, the
    sequential cumsum, the final division -- so wherever the device's exp (ocml) returns what NumPy's (glibc) returns the
    CDF must be NumPy's BITWISE: asserted for weights whose exp is exact or agrees on both sides (equal, one-hot, powers
    of two).  The two libraries may differ in the last place of some exp; then the CDFs differ by the few-ulp effect of
    those weights (asserted <= 4 ulp of 1) and the ancestors of N uniforms must still be NumPy's."""
    import torch
    from sgmcmc_ssm_amd import _capi
    rs = np.random.RandomState(N % 1000 + len(kind))
    lw = _logweights(kind, N, rs)
    dev = torch.device("cuda", 0)
    model, dtype = "svm", "f64"
    L = grid_layout(model, dtype, N, True)
    sb = ctx.scratch_bytes(model, dtype, "replay", N)
    assert sb == L["bytes"], (sb, L["bytes"])
    T = 1
    y = torch.zeros(T, dtype=torch.float64, device=dev)
    theta = torch.tensor([0.9, 1.0, 1.0, 0.0], dtype=torch.float64, device=dev)
    x0 = torch.from_numpy(rs.randn(N)).to(dev)
    lw_d = torch.from_numpy(lw).to(dev)
    u = torch.from_numpy(rs.random_sample(N)).to(dev)
    z = torch.from_numpy(rs.randn(N)).to(dev)
    out = torch.zeros(_capi.OUT_DOUBLES, dtype=torch.float64, device=dev)
    anc = torch.zeros(N, dtype=torch.int32, device=dev)
    tx = torch.zeros((T + 1) * N, dtype=torch.float64, device=dev)
    tlw = torch.zeros((T + 1) * N, dtype=torch.float64, device=dev)
    scratch = torch.zeros(sb, dtype=torch.uint8, device=dev)
    d = np.zeros(1, dtype=_capi.DEV_PROBLEM_DTYPE)
    d["y"], d["theta"], d["u"], d["z"] = y.data_ptr(), theta.data_ptr(), u.data_ptr(), z.data_ptr()
    d["init_x"], d["init_logw"] = x0.data_ptr(), lw_d.data_ptr()
    d["out"], d["scratch"] = out.data_ptr(), scratch.data_ptr()
    d["trace_x"], d["trace_logw"], d["trace_anc"] = tx.data_ptr(), tlw.data_ptr(), anc.data_ptr()
    d["prior_var"], d["lambduh"] = 1.0, 1.0
    d["T"], d["t1"], d["tL"], d["N"] = T, 0, T, N
    d["smoother"], d["stat"] = _capi.SMOOTHER["nemeth"], _capi.STAT["score"]
    desc = torch.from_numpy(d.view(np.uint8).reshape(1, -1)).to(dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    ctx.launch_device_grid(model, "prior", dtype, "replay", N, T, 1, desc.data_ptr(), st)
    torch.cuda.synchronize(dev)
    raw = scratch.cpu().numpy()
    cdf = raw[L["cdf"]:L["cdf"] + 8 * N].view(np.float64)
    nwalk = int(raw[L["head"]:L["head"] + 8 * 32].view(np.float64)[6])
    # NumPy on the same log-weights
    p = np.exp(lw - np.max(lw))
    p /= np.sum(p)
    ref = np.cumsum(p)
    ref /= ref[-1]
    same = cdf.view(np.uint64) == ref.view(np.uint64)
    anc_ref = np.searchsorted(ref, u.cpu().numpy(), side="right")
    anc_dev = anc.cpu().numpy()
    nflip = int(np.sum(anc_dev != np.minimum(anc_ref, N - 1)))
    # the device exp (ocml) is within an ulp of NumPy's (glibc) but not always equal: if every p agrees the CDF must be
    # bitwise NumPy's; otherwise the CDFs may differ in the last places AFTER the first differing p -- but never by more
    # than the few-ulp effect of those p, and the ancestors of N uniforms must still agree
    frac = float(np.mean(same))
    print("cdf[{0} N={1}] bitwise equal: {2:.6f}, walk elements {3}, ancestor flips {4}".format(kind, N, frac, nwalk, nflip))
    assert np.max(np.abs(cdf - ref)) <= 4 * np.finfo(float).eps, np.max(np.abs(cdf - ref))
    assert nflip == 0
    assert cdf[-1] == 1.0 and np.all(np.diff(cdf) >= 0)
    if kind in ("equal", "one_hot", "dyadic"):
        assert frac == 1.0           # exp of these arguments is exact (0, or powers of two) on both sides
    assert nwalk >= 1
    # the lone-workgroup form of the same kernel (PFGRAD_CDF_SINGLE=1; the default spreads the particle axis over the GPU
    # in four launches): which steps are walked may differ, the CDF may not -- bitwise, whatever the device's exp returned
    monkeypatch.setenv("PFGRAD_CDF_SINGLE", "1")
    scratch.zero_(); anc.zero_()
    ctx.launch_device_grid(model, "prior", dtype, "replay", N, T, 1, desc.data_ptr(), st)
    torch.cuda.synchronize(dev)
    raw1 = scratch.cpu().numpy()
    cdf1 = raw1[L["cdf"]:L["cdf"] + 8 * N].view(np.float64)
    assert np.array_equal(cdf1.view(np.uint64), cdf.view(np.uint64))
    assert np.array_equal(anc.cpu().numpy(), anc_dev)
    C = L["C"]
    assert np.array_equal(raw1[L["coarse"]:L["coarse"] + 8 * C], raw[L["coarse"]:L["coarse"] + 8 * C])


# ---------------------------------------------------------------------------------------------------------------------
# DEVICE generator: the launch replayed by the oracle from its recorded draws
# ---------------------------------------------------------------------------------------------------------------------
def grid_device_ancestors(logw, u, TILE):
    """The resampling of pfg_grid_step_dev_kernel (a restatement of the KERNEL's CDF layout, not of the reference: both
    draw ancestors i.i.d. from softmax(logw)): tile-wise maxima m_b and local scans cs_b = cumsum(exp(lw - m_b)) with totals
    W_b; PWn = cumsum(W_b exp(m_b - m)) / W (where tile b's CDF ends), scn_b = exp(m_b - m) / W; child u -> parent tile =
    #{b <= G-2: PWn[b] <= u}, position inside = #{j: PWn[b-1] + cs_b[j] scn_b <= u} (at most the tile's last particle)."""
    N = logw.shape[0]
    G = (N + TILE - 1) // TILE
    mb = np.array([np.max(logw[b * TILE:(b + 1) * TILE]) for b in range(G)])
    m = np.max(mb)
    sc = np.exp(mb - m)
    cs = [np.cumsum(np.exp(logw[b * TILE:(b + 1) * TILE] - mb[b])) for b in range(G)]
    Wb = np.array([c[-1] for c in cs])
    PW = np.cumsum(Wb * sc)
    invW = 1.0 / PW[-1]
    PWn, scn = PW * invW, sc * invW
    pt = np.searchsorted(PWn[:G - 1], u, side="right")
    anc = np.empty(N, dtype=np.int64)
    for b in np.unique(pt):
        sel = pt == b
        F = (PWn[b - 1] if b > 0 else 0.0) + cs[b] * scn[b]
        pos = np.minimum(np.searchsorted(F, u[sel], side="right"), F.shape[0] - 1)
        anc[sel] = b * TILE + pos
    return anc


DEVICE_CASES = [
    ("svm", "prior", [0.95, 1.4, 1.4], 20000, 8, "poyiadjis_N", 1.0),
    ("garch", "optimal", [0.0, 2.0, 2.0, 1.8], 50001, 5, "nemeth", 0.9),
    ("lgssm", "optimal", [0.9, 1.0, 1.2, 1.0], 100000, 4, "filter", 1.0),
    ("svm", "prior", [0.9, 1.2, 1.1], 300000, 4, "poyiadjis_N", 1.0),
    ("lgssm", "prior", [0.9, 0.7, 1.2, 1.0], 1200000, 3, "nemeth", 0.9),          # 2048-particle tiles (N > 2^19)
    ("svm", "prior", [0.9, 1.2, 1.1], 4194304, 3, "poyiadjis_N", 1.0),             # the maximum (8 tile partials per thread)
]


@pytest.mark.parametrize("case", DEVICE_CASES, ids=lambda c: "{0}-{1}-N{3}-{5}".format(*c))
def test_giant_device_launch_replayed_by_oracle(ctx, case):
    model, kernel, theta, N, T, pf, lam = case
    rs = np.random.RandomState(N % 9973 + 1)
    y = rs.normal(size=T) * (3.0 if model == "svm" else 1.0)
    t1, tL = 1, T - 1
    w = rs.uniform(1.0, 40.0, size=tL - t1)
    smoother = "filter" if pf == "filter" else "nemeth"
    q = dict(model=model, kernel=kernel, smoother=smoother, stat="score", dtype="f64", rng="device", N=N,
             t1=t1, tL=tL, lambduh=lam, prior_mean=0.0, prior_var=1.5, y=y, weights=w, theta=theta, seed=1234 + N, stream=7)
    o = ctx.run_batch([q], want_final=True, want_trace=True, want_draws=True)[0]
    TILE = 2048 if N > (1 << 19) else 1024
    ud = o["rec_ud"]
    assert np.all(np.diff(ud, axis=1) >= 0) and ud.min() > 0 and ud.max() < 1          # sorted uniforms, rank order
    flips = [0]

    def resampler(t, logw):
        a = grid_device_ancestors(logw, ud[t], TILE)
        flips[0] += int(np.sum(a != o["all_ancestors"][t]))
        return o["all_ancestors"][t].astype(np.int64)      # continue on the kernel's ancestors (flips are counted)

    r = po.pf_window(model, theta, y, N, o["rec_z0"], None, o["rec_z"], kernel=kernel, pf=pf, lambduh=lam, t1=t1, tL=tL,
                     weights=w, prior_mean=0.0, prior_var=1.5, save_all=True, resampler=resampler)
    assert flips[0] == 0, flips
    np.testing.assert_allclose(o["all_x_t"], r["all_x_t"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(o["all_log_weights"], r["all_log_weights"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(o["all_loglikelihood_estimate"], r["all_loglikelihood_estimate"], rtol=1e-8, atol=1e-8)
    ref = r["statistics"] if pf == "filter" else r["mean_statistic"]
    np.testing.assert_allclose(o["mean_stat"], ref, rtol=1e-7, atol=1e-7)
    # same key, no recording: bitwise the same result -- or, where the window is the Poyiadjis O(N) score, the score-only
    # twin of the timestep kernel (PFG_SMOOTHER_POYIADJIS_N: another specialisation, equal to the last place or two)
    o2 = ctx.run_batch([q])[0]
    if pf == "poyiadjis_N":
        assert ctx.last_variant().endswith("_score1")
        np.testing.assert_allclose(o2["mean_stat"], o["mean_stat"], rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(o["mean_stat"]).max())))
        assert abs(o2["loglik"] - o["loglik"]) <= 1e-13 * abs(o["loglik"])
    else:
        assert not ctx.last_variant().endswith("_score1")
        assert np.array_equal(o2["mean_stat"], o["mean_stat"]) and o2["loglik"] == o["loglik"]


def test_giant_device_sorted_uniforms_are_order_statistics(ctx):
    """The N uniforms of a step must be the order statistics of N i.i.d. uniforms: E[U_(r)] = r / (N + 1) along the whole
    rank range (a wrong tile offset shows there), normalised spacings ~ Exp(1), pooled KS against U(0,1)."""
    from scipy import stats as st
    N, T = 200000, 6
    q = dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="device", N=N, t1=0, tL=T,
             lambduh=1.0, prior_mean=0.0, prior_var=1.0, y=np.zeros(T), theta=[0.9, 1.0, 1.0], seed=99, stream=3)
    o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
    ud = o["rec_ud"]
    r = (np.arange(N) + 1.0) / (N + 1.0)
    # sd of U_(r) is sqrt(r (1 - r) / (N + 2)) <= 1.1e-3; the mean over T steps at a tenth of that scale
    dev = np.abs(ud.mean(axis=0) - r) / np.sqrt(r * (1 - r) / (N + 2) / T)
    assert dev.max() < 6.0, dev.max()
    sp = np.diff(np.concatenate((np.zeros((T, 1)), ud), axis=1), axis=1) * (N + 1)
    assert abs(sp.mean() - 1.0) < 5e-3 and abs(sp.var() - 1.0) < 2e-2
    assert abs(np.corrcoef(sp[:, :-1].ravel(), sp[:, 1:].ravel())[0, 1]) < 5e-3
    assert st.kstest(ud[0][::7], "uniform").pvalue > 1e-4
    z = o["rec_z"].ravel()
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1) < 5 * np.sqrt(2.0 / z.size)


def test_million_particles_kalman_ground_truth_and_resampling_counts(ctx):
    """N = 10^6 (the size the reference's bias experiments use as ground truth), LGSSM: the PF score must sit on the exact
    Kalman gradient within a few standard errors of a 10^6-particle estimate; REPLAY and DEVICE agree within that."""
    N, T = 1000000, 24
    rs = np.random.RandomState(11)
    theta = [0.9, 1.0, np.sqrt(1.0 / 0.7), 1.0]
    x = 0.0
    y = np.empty(T)
    for t in range(T):
        x = 0.9 * x + np.sqrt(0.7) * rs.randn()
        y[t] = x + rs.randn()
    q = dict(model="lgssm", kernel="optimal", smoother="nemeth", stat="score", dtype="f64", rng="device", N=N, t1=0, tL=T,
             lambduh=1.0, prior_mean=0.0, prior_var=0.7 / (1 - 0.81), y=y, theta=theta, seed=5, stream=1)
    outs = []
    for s in range(4):
        q["stream"] = s
        outs.append(ctx.run_batch([dict(q)])[0])
    assert ctx.last_variant() == "grid2048_score1"
    g = np.array([o["mean_stat"] for o in outs])
    ll = np.array([o["loglik"] for o in outs])
    # small-N runs of the one-workgroup kernels on the same data: their mean converges to the same place like 1/N
    qs = []
    for s in range(512):
        qq = dict(q)
        qq["N"], qq["stream"] = 4000, 1000 + s
        qs.append(qq)
    small = ctx.run_batch(qs)
    gs = np.array([o["mean_stat"] for o in small])
    se = gs.std(axis=0) / np.sqrt(len(qs))
    # bias of the N = 4000 estimator is O(1/N); the 10^6 runs scatter with sd(4000) / sqrt(250)
    sd_big = gs.std(axis=0) / np.sqrt(N / 4000.0)
    assert np.all(np.abs(g - g.mean(axis=0)) < 6 * sd_big + 1e-12), (g, sd_big)
    assert np.all(np.abs(g.mean(axis=0) - gs.mean(axis=0)) < 6 * se + 6 * sd_big + 5e-3 * np.abs(gs.mean(axis=0)) + 1e-2), (g.mean(axis=0), gs.mean(axis=0), se)
    assert np.ptp(ll) < 0.05 and abs(ll.mean() - np.mean([o["loglik"] for o in small])) < 0.05


# ---------------------------------------------------------------------------------------------------------------------
# the REFERENCE's own giant-N calls (tests/golden/giant.npz, written by make_golden.py from the reference itself):
# helper.pf_gradient_estimate(pf='poyiadjis_N', N=1000000) on the buffered 48-step window of
# gradient_error_fig_scripts/svm_grad_compare.py:58-82 (garch_grad_compare.py:66-93: 40 steps), through the drop-in Helper
# ---------------------------------------------------------------------------------------------------------------------
def _giant_cases():
    from conftest import Golden
    g = Golden("giant.npz")
    return [(g, m) for m in g.meta]


@pytest.mark.parametrize("idx", range(7))
def test_reference_giant_calls_seed_for_seed(idx):
    """np.random.seed(s); helper.pf_gradient_estimate(N = 10^5 | 3 10^5 | 10^6) = the reference's numbers to rtol 1e-9
    (the north-star bar is 1e-4 ||g||), and the NEXT np.random draw is the reference's: the call consumed exactly what
    the reference consumes.  A parallel scan in place of NumPy's sequential cumsum flips an ancestor every ~16 steps at
    N = 10^6 (pfg_grid_cdf.hpp) and a flipped ancestor decorrelates the run: this test is what that kernel is for."""
    from test_gpu_paris import _helper_for, _params_for
    from test_host_logic import vec
    g, m = _giant_cases()[idx]
    key = m["key"]
    helper = _helper_for(g, m)
    p = _params_for(m["model"], g.get(key, "theta"))
    kw = dict(observations=g.get(key, "y").reshape(-1, 1), parameters=p, subsequence_start=m["t1"], subsequence_end=m["tL"],
              weights=g.get(key, "weights"), pf=m["pf"], N=m["N"], kernel=m["kernel"], **m["kwargs"])
    if m["has_grad"]:
        np.random.seed(m["seed"])
        grad = helper.pf_gradient_estimate(**kw)
        nxt = np.random.random_sample()
        ref = g.get(key, "grad")
        got = vec(m["model"], grad)
        l2 = np.linalg.norm(got - ref)
        print("giant", key, m["model"], m["N"], "grad L2 err", l2, "of", np.linalg.norm(ref))
        assert l2 <= 1e-9 * max(1.0, np.linalg.norm(ref)), (m, got, ref)
        assert nxt == float(g.get(key, "next_draw"))
    if m["has_loglik"]:
        np.random.seed(m["seed"])
        ll = helper.pf_loglikelihood_estimate(**kw)
        nxt = np.random.random_sample()
        assert abs(ll - float(g.get(key, "loglik"))) <= 1e-9 * abs(float(g.get(key, "loglik"))), (m, ll)
        assert nxt == float(g.get(key, "next_draw_loglik"))


# ---------------------------------------------------------------------------------------------------------------------
# f32 particle state on the whole-GPU window
# ---------------------------------------------------------------------------------------------------------------------
F32_CASES = [
    ("svm", "prior", [0.95, 1.4, 1.4], "poyiadjis_N", 1.0),
    ("garch", "optimal", [0.0, 2.0, 2.0, 1.8], "nemeth", 0.9),
    ("garch", "prior", [0.0, 2.0, 2.0, 1.8], "nemeth", 0.9),
    ("lgssm", "optimal", [0.9, 1.0, 1.2, 1.0], "poyiadjis_N", 1.0),
    ("lgssm", "prior", [0.9, 0.7, 1.2, 1.0], "filter", 1.0),
]


def _f32_problem(model, kernel, theta, pf, lam, N, T, rng, rs):
    y = rs.normal(size=T)
    q = dict(model=model, kernel=kernel, smoother="filter" if pf == "filter" else "nemeth", stat="score", rng=rng, N=N,
             t1=1, tL=T - 1, lambduh=lam, prior_mean=0.0, prior_var=1.5, y=y, weights=rs.uniform(1.0, 5.0, size=T - 2),
             theta=theta, dtype="f32")
    if rng == "replay":
        q["z0"], q["u"], q["z"] = po.draw_streams(rs, N, T)
    else:
        q["seed"], q["stream"] = 77 + N, 5
    return q


@pytest.mark.parametrize("case", F32_CASES, ids=lambda c: "{0}-{1}-{3}".format(*c))
@pytest.mark.parametrize("N", [3000, 12000])
def test_f32_grid_window_equals_the_one_workgroup_f32_kernel(ctx, monkeypatch, case, N):
    """dtype='f32' (particles, statistics and log-weights in f32; weight sums, CDF and search in f64, as for every
    kernel), REPLAY: the whole-GPU window forced at a size the one-workgroup kernel serves too must resample the SAME
    ancestors at every step and return the same particles and log-weights bit for bit -- 3 and 12 tiles, every model
    and proposal kernel, the filter.  (Against f64 a REPLAY run cannot be compared beyond a few steps: one ancestor
    that f32 rounding flips moves every CDF edge behind it, measured here 1 -> 96 -> 6077 differing ancestors within
    three steps at N = 12000; seed-for-seed parity is an f64 property, SURVEY finding 2.)"""
    model, kernel, theta, pf, lam = case
    T = 6
    q = _f32_problem(model, kernel, theta, pf, lam, N, T, "replay", np.random.RandomState(5))
    one = ctx.run_batch([dict(q)], want_trace=True)[0]
    assert ctx.last_variant() == "mem1024"
    monkeypatch.setenv("PFGRAD_VARIANT", "grid")
    grid = ctx.run_batch([dict(q)], want_trace=True)[0]
    assert ctx.last_variant() == "grid1024"
    assert np.array_equal(one["all_ancestors"], grid["all_ancestors"])
    assert np.array_equal(one["all_x_t"], grid["all_x_t"]) and np.array_equal(one["all_log_weights"], grid["all_log_weights"])
    np.testing.assert_allclose(grid["mean_stat"], one["mean_stat"], rtol=1e-6, atol=1e-7)     # f64 sums in another order
    assert abs(grid["loglik"] - one["loglik"]) <= 1e-7 * abs(one["loglik"])


@pytest.mark.parametrize("case", F32_CASES[:4], ids=lambda c: "{0}-{1}-{3}".format(*c))
@pytest.mark.parametrize("rng,N", [("replay", 40000), ("device", 70001), ("device", 600000)])
def test_giant_f32_state_against_f64(ctx, case, rng, N):
    """At sizes only the whole-GPU window serves (1024- and 2048-particle tiles): f32 against f64 on the same random
    inputs.  Once f32 rounding has flipped an ancestor the two runs are different Monte-Carlo draws of the same
    estimator (see above), so this is a statistical check: gradient within 5 % of its norm, log-likelihood within 0.5 %
    -- a misplaced record or tile shows up as garbage, not as 1 %."""
    model, kernel, theta, pf, lam = case
    q = _f32_problem(model, kernel, theta, pf, lam, N, 6, rng, np.random.RandomState(N % 7919))
    b = ctx.run_batch([dict(q)])[0]
    assert ctx.last_variant() == ("grid2048" if N > (1 << 19) else "grid1024") + ("_score1" if rng == "device" and pf == "poyiadjis_N" else "")
    a = ctx.run_batch([dict(q, dtype="f64")])[0]
    assert np.all(np.isfinite(b["mean_stat"])) and np.isfinite(b["loglik"])
    scale = max(1.0, float(np.linalg.norm(a["mean_stat"])))
    assert np.linalg.norm(a["mean_stat"] - b["mean_stat"]) <= 5e-2 * scale, (a["mean_stat"], b["mean_stat"])
    assert abs(a["loglik"] - b["loglik"]) <= 5e-3 * max(1.0, abs(a["loglik"])), (a["loglik"], b["loglik"])
    assert not np.array_equal(a["mean_stat"], b["mean_stat"])           # it really is another arithmetic


# ---------------------------------------------------------------------------------------------------------------------
# resident windows (what bench.py --config g1 times)
# ---------------------------------------------------------------------------------------------------------------------
def test_resident_windows_equal_run_batch_and_graph_replay(ctx):
    """sgmcmc_ssm_amd.grid.ResidentWindows (descriptors, observations, parameters and scratch resident; T + 2 launches per
    repetition, the repetition counter on the device) returns BITWISE what pfg_run_batch returns for (seed, stream0 + b,
    step = repetition) -- the pinned path of test_giant_device_launch_replayed_by_oracle -- and its hipGraph form
    (launch_graph) bitwise what the eager launches return."""
    from sgmcmc_ssm_amd.grid import ResidentWindows
    N, T, B = 30000, 12, 3
    rs = np.random.RandomState(3)
    y = rs.normal(size=T) * 1.5
    th = np.array([[0.95, 1.4, 1.4], [0.9, 1.2, 1.1], [0.8, 1.0, 1.3]])
    w = rs.uniform(1.0, 3.0, size=6)
    kw = dict(t1=3, tL=9, weights=w, prior_var=2.0, seed=11, stream0=4)
    rw = ResidentWindows("svm", y, th, N, **kw)
    eager = []
    for rep in range(3):
        rw.launch()
        g, ll = rw.results()
        eager.append((g.copy(), ll.copy()))
        ref = ctx.run_batch([dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="device", N=N,
                                  t1=3, tL=9, lambduh=1.0, prior_mean=0.0, prior_var=2.0, y=y, weights=w, theta=th[b],
                                  seed=11, stream=4 + b, step=rep) for b in range(B)])
        for b in range(B):
            assert np.array_equal(g[b], ref[b]["mean_stat"]) and ll[b] == ref[b]["loglik"], (rep, b)
    assert not np.array_equal(eager[0][0], eager[1][0])                 # a repetition is a fresh draw
    rg = ResidentWindows("svm", y, th, N, **kw)
    rg.launch_graph(1)
    g, ll = rg.results()
    assert np.array_equal(g, eager[0][0]) and np.array_equal(ll, eager[0][1])
    rg.launch_graph(2)                                                    # repetitions 1 and 2 in one graph launch
    g, ll = rg.results()
    assert np.array_equal(g, eager[2][0]) and np.array_equal(ll, eager[2][1]) and rg.launches == 3
    rg.launch_graph(1)                                                    # the cached graph again: repetition 3
    assert rg.launches == 4 and not np.array_equal(rg.results()[0], eager[2][0])


def test_score_only_twin_refuses_other_estimators(ctx):
    """PFG_SMOOTHER_POYIADJIS_N launches of the whole-GPU window run the score-only twin of the timestep kernel; a window
    whose descriptor is not (NEMETH, lambduh = 1, score) must come back as NaNs, the others untouched."""
    import torch
    from sgmcmc_ssm_amd.grid import ResidentWindows
    N, T, B = 30000, 8, 3
    rs = np.random.RandomState(4)
    y = rs.normal(size=T)
    th = np.tile([0.95, 1.4, 1.4], (B, 1))
    good = ResidentWindows("svm", y, th, N, t1=2, tL=6, prior_var=2.0, seed=5, stream0=1)
    good.launch()
    g0, ll0 = good.results()
    assert good.ctx.last_variant() == "grid1024_score1" and np.isfinite(g0).all()
    bad = ResidentWindows("svm", y, th, N, t1=2, tL=6, prior_var=2.0, seed=5, stream0=1)
    bad._desc["lambduh"][1] = 0.9
    bad.desc_dev.copy_(torch.from_numpy(bad._desc.view(np.uint8).reshape(B, -1)))
    bad.launch()
    g1, ll1 = bad.results()
    assert np.isnan(g1[1]).all() and np.isnan(ll1[1])
    assert np.array_equal(g1[[0, 2]], g0[[0, 2]]) and np.array_equal(ll1[[0, 2]], ll0[[0, 2]])
    # stated as NEMETH the same descriptors run the general kernel
    bad._launch_smoother = "nemeth"
    bad.step_ctr.zero_()
    bad.launch()
    g2, ll2 = bad.results()
    assert bad.ctx.last_variant() == "grid1024" and np.isfinite(g2).all()
    np.testing.assert_allclose(g2[[0, 2]], g0[[0, 2]], rtol=1e-13, atol=1e-13)
