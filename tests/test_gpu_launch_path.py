"""The launch bench.py times, pinned deterministically.

bench.py (and every resident SGLD run) goes  ChainEnsemble.launch_pf -> pfg_launch_device  with descriptors
built in ensemble.py (pointers into resident arrays, stream = global chain id, a device-side step counter).
The recorded-draw parity tests (test_gpu_device_replay.py) go through pfg_run_batch, which builds its own
descriptors.  Same kernels, different marshalling and keying -- so here

 (a) every chain's result record of a ChainEnsemble launch must be BITWISE what pfg_run_batch returns for the
     window the descriptor describes with (seed, stream = global chain id, step): at step 0 and after the
     device counter has been bumped by real SGLD steps (parameters moved, too), for full-sequence chains and
     for buffered windows sampled on the host and on the device; a wrong stream / t1 / tL / weights pointer /
     step in either path breaks the equality;
 (b) one chain of a ChainEnsemble launch at BASELINE's full size (SVM T = N = 1000, the bench instantiation
     wg256x4s) gets record buffers in its descriptor; the production launch must ignore them, its trace-honouring
     twin (same descriptors, same marshalling, `launch_pf(traced=True)`) must return bitwise the production
     launch's result records, and the CPU oracle replays that launch from the recorded draws (zero ancestor flips, rtol 1e-8) -- the reference's T-loop (particle_filters/
     buffered_smoother.py:93-133, pf.py:138-181) on what sgmcmc_sampler.py:364-384 says a chain's gradient is.
"""
import numpy as np
import pytest

from oracle import pf_oracle as po
from test_host_logic import default_params, GEN

pytestmark = pytest.mark.gpu


def _series(model, T, seed=5):
    np.random.seed(seed)
    return GEN[model](T=T, parameters=default_params(model))["observations"]


def _descriptors(ens):
    """The descriptors as the kernel sees them (device copy), as a structured array."""
    from sgmcmc_ssm_amd import _capi
    ens.synchronize()
    return np.frombuffer(ens.desc_dev.cpu().numpy().tobytes(), dtype=_capi.DEV_PROBLEM_DTYPE)


def _problems(ens, d, step, theta):
    """Host-buffer problems (pfg_problem) for the windows the device descriptors `d` describe."""
    y_all = ens.y_dev.cpu().numpy()
    w_all = None if ens.weights_dev is None else ens.weights_dev.cpu().numpy().reshape(-1)
    out = []
    for c in range(ens.C):
        left = (int(d["y"][c]) - ens.y_dev.data_ptr()) // 8
        T, t1, tL = int(d["T"][c]), int(d["t1"][c]), int(d["tL"][c])
        w = None
        if int(d["weights"][c]) != 0:
            w0 = (int(d["weights"][c]) - ens.weights_dev.data_ptr()) // 8
            w = w_all[w0:w0 + (tL - t1)].copy()
        out.append(dict(model=ens.model, kernel=ens.kernel, smoother="nemeth", stat="score", dtype=ens.dtype,
                        rng="device", N=ens.N, t1=t1, tL=tL, lambduh=float(d["lambduh"][c]),
                        prior_mean=float(d["prior_mean"][c]), prior_var=float(d["prior_var"][c]),
                        flags=int(d["flags"][c]), y=y_all[left:left + T].copy(), weights=w, theta=theta[c].copy(),
                        seed=int(d["seed"][c]), stream=int(d["stream"][c]), step=step))
    return out


def _assert_launch_equals_run_batch(ens, step):
    """One PF launch of the ensemble vs pfg_run_batch on the same windows / keys: bitwise."""
    from sgmcmc_ssm_amd import _capi
    theta = ens.theta()
    assert int(ens.step_ctr.item()) == step
    ens.launch_pf()
    ens.synchronize()
    variant = ens.ctx.last_variant()
    got = ens.out_dev.cpu().numpy().copy()
    d = _descriptors(ens)
    assert np.all(d["stream"] == np.arange(ens.C, dtype=np.uint64) + np.uint64(ens.chain_offset))
    assert np.all(d["step_ctr"] == ens.step_ctr.data_ptr())
    ref = ens.ctx.run_batch(_problems(ens, d, step, theta))
    assert ens.ctx.last_variant() == variant               # same instantiation on both paths
    h = _capi.STAT_DIM[ens.model]
    for c in range(ens.C):
        assert np.array_equal(got[c, :h], ref[c]["mean_stat"]), (c, got[c, :h], ref[c]["mean_stat"])
        assert got[c, 4] == ref[c]["loglik"], (c, got[c, 4], ref[c]["loglik"])
    # and the key matters: the same windows at another step / with the streams shifted by one differ
    other = ens.ctx.run_batch(_problems(ens, d, step + 1, theta))
    assert not any(np.array_equal(got[c, :h], other[c]["mean_stat"]) for c in range(ens.C))
    return variant


@pytest.mark.parametrize("model,N,variant", [("svm", 1000, "wg256x4s"), ("garch", 1000, "wg512x2s"), ("lgssm", 100, "wg64x2s_score1")])
def test_full_sequence_launch_is_run_batch_bitwise(model, N, variant):
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series(model, 90)
    ens = ChainEnsemble(model, y, default_params(model), num_chains=96, N=N, epsilon=0.01, seed=1234, chain_offset=4096)
    assert _assert_launch_equals_run_batch(ens, 0) == variant
    ens.step(3)                    # three real SGLD steps: parameters move, the device counter is at 3
    ens.synchronize()
    assert len({tuple(r) for r in ens.theta()}) == ens.C
    _assert_launch_equals_run_batch(ens, 3)


@pytest.mark.parametrize("windows", ["host", "device"])
@pytest.mark.parametrize("model,N", [("svm", 1000), ("garch", 1000)])
def test_buffered_launch_is_run_batch_bitwise(model, N, windows):
    """S = 16, B = 4 windows (BASELINE config 3's shape) with importance weights from the resident table."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series(model, 300)
    ens = ChainEnsemble(model, y, default_params(model), num_chains=80, N=N, epsilon=0.01, seed=77, chain_offset=160,
                        subsequence_length=16, buffer_length=4, window_sampling=windows)
    if windows == "device":
        ens.launch_windows()
    _assert_launch_equals_run_batch(ens, 0)
    d0 = _descriptors(ens).copy()
    ens.step(2)
    ens.synchronize()
    if windows == "device":
        ens.launch_windows()           # the windows of step 2, as _enqueue_step would draw them
    else:
        ens._set_windows()
        import torch
        ens.desc_dev.copy_(torch.from_numpy(ens._desc.view(np.uint8).reshape(ens.C, -1)))
    d2 = _descriptors(ens)
    assert np.any(d2["y"] != d0["y"])                      # other windows than at step 0
    assert np.all(d2["tL"] - d2["t1"] == 16) and np.all(d2["weights"] != 0)
    _assert_launch_equals_run_batch(ens, 2)


def test_bench_launch_replayed_by_oracle_at_full_size():
    """(b): BASELINE configs[1] (SVM T = N = 1000) through ChainEnsemble -> pfg_launch_device on the bench
    instantiation, the device counter already bumped; chain 7's descriptor carries record buffers."""
    import torch
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    T = N = 1000
    p = default_params("svm")
    np.random.seed(12345)
    y = GEN["svm"](T=T, parameters=p)["observations"]
    ens = ChainEnsemble("svm", y, p, num_chains=72, N=N, epsilon=0.1, seed=2024, chain_offset=0)
    ens.step(1)                                             # counter at 1, parameters of every chain moved
    ens.synchronize()
    theta = ens.theta()
    c = 7
    dev = ens.device
    f64 = lambda *s: torch.zeros(s, dtype=torch.float64, device=dev)
    buf = dict(trace_x=f64(T + 1, N), trace_logw=f64(T + 1, N), trace_stats=f64(T + 1, N, 3), trace_ll=f64(T + 1),
               trace_anc=torch.zeros((T, N), dtype=torch.int32, device=dev),
               rec_u=torch.zeros((T, N), dtype=torch.int32, device=dev), rec_z=f64(T, N), rec_z0=f64(N))
    for k, t in buf.items():
        ens._desc[k][c] = t.data_ptr()
    ens.desc_dev.copy_(torch.from_numpy(ens._desc.view(np.uint8).reshape(ens.C, -1)))
    # the production launch (what bench.py times) ignores the record buffers; its twin with the trace
    # instrumentation compiled in honours them and must return bitwise the same result records
    ens.launch_pf()
    ens.synchronize()
    assert ens.ctx.last_variant() == "wg256x4s" and not ens.ctx.last_traced()
    production = ens.out_dev.cpu().numpy().copy()
    assert not np.any(buf["rec_u"].cpu().numpy()) and not np.any(buf["trace_x"].cpu().numpy())
    ens.launch_pf(traced=True)
    ens.synchronize()
    assert ens.ctx.last_variant() == "wg256x4s" and ens.ctx.last_traced()
    out = ens.out_dev.cpu().numpy()
    assert np.array_equal(out[:, :4], production[:, :4])                      # every chain's gradient, bitwise
    np.testing.assert_allclose(out[:, 4], production[:, 4], rtol=1e-12)       # log-lik: same terms, flushed per step when traced
    o = {k: t.cpu().numpy() for k, t in buf.items()}
    words = o["rec_u"].view(np.uint32)
    assert np.any(words != 0) and np.all(np.isfinite(o["rec_z"]))
    d = _descriptors(ens)
    ref = po.pf_window("svm", theta[c], y.reshape(-1), N, o["rec_z0"], None, o["rec_z"], kernel="prior", pf="poyiadjis_N",
                       lambduh=1.0, stat="score", t1=0, tL=T, weights=None, prior_mean=float(d["prior_mean"][c]),
                       prior_var=float(d["prior_var"][c]), save_all=True,
                       resampler=lambda t, logw: po.device_ancestors(logw, words[t], 256, 4, "fixed32"))
    assert int(np.sum(o["trace_anc"] != ref["all_ancestors"])) == 0
    np.testing.assert_allclose(o["trace_x"], ref["all_x_t"][:, :, 0], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(o["trace_logw"], ref["all_log_weights"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(o["trace_stats"], ref["all_statistics"], rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(o["trace_ll"], ref["all_loglikelihood_estimate"], rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(out[c, :3], ref["mean_statistic"], rtol=1e-8, atol=1e-7)
    assert np.linalg.norm(out[c, :3] - ref["mean_statistic"]) < 1e-6 * max(1.0, np.linalg.norm(ref["mean_statistic"]))
    np.testing.assert_allclose(out[c, 4], ref["loglikelihood_estimate"], rtol=1e-8)
    # the recorded chain is the timed computation: its neighbours (no record buffers) are what run_batch returns
    # for (seed, stream, step = 1), and so is the recorded chain's gradient, bitwise
    same = ens.ctx.run_batch(_problems(ens, d, 1, theta))
    for k in (c, c + 1, 0, ens.C - 1):
        assert np.array_equal(out[k, :3], same[k]["mean_stat"]), k
        assert production[k, 4] == same[k]["loglik"], k
