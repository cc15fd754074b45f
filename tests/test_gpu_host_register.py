"""pfg_host_register: replay streams staged by DMA from the caller's page-locked buffers give the same
result as the packed path, bit for bit; registration errors are reported, not fatal."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_registered_streams_same_result(golden_sampler):
    from sgmcmc_ssm_amd import _capi, particle_filters as pfm
    y = golden_sampler.get("lgssm", "y")
    theta = np.array([0.9, 1.0, 0.7 ** -0.5, 1.0])
    ctx = _capi.default_context(0)
    N = 1000
    q = pfm.make_problem("lgssm", "optimal", "poyiadjis_N", y, theta, N, prior_mean=0.0, prior_var=1.0,
                         random_state=np.random.RandomState(11))
    plain = dict(q, u=q["u"].copy(), z=q["z"].copy())           # ordinary pageable arrays: packed path
    locked = dict(q, u=q["u"].copy(), z=q["z"].copy())
    assert _capi.host_register(locked["u"]) and _capi.host_register(locked["z"])
    assert not _capi.host_register(locked["u"])                  # twice: refused
    try:
        a = ctx.run_batch([plain])[0]
        b = ctx.run_batch([locked])[0]
        c = ctx.run_batch([plain, locked])                       # mixed batch: packed and direct pieces interleave
    finally:
        assert _capi.host_unregister(locked["u"]) and _capi.host_unregister(locked["z"])
        assert not _capi.host_unregister(locked["u"])            # not registered any more
    for o in (b, c[0], c[1]):
        assert np.array_equal(a["mean_stat"], o["mean_stat"]) and a["loglik"] == o["loglik"]


def test_pooled_stream_buffers_are_registered_and_reused(monkeypatch):
    from sgmcmc_ssm_amd import particle_filters as pfm
    N, T = 1000, 200                       # 1.6 MB per array: above the pinning threshold
    monkeypatch.setattr(pfm, "_PIN_MAX_BYTES", pfm._pinned_bytes + (64 << 20))     # headroom whatever ran before
    pfm._stream_pool.pop((N, T), None)     # ... and a fresh pair, not one an earlier test left in the pool
    n0 = len(pfm._pinned)
    bufs = pfm._stream_buffers(N, T)
    assert len(pfm._pinned) == n0 + 1 and pfm._is_pinned(bufs)
    pfm._recycle_streams([{"_stream_bufs": bufs}])
    again = pfm._stream_buffers(N, T)
    assert again[0] is bufs[0] and len(pfm._pinned) == n0 + 1
    pfm._recycle_streams([{"_stream_bufs": again}])


def test_vectorised_marshalling_matches_the_per_problem_path(golden_sampler):
    """Batches of >= 64 device-generator windows are marshalled column by column (_run_batch_plain); the
    descriptors must be the ones the per-problem ctypes path builds: identical results, window by window."""
    from sgmcmc_ssm_amd import _capi, particle_filters as pfm
    y = golden_sampler.get("lgssm", "y")
    ctx = _capi.default_context(0)
    rs = np.random.RandomState(5)
    probs = []
    for b in range(96):
        T = int(rs.randint(20, len(y)))
        t1 = int(rs.randint(0, 5)); tL = int(rs.randint(T - 5, T + 1))
        w = rs.uniform(0.5, 2.0, size=tL - t1) if b % 3 == 0 else None
        theta = np.array([0.9, 1.0, 0.7 ** -0.5, 1.0]) * (1.0 + 0.01 * rs.randn(4))
        probs.append(pfm.make_problem("lgssm", "optimal" if b % 2 else "prior", "nemeth" if b % 5 == 0 else "poyiadjis_N",
                                      y[:T] if b % 4 else y, theta, 100 + b, t1=t1, tL=tL if b % 4 else None, weights=w if b % 4 else None,
                                      prior_mean=0.0, prior_var=1.0, rng="device", seed=1234, stream=b,
                                      **({"lambduh": 0.9} if b % 5 == 0 else {})))
    # one model / kernel per launch: split by kernel, keep the order
    for kern in ("prior", "optimal"):
        sel = [q for q in probs if q["kernel"] == kern]
        for q in sel:
            q["N"] = 128                      # one variant for the whole batch
        assert len(sel) < 64
        big = sel + sel                       # >= 64 problems: the vectorised path
        assert len(big) >= 64
        fast = ctx.run_batch(big)
        slow = ctx.run_batch(big, want_final=True)      # asks for the final particles: the per-problem path (same
        assert "x_t" in slow[0] and "x_t" not in fast[0]  # batch size, hence the same kernel variant and draws)
        for a, b_ in zip(fast, slow):
            assert np.array_equal(a["mean_stat"], b_["mean_stat"]) and a["loglik"] == b_["loglik"]
