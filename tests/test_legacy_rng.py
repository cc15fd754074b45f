"""libpfgrad's native generator of NumPy's legacy RandomState streams (pfg_legacy_streams) is
bit-identical to NumPy: values, final state (key, pos, cached Gaussian), global np.random included.
The REPLAY path's seed compatibility with the reference rests on this (host-only: no GPU needed)."""
import numpy as np
import pytest

from sgmcmc_ssm_amd import _capi, particle_filters


def numpy_streams(rs, N, T):
    z0 = rs.normal(size=N)
    u, z = np.empty((T, N)), np.empty((T, N))
    for t in range(T):
        u[t] = rs.random_sample(N)
        z[t] = rs.normal(size=N)
    return z0, u, z


@pytest.mark.parametrize("N,T,seed,pre", [
    (1000, 120, 1, 0), (1000, 60, 2, 1), (999, 37, 3, 0), (7, 5, 4, 1), (1, 1, 5, 0), (1001, 24, 6, 3),
    (333, 0, 8, 1), (4000, 12, 9, 2), (10000, 6, 10, 0), (2, 500, 11, 1), (313, 77, 12, 5), (156, 311, 13, 0)])
@pytest.mark.parametrize("threads", [1, 3])
def test_native_streams_bit_identical(N, T, seed, pre, threads):
    a, b = np.random.RandomState(seed), np.random.RandomState(seed)
    for _ in range(pre):                 # odd numbers of earlier normals leave a cached Gaussian behind
        a.normal(); b.normal(); a.random_sample(5); b.random_sample(5)
    ref = numpy_streams(a, N, T)
    z0, u, z = np.empty(N), np.empty((T, N)), np.empty((T, N))
    _capi.legacy_streams(b, N, T, z0, u, z, threads=threads)
    assert np.array_equal(ref[0], z0) and np.array_equal(ref[1], u) and np.array_equal(ref[2], z)
    sa, sb = a.get_state(), b.get_state()
    assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
    assert a.normal() == b.normal() and np.array_equal(a.random_sample(4), b.random_sample(4))
    assert a.randint(0, 1000) == b.randint(0, 1000)


@pytest.mark.parametrize("N,T,seed,pre,threads", [
    (1000, 1000, 21, 0, 0), (999, 130, 22, 1, 2), (64, 2000, 23, 3, 0), (17, 7001, 24, 0, 4), (100000, 1, 25, 1, 0),
    (1250, 81, 26, 2, 16), (1000, 400, 27, 1, 3), (2500, 100, 28, 0, 4), (999, 333, 29, 3, 8), (30, 9000, 30, 1, 16)])
def test_pipelined_generator_bit_identical(N, T, seed, pre, threads):
    """Streams long enough for the pipelined path (recurrence on a producer thread, transforms on workers):
    same numbers, same final state, also over two calls in a row."""
    a, b = np.random.RandomState(seed), np.random.RandomState(seed)
    for _ in range(pre):
        a.normal(); b.normal(); a.random_sample(3); b.random_sample(3)
    for rep in range(2):
        ref = numpy_streams(a, N, T)
        z0, u, z = np.empty(N), np.empty((T, N)), np.empty((T, N))
        _capi.legacy_streams(b, N, T, z0, u, z, threads=threads)
        assert np.array_equal(ref[0], z0) and np.array_equal(ref[1], u) and np.array_equal(ref[2], z)
        sa, sb = a.get_state(), b.get_state()
        assert np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]
    assert a.normal() == b.normal() and np.array_equal(a.random_sample(4), b.random_sample(4))


def test_global_np_random_and_draw_replay_streams():
    N, T = 80, 100                        # N * T above the native threshold
    np.random.seed(5)
    ref = numpy_streams(np.random, N, T)
    after = np.random.normal(size=3)
    np.random.seed(5)
    z0, u, z = particle_filters.draw_replay_streams(N, T)
    assert np.array_equal(ref[0], z0) and np.array_equal(ref[1], u) and np.array_equal(ref[2], z)
    assert np.array_equal(after, np.random.normal(size=3))
    rs = np.random.RandomState(6)          # small windows still go through NumPy: same numbers either way
    small = particle_filters.draw_replay_streams(10, 3, rs)
    assert np.array_equal(small[1], numpy_streams(np.random.RandomState(6), 10, 3)[1])


def test_worker_placement_leaves_the_caller_alone(monkeypatch):
    """The transform workers are placed on cores that share the caller's L3 (pfg_legacy_rng.hip: cores_near); the calling
    thread's own affinity mask is never touched, and PFGRAD_RNG_PIN=0 (placement left to the scheduler) gives the same
    numbers and the same final state."""
    import os
    N, T = 1000, 300
    before = os.sched_getaffinity(0)
    outs = []
    for pin in ("1", "0"):
        monkeypatch.setenv("PFGRAD_RNG_PIN", pin)
        rs = np.random.RandomState(77)
        z0, u, z = np.empty(N), np.empty((T, N)), np.empty((T, N))
        _capi.legacy_streams(rs, N, T, z0, u, z, threads=4)
        outs.append((z0, u, z, rs.get_state()))
        assert os.sched_getaffinity(0) == before
    a, b = outs
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[3][1], b[3][1]) and a[3][2:] == b[3][2:]
    ref = numpy_streams(np.random.RandomState(77), N, T)
    assert np.array_equal(ref[2], a[2]) and np.array_equal(ref[1], a[1])
