"""NumPy model of the kernel algorithm that reproduces a SEQUENTIAL fp64 cumulative sum in parallel,
bit for bit (csrc/pfg_grid_cdf.hpp).  TEST INFRASTRUCTURE: the CPU tests check this model against
np.cumsum on adversarial inputs; the GPU tests check the kernel against np.cumsum itself.

Why: np.random.choice (particle_filters/pf.py:26-30 of the reference) resamples against
cdf = cumsum(p); cdf /= cdf[-1] -- a sequential sum whose rounding errors accumulate like a random
walk (~ sqrt(N) ulp).  With N = 10^6 particles a parallel (tree) scan differs from it by enough to flip
an ancestor every ~16 timesteps (2 N^2 delta with delta ~ 3e-14), and one flipped ancestor decorrelates
the whole particle system one step later.  Seed-for-seed parity at that size needs the SAME roundings.

How: for s, p >= 0 with s a double in binade [2^e, 2^(e+1)) and s + p still inside it,
fl(s + p) = s + RN(p / u) u with u = 2^(e-52) (ulp of the binade), unless p / u lies exactly half way
between two integers (round-to-even then looks at the parity of the running sum).  So inside a binade
the sequential sum IS an integer prefix sum in units of u -- associative, hence parallel and exact.
The running sum is non-decreasing, so binades are visited in order; an approximate scan (any order,
relative error <= eta) certifies for almost every step that both s_{k-1} and s_k lie inside the same
binade ("safe" steps).  The few remaining steps -- binade crossings, steps the approximate scan cannot
certify, ties, the first element -- are the "walk" elements: they are chained sequentially with genuine
fp64 additions, everything between two of them is an integer prefix sum at the quantum of the running
sum the walk element left behind."""
import numpy as np

MIN_E = -1023          # "binade" of zero and the subnormals: [0, 2^-1022), quantum 2^-1074


def _binade(x):
    """floor(log2 x) for normal x, MIN_E for zero / subnormals."""
    m, e = np.frexp(x)                      # x = m 2^e, m in [0.5, 1)
    e = e.astype(np.int64) - 1
    return np.where((x == 0) | (e < -1022), MIN_E, e)


def _quantum_exp(e):
    return np.maximum(e, -1022) - 52


def approx_scan(p, block=4096):
    """an inclusive scan in ANOTHER summation order (block-local cumsum + block offsets), as a tiled GPU
    scan produces it: relative error ~ 1e-15, different roundings than np.cumsum"""
    N = p.shape[0]
    out = np.empty(N)
    off = 0.0
    for b in range(0, N, block):
        c = np.cumsum(p[b:b + block][::-1])[::-1] if False else np.cumsum(p[b:b + block])
        out[b:b + block] = off + c
        off = off + c[-1]
    return out


def classify(p, ct, eta):
    """safe[k], binade e[k], integer quanta q[k] (0 for walk elements) from the approximate scan ct"""
    N = p.shape[0]
    prev = np.concatenate(([0.0], ct[:-1]))
    lo = prev * (1.0 - eta)
    hi = ct * (1.0 + eta)
    e_lo, e_hi = _binade(lo), _binade(hi)
    safe = e_lo == e_hi
    qe = _quantum_exp(e_hi)
    # "null" steps: p below half the quantum of the LOWER candidate binade changes nothing whichever binade the
    # running sum is in (fl(s + p) = s), so no certificate is needed (zero weights, long tails of tiny ones)
    null = p < np.ldexp(1.0, (_quantum_exp(e_lo) - 1).astype(np.int64))
    safe |= null
    safe[0] = False
    scaled = np.ldexp(p, (-qe).astype(np.int64))          # exact scaling; < 2^53 for safe steps
    fl = np.floor(scaled)
    tie = (scaled - fl) == 0.5
    safe &= ~tie
    with np.errstate(invalid="ignore", over="ignore"):
        q = np.where(safe, np.rint(np.where(safe, scaled, 0.0)), 0.0).astype(np.uint64)
    return safe, q


def exact_cumsum(p, ct=None, eta=None, stats=None):
    """np.cumsum(p) for p >= 0, computed as the kernel does"""
    p = np.ascontiguousarray(p, dtype=np.float64)
    N = p.shape[0]
    if ct is None:
        ct = approx_scan(p)
    if eta is None:
        eta = (N + 128) * 2.0 ** -52
    safe, q = classify(p, ct, eta)
    Q = np.cumsum(q, dtype=np.uint64)                      # wraps mod 2^64; only differences are used
    walk = np.flatnonzero(~safe)
    s_walk = np.empty(walk.shape[0])
    # ---- the sequential chain over the walk elements (one thread in the kernel) ----
    s = p[0]
    s_walk[0] = s
    for j in range(1, walk.shape[0]):
        w, pw = walk[j], walk[j - 1]
        qe = int(_quantum_exp(_binade(np.float64(s))))
        dq = int(Q[w - 1]) - int(Q[pw])
        dq &= (1 << 64) - 1
        s_before = s + np.ldexp(np.float64(dq), qe)        # exact: multiples of the quantum inside one binade
        s = np.float64(s_before) + p[w]                    # the genuine fp64 addition
        s_walk[j] = s
    # ---- apply: every element from the last walk element at or before it ----
    base = np.maximum.accumulate(np.where(~safe, np.arange(N), -1))
    jb = np.searchsorted(walk, base)
    sb = s_walk[jb]
    qe = _quantum_exp(_binade(sb))
    dq = Q - Q[base]
    out = sb + np.ldexp(dq.astype(np.float64), qe.astype(np.int64))
    if stats is not None:
        stats["walk"] = int(walk.shape[0])
    return out
