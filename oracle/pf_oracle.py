"""CPU oracle for the buffered particle-filter score path (TEST INFRASTRUCTURE ONLY).

This file is a NumPy fp64 *restatement* of the reference algorithm
(`sgmcmc_ssm/particle_filters/*` + `sgmcmc_ssm/models/{svm,garch,lgssm}`), written
from the reference's behaviour, not copied from it.  It exists so that the HIP
path can be checked on a GPU box where the reference is absent.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module.  The product package never does: it fails
loudly when the HIP extension is missing.

Pinning: ``tests/golden/make_golden.py`` runs the *reference itself* (imported
from ``/root/reference`` in the build container) and stores inputs/outputs in
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle
against those fixtures with absolute error 0.0.

Reference map (file:line under /root/reference/sgmcmc_ssm):
  log_normalize            particle_filters/pf.py:374-377
  multinomial resampling   particle_filters/pf.py:26-30  (RandomState.choice semantics)
  pf step                  particle_filters/pf.py:7-38
  nemeth / poyiadjis_N     particle_filters/pf.py:138-181, buffered_smoother.py:175-180
  pf_filter                particle_filters/pf.py:40-82
  T-loop                   particle_filters/buffered_smoother.py:12-149
  average_statistic        particle_filters/buffered_smoother.py:151-154
  x0                       particle_filters/kernels.py:83-100, models/garch/kernels.py:7-18
  SVM kernel / score       models/svm/kernels.py:15-64, models/svm/helper.py:342-348
  GARCH kernels / score    models/garch/kernels.py:49-180, models/garch/helper.py:335-372
  LGSSM kernels / score    models/lgssm/kernels.py:11-122, models/lgssm/helper.py:1270-1277
  sufficient statistics    models/lgssm/helper.py:1338-1363, models/garch/helper.py:414-430
"""
import numpy as np
from scipy.special import expit

MODELS = ("svm", "garch", "lgssm")
# theta layouts (raw parameters, the order of Parameters.var_dict in the reference)
THETA_NAMES = {
    "svm": ("A", "LQinv", "LRinv"),                              # svm/parameters.py:21-25
    "lgssm": ("A", "C", "LQinv", "LRinv"),                       # lgssm/parameters.py:20-25
    "garch": ("log_mu", "logit_phi", "logit_lambduh", "LRinv"),  # garch/parameters.py:19-22
}
# column order of the score statistic (what pf_gradient_estimate unpacks)
SCORE_NAMES = {
    "svm": ("LRinv_vec", "LQinv_vec", "A"),                            # svm/helper.py:121-126
    "lgssm": ("LRinv_vec", "LQinv_vec", "C", "A"),                     # lgssm/helper.py:1136-1142
    "garch": ("LRinv_vec", "log_mu", "logit_phi", "logit_lambduh"),    # garch/helper.py:109-115
}
STATE_DIM = {"svm": 1, "lgssm": 1, "garch": 2}
DEFAULT_KERNEL = {"svm": "prior", "garch": "optimal", "lgssm": "optimal"}
LOG_2PI = np.log(2.0 * np.pi)


# --------------------------------------------------------------------------
# derived constants (what the reference's Parameters properties return)
# --------------------------------------------------------------------------
def derived(model, theta):
    """Quantities the kernels/statistics use, computed as the reference's Parameters
    properties do (variables/covariance.py:128-157, variables/garch_var.py:69-91).
    Values are kept as 1-element ndarrays of the reference's shapes ((1,1) matrices,
    (1,) GARCH variables) so every later expression runs through the same NumPy
    array code paths (e.g. ``arr**-1`` is an exact reciprocal) and rounds identically."""
    th = dict(zip(THETA_NAMES[model], [float(v) for v in theta]))
    d = {}
    mat = lambda v: np.array([[v]], dtype=float)
    LRinv = mat(th["LRinv"])
    d["LRinv"] = LRinv
    d["Rinv"] = LRinv.dot(LRinv.T) + 1e-16 * np.eye(1)
    d["R"] = d["Rinv"] ** -1
    if model in ("svm", "lgssm"):
        d["A"] = mat(th["A"])
        if model == "lgssm":
            d["C"] = mat(th["C"])
        LQinv = mat(th["LQinv"])
        d["LQinv"] = LQinv
        d["Qinv"] = LQinv.dot(LQinv.T) + 1e-16 * np.eye(1)
        d["Q"] = d["Qinv"] ** -1
    if model == "garch":
        for k in ("log_mu", "logit_phi", "logit_lambduh"):
            d[k] = np.atleast_1d(th[k]).astype(float)
        d["mu"] = np.exp(d["log_mu"])
        d["phi"] = expit(d["logit_phi"])
        d["lambduh"] = expit(d["logit_lambduh"])
        d["alpha"] = d["mu"] * (1 - d["phi"])
        d["beta"] = d["phi"] * d["lambduh"]
        d["gamma"] = d["phi"] * (1 - d["lambduh"])
    return d


# --------------------------------------------------------------------------
# pieces of one PF step
# --------------------------------------------------------------------------
def log_normalize(log_weights):
    """pf.py:374-377"""
    probs = np.exp(log_weights - np.max(log_weights))
    probs /= np.sum(probs)
    return probs


def multinomial_ancestors(p, u):
    """np.random.choice(range(N), size=N, replace=True, p=p) given its N uniforms
    (pf.py:26-30).  Legacy RandomState.choice: cdf = cumsum(p); cdf /= cdf[-1];
    searchsorted(cdf, u, side='right')."""
    cdf = np.cumsum(p)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, u, side="right")


def device_ancestors(logw, words, NT, PPT, cdf="fixed32"):
    """Multinomial resampling as the DEVICE-generator kernels of libpfgrad lay it out (this is a
    restatement of csrc/pfg_reg_kernel.hpp phases B-E / csrc/pfg_big_kernel.hpp, not of the
    reference: the reference's resampling is `multinomial_ancestors`; both draw ancestors i.i.d.
    from softmax(logw)).  The CDF runs over NT*PPT slots in THREAD-major order -- slot
    q = tid*PPT + k holds particle k*NT + tid, slots of particles >= N weigh 0 -- and child i is
    the count of CDF entries <= its own 32-bit word `words[i]`:
      cdf='fixed32' (LDS-resident kernels): entries floor(min(cs/W * 2^32, 2^32 - 1)), integer compare;
      cdf='f64'         entries cs/W in f64 against (word + 0.5) / 2^32;
      cdf='f64_uniform' (pf_big_kernel): entries cs/W in f64 against the recorded uniform itself."""
    N = logw.shape[0]
    NP = NT * PPT
    q = np.arange(NP)
    particle = (q % PPT) * NT + q // PPT
    p = np.exp(logw - np.max(logw))
    w = np.where(particle < N, p[np.minimum(particle, N - 1)], 0.0)
    cs = np.cumsum(w)
    W = cs[-1]
    if cdf == "fixed32":
        table = np.floor(np.minimum(cs * ((1.0 / W) * 4294967296.0), 4294967295.0)).astype(np.uint64)
        pos = np.searchsorted(table, words.astype(np.uint64), side="right")
    elif cdf == "f64":
        table = cs * (1.0 / W)
        pos = np.searchsorted(table, (words.astype(np.float64) + 0.5) * (1.0 / 4294967296.0), side="right")
    elif cdf == "f64_uniform":
        # pf_big_kernel: `words` are the uniforms themselves (sorted uniforms from exponential spacings)
        table = cs * (1.0 / W)
        pos = np.searchsorted(table, np.asarray(words, dtype=np.float64), side="right")
    else:
        raise ValueError(cdf)
    pos = np.minimum(pos, NP - 1)
    return np.minimum((pos % PPT) * NT + pos // PPT, N - 1)


def sample_x0(model, prior_mean, prior_var, z0):
    """kernels.py:83-100 (n=1), garch/kernels.py:7-18.  z0: (N,) standard normals.
    np.random.normal(loc, scale) is loc + scale*gauss."""
    N = z0.shape[0]
    x0 = float(prior_mean) + np.sqrt(float(prior_var)) * z0
    if model == "garch":
        x = np.zeros((N, 2))
        x[:, 0] = x0
        return x
    return x0.reshape(N, 1)


def kernel_rv(model, kernel, d, x, y, z):
    """Kernel.rv: x (N,n) parents, y (1,) array, z (N,) normals -> x_next (N,n)."""
    if model == "svm":
        if kernel != "prior":
            raise NotImplementedError("SVM optimal kernel not analytic")  # svm/helper.py:62
        # svm/kernels.py:34-37
        return d["LQinv"] ** -1 * z[:, None] + x * d["A"]
    if model == "lgssm":
        if kernel == "prior":
            # lgssm/kernels.py:30-33
            return d["LQinv"] ** -1 * z[:, None] + x * d["A"]
        # lgssm/kernels.py:87-97
        mean_prec = x * d["A"] * d["Qinv"] + y * d["C"] * d["Rinv"]
        prec = d["Qinv"] + (d["C"] ** 2) * d["Rinv"]
        return (prec) ** -0.5 * z[:, None] + mean_prec / prec
    if model == "garch":
        N = x.shape[0]
        sigma2_next = d["alpha"] + d["beta"] * x[:, 0] ** 2 + d["gamma"] * x[:, 1]
        x_next = np.zeros((N, 2))
        if kernel == "prior":
            # garch/kernels.py:60-68
            x_next[:, 0] = np.sqrt(sigma2_next) * z
        else:
            # garch/kernels.py:146-156
            var_next = (d["Rinv"] + sigma2_next ** -1) ** -1
            mean_next = var_next * (y * d["Rinv"])
            x_next[:, 0] = mean_next + np.sqrt(var_next) * z
        x_next[:, 1] = sigma2_next
        return x_next
    raise ValueError(model)


def kernel_reweight(model, kernel, d, x, x_next, y):
    """Kernel.reweight -> (N,) log weights."""
    if model == "svm":
        # svm/kernels.py:56-62  (diff = y)
        lw = (-0.5 * LOG_2PI
              + -0.5 * (y ** 2) * np.exp(-x_next) * d["Rinv"]
              + np.log(d["LRinv"])
              + -0.5 * x_next)
        return lw.reshape(-1)
    if model == "lgssm":
        if kernel == "prior":
            # lgssm/kernels.py:58-62
            diff = y - d["C"] * x_next
            lw = (-0.5 * LOG_2PI + -0.5 * (diff ** 2) * d["Rinv"] + np.log(d["LRinv"]))
        else:
            # lgssm/kernels.py:117-120 (formula assumes C == 1)
            diff = y - d["A"] * x
            variance = d["Qinv"] ** -1 + d["Rinv"] ** -1
            lw = -0.5 * (diff) ** 2 / variance - 0.5 * LOG_2PI - 0.5 * np.log(variance)
        return lw.reshape(-1)
    if model == "garch":
        if kernel == "prior":
            # garch/kernels.py:83-88
            diff = y - x_next[:, 0]
            lw = (-0.5 * LOG_2PI + -0.5 * (diff ** 2) * d["Rinv"] + np.log(d["LRinv"]))
        else:
            # garch/kernels.py:172-178
            var = x_next[:, 1] + d["R"]
            lw = (-0.5 * LOG_2PI + -0.5 * (y ** 2) / var + -0.5 * np.log(var))
        return lw.reshape(-1)
    raise ValueError(model)


def score_statistic(model, d, x, x_next, y):
    """*_complete_data_loglike_gradient scalar branches -> (N,h)."""
    if model == "svm":
        # svm/helper.py:342-348
        diff_x = x_next - d["A"] * x
        grad_A = d["Qinv"] * diff_x * x
        grad_LQinv = (d["LQinv"] ** -1) - (diff_x ** 2) * d["LQinv"]
        diff_y2 = y ** 2 / np.exp(x_next)
        grad_LRinv = (d["LRinv"] ** -1) - (diff_y2) * d["LRinv"]
        return np.hstack([grad_LRinv, grad_LQinv, grad_A])
    if model == "lgssm":
        # lgssm/helper.py:1270-1277
        diff_x = x_next - d["A"] * x
        grad_A = d["Qinv"] * diff_x * x
        grad_LQinv = (d["LQinv"] ** -1) - (diff_x ** 2) * d["LQinv"]
        diff_y = y - d["C"] * x_next
        grad_C = d["Rinv"] * diff_y * x_next
        grad_LRinv = (d["LRinv"] ** -1) - (diff_y ** 2) * d["LRinv"]
        return np.hstack([grad_LRinv, grad_LQinv, grad_C, grad_A])
    if model == "garch":
        # garch/helper.py:350-370
        mu, phi, lam = d["mu"], d["phi"], d["lambduh"]
        v = x_next[:, 1]
        grad_v = -0.5 * (v - x_next[:, 0] ** 2) / (v ** 2)
        grad_log_mu = grad_v * (1 - phi) * mu
        grad_logit_phi = (grad_v *
                          (-mu + lam * x[:, 0] ** 2 + (1 - lam) * x[:, 1]) * (1 - phi) * phi)
        grad_logit_lambduh = (grad_v * phi * (x[:, 0] ** 2 - x[:, 1]) * (1 - lam) * lam)
        diff_y = y - x_next[:, 0]
        grad_LRinv = (d["LRinv"] ** -1) - (diff_y ** 2) * d["LRinv"]
        return np.array([grad_LRinv[0], grad_log_mu, grad_logit_phi, grad_logit_lambduh]).T
    raise ValueError(model)


def sufficient_statistic(model, x, x_next):
    """gaussian_sufficient_statistics (lgssm/helper.py:1360-1362) /
    garch_sufficient_statistics (garch/helper.py:429) -> (N,3)."""
    if model == "garch":
        return np.array([x_next[:, 0], x_next[:, 0] ** 2, x_next[:, 0] ** 4]).T
    return np.hstack([x_next, x_next ** 2, x * x_next])


def prior_log_density(model, d, x_t, x_next):
    """Kernel.prior_log_density: log q(x_next | x_t) row-wise -> (L,).
    kernels.py:102-126 (latent Gaussian, n = 1), garch/kernels.py:20-37."""
    L = np.shape(x_t)[0]
    if model == "garch":
        sigma2_next = d["alpha"] + d["beta"] * x_t[:, 0] ** 2 + d["gamma"] * x_t[:, 1]
        ll = -0.5 * x_next[:, 0] ** 2 / sigma2_next - 0.5 * LOG_2PI - 0.5 * np.log(sigma2_next)
        return np.reshape(ll, (L))
    diff = x_next - d["A"] * x_t
    ll = -0.5 * (diff ** 2) * d["Qinv"] + -0.5 * LOG_2PI + np.log(d["LQinv"])
    return np.reshape(ll, (L))


def prior_log_density_max(model, d):
    """Kernel.get_prior_log_density_max: kernels.py:128-138, garch/kernels.py:39-47."""
    if model == "garch":
        return -0.5 * LOG_2PI - 0.5 * np.log(d["alpha"])
    return -0.5 * 1 * LOG_2PI + np.sum(np.log(np.diag(d["LQinv"])))


class NpDraws(object):
    """Uniform draws in the REFERENCE's order from a legacy generator: np.random.choice(size=L)
    and np.random.rand(L) each take L doubles, choice(size=1) takes one."""

    def __init__(self, rng):
        self.rng = rng

    def index_uniforms(self, t, j, r, L):
        return self.rng.random_sample(len(L))

    def accept_uniforms(self, t, j, r, L):
        return self.rng.random_sample(len(L))

    def manual_uniform(self, t, j, i):
        return self.rng.random_sample(1)[0]

    def child_uniforms(self, t, i, Ntilde):
        return self.rng.random_sample(Ntilde)


class PoolDraws(object):
    """Uniform draws addressed by (timestep, j, round, particle): what the device kernel reads.
    idx_u, acc_u: [T, Ntilde, R, N]; man_u: [T, Ntilde, N]."""

    def __init__(self, idx_u, acc_u, man_u):
        self.idx_u, self.acc_u, self.man_u = idx_u, acc_u, man_u

    def index_uniforms(self, t, j, r, L):
        return self.idx_u[t, j, r, L]

    def accept_uniforms(self, t, j, r, L):
        return self.acc_u[t, j, r, L]

    def manual_uniform(self, t, j, i):
        return self.man_u[t, j, i]


def paris_backward_indices(model, d, x, logw, x_next, Ntilde, draws, t,
                           max_accept_reject=None, manual_sample_threshold=None):
    """accept_reject_based_backward_sampling (pf.py:260-341): J[N, Ntilde]."""
    N = np.shape(x)[0]
    weights = log_normalize(logw)
    ll_max = prior_log_density_max(model, d)
    if max_accept_reject is None:
        max_accept_reject = int(100 * np.log10(N / 10))
    if manual_sample_threshold is None:
        manual_sample_threshold = int(10 * np.log10(N / 10))
    J = np.zeros((N, Ntilde), dtype=int)
    for j in range(Ntilde):
        L = [ii for ii in range(N)]
        converged = False
        for r in range(max_accept_reject):
            size_L = len(L)
            if size_L == 0:
                converged = True
                break
            if size_L <= manual_sample_threshold:
                break
            indices = multinomial_ancestors(weights, draws.index_uniforms(t, j, r, L))
            uniforms = draws.accept_uniforms(t, j, r, L)
            child_ll = prior_log_density(model, d, x[indices], x_next[L])
            threshold = np.exp(child_ll - ll_max)
            new_L = []
            for k in range(size_L):
                if uniforms[k] <= threshold[k]:
                    J[L[k], j] = indices[k]
                else:
                    new_L.append(L[k])
            L = new_L
        if not converged:
            for i in L:
                child_ll = prior_log_density(model, d, x, np.outer(np.ones(N), x_next[i]))
                u = np.array([draws.manual_uniform(t, j, i)])
                J[i, j] = multinomial_ancestors(log_normalize(logw + child_ll), u)[0]
    return J


def predictive_statistic(model, d, x_next, t, y_all, num_steps_ahead, normals):
    """[log Pr(y_{t+k} | x_{t+1-ish})]_{k=0..K} per particle -> (N, K+1).
    svm_predictive_loglikelihood (svm/helper.py:352-395, Ntilde = 1),
    gaussian_predictive_loglikelihood (lgssm/helper.py:1281-1336, n = m = 1),
    garch_predictive_loglikelihood (garch/helper.py:374-412).
    `normals(k)` returns the N standard normals the reference draws at lead k (SVM: the Monte-Carlo
    sample, drawn even at k = 0 where its scale is 0; GARCH: prior_kernel.rv; LGSSM: none)."""
    N = x_next.shape[0]
    T = y_all.shape[0]
    K = num_steps_ahead
    out = np.zeros((N, K + 1))
    if model == "svm":
        x_pred_mean = x_next + 0.0
        x_pred_cov = 0.0
        R, Q = d["R"], d["Q"]
        for k in range(K + 1):
            if t + k >= T:
                break
            diff = y_all[t + k]
            x_mc = (np.outer(x_pred_mean, np.ones(1)) + np.sqrt(x_pred_cov) * normals(k).reshape(N, 1))
            y_pred_cov = R * np.exp(x_mc)
            out[:, k] = np.mean(-0.5 * diff ** 2 / y_pred_cov + -0.5 * LOG_2PI - 0.5 * np.log(y_pred_cov), axis=1)
            x_pred_mean = d["A"] * x_pred_mean
            x_pred_cov = Q + d["A"] ** 2 * x_pred_cov
    elif model == "lgssm":
        x_pred_mean = x_next + 0.0
        x_pred_cov = np.zeros((1, 1))
        R, Q = d["R"], d["Q"]
        for k in range(K + 1):
            if t + k >= T:
                break
            diff = (np.outer(np.ones(N), y_all[t + k]) - np.dot(x_pred_mean, d["C"].T))
            y_pred_cov = R + np.dot(d["C"], np.dot(x_pred_cov, d["C"].T))
            pl = -0.5 * diff ** 2 / y_pred_cov + -0.5 * LOG_2PI - 0.5 * np.log(y_pred_cov)
            out[:, k] = pl[:, 0]
            x_pred_mean = np.dot(x_pred_mean, d["A"].T)
            x_pred_cov = Q + np.dot(d["A"], np.dot(x_pred_cov, d["A"].T))
    else:
        x_pred = x_next + 0
        R = d["R"]
        for k in range(K + 1):
            if t + k >= T:
                break
            diff = np.ones(N) * y_all[t + k] - x_pred[:, 0]
            y_pred_cov = R
            pl = -0.5 * diff ** 2 / y_pred_cov + -0.5 * LOG_2PI - 0.5 * np.log(y_pred_cov)
            out[:, k] = pl
            x_pred = kernel_rv("garch", "prior", d, x_pred, None, normals(k))
    return out


STAT_DIM = {
    ("svm", "score"): 3, ("lgssm", "score"): 4, ("garch", "score"): 4,
    ("svm", "suff"): 3, ("lgssm", "suff"): 3, ("garch", "suff"): 3,
}


# --------------------------------------------------------------------------
# RNG stream (order verified against the reference: finding 1 of SURVEY.md)
# --------------------------------------------------------------------------
def draw_streams(rng, N, T):
    """Consume the legacy stream as one PF run of the reference does: N normals (x0),
    then per timestep N uniforms (np.random.choice) followed by N normals (Kernel.rv).
    `rng` is `np.random` (global legacy state) or a `np.random.RandomState`."""
    z0 = rng.normal(size=N)
    u = np.empty((T, N))
    z = np.empty((T, N))
    for t in range(T):
        u[t] = rng.random_sample(N)
        z[t] = rng.normal(size=N)
    return z0, u, z


# --------------------------------------------------------------------------
# the T-loop (buffered_smoother.py:12-149) on pre-drawn streams
# --------------------------------------------------------------------------
def pf_window(model, theta, y, N, z0, u, z, kernel=None, pf="poyiadjis_N",
              lambduh=None, stat="score", t1=0, tL=None, weights=None,
              prior_mean=0.0, prior_var=1.0, save_all=False,
              Ntilde=2, max_accept_reject=None, manual_sample_threshold=None, paris_draws=None,
              elementwise_statistic=False, num_steps_ahead=5, pred_normals=None, resampler=None,
              accept_reject=True):
    """One buffered PF window.

    Args:
      model: 'svm' | 'garch' | 'lgssm';  theta: raw parameters (THETA_NAMES order)
      y: (T,) or (T,1) observations of the buffered window
      z0 (N,), u (T,N), z (T,N): the random streams (see draw_streams)
      pf: 'poyiadjis_N' (lambda=1), 'nemeth' (default lambda .95), 'filter'
      stat: 'score' | 'suff' | 'none'
      resampler: None = the reference's np.random.choice semantics on u[t]; else a callable
          (t, logw) -> ancestors (tests of the device-generator kernels: `device_ancestors` on the
          words the launch recorded), u is then unused
    Returns dict(x_t, log_weights, statistics, loglikelihood_estimate[, mean_statistic, all_*])
    """
    y = np.asarray(y, dtype=float).reshape(-1, 1)   # y[t] is a (1,) array as in the reference
    T = y.shape[0]
    if tL is None:
        tL = T
    if kernel is None:
        kernel = DEFAULT_KERNEL[model]
    if pf == "poyiadjis_N":
        lambduh = 1.0
    elif pf == "nemeth":
        lambduh = 0.95 if lambduh is None else lambduh
    elif pf == "paris":
        if paris_draws is None:
            raise ValueError("pf='paris' needs paris_draws (NpDraws or PoolDraws)")
    elif pf == "poyiadjis_N2":
        pass
    elif pf != "filter":
        raise ValueError("Unrecognized pf = {0}".format(pf))
    is_filter = (pf == "filter")
    is_paris = (pf == "paris")
    is_n2 = (pf == "poyiadjis_N2")
    d = derived(model, theta)
    if model == "svm" and abs(d["A"]) > 1:
        raise ValueError("Current AR parameter is |A| = {0} > 1".format(abs(d["A"])))
    if stat == "predictive":
        # pf_predictive_loglikelihood_estimate: filter + logsumexp accumulation (pf.py:72-76)
        if pf != "filter":
            raise ValueError("Only can use pf = 'filter' since we are filtering")
        h = num_steps_ahead + 1
    else:
        h = 3 if stat == "none" else STAT_DIM[(model, stat)]
    h_base = h
    if elementwise_statistic:
        # elementwise_statistic_wrapper (buffered_smoother.py:64-65, 201-210): one h-block per
        # window timestep, the block of step t sits at offset (t - t1) * h
        h = (tL - t1) * h_base

    x = sample_x0(model, prior_mean, prior_var, z0)
    logw = np.zeros(N)
    loglik = 0.0
    stats = np.zeros(h) if is_filter else np.zeros((N, h))
    if save_all:
        all_x, all_lw, all_s, all_ll, all_anc = [x], [logw], [stats], [loglik], []

    def widen(add, t):
        """elementwise_statistic_wrapper (buffered_smoother.py:201-210): the h_base columns of step t
        sit at offset (t - t1) * h_base of the wide statistic (any number of rows)."""
        if not elementwise_statistic:
            return add
        wide = np.zeros((add.shape[0], h))
        wide[:, (t - t1) * h_base:(t - t1 + 1) * h_base] = add
        return wide

    for t in range(T):
        inside = (t >= t1) and (t < tL)
        weight_t = 1.0
        if inside and weights is not None:
            weight_t = float(weights[t - t1])

        if not is_filter and not is_paris and not is_n2:
            # nemeth_smoother: S from the *previous* weights (pf.py:161)
            S = np.sum(stats.T * log_normalize(logw), axis=1)
        # pf(): resample every step, propose, weight (pf.py:26-38)
        if resampler is None:
            anc = multinomial_ancestors(log_normalize(logw), u[t])
        else:
            anc = resampler(t, logw)
        if save_all:
            all_anc.append(anc)
        parents = x[anc]
        x_next = kernel_rv(model, kernel, d, parents, y[t], z[t])
        new_logw = kernel_reweight(model, kernel, d, parents, x_next, y[t])

        if is_n2:
            # poyiadjis_smoother (pf.py:84-136): every child averages over ALL parents with the
            # backward weights  w_j q(child | x_j)  (normalised per child)
            bw = np.zeros((N, N))
            for i in range(N):
                child_ll = prior_log_density(model, d, x, np.outer(np.ones(N), x_next[i]))
                bw[i] = log_normalize(logw + child_ll)
            idx = np.array([ii for _ in range(N) for ii in range(N)])
            new_idx = np.array([ii for ii in range(N) for _ in range(N)])
            if inside and stat == "score":
                add = widen(score_statistic(model, d, x[idx], x_next[new_idx], y[t]), t)
            elif inside and stat == "suff":
                add = widen(sufficient_statistic(model, x[idx], x_next[new_idx]), t)
            else:
                add = np.zeros((N * N, h))
            add = add * weight_t
            stats = np.einsum('ijk,ij->ik', np.reshape(stats[idx] + add, (N, N, -1)), bw)
            x, logw = x_next, new_logw
            if inside:
                loglik += weight_t * np.log(np.mean(np.exp(logw)))
            if save_all:
                all_x.append(x); all_lw.append(logw); all_s.append(stats); all_ll.append(loglik)
            continue
        if is_paris:
            # paris_smoother (pf.py:183-258): rewire Ntilde backward-sampled parents per child
            if accept_reject:
                J = paris_backward_indices(model, d, x, logw, x_next, Ntilde, paris_draws, t,
                                           max_accept_reject, manual_sample_threshold)
            else:
                # paris_smoother(accept_reject=False), pf.py:226-236: the naive O(N^2) draw -- every child takes its
                # Ntilde parents from the exact backward categorical, np.random.choice(size=Ntilde) per child
                J = np.zeros((N, Ntilde), dtype=int)
                for i in range(N):
                    child_ll = prior_log_density(model, d, x, np.outer(np.ones(N), x_next[i]))
                    J[i] = multinomial_ancestors(log_normalize(logw + child_ll), paris_draws.child_uniforms(t, i, Ntilde))
            flat = J.flatten()
            rew_parents = x[flat]
            xi_next = x_next[np.array([ii for ii in range(N) for _ in range(Ntilde)])]
            if inside and stat == "score":
                add = widen(score_statistic(model, d, rew_parents, xi_next, y[t]), t)
            elif inside and stat == "suff":
                add = widen(sufficient_statistic(model, rew_parents, xi_next), t)
            else:
                add = np.zeros((N * Ntilde, h))
            add = add * weight_t
            stats = np.mean(np.reshape(stats[flat] + add, (N, Ntilde, -1)), axis=1)
            x, logw = x_next, new_logw
            if inside:
                loglik += weight_t * np.log(np.mean(np.exp(logw)))
            if save_all:
                all_x.append(x); all_lw.append(logw); all_s.append(stats); all_ll.append(loglik)
            continue
        if inside and stat == "predictive":
            add = predictive_statistic(model, d, x_next, t, y, num_steps_ahead, lambda k: pred_normals(t, k))
        elif inside and stat == "score":
            add = score_statistic(model, d, parents, x_next, y[t])
        elif inside and stat == "suff":
            add = sufficient_statistic(model, parents, x_next)
        else:
            add = np.zeros((N, h))      # zero_statistics (buffered_smoother.py:77-79)
        if inside and stat != "none":
            add = widen(add, t)
        add = add * weight_t            # additive_scale

        if is_filter and stat == "predictive":
            # pf.py:72-76 (logsumexp=True).  NB the reference's np.sum has NO axis: the sum runs
            # over every lead k AND every particle, and its log is added to every column (so
            # steps outside the window, where add = 0, add log(K+1)).  Reproduced as is.
            max_add = np.max(add.T, axis=1)
            stats = stats + max_add + np.log(np.sum(
                np.exp(add.T - max_add[:, np.newaxis]) * log_normalize(new_logw)))
        elif is_filter:
            # pf.py:78-80, new weights
            stats = stats + np.sum(add.T * log_normalize(new_logw), axis=1)
        else:
            # pf.py:175-179
            stats = (lambduh * stats[anc]
                     + (1.0 - lambduh) * np.outer(np.ones(N), S)
                     + add)
        x, logw = x_next, new_logw
        if inside:
            # buffered_smoother.py:124-126 (not max-stabilised in the reference)
            loglik += weight_t * np.log(np.mean(np.exp(logw)))
        if save_all:
            all_x.append(x); all_lw.append(logw); all_s.append(stats); all_ll.append(loglik)

    out = dict(x_t=x, log_weights=logw, statistics=stats,
               loglikelihood_estimate=loglik)
    if not is_filter:
        # average_statistic (buffered_smoother.py:151-154)
        out["mean_statistic"] = np.sum(stats.T * log_normalize(logw), axis=1)
    if save_all:
        out["all_x_t"] = np.array(all_x)
        out["all_log_weights"] = np.array(all_lw)
        out["all_statistics"] = np.array(all_s)
        out["all_loglikelihood_estimate"] = np.array(all_ll)
        out["all_ancestors"] = np.array(all_anc, dtype=int).reshape(-1, N)
    return out


def latent_var_distr(model, theta, y, N, rng=np.random, **kw):
    """Helper.pf_latent_var_distr (svm/helper.py:249-294, lgssm/helper.py:1145-1198,
    garch/helper.py:274-318) for smoothing (lag=None): elementwise sufficient statistics,
    averaged -> (x_mean (L,1), x_cov (L,1,1))."""
    squared = kw.pop("squared", False)
    if kw.get("pf") == "paris":
        kw.pop("pf")
        out = pf_window_paris_rng(model, theta, y, N, rng=rng, stat="suff", elementwise_statistic=True, **kw)
    else:
        out = pf_window_rng(model, theta, y, N, rng=rng, stat="suff", elementwise_statistic=True, **kw)
    avg = np.reshape(out["mean_statistic"], (-1, 3))
    if model == "garch" and squared:
        x_mean = avg[:, 1]
        x_cov = avg[:, 2] - x_mean ** 2
    else:
        x_mean = avg[:, 0]
        x_cov = avg[:, 1] - x_mean ** 2
    return np.reshape(x_mean, (x_mean.shape[0], 1)), np.reshape(x_cov, (x_cov.shape[0], 1, 1))


def pf_window_rng(model, theta, y, N, rng=np.random, **kw):
    """pf_window drawing its streams from `rng` in the reference's order; the drop-in
    equivalent of Helper.pf_gradient_estimate's inner buffered_pf_wrapper call.
    (Not for pf='paris': its draws interleave data-dependently with the filter's; use
    pf_window_paris_rng.)"""
    T = np.asarray(y).reshape(-1).shape[0]
    z0, u, z = draw_streams(rng, N, T)
    return pf_window(model, theta, y, N, z0, u, z, **kw)


class _LazyStreams(object):
    """u[t] / z[t] drawn from the generator at first access, so that a PaRIS run consumes the
    legacy stream in the reference's interleaved order: per timestep N uniforms, N normals
    (pf()), then the backward-sampling draws."""

    def __init__(self, rng, N, kind):
        self.rng, self.N, self.kind, self.cache = rng, N, kind, {}

    def __getitem__(self, t):
        if t not in self.cache:
            self.cache[t] = (self.rng.random_sample(self.N) if self.kind == "u"
                             else self.rng.normal(size=self.N))
        return self.cache[t]


def pf_window_paris_rng(model, theta, y, N, rng=np.random, **kw):
    """PaRIS window consuming `rng` exactly as the reference does."""
    z0 = rng.normal(size=N)
    return pf_window(model, theta, y, N, z0, _LazyStreams(rng, N, "u"), _LazyStreams(rng, N, "z"),
                     pf="paris", paris_draws=NpDraws(rng), **kw)


def pf_gradient_estimate(model, theta, y, N, rng=np.random, **kw):
    """Helper.pf_gradient_estimate -> dict keyed like the reference (SCORE_NAMES)."""
    out = pf_window_rng(model, theta, y, N, rng=rng, stat="score", **kw)
    return dict(zip(SCORE_NAMES[model], out["mean_statistic"]))


def pf_loglikelihood_estimate(model, theta, y, N, rng=np.random, **kw):
    """Helper.pf_loglikelihood_estimate."""
    out = pf_window_rng(model, theta, y, N, rng=rng, stat="suff", **kw)
    return out["loglikelihood_estimate"]


def pf_predictive_loglikelihood_estimate(model, theta, y, N, rng=np.random, num_steps_ahead=5, **kw):
    """Helper.pf_predictive_loglikelihood_estimate (svm/helper.py:187-247, garch/helper.py:172-231,
    lgssm/helper.py:1048-1087), consuming `rng` in the reference's order: per timestep N uniforms,
    N normals (pf()), then inside the window N normals per lead k (SVM / GARCH)."""
    z0 = rng.normal(size=N)
    out = pf_window(model, theta, y, N, z0, _LazyStreams(rng, N, "u"), _LazyStreams(rng, N, "z"), pf="filter",
                    stat="predictive", num_steps_ahead=num_steps_ahead,
                    pred_normals=(lambda t, k: rng.normal(size=N)) if model != "lgssm" else (lambda t, k: None),
                    **kw)
    pred = np.array(out["statistics"], dtype=float)
    pred[0] = out["loglikelihood_estimate"]
    return pred


def garch_prior_x(theta):
    """garch/helper.py:324-332 with forward_message=None."""
    d = derived("garch", theta)
    return 0.0, d["alpha"] / (1 - d["beta"] - d["gamma"])
