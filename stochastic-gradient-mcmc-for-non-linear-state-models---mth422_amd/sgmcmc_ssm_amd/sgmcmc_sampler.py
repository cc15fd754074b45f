"""SG-MCMC samplers whose noisy gradient comes from the HIP particle filter.

Host-side orchestration of the reference's `sgmcmc_ssm/sgmcmc_sampler.py` for `kind='pf'`:
window sampling, importance weights, prior gradient, 1/T scaling, the SGD / ADAGRAD / SGLD
parameter updates and the fit loops.  All of it costs O(#parameters) per step and stays in
Python; everything O(N*T) runs in libpfgrad.so.

Every window a gradient needs (minibatch_size windows x num_sequences sequences) is collected
first -- drawing from `np.random` in exactly the reference's order -- and then issued as ONE
kernel launch with one workgroup per window.

Behaviour kept from the reference on purpose (SURVEY.md 8b "quirks"):
  * kind='pf' always differentiates at `self.parameters`, even if `parameters=` is passed
    (sgmcmc_sampler.py:379).
  * SGLD noise is drawn after the particle filter has consumed its draws, one
    np.random.normal call per variable in `parameters.as_dict()` order (:540-546).
Consciously changed: NaN / blow-up checks look at every gradient entry, not only the last
dict key (:420-424 leaks the loop variable).
"""
import logging
import time

import numpy as np

from . import _capi
from . import particle_filters as _pf

logger = logging.getLogger(name=__name__)
NOISE_NUGGET = 1e-9

_ONLY_PF = ("only kind='pf' is implemented by the MI355X backend; kind='{0}' (analytic / "
            "complete-data paths of the reference) is out of scope (SURVEY.md section 8)")


# ----------------------------------------------------------------------------------------
# window sampling (sgmcmc_sampler.py:1969-2017)
# ----------------------------------------------------------------------------------------
def random_subsequence_and_weights(S, T, partition_style=None, random_state=None):
    """Draw a length-S window of a length-T series and the importance weights that make the
    windowed sum unbiased for the full sum.  Consumes one np.random integer draw.

    'uniform' (default): start ~ U{0..T-S}; weight_t = (T-S+1) / #windows covering t.
    'strict': T % S == 0, start in {0,S,2S,..}; weights T/S.   'naive': weights T/S.
    Returns (start, end, weights[S])."""
    style = 'uniform' if partition_style is None else partition_style
    rs = np.random if random_state is None else random_state
    if style == 'strict':
        if T % S != 0:
            raise ValueError("S {0} does not evenly divide T {1}".format(S, T))
        start = rs.choice(np.arange(0, T // S)) * S
        weights = np.ones(S, dtype=float) * T / S
    elif style == 'uniform':
        start = rs.randint(0, T - S + 1)
        t = np.arange(start, start + S)
        cap = np.ones_like(t) * min(S, T - S + 1)
        if start + S <= 2 * S:
            covering = np.min(np.array([t + 1, cap]), axis=0)
        elif start >= T - 2 * S - 1:
            covering = np.min(np.array([T - t, cap]), axis=0)
        else:
            covering = np.ones(S) * S
        weights = np.ones(S, dtype=float) * (T - S + 1) / covering
    elif style == 'naive':
        start = rs.randint(0, T - S + 1)
        weights = np.ones(S, dtype=float) * T / S
    else:
        raise ValueError("Unrecognized partition_style = '{0}'".format(style))
    return int(start), int(start + S), weights


# ----------------------------------------------------------------------------------------
# Helper: per-model particle-filter estimators (the hot-path entry)
# ----------------------------------------------------------------------------------------
class PFHelper(object):
    """Base of SVMHelper / GARCHHelper / LGSSMHelper: `pf_gradient_estimate` and
    `pf_loglikelihood_estimate` with the reference's signatures
    (models/svm/helper.py:67-185, models/garch/helper.py:59-170, models/lgssm/helper.py:1016-1143)."""
    model = None                 # 'svm' | 'garch' | 'lgssm'
    default_kernel = None
    kernels = ()
    score_names = ()             # dict keys of the gradient, in statistic-column order

    def __init__(self, n=1, m=1, forward_message=None, backward_message=None, **kwargs):
        if n != 1 or m != 1:
            raise NotImplementedError("the MI355X particle-filter backend covers n = m = 1 models")
        self.n, self.m = n, m
        if forward_message is None:
            forward_message = self._default_forward_message()
        self.default_forward_message = forward_message
        if backward_message is None:
            backward_message = dict(log_constant=0.0, mean_precision=np.zeros(self.n),
                                    precision=np.zeros((self.n, self.n)))
        self.default_backward_message = backward_message

    def _default_forward_message(self):
        return dict(log_constant=0.0, mean_precision=np.zeros(self.n), precision=np.eye(self.n) / 10)

    def _get_kernel(self, kernel):
        if kernel is None:
            kernel = self.default_kernel
        if kernel not in ("prior", "optimal"):
            raise ValueError("Unrecoginized kernel = {0}".format(kernel))
        if kernel not in self.kernels:
            raise NotImplementedError("{0} {1} kernel not analytic".format(self.model.upper(), kernel))
        return kernel

    def _prior_x(self, forward_message, parameters):
        """Prior mean / variance of x_{-1} from a Gaussian message.  The reference computes
        prior_mean = solve(prior_var, mean_precision) (svm/helper.py:99-104) -- only right
        because mean_precision == 0 in practice; reproduced as is."""
        if forward_message is None:
            forward_message = self.default_forward_message
        prior_var = np.linalg.inv(forward_message['precision'])
        prior_mean = np.linalg.solve(prior_var, forward_message['mean_precision'])
        return float(np.reshape(prior_mean, -1)[0]), float(prior_var[0, 0]), 0

    def pf_problem(self, observations, parameters, subsequence_start=0, subsequence_end=None,
                   weights=None, pf="poyiadjis_N", N=1000, kernel=None, forward_message=None,
                   stat="score", **kwargs):
        """One window as a libpfgrad problem (draws its replay streams now)."""
        kernel = self._get_kernel(kernel)
        prior_mean, prior_var, flags = self._prior_x(forward_message, parameters)
        return _pf.make_problem(self.model, kernel, pf, observations, parameters.theta(), N,
                                t1=subsequence_start, tL=subsequence_end, weights=weights,
                                prior_mean=prior_mean, prior_var=prior_var, stat=stat, flags=flags,
                                **kwargs)

    def grad_from_statistic(self, mean_statistic):
        return dict(zip(self.score_names, mean_statistic))

    def pf_gradient_estimate(self, observations, parameters, subsequence_start=0,
                             subsequence_end=None, weights=None, pf="poyiadjis_N", N=1000,
                             kernel=None, forward_message=None, **kwargs):
        """Particle-filter score estimate -> dict of gradients keyed like parameters.var_dict."""
        q = self.pf_problem(observations, parameters, subsequence_start, subsequence_end, weights,
                            pf, N, kernel, forward_message, stat="score", **kwargs)
        if q["smoother"] == "filter":
            raise ValueError("pf = 'filter' has no per-particle statistics to average "
                             "(the reference fails in average_statistic for it)")
        out = _pf.run_windows([q], ctx=kwargs.get("ctx", None))[0]
        return self.grad_from_statistic(out["mean_statistic"])

    def pf_loglikelihood_estimate(self, observations, parameters, subsequence_start=0,
                                  subsequence_end=None, weights=None, pf="poyiadjis_N", N=1000,
                                  kernel=None, forward_message=None, **kwargs):
        """Particle-filter marginal log-likelihood estimate (the sufficient statistics the
        reference accumulates alongside are not returned by it either)."""
        q = self.pf_problem(observations, parameters, subsequence_start, subsequence_end, weights,
                            pf, N, kernel, forward_message, stat="suff", **kwargs)
        return _pf.run_windows([q], ctx=kwargs.get("ctx", None))[0]["loglikelihood_estimate"]

    def pf_predictive_loglikelihood_estimate(self, observations, parameters, num_steps_ahead=5,
                                             subsequence_start=0, subsequence_end=None, pf="filter",
                                             N=1000, kernel=None, forward_message=None, **kwargs):
        """Particle-filter predictive log-likelihoods for leads k = 0..num_steps_ahead
        (svm/helper.py:187-247, garch/helper.py:172-231, lgssm/helper.py:1048-1087): the filter with
        the k-step-ahead statistic folded in by the reference's logsumexp update (pf.py:72-76,
        including its axis-less sum); entry 0 is replaced by the log-likelihood estimate."""
        if pf != "filter":
            raise ValueError("Only can use pf = 'filter' since we are filtering")
        q = self.pf_problem(observations, parameters, subsequence_start, subsequence_end, None,
                            pf, N, kernel, forward_message, stat="predictive",
                            num_steps_ahead=num_steps_ahead, **kwargs)
        out = _pf.run_windows([q], ctx=kwargs.get("ctx", None))[0]
        pred = np.array(out["statistics"], dtype=float)
        pred[0] = out["loglikelihood_estimate"]
        return pred


    def pf_latent_var_distr(self, observations, parameters, lag=None, subsequence_start=0,
                            subsequence_end=None, weights=None, pf="poyiadjis_N", N=1000, kernel=None,
                            forward_message=None, squared=False, **kwargs):
        """Smoothed marginals of the latent state on [subsequence_start, subsequence_end):
        (x_mean (L,1), x_cov (L,1,1)) -- svm/helper.py:249-294, lgssm/helper.py:1145-1198,
        garch/helper.py:274-318: the reference's `elementwise_statistic=True` run, i.e. one block of
        sufficient statistics per window timestep carried through the smoother's recursion.  On the GPU
        the filter runs once (recording particles, log-weights and every child's parent(s)) and a second
        device pass streams the [N, 3 L] statistic matrix through the recursion (pfg_problem.elementwise:
        every smoother the reference dispatches: 'poyiadjis_N', 'nemeth', 'paris', 'poyiadjis_N2').  lag=None only: the reference's lag=0 / pf='filter'
        branch fails in average_statistic (shape mismatch)."""
        if lag == 0 and pf != 'filter':
            raise ValueError("pf must be filter for lag = 0")
        elif lag is None and pf == 'filter':
            raise ValueError("pf must not be filter for smoothing")
        elif lag is not None and lag != 0:
            raise NotImplementedError("lag can only be None or 0")
        if pf not in ("poyiadjis_N", "nemeth", "paris", "poyiadjis_N2"):
            raise ValueError("Unrecognized pf = {0}".format(pf))
        kwargs.pop("tqdm", None)
        q = self.pf_problem(observations, parameters, subsequence_start, subsequence_end, weights,
                            pf, N, kernel, forward_message, stat="none", **kwargs)
        if "_result" in q:
            # pf='paris' in np.random's order: the window has just run step by step (that fixed how many uniforms
            # each backward sampling consumed and left the generator where the reference leaves it); the same
            # window once more in one launch, on the same numbers, with the elementwise pass behind it
            q2 = _pf.paris_replay_again(q)
            o = (kwargs.get("ctx", None) or _capi.default_context()).run_batch([q2], want_elementwise=True)[0]
            expected = q2.get("_consumed", q2["paris_stream"].shape[0])
            if o["paris_consumed"] != expected:
                raise RuntimeError("PaRIS replay consumed {0} of {1} uniforms".format(o["paris_consumed"], expected))
        else:
            o = (kwargs.get("ctx", None) or _capi.default_context()).run_batch([q], want_elementwise=True)[0]
        _pf._recycle_streams([q])
        avg = np.reshape(o["ew_mean"], (-1, 3))
        if self.model == "garch" and squared:
            x_mean, x_cov = avg[:, 1], avg[:, 2] - avg[:, 1] ** 2
        else:
            x_mean, x_cov = avg[:, 0], avg[:, 1] - avg[:, 0] ** 2
        return np.reshape(x_mean, (x_mean.shape[0], 1)), np.reshape(x_cov, (x_cov.shape[0], 1, 1))


# ----------------------------------------------------------------------------------------
# Sampler
# ----------------------------------------------------------------------------------------
class SGMCMCSampler(object):
    """Base class: SG-MCMC for one time series with particle-filter gradients."""

    def __init__(self, **kwargs):
        raise NotImplementedError()

    # -- init ------------------------------------------------------------------------------
    def prior_init(self):
        self.parameters = self.prior.sample_prior()
        return self.parameters

    # -- observations ------------------------------------------------------------------------
    @property
    def observations(self):
        return self._observations

    @observations.setter
    def observations(self, observations):
        self._check_observation_shape(observations)
        self._observations = observations

    def _check_observation_shape(self, observations):
        return

    def _get_observations(self, observations, check_shape=True):
        if observations is None:
            observations = self.observations
            if observations is None:
                raise ValueError("observations not specified")
        elif check_shape:
            self._check_observation_shape(observations)
        return observations

    def _get_T(self, **kwargs):
        T = kwargs.get('T')
        if T is None:
            observations = self._get_observations(kwargs.get('observations'))
            T = observations.shape[0]
        return T

    # -- windows -------------------------------------------------------------------------------
    def _random_subsequence_and_buffers(self, buffer_length, subsequence_length, T=None, random_state=None):
        """Window [start,end) plus up to buffer_length extra points each side
        (sgmcmc_sampler.py:259-288)."""
        if T is None:
            T = self._get_T()
        if buffer_length == -1:
            buffer_length = T
        if (subsequence_length == -1) or (T - subsequence_length <= 0):
            start, end, weights = 0, T, None
        else:
            start, end, weights = random_subsequence_and_weights(
                S=subsequence_length, T=T, partition_style=self.options.get('partition_style'),
                random_state=random_state)
        return dict(subsequence_start=start, subsequence_end=end,
                    left_buffer_start=max(0, start - buffer_length),
                    right_buffer_end=min(T, end + buffer_length), weights=weights)

    def _window_problem(self, buffer_dict, observations, stat, **kwargs):
        rel_start = buffer_dict['subsequence_start'] - buffer_dict['left_buffer_start']
        rel_end = buffer_dict['subsequence_end'] - buffer_dict['left_buffer_start']
        buffer_ = observations[buffer_dict['left_buffer_start']:buffer_dict['right_buffer_end']]
        return self.message_helper.pf_problem(
            observations=buffer_, parameters=self.parameters, subsequence_start=rel_start,
            subsequence_end=rel_end, weights=buffer_dict['weights'], stat=stat, **kwargs)

    @staticmethod
    def _require_pf(kind):
        if kind != 'pf':
            raise NotImplementedError(_ONLY_PF.format(kind))

    # -- log-likelihood --------------------------------------------------------------------------
    def _loglike_problems(self, kind='pf', subsequence_length=-1, minibatch_size=1, buffer_length=10,
                          num_samples=None, observations=None, parameters=None, check_shape=True,
                          **kwargs):
        self._require_pf(kind)
        observations = self._get_observations(observations, check_shape=check_shape)
        if kwargs.get("N", None) is None:
            kwargs['N'] = num_samples if num_samples is not None else 1000
        T = observations.shape[0]
        probs = []
        for _ in range(minibatch_size):
            # the reference draws a window and runs its filter inside one loop body (:210-237)
            bd = self._random_subsequence_and_buffers(buffer_length=buffer_length,
                                                      subsequence_length=subsequence_length, T=T)
            probs.append((self._window_problem(bd, observations, "suff", **kwargs), minibatch_size))
        return probs

    def predictive_loglikelihood(self, kind='pf', num_steps_ahead=10, subsequence_length=-1,
                                 minibatch_size=1, buffer_length=10, num_samples=1000,
                                 parameters=None, observations=None, **kwargs):
        """Predictive log-likelihoods [sum_t log Pr(y_{t+k} | y_{<=t})]_{k=0..num_steps_ahead}
        by particle filter (sgmcmc_sampler.py:50-128, kind='pf' branch; as there, the filter runs at
        self.parameters whatever `parameters` is)."""
        self._require_pf(kind)
        observations = self._get_observations(observations, check_shape=kwargs.pop('check_shape', True))
        T = observations.shape[0]
        if kwargs.get("N", None) is None:
            kwargs['N'] = num_samples
        pred_loglikelihood = np.zeros(num_steps_ahead + 1)
        for _ in range(minibatch_size):
            out = self._random_subsequence_and_buffers(buffer_length=buffer_length,
                                                      subsequence_length=subsequence_length, T=T)
            relative_start = out['subsequence_start'] - out['left_buffer_start']
            relative_end = out['subsequence_end'] - out['left_buffer_start']
            buffer_ = observations[out['left_buffer_start']:out['right_buffer_end']]
            add = self.message_helper.pf_predictive_loglikelihood_estimate(
                observations=buffer_, parameters=self.parameters, num_steps_ahead=num_steps_ahead,
                subsequence_start=relative_start, subsequence_end=relative_end, **kwargs)
            for ll in range(num_steps_ahead + 1):
                pred_loglikelihood[ll] += add[ll] * (T - ll) / (
                    out['subsequence_end'] - out['subsequence_start'] - ll)
        pred_loglikelihood *= 1.0 / minibatch_size
        return pred_loglikelihood

    def noisy_loglikelihood(self, **kwargs):
        """Subsequence approximation to the log-likelihood (kind='pf')."""
        kwargs.pop('tqdm', None)
        probs = self._loglike_problems(**kwargs)
        outs = _pf.run_windows([q for q, _ in probs])
        value = 0.0
        for o in outs:
            value += o["loglikelihood_estimate"]
        value *= 1.0 / probs[0][1]
        if np.isnan(value):
            raise ValueError("NaNs in loglikelihood")
        return value

    def noisy_logjoint(self, return_loglike=False, **kwargs):
        loglikelihood = self.noisy_loglikelihood(**kwargs)
        logprior = self.prior.logprior(self.parameters)
        if return_loglike:
            return dict(logjoint=loglikelihood + logprior, loglikelihood=loglikelihood)
        return loglikelihood + logprior

    # -- gradient ------------------------------------------------------------------------------------
    def _grad_problems(self, subsequence_length=-1, minibatch_size=1, buffer_length=0,
                       observations=None, buffer_dicts=None, kind='pf', num_samples=None,
                       parameters=None, **kwargs):
        """All windows of one gradient for ONE series -> [(problem, minibatch_size)].  RNG order
        as sgmcmc_sampler.py:390-418: every window is drawn first, then each filter's streams."""
        self._require_pf(kind)
        observations = self._get_observations(observations, check_shape=False)
        if kwargs.get("N", None) is None:
            kwargs['N'] = num_samples if num_samples is not None else 1000
        T = observations.shape[0]
        if buffer_dicts is None:
            buffer_dicts = [self._random_subsequence_and_buffers(
                buffer_length=buffer_length, subsequence_length=subsequence_length, T=T)
                for _ in range(minibatch_size)]
        elif len(buffer_dicts) != minibatch_size:
            raise ValueError("len(buffer_dicts != minibatch_size")
        probs = [(self._window_problem(bd, observations, "score", **kwargs), minibatch_size)
                 for bd in buffer_dicts]
        for q, _ in probs:
            q["_series_length"] = T          # host-side metadata (Seq rescaling); not sent to the device
        return probs

    def _run_grad_problems(self, groups):
        """groups: list (one per series) of [(problem, minibatch_size)].  One launch for all
        windows; sums in the reference's order (within a series, then across series)."""
        names = self.message_helper.score_names
        flat = [q for group in groups for q, _ in group]
        for q in flat:
            if q["smoother"] == "filter":
                raise ValueError("pf = 'filter' cannot be used for gradients")
        self._speculate_next_stream(flat)
        outs = iter(_pf.run_windows(flat))
        grad = None
        for group in groups:
            part = {var: np.zeros_like(value) for var, value in self.parameters.as_dict().items()}
            for _, minibatch_size in group:
                o = next(outs)
                for name, g in zip(names, o["mean_statistic"]):
                    part[name] += g * 1.0 / minibatch_size
            grad = part if grad is None else {var: grad[var] + part[var] for var in part}
        for var in grad:
            if np.any(np.isnan(grad[var])):
                raise ValueError("NaNs in gradient of {0}".format(var))
            if np.linalg.norm(grad[var]) > 1e16:
                logger.warning("Norm of noisy_grad_loglike[{1} > 1e16: {0}".format(grad[var], var))
        return grad

    def _noisy_grad_loglikelihood(self, **kwargs):
        self._last_grad_kwargs = kwargs
        return self._run_grad_problems([self._grad_problems(**kwargs)])

    # what the step functions draw from np.random between this gradient and the next one's streams; set by
    # sample_sgld / step_sgd / step_adagrad around their noisy_gradient call, None = do not speculate
    _rng_after_gradient = None

    def _speculate_next_stream(self, flat):
        """While the GPU runs this window, prefetch the next one's replay stream on a worker thread
        (particle_filters.speculation): same step type, same kwargs assumed; adopted only if np.random is
        then bit for bit where the clone was, so a wrong guess costs nothing but the worker's time."""
        after = self._rng_after_gradient
        kw = getattr(self, "_last_grad_kwargs", None)
        if (after is None or kw is None or len(flat) != 1 or flat[0].get("rng") != "replay"
                or "_stream_bufs" not in flat[0] or type(self)._grad_groups is not SGMCMCSampler._grad_groups
                or kw.get("buffer_dicts") is not None or kw.get("minibatch_size", 1) != 1):
            return
        N = int(flat[0]["N"])
        T = self._get_observations(kw.get("observations"), check_shape=False).shape[0]
        S, B = kw.get("subsequence_length", -1), kw.get("buffer_length", 0)
        if N * int(flat[0]["y"].shape[0]) < _pf._SPECULATE_MIN:
            return                                    # short windows: not worth a thread
        shapes = [np.shape(v) for v in self.parameters.as_dict().values()] if after == "noise" else []

        def between(rs):
            for shp in shapes:                        # _get_sgmcmc_noise: one legacy normal per element
                rs.normal(loc=0.0, scale=1.0, size=shp)
            bd = self._random_subsequence_and_buffers(buffer_length=B, subsequence_length=S, T=T, random_state=rs)
            return N, bd['right_buffer_end'] - bd['left_buffer_start']

        _pf.speculation.start(between)

    def noisy_gradient(self, preconditioner=None, is_scaled=True, **kwargs):
        """grad log-likelihood (particle filter, buffered) + grad log-prior, optionally / T
        (sgmcmc_sampler.py:427-464)."""
        kwargs.pop('tqdm', None)
        T_total = self._get_T(**kwargs)
        grad_loglike = self._noisy_grad_loglikelihood(**{k: v for k, v in kwargs.items() if k != 'T'})
        # a `parameters=` argument reaches the prior term and the preconditioner only: the
        # particle filter always runs at self.parameters (sgmcmc_sampler.py:379)
        at = kwargs.get('parameters', None)
        at = self.parameters if at is None else at
        grad_prior = self.prior.grad_logprior(parameters=at)
        grad = {var: grad_prior[var] + grad_loglike[var] for var in grad_prior}
        if preconditioner is None:
            if is_scaled:
                for var in grad:
                    grad[var] = grad[var] / T_total
        else:
            # sgmcmc_sampler.py:452-457: D(theta) * gradient, scaled by 1/T
            grad = preconditioner.precondition(grad, parameters=at,
                                               scale=(1.0 / T_total if is_scaled else 1.0))
        return grad

    def noisy_gradient_trace(self, parameters_list, is_scaled=True, **kwargs):
        """noisy_gradient at every Parameters of a stored trace, ALL in one launch (one workgroup
        per stored parameter x window).  Equivalent -- also in its np.random consumption -- to
            for p in parameters_list: sampler.parameters = p; sampler.noisy_gradient(**kwargs)
        which is the gradient loop of the KSD evaluation (svm/driver.py:1006-1027).
        Returns a list of gradient dicts; `self.parameters` is left untouched."""
        kwargs.pop('tqdm', None)
        T_total = self._get_T(**kwargs)
        grad_kwargs = {k: v for k, v in kwargs.items() if k != 'T'}
        keep = self.parameters
        groups_per_param = []
        try:
            for p in parameters_list:
                self.parameters = p
                groups_per_param.append(self._grad_groups(**grad_kwargs))
        finally:
            self.parameters = keep
        flat = [q for groups in groups_per_param for group in groups for q, _ in group]
        outs = iter(_pf.run_windows(flat))
        names = self.message_helper.score_names
        result = []
        for p, groups in zip(parameters_list, groups_per_param):
            grad = None
            for group in groups:
                part = {var: np.zeros_like(value) for var, value in p.as_dict().items()}
                for _, minibatch_size in group:
                    o = next(outs)
                    for name, g in zip(names, o["mean_statistic"]):
                        part[name] += g * 1.0 / minibatch_size
                grad = part if grad is None else {var: grad[var] + part[var] for var in part}
            grad = self._rescale_groups(grad, groups, **grad_kwargs)
            prior = self.prior.grad_logprior(parameters=p)
            total = {var: prior[var] + grad[var] for var in prior}
            if is_scaled:
                total = {var: total[var] / T_total for var in total}
            for var in total:
                if np.any(np.isnan(total[var])):
                    raise ValueError("NaNs in gradient of {0}".format(var))
            result.append(total)
        return result

    def _grad_groups(self, **kwargs):
        """Window problems of one gradient at self.parameters: list (per series) of groups."""
        return [self._grad_problems(**kwargs)]

    def _rescale_groups(self, grad, groups, **kwargs):
        return grad

    # -- steps ---------------------------------------------------------------------------------------
    def step_sgd(self, epsilon, **kwargs):
        self._rng_after_gradient = "nothing"
        try:
            delta = self.noisy_gradient(**kwargs)
        finally:
            self._rng_after_gradient = None
        for var in self.parameters.var_dict:
            self.parameters.var_dict[var] += epsilon * delta[var]
        return self.parameters

    def step_adagrad(self, epsilon, **kwargs):
        if not hasattr(self, "_adagrad_moments"):
            self._adagrad_moments = dict(t=0, G=0.0)
        self._rng_after_gradient = "nothing"
        try:
            g = self.parameters.from_dict_to_vector(self.noisy_gradient(**kwargs))
        finally:
            self._rng_after_gradient = None
        G = self._adagrad_moments['G'] + g ** 2
        delta = self.parameters.from_vector_to_dict(g / np.sqrt(G + NOISE_NUGGET), **self.parameters.dim)
        for var in self.parameters.var_dict:
            self.parameters.var_dict[var] += epsilon * delta[var]
        self._adagrad_moments['t'] += 1
        self._adagrad_moments['G'] = G
        return self.parameters

    def _get_sgmcmc_noise(self, is_scaled=True, preconditioner=None, **kwargs):
        scale = 1.0 / self._get_T(**kwargs) if is_scaled else 1.0
        if preconditioner is not None:
            return preconditioner.precondition_noise(parameters=self.parameters, scale=scale)
        return {var: np.random.normal(loc=0, scale=np.sqrt(scale), size=value.shape)
                for var, value in self.parameters.as_dict().items()}

    def sample_sgld(self, epsilon, **kwargs):
        """theta += eps * noisy_gradient + sqrt(2 eps) * N(0, 1/T)  (sgmcmc_sampler.py:549-567)."""
        if "preconditioner" in kwargs:
            raise ValueError("Use SGRLD instead")
        self._rng_after_gradient = "noise"
        try:
            delta = self.noisy_gradient(**kwargs)
        finally:
            self._rng_after_gradient = None
        white_noise = self._get_sgmcmc_noise(**kwargs)
        for var in self.parameters.var_dict:
            self.parameters.var_dict[var] += epsilon * delta[var] + np.sqrt(2.0 * epsilon) * white_noise[var]
        return self.parameters

    def sample_sgld_cv(self, epsilon, centering_parameters, centering_gradient, **kwargs):
        """SGLD with control variates (sgmcmc_sampler.py:569-611): gradient = centering_gradient +
        sub_gradient(parameters) - sub_gradient(centering_parameters) on the SAME windows.  As in
        the reference, with kind='pf' both particle filters run at self.parameters (only the prior
        term sees centering_parameters) and each consumes its own random draws."""
        if "preconditioner" in kwargs:
            raise ValueError("Use SGRLD instead")
        buffer_dicts = [self._random_subsequence_and_buffers(
            buffer_length=kwargs.get('buffer_length', 0),
            subsequence_length=kwargs.get('subsequence_length', -1),
            T=self._get_T(**kwargs)) for _ in range(kwargs.get('minibatch_size', 1))]
        cur = self.noisy_gradient(buffer_dicts=buffer_dicts, **kwargs)
        cen = self.noisy_gradient(parameters=centering_parameters, buffer_dicts=buffer_dicts, **kwargs)
        delta = {var: centering_gradient[var] + cur[var] - cen[var] for var in cur}
        white_noise = self._get_sgmcmc_noise(**kwargs)
        for var in self.parameters.var_dict:
            self.parameters.var_dict[var] += epsilon * delta[var] + np.sqrt(2.0 * epsilon) * white_noise[var]
        return self.parameters

    def step_precondition_sgd(self, epsilon, preconditioner, **kwargs):
        """theta += eps * D(theta) noisy_gradient  (sgmcmc_sampler.py:486-502)."""
        delta = self.noisy_gradient(preconditioner=preconditioner, **kwargs)
        for var in self.parameters.var_dict:
            self.parameters.var_dict[var] += epsilon * delta[var]
        return self.parameters

    def sample_sgrld(self, epsilon, preconditioner, **kwargs):
        """theta += eps (D grad + correction) + sqrt(2 eps) N(0, D / T)  (sgmcmc_sampler.py:613-640).
        The particle-filter gradient runs on the GPU; D(theta) is d = 1 host algebra."""
        scale = 1.0 / self._get_T(**kwargs) if kwargs.get("is_scaled", True) else 1.0
        delta = self.noisy_gradient(preconditioner=preconditioner, **kwargs)
        white_noise = self._get_sgmcmc_noise(preconditioner=preconditioner, **kwargs)
        correction = preconditioner.correction_term(self.parameters, scale=scale)
        for var in self.parameters.var_dict:
            self.parameters.var_dict[var] += (epsilon * (delta[var] + correction[var])
                                              + np.sqrt(2.0 * epsilon) * white_noise[var])
        return self.parameters

    def _get_preconditioner(self, preconditioner=None):
        if preconditioner is None:
            raise NotImplementedError("No Default Preconditioner for {}".format(self.name))
        return preconditioner

    def sample_gibbs(self):
        raise NotImplementedError()

    def project_parameters(self, **kwargs):
        self.parameters.project_parameters(**self.options, **kwargs)
        return self.parameters

    # -- fit loops -----------------------------------------------------------------------------------
    def get_iter_step(self, iter_type, steps_per_iteration=1, **kwargs):
        """(function names, kwargs) of one iteration, as sgmcmc_sampler.py:896-947."""
        project_kwargs = kwargs.get("project_kwargs", {})
        if iter_type == 'custom':
            names, kws = kwargs.get("iter_func_names"), kwargs.get("iter_func_kwargs")
        elif iter_type in ('SGD', 'ADAGRAD', 'SGLD', 'SGRD', 'SGRLD'):
            grad_kwargs = dict(epsilon=kwargs['epsilon'],
                               subsequence_length=kwargs['subsequence_length'],
                               buffer_length=kwargs['buffer_length'],
                               minibatch_size=kwargs.get('minibatch_size', 1),
                               kind=kwargs.get("kind", "pf"),
                               num_samples=kwargs.get("num_samples", None),
                               **kwargs.get("pf_kwargs", {}))
            if 'num_sequences' in kwargs:
                grad_kwargs['num_sequences'] = kwargs['num_sequences']
            if iter_type in ('SGRD', 'SGRLD'):
                grad_kwargs['preconditioner'] = self._get_preconditioner(kwargs.get('preconditioner'))
            step = dict(SGD='step_sgd', ADAGRAD='step_adagrad', SGLD='sample_sgld',
                        SGRD='step_precondition_sgd', SGRLD='sample_sgrld')[iter_type]
            names, kws = [step, 'project_parameters'], [grad_kwargs, project_kwargs]
        elif iter_type == 'Gibbs':
            raise NotImplementedError("iter_type '{0}' is not on the particle-filter path".format(iter_type))
        else:
            raise ValueError("Unrecognized iter_type {0}".format(iter_type))
        return names * steps_per_iteration, kws * steps_per_iteration

    # -- resident fit: rng='device' SGLD runs on the GPU without the host in the loop ----------------------------
    _RESIDENT_PF_KEYS = frozenset(("pf", "N", "kernel", "lambduh", "rng", "dtype", "resampling", "tqdm", "resident"))

    def _resident_plan(self, iter_type, kwargs):
        """fit / fit_timed / fit_evaluate with iter_type='SGLD' and pf_kwargs=dict(rng='device', ...): the step
        (sample_sgld + project_parameters, sgmcmc_sampler.py:549-567, 650-656 of the reference) needs nothing from the
        host -- window start, particle filter, Langevin noise and projection are all device kernels of ChainEnsemble
        -- so the loop runs RESIDENT: a one-chain ensemble, K steps per hipGraph replay, parameters copied back at the
        reference's save points.  Returns the ensemble's arguments, or None when the step has to stay on the host
        (rng='replay': np.random's order is the contract there; other iter types; minibatches; options the device
        update does not know).  pf_kwargs['resident'] = False keeps the host loop."""
        pfkw = dict(kwargs.get("pf_kwargs", {}))
        if iter_type != "SGLD" or pfkw.get("rng", "replay") not in ("device", "philox") or pfkw.get("resident", True) is False:
            return None
        if kwargs.get("kind", "pf") != "pf" or kwargs.get("minibatch_size", 1) != 1:
            return None
        if getattr(self, "options", None) or kwargs.get("project_kwargs"):
            return None
        if set(pfkw) - self._RESIDENT_PF_KEYS or pfkw.get("pf", "poyiadjis_N") not in ("poyiadjis_N", "nemeth"):
            return None
        obs = self.observations
        is_list = isinstance(obs, (list, tuple))
        if is_list != isinstance(self, SeqSGMCMCSampler) or obs is None:
            return None
        if is_list and kwargs.get("num_sequences", -1) != 1:
            return None             # the ensemble's sequence lists draw ONE sequence per step (num_sequences = 1)
        if int(pfkw.get("N", 1000)) > 16384:
            return None
        return dict(N=int(pfkw.get("N", 1000)), pf=pfkw.get("pf", "poyiadjis_N"), lambduh=pfkw.get("lambduh", None),
                    kernel=pfkw.get("kernel", None), dtype=pfkw.get("dtype", "f64"),
                    resampling=pfkw.get("resampling", "multinomial"), epsilon=float(kwargs["epsilon"]),
                    S=int(kwargs["subsequence_length"]), B=int(kwargs["buffer_length"]),
                    spi=int(kwargs.get("steps_per_iteration", 1)), is_list=is_list)

    def _resident_ensemble(self, plan):
        """The one-chain ensemble of a resident fit.  Its device generator is keyed from np.random (two randint draws), so
        np.random.seed(s) before fit reproduces the run; (seed, chain 0) is what a ChainEnsemble built by hand with the
        same arguments runs -- bit for bit (tests/test_gpu_sampler.py)."""
        from .ensemble import ChainEnsemble
        seed = int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31)
        return ChainEnsemble(self.message_helper.model, self.observations, self.parameters, num_chains=1, N=plan["N"],
                             pf=plan["pf"], lambduh=plan["lambduh"], kernel=plan["kernel"], epsilon=plan["epsilon"],
                             prior=self.prior, subsequence_length=plan["S"], buffer_length=plan["B"], dtype=plan["dtype"],
                             seed=seed, chain_offset=0, forward_message=getattr(self, "forward_message", None),
                             resampling=plan["resampling"], window_sampling="host" if plan["is_list"] else "device")

    @staticmethod
    def _graph_steps(ens, thin):
        """Largest K <= 64 dividing `thin` (steps between two saved states) when the ensemble's step can be captured."""
        if ens.S != -1 and ens.window_sampling != "device":
            return 0
        for K in range(min(64, thin), 0, -1):
            if thin % K == 0:
                return K
        return 0

    def _fit_resident(self, plan, num_iters, output_all):
        ens = self._resident_ensemble(plan)
        steps = num_iters * plan["spi"]
        history = [self.parameters.copy()] if output_all else None
        if steps > 0:
            thin = plan["spi"] if output_all else steps
            samples = ens.run(steps, thin=thin, graph_steps=self._graph_steps(ens, thin))
            if output_all:
                history += [ens._params_from_theta(th[0]) for th in samples]
            self.parameters = ens._params_from_theta(samples[-1][0])
        return history if output_all else self.parameters.copy()

    def _run_iter(self, names, kws):
        for name, kw in zip(names, kws):
            getattr(self, name)(**kw)

    def fit(self, iter_type, num_iters, output_all=False, observations=None, init_parameters=None,
            tqdm=None, catch_interrupt=False, **kwargs):
        """num_iters iterations; returns the final Parameters or, with output_all, the list of
        num_iters + 1 Parameters (sgmcmc_sampler.py:659-721)."""
        if observations is not None:
            self.observations = observations
        if init_parameters is not None:
            self.parameters = init_parameters.copy()
        names, kws = self.get_iter_step(iter_type, **kwargs)
        plan = self._resident_plan(iter_type, kwargs)
        if plan is not None:
            return self._fit_resident(plan, num_iters, output_all)
        history = [self.parameters.copy()] if output_all else None
        steps = range(1, num_iters + 1)
        if tqdm is not None:
            steps = tqdm(steps)
            steps.set_description("fit using {0} iters".format(iter_type))
        for it in steps:
            try:
                self._run_iter(names, kws)
                if output_all:
                    history.append(self.parameters.copy())
            except KeyboardInterrupt as e:
                if not catch_interrupt:
                    raise e
                logger.warning("Interrupt in fit:\n{0}\nStopping early after {1} iters".format(e, it))
                return history[:it] if output_all else self.parameters.copy()
        return history if output_all else self.parameters.copy()

    def fit_evaluate(self, iter_type, metric_functions=None, max_num_iters=None, max_time=60,
                     min_save_time=1, observations=None, init_parameters=None, tqdm=None,
                     tqdm_iter=False, catch_interrupt=False, total_max_time=None, **kwargs):
        """Time-budgeted fit: keep stepping, save a copy of the parameters at most every
        min_save_time seconds until max_time seconds of sampler time are used
        (sgmcmc_sampler.py:762-894).  Returns three pandas DataFrames
        (parameters_list[iteration, parameters], times[iteration, time], metrics[...]).
        metric_functions: callables f(sampler) -> dict | list of dict(metric, variable, value)."""
        import pandas as pd
        if observations is not None:
            self.observations = observations
        if init_parameters is not None:
            self.parameters = init_parameters.copy()
        if metric_functions is None:
            metric_functions = []
        elif callable(metric_functions):
            metric_functions = [metric_functions]
        names, kws = self.get_iter_step(iter_type, **kwargs)
        num_saves = int(max_time // min_save_time)
        if max_num_iters is not None:
            num_saves = min(num_saves, max_num_iters)

        rows = []

        def evaluate(iteration):
            for f in metric_functions:
                res = f(self)
                for r in ([res] if isinstance(res, dict) else res):
                    rows.append(dict(iteration=iteration, metric=r['metric'], variable=r['variable'],
                                     value=r['value']))

        saved, times, iterations = [self.parameters.copy()], [0.0], [0]
        evaluate(0)
        iteration, total_time = 0, 0.0
        plan = self._resident_plan(iter_type, kwargs)
        ens = self._resident_ensemble(plan) if plan is not None else None
        burst = 0
        if ens is not None:
            # resident: `burst` iterations per call (one hipGraph replay of K steps each when the step can be captured),
            # the clock is read after every burst; metrics see the parameters of the save point, as in the host loop
            K = self._graph_steps(ens, 64 * plan["spi"])
            burst = max(1, (K or plan["spi"]) // plan["spi"])

            def run_burst():
                th = ens.run(burst * plan["spi"], thin=burst * plan["spi"], graph_steps=K)
                self.parameters = ens._params_from_theta(th[-1][0])
        start, last_save = time.time(), time.time()
        saves = range(1, num_saves + 1)
        if tqdm is not None:
            saves = tqdm(saves)
        try:
            for _ in saves:
                if ens is not None:
                    # (the reference caps a save interval at 1000 iterations, sgmcmc_sampler.py:845-867 -- with its
                    # 20 ms steps never reached; at 70 us per resident step the cap would end a 60-s fit after 4 s, so
                    # the resident loop keeps to the interval's meaning: step until min_save_time has passed, save)
                    done = 0
                    while True:
                        run_burst()
                        done += burst
                        if time.time() - last_save > min_save_time:
                            total_time += time.time() - last_save
                            iteration += done
                            saved.append(self.parameters.copy())
                            times.append(total_time)
                            iterations.append(iteration)
                            evaluate(iteration)
                            last_save = time.time()
                            break
                    if total_time > max_time:
                        break
                    if total_max_time is not None and time.time() - start > total_max_time:
                        break
                    continue
                for step in range(1000):
                    self._run_iter(names, kws)
                    if time.time() - last_save > min_save_time:
                        total_time += time.time() - last_save
                        iteration += step + 1
                        saved.append(self.parameters.copy())
                        times.append(total_time)
                        iterations.append(iteration)
                        evaluate(iteration)
                        last_save = time.time()
                        break
                if total_time > max_time:
                    break
                if total_max_time is not None and time.time() - start > total_max_time:
                    break
        except KeyboardInterrupt as e:
            if not catch_interrupt:
                raise e
            logger.warning("Interrupt in fit_timed:\n{0}\nStopping early".format(e))
        parameters_list = pd.DataFrame(dict(iteration=iterations, parameters=saved))
        times_df = pd.DataFrame(dict(iteration=iterations, time=times))
        metrics = pd.DataFrame(rows, columns=['iteration', 'metric', 'variable', 'value'])
        return parameters_list, times_df, metrics

    def fit_timed(self, iter_type, max_time=60, min_save_time=1, observations=None,
                  init_parameters=None, tqdm=None, tqdm_iter=False, catch_interrupt=False, **kwargs):
        """-> (list of Parameters, list of cumulative fit times)  (sgmcmc_sampler.py:723-760)."""
        plist, times, _ = self.fit_evaluate(
            iter_type=iter_type, max_time=max_time, min_save_time=min_save_time,
            observations=observations, init_parameters=init_parameters, tqdm=tqdm,
            tqdm_iter=tqdm_iter, catch_interrupt=catch_interrupt, **kwargs)
        return plist['parameters'].tolist(), times['time'].tolist()

    # -- predict (kind='pf', latent marginals) ---------------------------------------------------------
    def predict(self, target='latent', distr=None, lag=None, return_distr=None, num_samples=None,
                kind='pf', observations=None, parameters=None, **kwargs):
        """Smoothed latent marginals by particle filter (sgmcmc_sampler.py:956-1069, kind='pf',
        target='latent', return_distr=True): -> (x_mean, x_cov) from Helper.pf_latent_var_distr."""
        self._require_pf(kind)
        if target != 'latent':
            raise NotImplementedError("predict(target='{0}', kind='pf') is not built (pf_y_distr)".format(target))
        if return_distr is False:
            raise ValueError("return_distr must be True for kind = pf")
        observations = self._get_observations(observations)
        if parameters is None:
            parameters = self.parameters
        kwargs.pop('tqdm', None)
        return self.message_helper.pf_latent_var_distr(lag=lag, observations=observations,
                                                       parameters=parameters, **kwargs)

    def exact_loglikelihood(self, *args, **kwargs):
        raise NotImplementedError(_ONLY_PF.format('marginal'))


class SeqSGMCMCSampler(object):
    """Mixin: `observations` is a list of independent sequences (sgmcmc_sampler.py:1157-1283).
    All windows of all chosen sequences go to the GPU in one launch."""

    def _get_T(self, **kwargs):
        T = kwargs.get('T')
        if T is None:
            observations = self._get_observations(kwargs.get('observations'))
            T = int(np.sum([np.shape(observation)[0] for observation in observations]))
        return T

    def _check_observation_shape(self, observations):
        if observations is not None:
            for ii, observation in enumerate(observations):
                try:
                    super()._check_observation_shape(observations=observation)
                except ValueError as e:
                    raise ValueError("Error in observations[{0}] :\n{1}".format(ii, e))

    def predict(self, target='latent', distr=None, lag=None, return_distr=None, num_samples=None,
                kind='pf', observations=None, parameters=None, tqdm=None, **kwargs):
        """One (x_mean, x_cov) per sequence (sgmcmc_sampler.py:1285-1423, kind='pf')."""
        self._require_pf(kind)
        if target != 'latent':
            raise NotImplementedError("predict(target='{0}', kind='pf') is not built (pf_y_distr)".format(target))
        if return_distr is False:
            raise ValueError("return_distr must be True for kind = pf")
        observations = self._get_observations(observations)
        if parameters is None:
            parameters = self.parameters
        return [self.message_helper.pf_latent_var_distr(lag=lag, observations=observation,
                                                        parameters=parameters, **kwargs)
                for observation in observations]

    def _choose_sequences(self, observations, num_sequences):
        indices = np.arange(len(observations))
        if num_sequences != -1:
            indices = np.random.choice(indices, num_sequences, replace=False)
        return indices

    def noisy_loglikelihood(self, num_sequences=-1, observations=None, tqdm=None, **kwargs):
        observations = self._get_observations(observations)
        kwargs.pop('check_shape', None)
        groups, S = [], 0.0
        for index in self._choose_sequences(observations, num_sequences):
            S += observations[index].shape[0]
            # check_shape=False: the reference re-checks ONE sequence as if it were a list of
            # sequences here, which raises IndexError for LGSSM (lgssm/sampler.py:61); fixed.
            groups.append(SGMCMCSampler._loglike_problems(
                self, observations=observations[index], check_shape=False, **kwargs))
        outs = iter(_pf.run_windows([q for group in groups for q, _ in group]))
        value = 0.0
        for group in groups:
            part = 0.0
            for _ in group:
                part += next(outs)["loglikelihood_estimate"]
            value += part * (1.0 / group[0][1])
        if np.isnan(value):
            raise ValueError("NaNs in loglikelihood")
        if num_sequences != -1:
            value *= self._get_T(**kwargs) / S
        return value

    def predictive_loglikelihood(self, num_sequences=-1, observations=None, tqdm=None, **kwargs):
        """Sum of the per-sequence predictive log-likelihoods (sgmcmc_sampler.py:1224-1247)."""
        observations = self._get_observations(observations)
        value, S = 0, 0.0
        for index in self._choose_sequences(observations, num_sequences):
            S += observations[index].shape[0]
            value = value + SGMCMCSampler.predictive_loglikelihood(
                self, observations=observations[index], check_shape=False, **kwargs)
        if num_sequences != -1:
            value = value * (self._get_T(**kwargs) / S)
        return value

    def _grad_groups(self, num_sequences=-1, **kwargs):
        observations = self.observations
        groups, S = [], 0.0
        for index in self._choose_sequences(observations, num_sequences):
            groups.append(SGMCMCSampler._grad_problems(self, observations=observations[index], **kwargs))
            S += observations[index].shape[0]
        return groups

    def _rescale_groups(self, grad, groups, num_sequences=-1, **kwargs):
        if num_sequences != -1:
            S = self._seq_lengths_of(groups)
            grad = {var: grad[var] * self._get_T(**kwargs) / S for var in grad}
        return grad

    def _seq_lengths_of(self, groups):
        return float(sum(group[0][0]["_series_length"] for group in groups))

    def _noisy_grad_loglikelihood(self, num_sequences=-1, **kwargs):
        groups = self._grad_groups(num_sequences=num_sequences, **kwargs)
        grad = self._run_grad_problems(groups)
        return self._rescale_groups(grad, groups, num_sequences=num_sequences, **kwargs)
