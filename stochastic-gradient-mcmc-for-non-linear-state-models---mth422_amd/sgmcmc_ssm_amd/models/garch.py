"""GARCH(1,1) latent process observed in noise:
    sigma2_t = alpha + beta x_{t-1}^2 + gamma sigma2_{t-1},  x_t ~ N(0, sigma2_t),  y_t ~ N(x_t, R)
parameterised by log_mu, logit_phi, logit_lambduh (alpha = mu(1-phi), beta = phi*lambduh,
gamma = phi(1-lambduh)) and LRinv_vec.

Exports GARCHParameters, GARCHPrior, GARCHHelper, GARCHSampler, SeqGARCHSampler,
generate_garch_data (reference: models/garch/{parameters,helper,sampler}.py).  Both proposals
(models/garch/kernels.py:5-180) and the score (helper.py:335-372) are model id
PFG_MODEL_GARCH in libpfgrad.so; the particle state is (x, sigma2)."""
import numpy as np
from scipy.special import logit

from .. import _capi
from ..base_parameters import (BaseParameters, BasePrior, GARCHVars, CholPrecisionVar,
                               GARCHVarsPrior, WishartPrecisionPrior, install_properties)
from ..sgmcmc_sampler import SGMCMCSampler, SeqSGMCMCSampler, PFHelper


@install_properties
class GARCHParameters(BaseParameters):
    """log_mu, logit_phi, logit_lambduh (each (1,)), LRinv_vec."""
    _specs = (GARCHVars(), CholPrecisionVar('R', 'm'))

    def __str__(self):
        return "GARCHParameters:\nalpha:{0}, beta:{1}, gamma:{2}, tau:{3}\n".format(
            np.around(float(self.alpha[0]), 6), np.around(float(self.beta[0]), 6),
            np.around(float(self.gamma[0]), 6), np.around(float(self.tau[0, 0]), 6))

    @property
    def tau(self):
        return self.LRinv ** -1

    @staticmethod
    def convert_alpha_beta_gamma(alpha, beta, gamma):
        """(alpha, beta, gamma) -> (log_mu, logit_phi, logit_lambduh):
        mu = alpha/(1-beta-gamma), phi = beta+gamma, lambduh = beta/(beta+gamma)."""
        if alpha <= 0 or beta <= 0 or gamma <= 0:
            raise ValueError("Cannot have alpha, beta, or gamma <= 0")
        if beta + gamma >= 1:
            raise ValueError("Cannot have beta + gamma >- 1")
        return np.log(alpha / (1 - beta - gamma)), logit(beta + gamma), logit(beta / (beta + gamma))


class GARCHPrior(BasePrior):
    _Parameters = GARCHParameters
    _blocks = (GARCHVarsPrior(), WishartPrecisionPrior('R', 'm'))


def generate_garch_data(T, parameters, initial_message=None, tqdm=None):
    """Simulate T steps with the reference's np.random call order (garch/parameters.py:70-134).
    Returns dict(observations, latent_vars, sigma2s, parameters, initial_message)."""
    alpha, beta, gamma, R = parameters.alpha, parameters.beta, parameters.gamma, parameters.R
    if initial_message is None:
        initial_message = dict(log_constant=0.0, mean_precision=np.zeros(1),
                               precision=np.atleast_2d((1 - beta - gamma) / alpha))
    x_prev = np.random.multivariate_normal(
        mean=np.linalg.solve(initial_message['precision'], initial_message['mean_precision']),
        cov=np.linalg.inv(initial_message['precision']))
    x = np.zeros((T, 1), dtype=float)
    sigma2s = np.zeros((T), dtype=float)
    y = np.zeros((T, 1), dtype=float)
    sigma2_prev = 0
    for t in range(T):
        sigma2s[t] = (alpha + beta * x_prev ** 2 + gamma * sigma2_prev)[0]
        x[t] = np.random.multivariate_normal(mean=np.zeros(1), cov=np.array([[sigma2s[t]]]))
        y[t] = np.random.multivariate_normal(mean=x[t], cov=R)
        x_prev, sigma2_prev = x[t], sigma2s[t]
    return dict(observations=y, latent_vars=x, sigma2s=sigma2s, parameters=parameters,
                initial_message=initial_message)


class GARCHHelper(PFHelper):
    """pf_gradient_estimate -> dict(LRinv_vec, log_mu, logit_phi, logit_lambduh)
    (models/garch/helper.py:109-115); default kernel 'optimal' (:48-57)."""
    model = "garch"
    default_kernel = "optimal"
    kernels = ("prior", "optimal")
    score_names = ("LRinv_vec", "log_mu", "logit_phi", "logit_lambduh")

    def _default_forward_message(self):
        return None       # no message: start from the stationary law (garch/helper.py:324-327)

    def _prior_x(self, forward_message, parameters):
        if forward_message is None:
            forward_message = self.default_forward_message
        if forward_message is None:
            var = parameters.alpha / (1 - parameters.beta - parameters.gamma)
            return 0.0, float(var[0]), 0
        return super()._prior_x(forward_message, parameters)


class GARCHSampler(SGMCMCSampler):
    def __init__(self, n=1, m=1, observations=None, prior=None, parameters=None,
                 forward_message=None, name="GARCHSampler", **kwargs):
        self.options = kwargs
        self.n, self.m, self.name = n, m, name
        self.setup(observations=observations, prior=prior, parameters=parameters,
                   forward_message=forward_message)

    def setup(self, observations, prior, parameters=None, forward_message=None):
        self.observations = observations
        self.prior = GARCHPrior.generate_default_prior(n=self.n, m=self.m) if prior is None else prior
        if parameters is None:
            self.parameters = self.prior.sample_prior()
        else:
            if not isinstance(parameters, GARCHParameters):
                raise ValueError("parameters is not a GARCHParameter")
            self.parameters = parameters
        self.forward_message = forward_message
        self.backward_message = dict(log_constant=0.0, mean_precision=np.zeros(self.n),
                                     precision=np.zeros((self.n, self.n)))
        self.message_helper = GARCHHelper(n=self.n, m=self.m, forward_message=forward_message,
                                          backward_message=self.backward_message)


class SeqGARCHSampler(SeqSGMCMCSampler, GARCHSampler):
    pass
