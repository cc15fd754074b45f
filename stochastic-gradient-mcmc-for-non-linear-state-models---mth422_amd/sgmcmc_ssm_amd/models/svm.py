"""Stochastic volatility model:  x_t = A x_{t-1} + N(0, Q),   y_t ~ N(0, exp(x_t) R).

Exports the reference's `sgmcmc_ssm.models.svm` names for the particle-filter path:
SVMParameters, SVMPrior, SVMHelper, SVMSampler, SeqSVMSampler, generate_svm_data
(reference: models/svm/{parameters,helper,sampler}.py).  The bootstrap ("prior") proposal,
its weights and the complete-data score (models/svm/kernels.py:15-64, helper.py:342-348) run
inside libpfgrad.so as model id PFG_MODEL_SVM."""
import numpy as np

from ..base_parameters import (BaseParameters, BasePrior, MatrixVar, CholPrecisionVar,
                               WishartPrecisionPrior, MatrixNormalPrior, install_properties)
from ..sgmcmc_sampler import SGMCMCSampler, SeqSGMCMCSampler, PFHelper


@install_properties
class SVMParameters(BaseParameters):
    """A (1x1), LQinv_vec, LRinv_vec.  SVMParameters(A=, Q=|LQinv=|LQinv_vec=, R=|LRinv=|LRinv_vec=)."""
    _specs = (MatrixVar('A', ('n',)), CholPrecisionVar('Q', 'n'), CholPrecisionVar('R', 'm'))

    def __str__(self):
        return "SVMParameters:\nA:{0}, Q:{1}, R:{2}\n".format(self.A[0, 0], self.Q[0, 0], self.R[0, 0])

    @property
    def phi(self):
        return self.A

    @property
    def sigma(self):
        return self.LQinv ** -1

    @property
    def tau(self):
        return self.LRinv ** -1


class SVMPrior(BasePrior):
    """Wishart on Qinv and Rinv, matrix-normal on A given Q (models/svm/parameters.py:62-72)."""
    _Parameters = SVMParameters
    _blocks = (WishartPrecisionPrior('Q', 'n'), WishartPrecisionPrior('R', 'm'),
               MatrixNormalPrior('A', ('n',), row_cov='Q'))


def stationary_precision(Qinv, A, num_iters=50):
    """Fixed-point iteration for the stationary precision of x_t = A x_{t-1} + N(0, Q)
    (what the reference's var_stationary_precision computes, _utils.py:175-183)."""
    precision = Qinv
    QinvA = np.dot(Qinv, A)
    AtQinvA = np.dot(A.T, QinvA)
    for _ in range(num_iters):
        precision = Qinv - np.dot(QinvA, np.linalg.solve(precision + AtQinvA, QinvA.T))
    return precision


def generate_svm_data(T, parameters, initial_message=None, tqdm=None):
    """Simulate T steps.  Draws from np.random in the reference's order (one
    multivariate_normal for x_{-1}, then per t one for x_t and one for y_t), so equal seeds
    give equal series.  Returns dict(observations (T,1), latent_vars (T,1), parameters,
    initial_message)."""
    A, Q, R = parameters.A, parameters.Q, parameters.R
    n = A.shape[0]
    if initial_message is None:
        initial_message = dict(log_constant=0.0, mean_precision=np.zeros(n),
                               precision=stationary_precision(parameters.Qinv, A, 10))
    x_prev = np.random.multivariate_normal(
        mean=np.linalg.solve(initial_message['precision'], initial_message['mean_precision']),
        cov=np.linalg.inv(initial_message['precision']))
    x = np.zeros((T, n), dtype=float)
    y = np.zeros((T, R.shape[0]), dtype=float)
    for t in range(T):
        x[t] = np.random.multivariate_normal(mean=np.dot(A, x_prev), cov=Q)
        y[t] = np.random.multivariate_normal(mean=np.zeros(1), cov=np.exp(x[t]) * R)
        x_prev = x[t]
    return dict(observations=y, latent_vars=x, parameters=parameters, initial_message=initial_message)


class SVMHelper(PFHelper):
    """pf_gradient_estimate -> dict(LRinv_vec, LQinv_vec, A)  (models/svm/helper.py:121-126)."""
    model = "svm"
    default_kernel = "prior"
    kernels = ("prior",)
    score_names = ("LRinv_vec", "LQinv_vec", "A")


class SVMSampler(SGMCMCSampler):
    def __init__(self, n=1, m=1, observations=None, prior=None, parameters=None,
                 forward_message=None, name="SVMSampler", **kwargs):
        self.options = kwargs
        self.n, self.m, self.name = n, m, name
        self.setup(observations=observations, prior=prior, parameters=parameters,
                   forward_message=forward_message)

    def setup(self, observations=None, prior=None, parameters=None, forward_message=None):
        """models/svm/sampler.py:22-65: default prior, prior draw when no parameters, and the
        fixed N(0, 10) forward message every window starts from."""
        self.observations = observations
        self.prior = SVMPrior.generate_default_prior(n=self.n, m=self.m) if prior is None else prior
        if parameters is None:
            self.parameters = self.prior.sample_prior().project_parameters()
        else:
            if not isinstance(parameters, SVMParameters):
                raise ValueError("parameters is not a SVMParameter")
            self.parameters = parameters
        if forward_message is None:
            forward_message = dict(log_constant=0.0, mean_precision=np.zeros(self.n),
                                   precision=np.eye(self.n) / 10)
        self.forward_message = forward_message
        self.backward_message = dict(log_constant=0.0, mean_precision=np.zeros(self.n),
                                     precision=np.zeros((self.n, self.n)))
        self.message_helper = SVMHelper(n=self.n, m=self.m, forward_message=forward_message,
                                        backward_message=self.backward_message)


class SeqSVMSampler(SeqSGMCMCSampler, SVMSampler):
    pass
