"""1-D linear-Gaussian SSM:  x_t = A x_{t-1} + N(0, Q),   y_t = C x_t + N(0, R).

Exports LGSSMParameters, LGSSMPrior, LGSSMHelper, LGSSMSampler, SeqLGSSMSampler,
generate_lgssm_data (reference: models/lgssm/{parameters,helper,sampler}.py).  Particle-filter
entries only: the "prior" and "optimal" proposals (models/lgssm/kernels.py:11-122) and the
score (helper.py:1270-1277) are model id PFG_MODEL_LGSSM in libpfgrad.so.  The reference's
Kalman (exact) code is not accelerated; its outputs serve as test fixtures."""
import numpy as np

from ..base_parameters import (BaseParameters, BasePrior, MatrixVar, CholPrecisionVar,
                               WishartPrecisionPrior, MatrixNormalPrior, install_properties,
                               BasePreconditioner, MatrixPrecond, CholPrecisionPrecond)
from ..sgmcmc_sampler import SGMCMCSampler, SeqSGMCMCSampler, PFHelper
from .svm import stationary_precision


@install_properties
class LGSSMParameters(BaseParameters):
    """A, C (1x1), LQinv_vec, LRinv_vec."""
    _specs = (MatrixVar('A', ('n',)), MatrixVar('C', ('m', 'n'), stable=False),
              CholPrecisionVar('Q', 'n'), CholPrecisionVar('R', 'm'))

    def __str__(self):
        return "LGSSMParameters:\nA:\n{0}\nC:\n{1}\nQ:\n{2}\nR:\n{3}".format(self.A, self.C, self.Q, self.R)

    def project_parameters(self, **kwargs):
        # C is pinned to the identity unless told otherwise (lgssm/parameters.py:39-42)
        if 'C' not in kwargs:
            kwargs['C'] = dict(fixed_eye=True)
        return super().project_parameters(**kwargs)


class LGSSMPrior(BasePrior):
    _Parameters = LGSSMParameters
    _blocks = (WishartPrecisionPrior('Q', 'n'), WishartPrecisionPrior('R', 'm'),
               MatrixNormalPrior('A', ('n',), row_cov='Q'),
               MatrixNormalPrior('C', ('m', 'n'), row_cov='R'))


class LGSSMPreconditioner(BasePreconditioner):
    """SGRLD / SGRD preconditioner (lgssm/parameters.py:58-67); noise is drawn A, C, Q, R."""
    _blocks = (MatrixPrecond('A', 'Q'), MatrixPrecond('C', 'R'),
               CholPrecisionPrecond('Q'), CholPrecisionPrecond('R'))


def generate_lgssm_data(T, parameters, initial_message=None, tqdm=None):
    """Simulate T steps with the reference's np.random call order (lgssm/parameters.py:67-129)."""
    A, C, Q, R = parameters.A, parameters.C, parameters.Q, parameters.R
    m, n = np.shape(C)
    if initial_message is None:
        initial_message = dict(log_constant=0.0, mean_precision=np.zeros(n),
                               precision=stationary_precision(parameters.Qinv, A, 10))
    x_prev = np.random.multivariate_normal(
        mean=np.linalg.solve(initial_message['precision'], initial_message['mean_precision']),
        cov=np.linalg.inv(initial_message['precision']))
    x = np.zeros((T, n), dtype=float)
    y = np.zeros((T, m), dtype=float)
    for t in range(T):
        x[t] = np.random.multivariate_normal(mean=np.dot(A, x_prev), cov=Q)
        y[t] = np.random.multivariate_normal(mean=np.dot(C, x[t]), cov=R)
        x_prev = x[t]
    return dict(observations=y, latent_vars=x, parameters=parameters, initial_message=initial_message)


class LGSSMHelper(PFHelper):
    """pf_gradient_estimate -> dict(LRinv_vec, LQinv_vec, C, A)  (models/lgssm/helper.py:1136-1142);
    default kernel 'optimal' for n*m = 1 (:1200-1214)."""
    model = "lgssm"
    default_kernel = "optimal"
    kernels = ("prior", "optimal")
    score_names = ("LRinv_vec", "LQinv_vec", "C", "A")


class LGSSMSampler(SGMCMCSampler):
    def __init__(self, n=1, m=1, observations=None, prior=None, parameters=None,
                 forward_message=None, backward_message=None, name="LGSSMSampler", **kwargs):
        self.options = kwargs
        self.n, self.m, self.name = n, m, name
        self.setup(observations=observations, prior=prior, parameters=parameters,
                   forward_message=forward_message, backward_message=backward_message)

    def setup(self, observations=None, prior=None, parameters=None, forward_message=None,
              backward_message=None):
        self.observations = observations
        self.prior = LGSSMPrior.generate_default_prior(n=self.n, m=self.m) if prior is None else prior
        self.parameters = self.prior.sample_prior() if parameters is None else parameters
        if forward_message is None:
            forward_message = dict(log_constant=0.0, mean_precision=np.zeros(self.n),
                                   precision=np.eye(self.n) / 10)
        self.forward_message = forward_message
        if backward_message is None:
            backward_message = dict(log_constant=0.0, mean_precision=np.zeros(self.n),
                                    precision=np.zeros((self.n, self.n)))
        self.backward_message = backward_message
        self.message_helper = LGSSMHelper(n=self.n, m=self.m, forward_message=forward_message,
                                          backward_message=backward_message)

    def _check_observation_shape(self, observations):
        if observations is None:
            return
        if np.shape(observations)[1] != self.m:
            raise ValueError("observations second dimension does not match m")

    def _get_preconditioner(self, preconditioner=None):
        return LGSSMPreconditioner() if preconditioner is None else preconditioner


class SeqLGSSMSampler(SeqSGMCMCSampler, LGSSMSampler):
    pass
