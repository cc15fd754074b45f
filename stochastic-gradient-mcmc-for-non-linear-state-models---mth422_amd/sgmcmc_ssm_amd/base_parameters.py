"""Parameters / Prior machinery for the three 1-D state-space models on the PF path.

Own implementation of the *interface* of the reference's `sgmcmc_ssm/base_parameters.py`
+ `variables/{matrices,covariance,garch_var}.py`, restricted to what the particle-filter
gradient path needs (n = m = 1; the reference's vector / HMM / SLDS helpers are out of scope,
SURVEY.md section 2).  Shapes are kept exactly as the reference keeps them so the classes are
drop-in for its experiment scripts: matrices are (1,1) ndarrays, Cholesky-precision vectors
`L*inv_vec` and the GARCH variables are (1,) ndarrays, and `var_dict` keeps the reference's
insertion order (svm/parameters.py:21-25, lgssm/parameters.py:20-25, garch/parameters.py:19-22)
because the SGLD noise is drawn per variable in that order (sgmcmc_sampler.py:540-546).

Host-side only: O(#parameters) work per SGLD step.
"""
from copy import deepcopy
import functools
import logging

import numpy as np


@functools.lru_cache(maxsize=None)
def _tril(n):
    """np.tril_indices(n), computed once per size (it showed up as a quarter of a short-window SGLD step)."""
    return np.tril_indices(int(n))


@functools.lru_cache(maxsize=None)
def _triu1(n):
    return np.triu_indices(int(n), 1)

import scipy.stats
from scipy.special import expit, logit

logger = logging.getLogger(name=__name__)


# ----------------------------------------------------------------------------------------
# variable specs: how one block of var_dict is parsed, exposed and projected
# ----------------------------------------------------------------------------------------
class MatrixVar(object):
    """A dense matrix variable stored under its own name (A: square, C: rectangular).
    Reference behaviour: variables/matrices.py:448-507 (square), :907-966 (rect)."""

    def __init__(self, name, dims, stable=True):
        self.name, self.dims, self.stable = name, dims, stable

    def keys(self):
        return (self.name,)

    def parse(self, params, kwargs):
        if self.name not in kwargs:
            raise ValueError("{} not provided".format(self.name))
        val = np.array(kwargs[self.name]).astype(float)
        if val.ndim != 2:
            raise ValueError("{} must be a matrix".format(self.name))
        if len(self.dims) == 1 and val.shape[0] != val.shape[1]:
            raise ValueError("{} must be square matrices".format(self.name))
        params.var_dict[self.name] = val
        if len(self.dims) == 1:
            params._set_check_dim(**{self.dims[0]: val.shape[0]})
        else:
            params._set_check_dim(**{self.dims[0]: val.shape[0], self.dims[1]: val.shape[1]})

    def size(self, dim):
        return dim[self.dims[0]] ** 2 if len(self.dims) == 1 else dim[self.dims[0]] * dim[self.dims[1]]

    def shape(self, dim):
        return (dim[self.dims[0]],) * 2 if len(self.dims) == 1 else (dim[self.dims[0]], dim[self.dims[1]])

    def project(self, params, **kwargs):
        opts = kwargs.get(self.name, {})
        if opts.get('thresh', self.stable):
            # |A| <= cutoff (for n = 1 the spectral radius is |A|): _utils.py:165-170
            A = params.var_dict[self.name]
            cutoff = opts.get('eigenvalue_cutoff', 0.9999)
            if A.shape == (1, 1):
                radius = np.abs(A[0, 0])
            else:
                radius = np.max(np.abs(np.linalg.eigvals(A)))
            if radius > cutoff:
                logger.info("Thresholding |{2}|: {0} > {1}".format(radius, cutoff, self.name))
                A *= (cutoff / radius)
            params.var_dict[self.name] = A
        if opts.get('fixed') is not None:
            params.var_dict[self.name] = opts['fixed'].copy()
        if opts.get('fixed_eye', False):
            A = params.var_dict[self.name]
            k = min(A.shape)
            A[0:k, 0:k] = np.eye(k)
            params.var_dict[self.name] = A

    def properties(self):
        name = self.name
        return {name: property(lambda self_: self_.var_dict[name],
                               lambda self_, v: self_.var_dict.__setitem__(name, v))}


def _tril_to_mat(vec):
    n = int(np.sqrt(len(vec) * 2))
    mat = np.zeros((n, n), dtype=float)
    mat[_tril(n)] = vec
    return mat


class CholPrecisionVar(object):
    """Covariance X stored as the lower-triangular Cholesky factor of its precision,
    flattened: var_dict['LXinv_vec'].  Accepts X=, LXinv= or LXinv_vec= at construction.
    Reference behaviour: variables/covariance.py:19-157."""

    def __init__(self, name, dim):
        self.name, self.dim = name, dim
        self.vec, self.chol, self.prec = 'L{}inv_vec'.format(name), 'L{}inv'.format(name), '{}inv'.format(name)

    def keys(self):
        return (self.vec,)

    def parse(self, params, kwargs):
        if self.vec in kwargs:
            vec = np.array(kwargs[self.vec]).astype(float)
            n = int(np.sqrt(len(vec) * 2))
        elif self.chol in kwargs:
            L = np.array(kwargs[self.chol]).astype(float)
            if L.ndim != 2 or L.shape[0] != L.shape[1]:
                raise ValueError("{} must be square matrix".format(self.chol))
            n, vec = L.shape[0], L[_tril(L.shape[0])]
        elif self.name in kwargs:
            X = np.array(kwargs[self.name]).astype(float)
            if X.ndim != 2 or X.shape[0] != X.shape[1]:
                raise ValueError("{} must be square matrix".format(self.name))
            L = np.linalg.cholesky(np.linalg.inv(X))
            n, vec = X.shape[0], L[_tril(L.shape[0])]
        else:
            raise ValueError("{} not provided".format(self.chol))
        params.var_dict[self.vec] = vec
        params._set_check_dim(**{self.dim: n})

    def size(self, dim):
        n = dim[self.dim]
        return (n + 1) * n // 2

    def shape(self, dim):
        return (self.size(dim),)

    def project(self, params, **kwargs):
        opts = kwargs.get(self.name, {})
        if opts.get('fixed') is not None:
            setattr(params, self.vec, opts['fixed'].copy())
        if opts.get('thresh', True):
            # reflect a negative Cholesky diagonal: chol(L L' + 1e-16 I), covariance.py:68-80
            L = getattr(params, self.chol)
            L[_triu1(L.shape[0])] = 0
            if np.any(np.diag(L) < 0.0):
                logger.info("Reflecting {0}: {1} < 0.0".format(self.chol, L))
                L[:] = np.linalg.cholesky(np.dot(L, L.T) + np.eye(L.shape[0]) * 1e-16)
            setattr(params, self.chol, L)

    def properties(self):
        vec, chol, prec = self.vec, self.chol, self.prec

        def set_chol(self_, value):
            self_.var_dict[vec] = value[_tril(value.shape[0])]

        def get_prec(self_):
            L = _tril_to_mat(self_.var_dict[vec])
            return L.dot(L.T) + 1e-16 * np.eye(L.shape[0])

        def get_cov(self_):
            P = get_prec(self_)
            return P ** -1 if np.size(P) == 1 else np.linalg.inv(P)

        return {
            vec: property(lambda self_: self_.var_dict[vec],
                          lambda self_, v: self_.var_dict.__setitem__(vec, v)),
            chol: property(lambda self_: _tril_to_mat(self_.var_dict[vec]), set_chol),
            prec: property(get_prec),
            self.name: property(get_cov),
        }


class GARCHVars(object):
    """log_mu, logit_phi, logit_lambduh as (1,) arrays + derived mu, phi, lambduh,
    alpha = mu(1-phi), beta = phi*lambduh, gamma = phi(1-lambduh).
    Reference behaviour: variables/garch_var.py:21-91."""
    names = ('log_mu', 'logit_phi', 'logit_lambduh')

    def keys(self):
        return self.names

    def parse(self, params, kwargs):
        for name in self.names:
            if name not in kwargs:
                raise ValueError("{} not provided".format(name))
            params.var_dict[name] = np.atleast_1d(kwargs[name]).astype(float)

    def size(self, dim):
        return 3

    def project(self, params, **kwargs):
        for name in self.names:
            opts = kwargs.get(name, {})
            if opts.get('fixed') is not None:
                params.var_dict[name] = opts['fixed'].copy()

    def properties(self):
        props = {}
        for name in self.names:
            props[name] = property(lambda self_, n=name: self_.var_dict[n],
                                   lambda self_, v, n=name: self_.var_dict.__setitem__(n, v))
        props['mu'] = property(lambda s: np.exp(s.var_dict['log_mu']))
        props['phi'] = property(lambda s: expit(s.var_dict['logit_phi']))
        props['lambduh'] = property(lambda s: expit(s.var_dict['logit_lambduh']))
        props['alpha'] = property(lambda s: s.mu * (1 - s.phi))
        props['beta'] = property(lambda s: s.phi * s.lambduh)
        props['gamma'] = property(lambda s: s.phi * (1 - s.lambduh))
        return props


def install_properties(cls):
    """Class decorator: expose every variable spec's properties + one property per dim."""
    dims = []
    for spec in cls._specs:
        for key, prop in spec.properties().items():
            setattr(cls, key, prop)
        for d in getattr(spec, 'dims', None) or ([spec.dim] if hasattr(spec, 'dim') else []):
            if d not in dims:
                dims.append(d)
    for d in dims:
        setattr(cls, d, property(lambda self_, d=d: self_.dim[d]))
    return cls


class BaseParameters(object):
    """Parameters container: `var_dict` (name -> ndarray, insertion-ordered), `dim`.
    Interface of the reference's BaseParameters (base_parameters.py:12-103)."""
    _specs = ()

    def __init__(self, **kwargs):
        self.dim = {}
        self.var_dict = {}
        for spec in self._specs:
            spec.parse(self, kwargs)

    def _set_check_dim(self, **kwargs):
        for key, value in kwargs.items():
            if key in self.dim and self.dim[key] != value:
                raise ValueError("{0} does not match existing dims {1} != {2}".format(
                    key, value, self.dim[key]))
            self.dim[key] = value

    def as_dict(self, copy=True):
        return self.var_dict.copy() if copy else self.var_dict

    def as_vector(self):
        return self.from_dict_to_vector(self.var_dict, **self.dim)

    def from_vector(self, vector):
        self.var_dict.update(self.from_vector_to_dict(vector, **self.dim))

    @classmethod
    def from_dict_to_vector(cls, var_dict, **dim):
        chunks = []
        for spec in cls._specs:
            for key in spec.keys():
                chunks.append(np.atleast_1d(var_dict[key]).flatten())
        return np.concatenate(chunks)

    @classmethod
    def from_vector_to_dict(cls, vector, **dim):
        out, i = {}, 0
        for spec in cls._specs:
            if isinstance(spec, GARCHVars):
                for key in spec.keys():
                    out[key] = np.reshape(vector[i:i + 1], (1))
                    i += 1
            else:
                size = spec.size(dim)
                out[spec.keys()[0]] = np.reshape(vector[i:i + size], spec.shape(dim))
                i += size
        return out

    def __iadd__(self, other):
        if not isinstance(other, dict):
            raise TypeError("Addition only defined for dict not {0}".format(type(other)))
        for key in self.var_dict:
            self.var_dict[key] += other[key]
        return self

    def __add__(self, other):
        out = self.copy()
        out += other
        return out

    def __radd__(self, other):
        return self + other

    def copy(self):
        return type(self)(**deepcopy(self.var_dict))

    def project_parameters(self, **kwargs):
        for spec in self._specs:
            spec.project(self, **kwargs)
        return self

    def theta(self):
        """Raw parameters as the flat f64 vector libpfgrad takes (include/pfgrad.h layouts)."""
        return np.array([float(np.asarray(self.var_dict[k]).reshape(-1)[0])
                         for spec in self._specs for k in spec.keys()])


# ----------------------------------------------------------------------------------------
# priors
# ----------------------------------------------------------------------------------------
class WishartPrecisionPrior(object):
    """Xinv ~ Wishart(df_Xinv, scale_Xinv).  variables/covariance.py:159-317."""

    def __init__(self, name, dim):
        self.name, self.dim = name, dim
        self.scale, self.df = 'scale_{0}inv'.format(name), 'df_{0}inv'.format(name)
        self.vec, self.chol, self.prec = 'L{}inv_vec'.format(name), 'L{}inv'.format(name), '{}inv'.format(name)

    def set_hyperparams(self, prior, **kwargs):
        if self.scale not in kwargs:
            raise ValueError("{} must be provided".format(self.scale))
        if self.df not in kwargs:
            raise ValueError("{} must be provided".format(self.df))
        n, n2 = np.shape(kwargs[self.scale])
        if n != n2:
            raise ValueError("{} must be square".format(self.scale))
        prior._set_check_dim(**{self.dim: n})
        prior.hyperparams[self.scale] = kwargs[self.scale]
        prior.hyperparams[self.df] = kwargs[self.df]

    def sample(self, prior, var_dict):
        scale, df = prior.hyperparams[self.scale], prior.hyperparams[self.df]
        draw = scipy.stats.wishart(df=df, scale=scale).rvs()
        P = np.array([[draw]]) if np.size(scale) == 1 else draw
        L = np.linalg.cholesky(P)
        var_dict[self.vec] = L[_tril(L.shape[0])]

    def logprior(self, prior, parameters):
        return scipy.stats.wishart.logpdf(getattr(parameters, self.prec),
                                          df=prior.hyperparams[self.df], scale=prior.hyperparams[self.scale])

    def grad(self, prior, grad, parameters):
        scale, df = prior.hyperparams[self.scale], prior.hyperparams[self.df]
        L = getattr(parameters, self.chol)
        g = (df - L.shape[0] - 1) * np.linalg.inv(L.T) - np.linalg.solve(scale, L)
        grad[self.vec] = g[_tril(g.shape[0])]

    def default_kwargs(self, out, var, **dims):
        n = dims[self.dim]
        df = n + 1.0 + var ** -1
        out[self.scale] = np.eye(n) / df
        out[self.df] = df

    def from_parameters(self, out, parameters, var):
        P = getattr(parameters, self.prec)
        df = np.shape(P)[-1] + 1.0 + var ** -1
        out[self.scale] = P / df
        out[self.df] = df


class MatrixNormalPrior(object):
    """M ~ MatrixNormal(mean_M, rowcov = X (the paired covariance), colcov = diag(var_col_M)).
    variables/matrices.py:509-630 (square), :969-1097 (rect)."""

    def __init__(self, name, dims, row_cov):
        self.name, self.dims, self.row_cov = name, dims, row_cov
        self.mean, self.var_col = 'mean_{0}'.format(name), 'var_col_{0}'.format(name)
        self.row_vec = 'L{0}inv_vec'.format(row_cov)

    def set_hyperparams(self, prior, **kwargs):
        if self.mean not in kwargs:
            raise ValueError("{} must be provided".format(self.mean))
        if self.var_col not in kwargs:
            raise ValueError("{} must be provided".format(self.var_col))
        r, c = np.shape(kwargs[self.mean])
        if c != np.size(kwargs[self.var_col]):
            raise ValueError("prior dimensions don't match")
        if len(self.dims) == 1:
            if r != c:
                raise ValueError("{} must be square".format(self.mean))
            prior._set_check_dim(**{self.dims[0]: r})
        else:
            prior._set_check_dim(**{self.dims[0]: r, self.dims[1]: c})
        prior.hyperparams[self.mean] = kwargs[self.mean]
        prior.hyperparams[self.var_col] = kwargs[self.var_col]

    def sample(self, prior, var_dict):
        if self.row_vec not in var_dict:
            raise ValueError("Missing {}: sample {} first".format(self.row_vec, self.row_cov))
        L = _tril_to_mat(var_dict[self.row_vec])
        P = L.dot(L.T) + 1e-9 * np.eye(L.shape[0])
        var_dict[self.name] = scipy.stats.matrix_normal(
            mean=prior.hyperparams[self.mean], rowcov=np.linalg.inv(P),
            colcov=np.diag(prior.hyperparams[self.var_col])).rvs()

    def logprior(self, prior, parameters):
        mean, var_col = prior.hyperparams[self.mean], prior.hyperparams[self.var_col]
        Lrow = _tril_to_mat(parameters.var_dict[self.row_vec])
        Lcol = np.diag(var_col ** -0.5)
        X = parameters.var_dict[self.name]
        r, c = np.shape(X)
        lp = -0.5 * r * c * np.log(2 * np.pi)
        lp += -0.5 * np.sum(np.dot(Lrow.T, np.dot(X - mean, Lcol)) ** 2)
        lp += c * np.sum(np.log(np.diag(Lrow)))
        lp += r * np.sum(np.log(np.diag(Lcol)))
        return lp

    def grad(self, prior, grad, parameters):
        mean, var_col = prior.hyperparams[self.mean], prior.hyperparams[self.var_col]
        P = getattr(parameters, '{}inv'.format(self.row_cov))
        grad[self.name] = -1.0 * np.dot(P, getattr(parameters, self.name) - mean) * var_col ** -1

    def default_kwargs(self, out, var, **dims):
        r = dims[self.dims[0]]
        c = r if len(self.dims) == 1 else dims[self.dims[1]]
        out[self.mean] = np.zeros((r, c))
        out[self.var_col] = np.ones(c) * var

    def from_parameters(self, out, parameters, var):
        M = getattr(parameters, self.name)
        out[self.mean] = M.copy()
        out[self.var_col] = np.ones(M.shape[1]) * var


class GARCHVarsPrior(object):
    """mu ~ InvGamma(shape_mu, scale_mu); (1+phi)/2, (1+lambduh)/2 ~ Beta.  garch_var.py:93-189."""
    hyper = ('scale_mu', 'shape_mu', 'alpha_phi', 'beta_phi', 'alpha_lambduh', 'beta_lambduh')

    def set_hyperparams(self, prior, **kwargs):
        for name in self.hyper:
            if name not in kwargs:
                raise ValueError("{} must be provided".format(name))
            prior.hyperparams[name] = kwargs[name]

    def sample(self, prior, var_dict):
        h = prior.hyperparams
        var_dict['log_mu'] = np.log(scipy.stats.invgamma(a=h['shape_mu'], scale=h['scale_mu']).rvs())
        var_dict['logit_phi'] = logit(scipy.stats.beta(a=h['alpha_phi'], b=h['beta_phi']).rvs())
        var_dict['logit_lambduh'] = logit(scipy.stats.beta(a=h['alpha_lambduh'], b=h['beta_lambduh']).rvs())

    def logprior(self, prior, parameters):
        h = prior.hyperparams
        lp = scipy.stats.invgamma(a=h['shape_mu'], scale=h['scale_mu']).logpdf(parameters.mu)
        lp = lp + scipy.stats.beta(a=h['alpha_phi'], b=h['beta_phi']).logpdf((1 + parameters.phi) / 2.0)
        lp = lp + scipy.stats.beta(a=h['alpha_lambduh'], b=h['beta_lambduh']).logpdf((1 + parameters.lambduh) / 2.0)
        return lp

    def grad(self, prior, grad, parameters):
        h = prior.hyperparams
        mu, phi, lam = parameters.mu, parameters.phi, parameters.lambduh
        grad['log_mu'] = - h['shape_mu'] - 1 + h['scale_mu'] / mu
        grad['logit_phi'] = ((h['alpha_phi'] - 1) / (1 + phi) - (h['beta_phi'] - 1) / (1 - phi)) * phi * (1 - phi)
        grad['logit_lambduh'] = ((h['alpha_lambduh'] - 1) / (1 + lam) -
                                 (h['beta_lambduh'] - 1) / (1 - lam)) * lam * (1 - lam)

    def default_kwargs(self, out, var, **dims):
        var = min(var, 1)
        out['scale_mu'] = var + 2
        out['shape_mu'] = out['scale_mu'] + 1
        out['alpha_phi'] = 1 + 19 * var ** -1
        out['beta_phi'] = out['alpha_phi'] / 9
        out['alpha_lambduh'] = 1 + 19 * var ** -1
        out['beta_lambduh'] = out['alpha_lambduh'] / 9

    def from_parameters(self, out, parameters, var):
        self.default_kwargs(out, var)


class BasePrior(object):
    """Interface of the reference's BasePrior (base_parameters.py:136-243)."""
    _Parameters = BaseParameters
    _blocks = ()

    def __init__(self, **kwargs):
        self.dim = {}
        self.hyperparams = {}
        for block in self._blocks:
            block.set_hyperparams(self, **kwargs)

    def _set_check_dim(self, **kwargs):
        for key, value in kwargs.items():
            if key in self.dim and self.dim[key] != value:
                raise ValueError("{0} does not match existing dims {1} != {2}".format(
                    key, value, self.dim[key]))
            self.dim[key] = value

    def sample_prior(self, **kwargs):
        var_dict = {}
        for block in self._blocks:
            block.sample(self, var_dict)
        return self._Parameters(**var_dict)

    def logprior(self, parameters, **kwargs):
        total = 0.0
        for block in self._blocks:
            total = total + block.logprior(self, parameters)
        return float(np.asarray(total).reshape(-1)[0])

    def grad_logprior(self, parameters, **kwargs):
        grad = {}
        for block in self._blocks:
            block.grad(self, grad, parameters)
        return grad

    @classmethod
    def generate_default_prior(cls, var=100.0, **kwargs):
        out = {}
        for block in cls._blocks:
            block.default_kwargs(out, var, **kwargs)
        return cls(**out)

    @classmethod
    def generate_prior(cls, parameters, from_mean=False, var=1.0):
        """Prior centred on `parameters` (from_mean=True) or the default prior of its shape."""
        if not from_mean:
            return cls.generate_default_prior(var=var, **parameters.dim)
        out = {}
        for block in cls._blocks:
            block.from_parameters(out, parameters, var)
        return cls(**out)


# ----------------------------------------------------------------------------------------
# Preconditioners (SGRLD / SGRD): D(theta) per variable block
# ----------------------------------------------------------------------------------------
class MatrixPrecond(object):
    """Matrix variable with row covariance `row_cov`: D = Q (x) I, noise = LQinv^-T z, no
    correction (variables/matrices.py:632-656 square, :1099-1125 rectangular)."""

    def __init__(self, name, row_cov):
        self.name, self.row_cov = name, row_cov

    def precondition(self, out, grad, parameters):
        out[self.name] = np.dot(getattr(parameters, self.row_cov), grad[self.name])

    def noise(self, out, parameters):
        L = getattr(parameters, "L{0}inv".format(self.row_cov))
        shape = np.shape(getattr(parameters, self.name))
        out[self.name] = np.linalg.solve(L.T, np.random.normal(loc=0, size=shape))

    def correction(self, out, parameters):
        out[self.name] = np.zeros_like(getattr(parameters, self.name), dtype=float)


class CholPrecisionPrecond(object):
    """Cholesky factor of a precision: D = Qinv / 2, noise = sqrt(1/2) LQinv z, correction
    (n + 1)/2 LQinv_vec (variables/covariance.py:286-317)."""

    def __init__(self, name):
        self.inv, self.vec = "{0}inv".format(name), "L{0}inv_vec".format(name)

    def precondition(self, out, grad, parameters):
        Qinv = getattr(parameters, self.inv)
        G = np.zeros(Qinv.shape)
        G[_tril(G.shape[0])] = grad[self.vec]
        P = np.dot(0.5 * Qinv, G)
        out[self.vec] = P[_tril(P.shape[0])]

    def noise(self, out, parameters):
        L = _tril_to_mat(getattr(parameters, self.vec))
        Z = np.dot(np.sqrt(0.5) * L, np.random.normal(loc=0, size=L.shape))
        out[self.vec] = Z[_tril(Z.shape[0])]

    def correction(self, out, parameters):
        vec = getattr(parameters, self.vec)
        n = int(np.sqrt(len(vec) * 2))
        out[self.vec] = 0.5 * (n + 1) * vec


class BasePreconditioner(object):
    """precondition / precondition_noise / correction_term over the blocks, each scaled as the
    reference scales them (base_parameters.py:260-297)."""
    _blocks = ()

    def precondition(self, grad, parameters, scale=1.0, **kwargs):
        out = {}
        for blk in self._blocks:
            blk.precondition(out, grad, parameters)
        for var in out:
            out[var] = out[var] * scale
        return out

    def precondition_noise(self, parameters, scale=1.0):
        out = {}
        for blk in self._blocks:
            blk.noise(out, parameters)
        for var in out:
            out[var] = out[var] * scale ** 0.5
        return out

    def correction_term(self, parameters, scale=1.0):
        out = {}
        for blk in self._blocks:
            blk.correction(out, parameters)
        for var in out:
            out[var] = out[var] * scale
        return out
