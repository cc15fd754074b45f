// libpfgrad device code: model constants, the per-particle step (proposal, weight, score / sufficient
// statistic) of SVM / GARCH / LGSSM, additive statistic and backward kernel ratio for PaRIS / O(N^2).
#pragma once
#include "pfg_math.hpp"

namespace pfg {

// ------------------------------------------------------------------------------------
// models.  Consts are derived from raw theta exactly as the reference's Parameters
// properties do (variables/covariance.py:128-157, variables/garch_var.py:69-91).
// ------------------------------------------------------------------------------------
template <int MODEL> struct ModelDims;
template <> struct ModelDims<PFG_MODEL_SVM>   { static constexpr int NS = 1, H = 3; };
template <> struct ModelDims<PFG_MODEL_GARCH> { static constexpr int NS = 2, H = 4; };
template <> struct ModelDims<PFG_MODEL_LGSSM> { static constexpr int NS = 1, H = 4; };

template <typename REAL> struct Consts {
    // common
    REAL LRinv, iLRinv, Rinv, R, logLRinv, c0;          // c0 = -0.5*log(2pi)
    // svm / lgssm
    REAL A, C, LQinv, iLQinv, Qinv;
    REAL opt_sd, opt_prec, opt_var, opt_logvar;         // lgssm optimal kernel
    // garch
    REAL mu, phi, lam, alpha, beta, gamma;
    REAL logalpha, logLQinv;                            // PaRIS backward kernel
};

template <int MODEL, typename REAL>
__device__ __forceinline__ Consts<REAL> make_consts(const double *__restrict__ th) {
    Consts<double> d = {};
    d.c0 = -0.5 * LOG_2PI;
    double LRinv;
    if (MODEL == PFG_MODEL_SVM) {
        d.A = th[0]; d.LQinv = th[1]; LRinv = th[2];
    } else if (MODEL == PFG_MODEL_LGSSM) {
        d.A = th[0]; d.C = th[1]; d.LQinv = th[2]; LRinv = th[3];
    } else {
        LRinv = th[3];
        d.mu = exp(th[0]);
        d.phi = 1.0 / (1.0 + exp(-th[1]));
        d.lam = 1.0 / (1.0 + exp(-th[2]));
        d.alpha = d.mu * (1.0 - d.phi);
        d.beta = d.phi * d.lam;
        d.gamma = d.phi * (1.0 - d.lam);
    }
    d.LRinv = LRinv;
    d.iLRinv = 1.0 / LRinv;
    d.Rinv = LRinv * LRinv + 1e-16;
    d.R = 1.0 / d.Rinv;
    d.logLRinv = log(LRinv);
    if (MODEL != PFG_MODEL_GARCH) {
        d.iLQinv = 1.0 / d.LQinv;
        d.Qinv = d.LQinv * d.LQinv + 1e-16;
        d.logLQinv = log(d.LQinv);
    } else {
        d.logalpha = log(d.alpha);
    }
    if (MODEL == PFG_MODEL_LGSSM) {
        d.opt_prec = d.Qinv + (d.C * d.C) * d.Rinv;
        d.opt_sd = pow(d.opt_prec, -0.5);
        d.opt_var = 1.0 / d.Qinv + 1.0 / d.Rinv;
        d.opt_logvar = log(d.opt_var);
    }
    // wave-uniform by construction: pin every constant in scalar registers (frees ~2 VGPRs each)
    {
        double *f = reinterpret_cast<double *>(&d);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(d) / sizeof(double)); ++q) f[q] = uniform_f64(f[q]);
    }
    Consts<REAL> c;
    c.LRinv = (REAL)d.LRinv; c.iLRinv = (REAL)d.iLRinv; c.Rinv = (REAL)d.Rinv; c.R = (REAL)d.R;
    c.logLRinv = (REAL)d.logLRinv; c.c0 = (REAL)d.c0;
    c.A = (REAL)d.A; c.C = (REAL)d.C; c.LQinv = (REAL)d.LQinv; c.iLQinv = (REAL)d.iLQinv;
    c.Qinv = (REAL)d.Qinv;
    c.opt_sd = (REAL)d.opt_sd; c.opt_prec = (REAL)d.opt_prec; c.opt_var = (REAL)d.opt_var;
    c.opt_logvar = (REAL)d.opt_logvar;
    c.mu = (REAL)d.mu; c.phi = (REAL)d.phi; c.lam = (REAL)d.lam;
    c.alpha = (REAL)d.alpha; c.beta = (REAL)d.beta; c.gamma = (REAL)d.gamma;
    c.logalpha = (REAL)d.logalpha; c.logLQinv = (REAL)d.logLQinv;
    return c;
}

// One particle: parent state xp -> proposal x' (Kernel.rv), log weight (Kernel.reweight) and
// additive statistic (STAT = PFG_STAT_SCORE: complete-data score; otherwise the sufficient
// statistics), all from the same registers, straight-line.  add[] is NOT yet scaled by weight_t.
template <int MODEL, int KERNEL, int STAT, typename REAL, typename MATH>
__device__ __forceinline__ void particle_step(const Consts<REAL> &c, const MATH &mth, const REAL *xp,
                                              REAL y, REAL z, REAL *xn, REAL &lw, REAL *add) {
    constexpr int H = ModelDims<MODEL>::H;
    const REAL half = (REAL)0.5;
#pragma unroll
    for (int h = 0; h < H; ++h) add[h] = (REAL)0;
    if (MODEL == PFG_MODEL_SVM) {
        // svm/kernels.py:34-37, :56-62; svm/helper.py:342-348
        REAL xpA = xp[0] * c.A;
        REAL x1 = c.iLQinv * z + xpA;
        REAL e = mth.exp_finite(-x1);          // x1 is finite
        REAL y2 = y * y;
#ifdef PFG_FAST_ALGEBRA
        // device-generator units (no operation-order parity to keep): the same expressions with
        // the wave-uniform factors of the step collected (they are computed once per step)
        const REAL k0 = c.c0 + c.logLRinv, ke = (-half * y2) * c.Rinv;
        lw = fma(ke, e, fma(-half, x1, k0));
#else
        lw = ((c.c0 + ((-half * y2) * e) * c.Rinv) + c.logLRinv) + (-half * x1);
#endif
        xn[0] = x1;
        if (STAT == PFG_STAT_SCORE) {
            REAL dx = x1 - c.A * xp[0];
            add[2] = (c.Qinv * dx) * xp[0];
            add[1] = c.iLQinv - (dx * dx) * c.LQinv;
#ifdef PFG_FAST_ALGEBRA
            add[0] = fma(-(y2 * c.LRinv), e, c.iLRinv);
#else
            REAL dy2 = y2 * e;                       // y^2 / exp(x')
            add[0] = c.iLRinv - dy2 * c.LRinv;
#endif
        } else {
            add[0] = x1; add[1] = x1 * x1; add[2] = xp[0] * x1;
        }
    } else if (MODEL == PFG_MODEL_LGSSM) {
        REAL x1;
        if (KERNEL == PFG_KERNEL_PRIOR) {
            // lgssm/kernels.py:30-33, :58-62
            x1 = c.iLQinv * z + xp[0] * c.A;
            REAL diff = y - c.C * x1;
            lw = (c.c0 + (-half * (diff * diff)) * c.Rinv) + c.logLRinv;
        } else {
            // lgssm/kernels.py:87-97, :117-120
#ifdef PFG_FAST_ALGEBRA
            // device-generator units: the two divisors are wave-uniform -> reciprocals, once per step
            const REAL rprec = (REAL)1 / c.opt_prec, rvar = (REAL)1 / c.opt_var;
            REAL mp = (xp[0] * c.A) * c.Qinv + (y * c.C) * c.Rinv;
            x1 = fma(c.opt_sd, z, mp * rprec);
            REAL diff = y - c.A * xp[0];
            lw = fma(-half * (diff * diff), rvar, -half * (REAL)LOG_2PI - half * c.opt_logvar);
#else
            REAL mp = (xp[0] * c.A) * c.Qinv + (y * c.C) * c.Rinv;
            x1 = c.opt_sd * z + mp / c.opt_prec;
            REAL diff = y - c.A * xp[0];
            lw = ((-half * (diff * diff)) / c.opt_var - half * (REAL)LOG_2PI) - half * c.opt_logvar;
#endif
        }
        xn[0] = x1;
        if (STAT == PFG_STAT_SCORE) {
            // lgssm/helper.py:1270-1277, order [LRinv, LQinv, C, A]
            REAL dx = x1 - c.A * xp[0];
            add[3] = (c.Qinv * dx) * xp[0];
            add[1] = c.iLQinv - (dx * dx) * c.LQinv;
            REAL dy = y - c.C * x1;
            add[2] = (c.Rinv * dy) * x1;
            add[0] = c.iLRinv - (dy * dy) * c.LRinv;
        } else {
            add[0] = x1; add[1] = x1 * x1; add[2] = xp[0] * x1;
        }
    } else {
        // garch/kernels.py:60-68 / :146-156, reweight :83-88 / :172-178
        REAL xx = xp[0] * xp[0];
        REAL s2 = (c.alpha + c.beta * xx) + c.gamma * xp[1];
        REAL x1;
        if (KERNEL == PFG_KERNEL_PRIOR) {
#ifdef PFG_FAST_ALGEBRA
            x1 = mth.sqrt_pos(s2) * z;
#else
            x1 = mth.sqrt(s2) * z;
#endif
            REAL diff = y - x1;
            lw = (c.c0 + (-half * (diff * diff)) * c.Rinv) + c.logLRinv;
        } else {
#ifdef PFG_FAST_ALGEBRA
            // device-generator units: 1/(Rinv + 1/s2) = s2 R / (s2 + R), so the proposal variance shares its
            // reciprocal with the weight; s2 > 0 and s2 + R > 0 are finite: rcp + Newton, rsq + Newton
            // (two reciprocals and no IEEE division per particle instead of four divisions)
            const REAL v2 = s2 + c.R;
            const REAL rv2 = mth.rcp_pos(v2);
            const REAL var = (s2 * c.R) * rv2;
            x1 = fma(mth.sqrt_pos(var), z, var * (y * c.Rinv));
            lw = fma(-half * (y * y), rv2, c.c0) + (-half * mth.log(v2));
#else
            REAL var = (REAL)1 / (c.Rinv + (REAL)1 / s2);
            REAL mean = var * (y * c.Rinv);
            x1 = mean + mth.sqrt(var) * z;
            REAL v2 = s2 + c.R;
            lw = (c.c0 + (-half * (y * y)) / v2) + (-half * mth.log(v2));
#endif
        }
        xn[0] = x1; xn[1] = s2;
        if (STAT == PFG_STAT_SCORE) {
            // garch/helper.py:350-370, order [LRinv, log_mu, logit_phi, logit_lambduh]
            REAL v = s2;
#ifdef PFG_FAST_ALGEBRA
            const REAL rv = mth.rcp_pos(v);
            const REAL gv = (-half * (v - x1 * x1)) * (rv * rv);
            const REAL omp = (REAL)1 - c.phi, oml_ = (REAL)1 - c.lam;
            add[1] = gv * (omp * c.mu);
            add[2] = (gv * fma(c.lam, xx, fma(oml_, xp[1], -c.mu))) * (omp * c.phi);
            add[3] = (gv * (xx - xp[1])) * ((c.phi * oml_) * c.lam);
            REAL dy = y - x1;
            add[0] = fma(-(dy * dy), c.LRinv, c.iLRinv);
#else
            REAL gv = (-half * (v - x1 * x1)) / (v * v);
            add[1] = (gv * ((REAL)1 - c.phi)) * c.mu;
            add[2] = ((gv * ((-c.mu + c.lam * xx) + ((REAL)1 - c.lam) * xp[1])) * ((REAL)1 - c.phi)) * c.phi;
            add[3] = (((gv * c.phi) * (xx - xp[1])) * ((REAL)1 - c.lam)) * c.lam;
            REAL dy = y - x1;
            add[0] = c.iLRinv - (dy * dy) * c.LRinv;
#endif
        } else {
            REAL x2 = x1 * x1;
            add[0] = x1; add[1] = x2; add[2] = x2 * x2;
        }
    }
}

// Additive statistic h(parent, child) alone (PaRIS evaluates it for rewired parents): the same
// expressions as in particle_step.  `aux` carries the child's sub-expression the proposal step
// already has (SVM: exp(-x')).
template <int MODEL, int STAT, typename REAL>
__device__ __forceinline__ void additive_stat(const Consts<REAL> &c, const REAL *xp, const REAL *xn, REAL y,
                                              REAL aux, REAL *add) {
    constexpr int H = ModelDims<MODEL>::H;
    const REAL half = (REAL)0.5;
#pragma unroll
    for (int h = 0; h < H; ++h) add[h] = (REAL)0;
    if (STAT != PFG_STAT_SCORE) {
        if (MODEL == PFG_MODEL_GARCH) { REAL x2 = xn[0] * xn[0]; add[0] = xn[0]; add[1] = x2; add[2] = x2 * x2; }
        else { add[0] = xn[0]; add[1] = xn[0] * xn[0]; add[2] = xp[0] * xn[0]; }
        return;
    }
    if (MODEL == PFG_MODEL_SVM) {
        REAL dx = xn[0] - c.A * xp[0];
        add[2] = (c.Qinv * dx) * xp[0];
        add[1] = c.iLQinv - (dx * dx) * c.LQinv;
        add[0] = c.iLRinv - ((y * y) * aux) * c.LRinv;
    } else if (MODEL == PFG_MODEL_LGSSM) {
        REAL dx = xn[0] - c.A * xp[0];
        add[3] = (c.Qinv * dx) * xp[0];
        add[1] = c.iLQinv - (dx * dx) * c.LQinv;
        REAL dy = y - c.C * xn[0];
        add[2] = (c.Rinv * dy) * xn[0];
        add[0] = c.iLRinv - (dy * dy) * c.LRinv;
    } else {
        REAL xx = xp[0] * xp[0];
        REAL v = xn[1];
        REAL gv = (-half * (v - xn[0] * xn[0])) / (v * v);
        add[1] = (gv * ((REAL)1 - c.phi)) * c.mu;
        add[2] = ((gv * ((-c.mu + c.lam * xx) + ((REAL)1 - c.lam) * xp[1])) * ((REAL)1 - c.phi)) * c.phi;
        add[3] = (((gv * c.phi) * (xx - xp[1])) * ((REAL)1 - c.lam)) * c.lam;
        REAL dy = y - xn[0];
        add[0] = c.iLRinv - (dy * dy) * c.LRinv;
    }
}

// log q(child | parent) - max q: the accept-reject exponent of PaRIS backward sampling
// (Kernel.prior_log_density - get_prior_log_density_max; kernels.py:102-138, garch/kernels.py:20-47)
template <int MODEL, typename REAL, typename MATH>
__device__ __forceinline__ REAL backward_log_ratio(const Consts<REAL> &c, const MATH &mth, const REAL *xp,
                                                   const REAL *xn) {
    const REAL half = (REAL)0.5;
    if (MODEL == PFG_MODEL_GARCH) {
        REAL s2 = (c.alpha + c.beta * (xp[0] * xp[0])) + c.gamma * xp[1];
        REAL ll = ((-half * (xn[0] * xn[0])) / s2 - half * (REAL)LOG_2PI) - half * mth.log(s2);
        return ll - (-half * (REAL)LOG_2PI - half * c.logalpha);
    }
    REAL diff = xn[0] - c.A * xp[0];
    REAL ll = ((-half * (diff * diff)) * c.Qinv + -half * (REAL)LOG_2PI) + c.logLQinv;
    return ll - (-half * (REAL)LOG_2PI + c.logLQinv);
}

}  // namespace pfg
