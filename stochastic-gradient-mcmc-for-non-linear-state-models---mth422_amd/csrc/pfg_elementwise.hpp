// libpfgrad device code: elementwise sufficient statistics of a window (pf_latent_var_distr).
//
// The reference's `elementwise_statistic=True` run (buffered_smoother.py:64-65, 201-210) widens the
// additive statistic to 3 (tL - t1) columns -- block t holds [x', x'^2, x x'] (GARCH [x', x'^2, x'^4],
// lgssm/helper.py:1338-1363, garch/helper.py:414-430) of window step t -- and carries the N x 3L matrix
// through the smoother's recursion (pf.py:138-181 nemeth / poyiadjis_N, :183-258 paris).  The filter's
// trajectory does not depend on the statistic, so it runs ONCE (recording particles, log-weights and the
// parent of every child: the resampling ancestor, or PaRIS' backward-sampled parents) and this second pass
// streams the matrix through HBM, one launch per timestep:
//     S'[i][:] = lambda/Nt sum_j S[J_j(i)][:] + (1 - lambda) Sbar[:]        Sbar = sum_k w_k S[k][:]
//     S'[i][3(t-t1)..+3] += w_t / Nt sum_j h(x_t[J_j(i)], x_{t+1}[i])        (t inside the window)
// Row-major [N][3L]: a child copies whole parent rows, so every access is coalesced over the column axis;
// 16 B per matrix element per step of traffic (read parent row, write own row).  fp64 throughout.
#pragma once
#include "pfg_math.hpp"

namespace pfg {

// normalised weights of one timestep: w = softmax(logw) (log_normalize, pf.py:374-377).  One workgroup.
__global__ __launch_bounds__(1024) void ews_softmax_kernel(int N, const double *__restrict__ logw, double *__restrict__ w) {
    __shared__ double red[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double m = -INFINITY;
    for (int i = tid; i < N; i += 1024) m = fmax(m, logw[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = red[0];
    for (int q = 1; q < 16; ++q) m = fmax(m, red[q]);
    __syncthreads();
    double s = 0.0;
    for (int i = tid; i < N; i += 1024) s += exp(logw[i] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    double tot = 0.0;
    for (int q = 0; q < 16; ++q) tot += red[q];
    for (int i = tid; i < N; i += 1024) w[i] = exp(logw[i] - m) / tot;
}

// out[c] = sum_k w[k] S[k][c]: one thread per column, rows walked in order (coalesced over columns;
// fixed summation order: reproducible)
__global__ __launch_bounds__(256) void ews_colsum_kernel(int N, int Wd, const double *__restrict__ S,
                                                         const double *__restrict__ w, double *__restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Wd) return;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int k = 0;
    for (; k + 3 < N; k += 4) {
        a0 = fma(w[k], S[(size_t)k * Wd + c], a0);
        a1 = fma(w[k + 1], S[(size_t)(k + 1) * Wd + c], a1);
        a2 = fma(w[k + 2], S[(size_t)(k + 2) * Wd + c], a2);
        a3 = fma(w[k + 3], S[(size_t)(k + 3) * Wd + c], a3);
    }
    for (; k < N; ++k) a0 = fma(w[k], S[(size_t)k * Wd + c], a0);
    out[c] = (a0 + a1) + (a2 + a3);
}

// one timestep of the recursion; grid = (ceil(Wd / 256), N): block row i = child i
//   parents: [Nt][N] int32 for this step; x_t [N][NS], x_next [N][NS]; col0 = 3 (t - t1) or -1 outside the window
template <int MODEL>
__global__ __launch_bounds__(256) void ews_step_kernel(int N, int Wd, int Nt, double lam, double wt, int col0,
                                                       const int32_t *__restrict__ parents,
                                                       const double *__restrict__ x_t, const double *__restrict__ x_next,
                                                       const double *__restrict__ Sbar, const double *__restrict__ S,
                                                       double *__restrict__ Sn) {
    constexpr int NS = MODEL == PFG_MODEL_GARCH ? 2 : 1;
    const int i = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Wd) return;
    double acc = 0.0;
    for (int j = 0; j < Nt; ++j) {
        const int J = parents[(size_t)j * N + i];
        acc += S[(size_t)J * Wd + c];
    }
    const double invNt = 1.0 / (double)Nt;
    double v = acc * invNt;
    if (lam != 1.0) v = lam * v + (1.0 - lam) * Sbar[c];
    if (col0 >= 0 && c >= col0 && c < col0 + 3) {
        const int q = c - col0;
        const double xn = x_next[(size_t)i * NS];
        double h = 0.0;
        for (int j = 0; j < Nt; ++j) {
            const int J = parents[(size_t)j * N + i];
            const double xp = x_t[(size_t)J * NS];
            double hv;
            if (MODEL == PFG_MODEL_GARCH) hv = q == 0 ? xn : (q == 1 ? xn * xn : (xn * xn) * (xn * xn));
            else hv = q == 0 ? xn : (q == 1 ? xn * xn : xp * xn);
            h += hv;
        }
        v += wt * (h * invNt);
    }
    Sn[(size_t)i * Wd + c] = v;
}

}  // namespace pfg
