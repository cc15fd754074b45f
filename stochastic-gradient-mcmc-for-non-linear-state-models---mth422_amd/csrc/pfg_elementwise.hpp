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
#include "pfg_models.hpp"

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

// Poyiadjis O(N^2) (pf.py:84-136) with elementwise statistics: child i averages over ALL parents with the
// backward weights  bw_ij = softmax_j(logw_j + log q(x'_i | x_j)):
//     S'[i][:] = sum_j bw_ij S[j][:]        block t:  += w_t sum_j bw_ij h(x_j, x'_i)
// One workgroup per child: the N backward weights go to LDS once, then every thread owns columns and
// walks the parents (rows of S stream through L2, coalesced over the column axis).  N <= 4096.
template <int MODEL>
__global__ __launch_bounds__(256) void ews_n2_step_kernel(int N, int Wd, double wt, int col0,
                                                          const double *__restrict__ theta,
                                                          const double *__restrict__ x_t, const double *__restrict__ logw,
                                                          const double *__restrict__ x_next,
                                                          const double *__restrict__ S, double *__restrict__ Sn) {
    constexpr int NS = ModelDims<MODEL>::NS;
    __shared__ double bw[4096];
    __shared__ double red[4];
    __shared__ double bx_s;
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const Consts<double> c = make_consts<MODEL, double>(theta);
    const Math<double, false> mth = {};
    double xc[NS];
#pragma unroll
    for (int d = 0; d < NS; ++d) xc[d] = x_next[(size_t)i * NS + d];
    // log weights of the parents as seen from this child, their maximum
    double mx = -INFINITY;
    for (int j = tid; j < N; j += 256) {
        double xj[NS];
#pragma unroll
        for (int d = 0; d < NS; ++d) xj[d] = x_t[(size_t)j * NS + d];
        const double v = logw[j] + backward_log_ratio<MODEL, double>(c, mth, xj, xc);
        bw[j] = v;
        mx = fmax(mx, v);
    }
    mx = wave_max(mx);
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    __syncthreads();
    double sum = 0.0, sx = 0.0;
    for (int j = tid; j < N; j += 256) {
        const double e = exp(bw[j] - mx);
        bw[j] = e;
        sum += e;
        sx = fma(e, x_t[(size_t)j * NS], sx);
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    const double inv = 1.0 / ((red[0] + red[1]) + (red[2] + red[3]));
    __syncthreads();
    sx = wave_sum(sx);
    if (lane == 0) red[wave] = sx;
    __syncthreads();
    if (tid == 0) bx_s = ((red[0] + red[1]) + (red[2] + red[3])) * inv;       // sum_j bw_ij x_j
    __syncthreads();
    const double bx = bx_s;
    for (int col = tid; col < Wd; col += 256) {
        double a0 = 0.0, a1 = 0.0;
        int j = 0;
        for (; j + 1 < N; j += 2) {
            a0 = fma(bw[j], S[(size_t)j * Wd + col], a0);
            a1 = fma(bw[j + 1], S[(size_t)(j + 1) * Wd + col], a1);
        }
        if (j < N) a0 = fma(bw[j], S[(size_t)j * Wd + col], a0);
        double v = (a0 + a1) * inv;
        if (col0 >= 0 && col >= col0 && col < col0 + 3) {
            const int q = col - col0;
            const double xn = xc[0];
            double h;
            if (MODEL == PFG_MODEL_GARCH) h = q == 0 ? xn : (q == 1 ? xn * xn : (xn * xn) * (xn * xn));
            else h = q == 0 ? xn : (q == 1 ? xn * xn : bx * xn);
            v += wt * h;
        }
        Sn[(size_t)i * Wd + col] = v;
    }
}

}  // namespace pfg
