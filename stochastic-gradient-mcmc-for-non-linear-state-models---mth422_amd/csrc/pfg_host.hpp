// Host-side declarations shared by the translation units of libpfgrad.so: the context, error
// plumbing, the kernel-variant table and the per-(model, kernel) launch entry the instantiation
// units (pfg_inst_*.hip) define.  Splitting the ~250 kernel instantiations over five units lets
// the build compile them in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "pfgrad.h"

namespace pfg_host {

inline thread_local std::string g_create_error;

struct Arena {           // growable device buffer
    void *ptr = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 4096;
        hipError_t e = hipMalloc(&ptr, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; cap = 0; }
};

struct HostArena {       // growable PINNED host buffer (hipHostMalloc): H2D / D2H copies from it are truly
    double *ptr = nullptr;   // asynchronous and run at link speed; a pageable std::vector is staged by the runtime
    size_t cap = 0;          // in doubles
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (ptr) (void)hipHostFree(ptr);
        ptr = nullptr; cap = 0;
        const size_t want = n + n / 4 + 512;
        void *p = nullptr;
        hipError_t e = hipHostMalloc(&p, want * sizeof(double), hipHostMallocDefault);
        if (e == hipSuccess) { ptr = static_cast<double *>(p); cap = want; }
        return e;
    }
    double *data() { return ptr; }
    double &operator[](size_t i) { return ptr[i]; }
    void release() { if (ptr) (void)hipHostFree(ptr); ptr = nullptr; cap = 0; }
};

}  // namespace pfg_host

struct pfg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    pfg_host::Arena in, out, desc, scratch, work;      // work: device-only buffers (elementwise-statistics pass)
    pfg_host::HostArena h_in, h_out;
    std::vector<double> h_in_pageable;   // staging when the pinned arena cannot be had (hipHostMalloc refused)
    std::vector<pfg_dev_problem> h_desc;
    const char *last_variant = "none";   // tag of the kernel variant the latest dispatch launched
    bool last_traced = false;            // ... and whether that was a trace-honouring instantiation
    bool score1 = false;                 // the dispatch in flight is a PFG_SMOOTHER_POYIADJIS_N launch (see launch_one)
    // largest dynamic-LDS size hipFuncAttributeMaxDynamicSharedMemorySize has been set to, per kernel: the
    // attribute is per (function, device) and a context is bound to one device
    std::unordered_map<const void *, size_t> lds_set;
};

namespace pfg_host {

inline int fail(pfg_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define PFG_HIP(ctx, call)                                                            \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess)                                                         \
            return fail(ctx, e_ == hipErrorOutOfMemory ? PFG_ERR_NOMEM : PFG_ERR_DEVICE, \
                        std::string(#call) + ": " + hipGetErrorString(e_));           \
    } while (0)

constexpr size_t kLdsLimit = 160 * 1024;
// variant ids below zero: kernels other than the LDS-resident table entries
constexpr int kVariantMem = -2;     // large-N kernel (state in an HBM scratch)
constexpr int kVariantParis = -3, kVariantSystematic = -4, kVariantN2 = -5, kVariantBig = -6;
constexpr int kVariantMemLw4 = -8;  // large-N kernel, N <= 4096, no predictive statistic: log-weights in registers
constexpr int kVariantGrid = -7;    // whole-GPU window for N above the one-workgroup kernels' maximum (pfg_grid_kernel.hpp)

// Launch of every kernel of one (model, proposal kernel, generator): defined (and explicitly
// instantiated) in pfg_inst_*.hip via pfg_launch.hpp, declared here for the dispatcher in pfgrad.hip.
template <int MODEL, int KERNEL, int RNG>
int launch_mkr(pfg_ctx *ctx, int dtype, int v, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st, bool traced);

// the whole-GPU window of one (model, kernel, generator): T_max + 2 (REPLAY: 5 T_max + 2) launches, see pfg_launch.hpp
template <int MODEL, int KERNEL, int RNG>
int launch_grid_mkr(pfg_ctx *ctx, int dtype, int n_max, int t_max, int B, const pfg_dev_problem *dp, hipStream_t st, int phase);

}  // namespace pfg_host
