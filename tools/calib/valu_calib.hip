// VALU issue calibration for gfx950: known instruction streams (v_fma_f64 / v_fma_f32 / v_exp_f32 /
// v_xor_b32 / a 50:50 f64:b32 mix) at 1, 2 and 4 waves per SIMD.  Prints, per case, the cycles one
// SIMD spends per wave-instruction (from s_memtime inside the kernel and from hipEvents), and is
// run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES ...` to read what the
// counters report for a stream whose instruction count and duration are known.
// Build: hipcc -O3 --offload-arch=gfx950 tools/calib/valu_calib.hip -o tools/calib/valu_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int UNROLL = 18;     // independent accumulators per lane (18: two periods of the 2-in-9 mix)

template <int KIND>
__global__ __launch_bounds__(256) void stream_kernel(int iters, double *out, unsigned long long *cyc) {
    extern __shared__ unsigned char pad[];
    double a64[UNROLL];
    float a32[UNROLL];
    unsigned u32[UNROLL];
    const double b64 = 1.0000001, c64 = 1e-9;
    const float b32 = 1.0001f, c32 = 1e-6f;
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) { a64[j] = threadIdx.x + j; a32[j] = threadIdx.x + j; u32[j] = threadIdx.x * 7 + j; }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < UNROLL; ++j) {
            if (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a64[j]) : "v"(b64), "v"(c64));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a32[j]) : "v"(b32), "v"(c32));
            if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a32[j]));
            if (KIND == 3) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u32[j]) : "v"(u32[(j + 1) % UNROLL]));
            if (KIND == 4) {   // alternate f64 / b32
                if (j & 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a64[j]) : "v"(b64), "v"(c64));
                else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u32[j]) : "v"(u32[(j + 2) % UNROLL]));
            }
            if (KIND == 7) {   // the bench kernel's class mix: 22 % fp64 arithmetic, 78 % 32-bit (2 of every 9)
                if (j % 9 == 2 || j % 9 == 6) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a64[j % 8]) : "v"(b64), "v"(c64));
                else if (j & 1) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u32[j]) : "v"(u32[(j + 3) % UNROLL]));
                else asm volatile("v_add_u32 %0, %0, %1" : "+v"(u32[j]) : "v"(u32[(j + 5) % UNROLL]));
            }
            if (KIND == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a64[j]) : "v"(c64));
            if (KIND == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a64[j]) : "v"(b64));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int j = 0; j < UNROLL; ++j) s += a64[j] + (double)a32[j] + (double)u32[j];
    if (s == 123.456) out[0] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
        cyc[gridDim.x * 4 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r1 - r0;     // 100 MHz ticks
        cyc[gridDim.x * 8 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r0;          // absolute start / end: the launch's span
        cyc[gridDim.x * 12 + blockIdx.x * 4 + (threadIdx.x >> 6)] = r1;
    }
}

static double inst_per_wave_of(int iters) { return (double)iters * UNROLL; }

template <int KIND>
void run(const char *name, int blocks_per_cu, int iters) {
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    const int grid = cus * blocks_per_cu;
    const size_t lds = (size_t)(160 * 1024 / blocks_per_cu) - 1024;     // pins blocks_per_cu workgroups per CU
    double *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 8)); CK(hipMalloc(&cyc, (size_t)grid * 16 * 8));
    auto k = stream_kernel<KIND>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, iters / 8, out, cyc);   // warm-up
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds, 0, iters, out, cyc);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4), hr((size_t)grid * 4);
    CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), cyc + (size_t)grid * 4, hr.size() * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> hs((size_t)grid * 4), he((size_t)grid * 4);
    CK(hipMemcpy(hs.data(), cyc + (size_t)grid * 8, hs.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(he.data(), cyc + (size_t)grid * 12, he.size() * 8, hipMemcpyDeviceToHost));
    const double span_ticks = (double)(*std::max_element(he.begin(), he.end()) - *std::min_element(hs.begin(), hs.end()));
    std::sort(h.begin(), h.end());
    std::sort(hr.begin(), hr.end());
    const double med = (double)h[h.size() / 2];
    const double ghz = med / (double)hr[hr.size() / 2] * 0.1;     // s_memtime ticks per 100 MHz s_memrealtime tick
    // the same cost in REAL time (s_memrealtime is a constant 100 MHz clock): independent of what an s_memtime tick is
    const double ns_per_inst_simd = (double)hr[hr.size() / 2] * 10.0 / (inst_per_wave_of(iters) * blocks_per_cu);
    const double inst_per_wave = (double)iters * UNROLL;
    // waves per SIMD = blocks_per_cu (256 threads = 4 waves = one per SIMD)
    const double cyc_per_inst_simd = med / (inst_per_wave * blocks_per_cu);
    const double total_inst = inst_per_wave * 4.0 * grid;
    printf("%-14s waves/SIMD %d  grid %5d  %8.3f ms  wave-ticks(median, s_memtime) %.3e  => %.2f s_memtime ticks = %.3f ns per wave-instruction per SIMD"
           "  s_memtime / s_memrealtime = %.3f GHz (loop %.3f ms)  wave-instructions %.4e\n",
           name, blocks_per_cu, grid, ms, med, cyc_per_inst_simd, ns_per_inst_simd,
           ghz, med / ghz * 1e-6, total_inst);
    // the SIMDs arbitrate by age: the waves of a SIMD do not share it equally, the oldest ends first.  What a SIMD needs
    // for blocks_per_cu waves is the time from the first start to the last end (every SIMD of the GPU holds the same
    // load), not the median wave's own duration -- which under-states the cost per instruction
    const double ns_span = span_ticks * 10.0 / (inst_per_wave * blocks_per_cu);
    printf("    wave durations min / median / max = %.3e / %.3e / %.3e s_memtime ticks; launch span %.3f ms => %.3f ns = %.2f ticks per wave-instruction per SIMD (span-based)\n",
           (double)h.front(), med, (double)h.back(), span_ticks * 1e-5, ns_span, ns_span * ghz);
    fflush(stdout);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    if (argc > 2) {
        // one long stream at 4 waves per SIMD (for a clock sampler running beside it): valu_calib <iters> <kind>
        const int kind = atoi(argv[2]);
        if (kind == 0) run<0>("v_fma_f64", 4, iters);
        if (kind == 1) run<1>("v_fma_f32", 4, iters);
        if (kind == 3) run<3>("v_xor_b32", 4, iters);
        return 0;
    }
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w, iters);
        run<5>("v_add_f64", w, iters);
        run<6>("v_mul_f64", w, iters);
        run<1>("v_fma_f32", w, iters);
        run<2>("v_exp_f32", w, iters);
        run<3>("v_xor_b32", w, iters);
        run<4>("f64:b32 1:1", w, iters);
        run<7>("f64:b32 2:7", w, iters);
    }
    return 0;
}
