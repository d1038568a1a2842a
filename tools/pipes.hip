// pipes: do the matrix pipe, the VALU and the transcendental unit of a gfx950 SIMD overlap?  (What bounds enc_attn_kernel.)
// Every test runs ITER iterations of a register-only loop body on 256 workgroups x 256 threads (one wave per SIMD) or x 512
// (two per SIMD) and reports cycles per loop body per wave at the clock the box runs (s_memtime-free: wall time x an
// assumed 2.4 GHz is printed next to the time itself, ratios between tests are what matters).
//   fma     32 independent v_fma_f32
//   exp     32 independent v_exp_f32
//   pkfma   16 v_pk_fma_f32 (the same 32 FMAs)
//   mfma    8 v_mfma_f32_32x32x16_f16 (two accumulator chains)
//   mfma+exp, mfma+fma   both bodies in ONE wave, independent of each other
//   split   waves 0-3 of a 512-thread workgroup run mfma, waves 4-7 run exp (different waves of the same SIMD)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

enum { T_FMA, T_EXP, T_PKFMA, T_MFMA, T_MFMA_EXP, T_MFMA_FMA, T_SPLIT };

template <int T>
__global__ __launch_bounds__(512) void body(float *sink, int iters) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a[32];
    f32x2 p[16];
#pragma unroll
    for (int i = 0; i < 32; i++) a[i] = 0.001f * (float)(lane + i);
#pragma unroll
    for (int i = 0; i < 16; i++) p[i] = (f32x2){0.001f * (float)(lane + i), 0.002f * (float)(lane + i)};
    f32x16 c0, c1;
#pragma unroll
    for (int i = 0; i < 16; i++) { c0[i] = 0.f; c1[i] = 0.f; }
    half8 x, y;
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = (half_t)(0.01f * (float)(lane & 7)); y[i] = (half_t)(0.02f * (float)(i + 1)); }
    const bool do_mfma = T == T_MFMA || T == T_MFMA_EXP || T == T_MFMA_FMA || (T == T_SPLIT && w < 4);
    const bool do_exp = T == T_EXP || T == T_MFMA_EXP || (T == T_SPLIT && w >= 4);
    const bool do_fma = T == T_FMA || T == T_MFMA_FMA;
    for (int it = 0; it < iters; it++) {
        if (do_mfma) {
#pragma unroll
            for (int k = 0; k < 4; k++) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(y, x, c1, 0, 0, 0);
            }
        }
        if (do_exp) {
#pragma unroll
            for (int i = 0; i < 32; i++) a[i] = __builtin_amdgcn_exp2f(a[i]);
        }
        if (do_fma) {
#pragma unroll
            for (int i = 0; i < 32; i++) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
        }
        if (T == T_PKFMA) {
            const f32x2 m = {0.999f, 0.999f}, b = {0.001f, 0.001f};
#pragma unroll
            for (int i = 0; i < 16; i++) p[i] = __builtin_elementwise_fma(p[i], m, b);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 16; i++) s += p[i][0] + p[i][1] + c0[i] + c1[i];
    if (s == 12345.678f) sink[0] = s;
}

template <int T>
static double run(const char *name, int threads, int iters, float *sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(body<T>, dim3(256), dim3(threads), 0, 0, sink, iters);
    CK(hipDeviceSynchronize());
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(body<T>, dim3(256), dim3(threads), 0, 0, sink, iters);
    hipEventRecord(b, 0); CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ns_per_body = ms * 1e6 / iters;
    printf("%-44s %d waves/SIMD: %8.1f ns per loop body = %7.0f cycles @2.4 GHz\n", name, threads / 256, ns_per_body, ns_per_body * 2.4);
    return ns_per_body;
}

int main() {
    float *sink; CK(hipMalloc(&sink, 4));
    const int iters = 20000;
    for (int threads : {256, 512}) {
        run<T_FMA>("fma: 32 v_fma_f32", threads, iters, sink);
        run<T_PKFMA>("pkfma: 16 v_pk_fma_f32 (32 FMAs)", threads, iters, sink);
        run<T_EXP>("exp: 32 v_exp_f32", threads, iters, sink);
        run<T_MFMA>("mfma: 8 x 32x32x16 f16 (2 chains)", threads, iters, sink);
        run<T_MFMA_EXP>("mfma + exp in one wave", threads, iters, sink);
        run<T_MFMA_FMA>("mfma + fma in one wave", threads, iters, sink);
    }
    run<T_SPLIT>("split: 4 waves mfma, 4 waves exp (512 thr)", 512, iters, sink);
    return 0;
}
