// gbench: the encoder GEMM kernels on the distil-large-v3 b32 shapes, random fp16 data, TFLOP/s per shape.
// GEMM_128=1 times the 128 x 128 kernel (two workgroups per CU) on the same shapes: 0.48-0.79 PFLOP/s against 0.63-1.07.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../norma_amd/csrc/nh_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill_rand(half_t *p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (half_t)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.f));
    }
}
int main(int argc, char **argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    const bool k128 = getenv("GEMM_128") && atoi(getenv("GEMM_128"));  // time the 128 x 128 kernel (two workgroups per CU) instead
    auto launch = [&](const GemmParams &p) { if (k128) launch_gemm_128(p, st); else launch_gemm(p, st); };
    const int M = argc > 1 ? atoi(argv[1]) : 48000;
    struct Shape { int N, K, epi; const char *name; };
    // gbench M N K epi: one custom shape (N <= 5120, K <= 5120) instead of the encoder's six
    Shape custom[1] = {{argc > 4 ? atoi(argv[2]) : 0, argc > 4 ? atoi(argv[3]) : 0, argc > 4 ? atoi(argv[4]) : 0, "custom"}};
    Shape shapes_all[] = {
        {1280, 1280, EPI_RESID_F32, "out-proj  N=1280 K=1280 resid"}, {3840, 1280, EPI_F16, "qkv       N=3840 K=1280 f16+VT"},
        {5120, 1280, EPI_GELU_F16, "fc1       N=5120 K=1280 gelu"}, {1280, 5120, EPI_RESID_F32, "fc2       N=1280 K=5120 resid"},
        {1280, 3840, EPI_CONV2_F32, "conv2     N=1280 K=3840 conv2"}, {2560, 1280, EPI_F16, "cross-kv  N=2560 K=1280 f16"}};
    half_t *A, *W; float *X, *bias, *pos; half_t *O0, *O1, *O2;
    CK(hipMalloc(&A, (size_t)M * 5120 * 2)); CK(hipMalloc(&W, (size_t)5120 * 5120 * 2)); CK(hipMalloc(&X, (size_t)M * 1280 * 4));
    CK(hipMalloc(&bias, 5120 * 4)); CK(hipMalloc(&pos, (size_t)1500 * 1280 * 4));
    CK(hipMalloc(&O0, (size_t)M * 5120 * 2)); CK(hipMalloc(&O1, (size_t)M * 1280 * 2)); CK(hipMalloc(&O2, (size_t)(M / 1500 + 1) * 1280 * NH_SP * 2));
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, A, (size_t)M * 5120, 1u);
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, W, (size_t)5120 * 5120, 7u);
    CK(hipMemset(X, 0, (size_t)M * 1280 * 4)); CK(hipMemset(bias, 0, 5120 * 4)); CK(hipMemset(pos, 0, (size_t)1500 * 1280 * 4));
    CK(hipStreamSynchronize(st));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const Shape *shapes = argc > 4 ? custom : shapes_all;
    const int nshapes = argc > 4 ? 1 : 6;
    for (int si = 0; si < nshapes; si++) {
        const Shape &s = shapes[si];
        GemmParams p{};
        p.A = A; p.lda = s.K; p.a_rpb = M; p.W = W; p.bias = bias; p.M = M; p.N = s.N; p.K = s.K; p.epi = s.epi;
        p.out[0] = (s.epi == EPI_RESID_F32 || s.epi == EPI_CONV2_F32) ? (void *)X : (void *)O0; p.out[1] = O1; p.out[2] = O2;
        p.seg_n = (s.N == 3840 || s.N == 2560) ? 1280 : s.N; p.ldo = (s.epi == EPI_RESID_F32 || s.epi == EPI_CONV2_F32) ? 1280 : p.seg_n;
        p.o_rpb = M; p.vt_seg = s.N == 3840 ? 2 : -1; p.S = 1500; p.H = 20; p.pos = pos;
        if (s.N == 3840) { p.out[0] = O0; p.out[1] = O1; }
        for (int i = 0; i < 3; i++) launch(p);
        CK(hipStreamSynchronize(st));
        const int reps = 10;
        hipEventRecord(a, st);
        for (int i = 0; i < reps; i++) launch(p);
        hipEventRecord(b, st); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b);
        double tf = 2.0 * M * s.N * s.K * reps / (ms * 1e-3) / 1e12;
        printf("%-34s %8.1f us  %7.1f TFLOP/s\n", s.name, ms * 1e3 / reps, tf);
    }
    return 0;
}
