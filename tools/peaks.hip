// peaks: attained (not nominal) ceilings of the box, SURVEY.md 8(d): an MFMA-only loop, an HBM read stream, and the
// vendor library (rocBLAS -> hipBLASLt/Tensile) on the encoder GEMM shapes as an outside reference point.
// Measurement tool only; nothing here is linked into libnorma_hip.so.
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half_t;
typedef half_t half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void fill_rand(half_t *p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (half_t)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.f));
    }
}

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(float *out, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; i++) {  // pseudo-random operands: MFMA power (and so the sustained clock) depends on bit toggling
        unsigned x = (threadIdx.x * 8 + i) * 2654435761u + blockIdx.x; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        a[i] = (half_t)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.f)); b[i] = (half_t)(((int)(x >> 16) - 32768) * (1.0f / 32768.f));
    }
    if (SHAPE == 16) {
        f32x4 acc[32];
        for (int i = 0; i < 32; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int i = 0; i < 32; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        f32x4 s = acc[0];
        for (int i = 1; i < 32; i++) s += acc[i];
        out[blockIdx.x * 512 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        float s = 0.f;
        for (int i = 0; i < 8; i++) for (int j = 0; j < 16; j++) s += acc[i][j];
        out[blockIdx.x * 512 + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void stream_read(const f32x4 *__restrict__ p, size_t n, float *out) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    size_t i = blockIdx.x * (size_t)256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n; i += 4 * stride) {
        f32x4 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride);
        f32x4 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        s += a + b + c + d;
    }
    if (s[0] + s[1] + s[2] + s[3] == 1.2345f) out[0] = s[0];
}

int main(int argc, char **argv) {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms, *out; CK(hipMalloc(&out, 1024 * 512 * 4));
    // ---- MFMA-only: 256 CUs x 8 waves (2 per SIMD), 32 independent accumulators per wave
    for (int wpb = 4; wpb <= 8; wpb += 4) {
        const int iters = 4000;
        for (int shape = 16; shape <= 32; shape += 16) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, st);
                if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(256), dim3(64 * wpb), 0, st, out, iters);
                else hipLaunchKernelGGL(mfma_loop<32>, dim3(256), dim3(64 * wpb), 0, st, out, iters);
                hipEventRecord(e1, st); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
            }
            const double flop = 256.0 * wpb * iters * (shape == 16 ? 32 * 2.0 * 16 * 16 * 32 : 16 * 2.0 * 32 * 32 * 16);
            printf("mfma-only %s, %d waves/CU: %7.1f TFLOP/s  (%.2f ms)\n", shape == 16 ? "16x16x32 f16" : "32x32x16 f16", wpb,
                   flop / (ms * 1e-3) / 1e12, ms);
        }
    }
    // ---- HBM read stream: 4 GiB, far beyond the 256 MiB Infinity Cache
    {
        const size_t bytes = (size_t)4 << 30; f32x4 *buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
        for (int blocks = 2048; blocks <= 8192; blocks *= 2) {
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0, st);
                hipLaunchKernelGGL(stream_read, dim3(blocks), dim3(256), 0, st, buf, bytes / 16, out);
                hipEventRecord(e1, st); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("hbm read stream, %d blocks: %7.1f GB/s\n", blocks, bytes / (ms * 1e-3) / 1e9);
        }
        CK(hipFree(buf));
    }
    // ---- vendor GEMM on the encoder shapes: C[M][N] (f16) = A[M][K] . W[N][K]^T, f32 accumulate
    {
        const int M = argc > 1 ? atoi(argv[1]) : 48000;
        rocblas_handle h; if (rocblas_create_handle(&h) != rocblas_status_success) { printf("rocblas: no handle\n"); return 0; }
        rocblas_set_stream(h, st);
        half_t *A, *W, *C; CK(hipMalloc(&A, (size_t)M * 5120 * 2)); CK(hipMalloc(&W, (size_t)5120 * 5120 * 2)); CK(hipMalloc(&C, (size_t)M * 5120 * 2));
        hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, A, (size_t)M * 5120, 1u);
        hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, W, (size_t)5120 * 5120, 7u);
        CK(hipStreamSynchronize(st));
        struct { int N, K; const char *name; } shapes[] = {{1280, 1280, "out-proj"}, {3840, 1280, "qkv"}, {5120, 1280, "fc1"}, {1280, 5120, "fc2"}, {1280, 3840, "conv2"}, {2560, 1280, "cross-kv"}};
        const float alpha = 1.f, beta = 0.f;
        for (auto &s : shapes) {
            // column-major view: C^T[N][M] = W (op T: stored [N][K] row-major = K x N col-major) ... compute C^T = W . A^T
            auto run = [&]() {
                return rocblas_gemm_ex(h, rocblas_operation_transpose, rocblas_operation_none, s.N, M, s.K, &alpha, W, rocblas_datatype_f16_r, s.K,
                                       A, rocblas_datatype_f16_r, s.K, &beta, C, rocblas_datatype_f16_r, s.N, C, rocblas_datatype_f16_r, s.N,
                                       rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
            };
            for (int i = 0; i < 3; i++) if (run() != rocblas_status_success) { printf("rocblas gemm failed\n"); return 0; }
            CK(hipStreamSynchronize(st));
            const int reps = 10;
            hipEventRecord(e0, st);
            for (int i = 0; i < reps; i++) run();
            hipEventRecord(e1, st); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
            printf("rocblas %-9s N=%4d K=%4d %8.1f us  %7.1f TFLOP/s\n", s.name, s.N, s.K, ms * 1e3 / reps, 2.0 * M * s.N * s.K * reps / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
