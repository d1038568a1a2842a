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

// the logits GEMV's weight stream without the GEMV: 16-row tiles of a [V][1280] fp16 matrix, 10 loads of 16 B in flight per
// lane.  TILED = 0: rows as stored (a wave instruction = 16 rows x 64 B, 2560 B apart); 1: tile-major repack (1 KiB contiguous)
template <int TILED>
__global__ __launch_bounds__(512) void gemv_stream(const half_t *__restrict__ W, int V, float *out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, fr = lane & 15, fq = lane >> 4;
    const int tiles = V / 16, nw = gridDim.x * 8;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int tile = blockIdx.x * 8 + w; tile < tiles; tile += nw) {
        const half_t *base = TILED ? W + (size_t)tile * 16 * 1280 + lane * 8 : W + ((size_t)tile * 16 + fr) * 1280 + 8 * fq;
        for (int s0 = 0; s0 < 40; s0 += 10) {
            f4 a[10];
#pragma unroll
            for (int u = 0; u < 10; u++) a[u] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(base + (TILED ? 512 : 32) * (s0 + u)));
#pragma unroll
            for (int u = 0; u < 10; u++) acc += a[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) out[0] = acc[0];
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
    // ---- GEMV-shaped weight stream (the decode step's logits layer): stored rows vs tile-major repack
    {
        const int V = 51872; half_t *Wg; CK(hipMalloc(&Wg, (size_t)4 * V * 1280 * 2)); CK(hipMemset(Wg, 1, (size_t)4 * V * 1280 * 2));
        for (int tiled = 0; tiled < 2; tiled++) {
            for (int rep = 0; rep < 5; rep++) {
                const half_t *Wr = Wg + (size_t)(rep % 4) * V * 1280;  // rotate over 4 copies: no Infinity-Cache reuse
                hipEventRecord(e0, st);
                if (tiled) hipLaunchKernelGGL(gemv_stream<1>, dim3(256), dim3(512), 0, st, Wr, V, out);
                else hipLaunchKernelGGL(gemv_stream<0>, dim3(256), dim3(512), 0, st, Wr, V, out);
                hipEventRecord(e1, st); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("gemv-shaped stream of 133 MB, %s: %6.1f us  %7.1f GB/s\n", tiled ? "tile-major (1 KiB per wave instruction)" : "stored rows (16 x 64 B per wave instruction)",
                   ms * 1e3, (double)V * 1280 * 2 / (ms * 1e-3) / 1e9);
        }
        CK(hipFree(Wg));
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
