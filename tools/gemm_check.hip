// gemm_check: race screen for the persistent ping-pong GEMM (k_gemm.hip).  The 256 x 256 kernel and the 128 x 128 kernel
// accumulate every output element over k in the same order (one v_mfma_f32_16x16x32_f16 per 32-deep step, ascending), so
// their results must agree bit for bit; a half-tile read before its LDS-DMA landed, or re-staged before its last read,
// shows up as a differing tile.  Many launches, several shapes (K = 128 .. 5120, M not a multiple of 256), fresh random
// data each round.  Usage: tools/bin/gemm_check [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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
__global__ void count_diff(const unsigned *a, const unsigned *b, size_t n, unsigned long long *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}
__global__ void count_nonzero(const unsigned *a, size_t n, unsigned long long *out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != 0;
    if (c) atomicAdd(out, c);
}
int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 20;
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Shape { int M, N, K, epi; } shapes[] = {{3000, 256, 128, EPI_F16}, {3000, 512, 256, EPI_F16}, {6000, 1280, 1280, EPI_F16},
                                                   {48000, 1280, 1280, EPI_GELU_F16}, {24000, 5120, 1280, EPI_GELU_F16}, {12345, 1280, 5120, EPI_F16},
                                                   {48000, 1280, 3840, EPI_F16}, {777, 256, 384, EPI_F16},
                                                   {96000, 256, 256, EPI_F16} /* M * M >= 2^32: the row-index divisions' magic multipliers (nh_magic) */};
    const size_t maxA = (size_t)48000 * 5120, maxW = (size_t)5120 * 5120, maxO = (size_t)48000 * 5120;
    half_t *A, *W, *O1, *O2; float *bias; unsigned long long *nd;
    CK(hipMalloc(&A, maxA * 2)); CK(hipMalloc(&W, maxW * 2)); CK(hipMalloc(&O1, maxO * 2)); CK(hipMalloc(&O2, maxO * 2));
    CK(hipMalloc(&bias, 5120 * 4)); CK(hipMemset(bias, 0, 5120 * 4)); CK(hipMalloc(&nd, 8));
    unsigned long long total_bad = 0; long launches = 0;
    for (int r = 0; r < rounds; r++) {
        for (auto &s : shapes) {
            hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, A, (size_t)s.M * s.K, 17u * r + 1u);
            hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, W, (size_t)s.N * s.K, 31u * r + 7u);
            GemmParams p{};
            p.A = A; p.lda = s.K; p.a_rpb = s.M; p.W = W; p.bias = bias; p.M = s.M; p.N = s.N; p.K = s.K; p.epi = s.epi;
            p.seg_n = s.N; p.ldo = s.N; p.o_rpb = s.M; p.vt_seg = -1; p.S = 1500; p.H = s.N / 64;
            CK(hipMemsetAsync(O1, 0, (size_t)s.M * s.N * 2, st)); CK(hipMemsetAsync(O2, 0, (size_t)s.M * s.N * 2, st));
            p.out[0] = O1;
            for (int rep = 0; rep < 3; rep++) launch_gemm(p, st);          // 256 x 256 persistent ping-pong (same output each time)
            p.out[0] = O2;
            launch_gemm_128(p, st);                                          // the independent 128 x 128 kernel
            CK(hipMemsetAsync(nd, 0, 8, st));
            hipLaunchKernelGGL(count_diff, dim3(1024), dim3(256), 0, st, (const unsigned *)O1, (const unsigned *)O2, (size_t)s.M * s.N / 2, nd);
            unsigned long long bad = 0; CK(hipMemcpyAsync(&bad, nd, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            CK(hipMemsetAsync(nd, 0, 8, st));
            hipLaunchKernelGGL(count_nonzero, dim3(1024), dim3(256), 0, st, (const unsigned *)O1, (size_t)s.M * s.N / 2, nd);
            unsigned long long nz = 0; CK(hipMemcpyAsync(&nz, nd, 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
            if (nz * 10 < (unsigned long long)s.M * s.N / 2 * 9) { printf("round %d M=%d N=%d K=%d: output mostly zero (%llu)\n", r, s.M, s.N, s.K, nz); total_bad++; }
            launches += 3;
            if (bad) printf("round %d  M=%d N=%d K=%d epi=%d: %llu differing dwords\n", r, s.M, s.N, s.K, s.epi, bad);
            total_bad += bad;
        }
    }
    printf("%ld launches of the 256 x 256 kernel checked against the 128 x 128 kernel: %llu differing dwords\n", launches, total_bad);
    return total_bad != 0;
}
