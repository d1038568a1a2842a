// gstamps: WHERE does a launch of the persistent 256 x 256 GEMM spend its time?  Links a -DG2_STAMPS build of k_gemm.hip in
// which lane 0 of every workgroup stamps s_memrealtime (100 MHz) at its start and after every tile, plus its XCC id.
// Prints, per encoder shape: the spread of the workgroups' finish times (static tile assignment: a launch ends with its
// slowest workgroup), tile durations by position in the sequence, and per-XCD means.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
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
    const int M = argc > 1 ? atoi(argv[1]) : 48000;
    struct Shape { int N, K, epi; const char *name; };
    Shape shapes[] = {{1280, 1280, EPI_RESID_F32, "out-proj"}, {3840, 1280, EPI_F16, "qkv"}, {5120, 1280, EPI_GELU_F16, "fc1"},
                      {1280, 5120, EPI_RESID_F32, "fc2"}, {2560, 1280, EPI_F16, "cross-kv"}};
    half_t *A, *W; float *X, *bias, *pos; half_t *O0, *O1, *O2; unsigned long long *dbg;
    CK(hipMalloc(&A, (size_t)M * 5120 * 2)); CK(hipMalloc(&W, (size_t)5120 * 5120 * 2)); CK(hipMalloc(&X, (size_t)M * 1280 * 4));
    CK(hipMalloc(&bias, 5120 * 4)); CK(hipMalloc(&pos, (size_t)1500 * 1280 * 4)); CK(hipMalloc(&dbg, 256 * 32 * 8));
    CK(hipMalloc(&O0, (size_t)M * 5120 * 2)); CK(hipMalloc(&O1, (size_t)M * 1280 * 2)); CK(hipMalloc(&O2, (size_t)(M / 1500 + 1) * 1280 * NH_SP * 2));
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, A, (size_t)M * 5120, 1u);
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, W, (size_t)5120 * 5120, 7u);
    CK(hipMemset(X, 0, (size_t)M * 1280 * 4)); CK(hipMemset(bias, 0, 5120 * 4)); CK(hipMemset(pos, 0, (size_t)1500 * 1280 * 4));
    CK(hipStreamSynchronize(st));
    for (const Shape &s : shapes) {
        GemmParams p{};
        p.A = A; p.lda = s.K; p.a_rpb = M; p.W = W; p.bias = bias; p.M = M; p.N = s.N; p.K = s.K; p.epi = s.epi;
        p.out[0] = (s.epi == EPI_RESID_F32) ? (void *)X : (void *)O0; p.out[1] = O1; p.out[2] = O2;
        p.seg_n = (s.N == 3840 || s.N == 2560) ? 1280 : s.N; p.head_major = s.N == 2560; p.ldo = s.epi == EPI_RESID_F32 ? 1280 : p.seg_n;
        p.o_rpb = M; p.vt_seg = s.N == 3840 ? 2 : -1; p.S = 1500; p.H = 20; p.pos = pos; p.dbg = dbg;
        for (int i = 0; i < 4; i++) launch_gemm(p, st);      // steady clocks and caches; the last launch is the one read
        CK(hipMemsetAsync(dbg, 0, 256 * 32 * 8, st));
        launch_gemm(p, st);
        std::vector<unsigned long long> h(256 * 32);
        CK(hipMemcpyAsync(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost, st)); CK(hipStreamSynchronize(st));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < 256; b++) t0 = std::min(t0, h[b * 32]);
        std::vector<double> fin, start;
        double xs[8] = {0}, xt[8] = {0}; int xn[8] = {0};
        std::vector<std::vector<double>> dur(32);
        for (int b = 0; b < 256; b++) {
            const int n = (int)h[b * 32 + 2], x = (int)h[b * 32 + 1] & 7;
            if (!n) continue;
            start.push_back((h[b * 32] - t0) * 0.01);
            const double f = (h[b * 32 + 4 + n - 1] - t0) * 0.01;
            fin.push_back(f); xs[x] += f; xn[x]++; xt[x] += n;
            for (int i = 0; i < n; i++) dur[i].push_back((h[b * 32 + 4 + i] - (i ? h[b * 32 + 4 + i - 1] : h[b * 32])) * 0.01);
        }
        std::sort(fin.begin(), fin.end()); std::sort(start.begin(), start.end());
        printf("%-9s M=%d N=%d K=%d: %zu workgroups; start spread %.1f us; finish min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f us (max/median %.3f)\n",
               s.name, M, s.N, s.K, fin.size(), start.back(), fin.front(), fin[fin.size() / 10], fin[fin.size() / 2], fin[fin.size() * 9 / 10], fin.back(), fin.back() / fin[fin.size() / 2]);
        printf("          per XCD (mean finish us / mean tiles):");
        for (int x = 0; x < 8; x++) if (xn[x]) printf("  %d: %.1f/%.2f", x, xs[x] / xn[x], xt[x] / xn[x]);
        {
            double ml = 0, tot = 0; int nt = 0;
            for (int b = 0; b < 256; b++) { const int n = (int)h[b * 32 + 2]; if (!n) continue; ml += h[b * 32 + 3] * 0.01; tot += (h[b * 32 + 4 + n - 1] - h[b * 32]) * 0.01; nt += n; }
            printf("\n          per tile: %.1f us from tile start to the end of the main loop (acc init + K-loop), %.1f us from there to the next tile's start (prologue issue + epilogue)", ml / nt, (tot - ml) / nt);
        }
        printf("\n          tile durations by position (median, max):");
        for (int i = 0; i < 32 && !dur[i].empty(); i++) { std::sort(dur[i].begin(), dur[i].end()); printf("  %.1f,%.1f", dur[i][dur[i].size() / 2], dur[i].back()); }
        printf("\n");
    }
    return 0;
}
