// sktrace: where does the time of one small decode-step GEMV go?  A faithful copy of the structure of
// skinny_gemm_kernel<2, 4, 1> (tile-major weights, 4 waves split K = 1280, LDS reduce, f32 residual epilogue) with
// wall-clock stamps (s_memrealtime, 100 MHz, common to all CUs) at every stage, run as a dependent chain replayed from a
// hipGraph exactly like the decode step (each launch reads what the previous one wrote; 28 weight matrices = 92 MB
// cycle through, so they come from the Infinity Cache as in the real loop).  Prints, for a launch in the middle of the
// chain, the stage times relative to the END of the previous kernel (last stamp of any of its workgroups).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../norma_amd/csrc/nh_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define NST 8
__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }

template <int MODE>  // 0: full; 1: no x loads (constant activations); 2: no weights
__global__ __launch_bounds__(256) void gemv_trace(const half_t *__restrict__ Wt, const half_t *__restrict__ x, float *__restrict__ resid,
                                                  half_t *__restrict__ xout, const float *__restrict__ bias, int K, int N, int R,
                                                  unsigned long long *stamps) {
    __shared__ f32x4 red[4][2][64];
    unsigned long long t[NST];
    t[0] = now();
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16, kslice = K / 4, kbeg = w * kslice, steps = kslice >> 5;
    f32x4 pre0 = {0, 0, 0, 0}, pre1 = {0, 0, 0, 0};
    {
        const int er = tid >> 2, en = n0 + 4 * (tid & 3);
        if (tid < 128 && er < R) { pre0 = *reinterpret_cast<const f32x4 *>(bias + en); pre1 = *reinterpret_cast<const f32x4 *>(resid + (long)er * N + en); }
    }
    const half_t *wp = Wt + ((long)blockIdx.x * (K >> 5) + (kbeg >> 5)) * 512 + lane * 8;
    const half_t *xp0 = x + (long)fr * K + kbeg + 8 * fq, *xp1 = x + (long)(16 + fr) * K + kbeg + 8 * fq;
    half8 a[10], b0[10], b1[10];
    t[1] = now();
#pragma unroll
    for (int u = 0; u < 10; u++) {
        if (MODE != 2) a[u] = *reinterpret_cast<const half8 *>(wp + 512L * u); else a[u] = (half8){1, 1, 1, 1, 1, 1, 1, 1};
        if (MODE != 1) { b0[u] = *reinterpret_cast<const half8 *>(xp0 + 32 * u); b1[u] = *reinterpret_cast<const half8 *>(xp1 + 32 * u); }
        else { b0[u] = (half8){1, 1, 1, 1, 1, 1, 1, 1}; b1[u] = b0[u]; }
    }
    t[2] = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[3] = now();
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 10; u++) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b0[u], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b1[u], acc1, 0, 0, 0);
    }
    red[w][0][lane] = acc0; red[w][1][lane] = acc1;
    __syncthreads();
    t[4] = now();
    const int r = tid >> 2, nq = tid & 3;
    if (tid < 128 && r < R) {
        const int src_lane = 16 * nq + (r & 15), cb = r >> 4;
        f32x4 v = red[0][cb][src_lane];
        for (int ww = 1; ww < 4; ww++) v += red[ww][cb][src_lane];
        v += pre0 + pre1;
        *reinterpret_cast<f32x4 *>(resid + (long)r * N + n0 + 4 * nq) = v;
        half4 hv = {(half_t)(v[0] * 1e-3f), (half_t)(v[1] * 1e-3f), (half_t)(v[2] * 1e-3f), (half_t)(v[3] * 1e-3f)};
        *reinterpret_cast<half4 *>(xout + (long)r * N + n0 + 4 * nq) = hv;
    }
    t[5] = now();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t[6] = now();
    if (stamps && tid == 0) {
        for (int i = 0; i < 7; i++) stamps[(long)blockIdx.x * NST + i] = t[i];
    }
    (void)steps;
}

int main(int argc, char **argv) {
    const int K = 1280, N = 1280, R = 32, tiles = N / 16, NW = 28, CH = 40;
    hipStream_t st; CK(hipStreamCreate(&st));
    half_t *W; CK(hipMalloc(&W, (size_t)NW * N * K * 2)); CK(hipMemset(W, 0x11, (size_t)NW * N * K * 2));
    half_t *xa, *xb; CK(hipMalloc(&xa, R * K * 2)); CK(hipMalloc(&xb, R * K * 2)); CK(hipMemset(xa, 0, R * K * 2)); CK(hipMemset(xb, 0, R * K * 2));
    float *resid, *bias; CK(hipMalloc(&resid, R * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMemset(resid, 0, R * N * 4)); CK(hipMemset(bias, 0, N * 4));
    unsigned long long *stamps; CK(hipMalloc(&stamps, (size_t)CH * tiles * NST * 8)); CK(hipMemset(stamps, 0, (size_t)CH * tiles * NST * 8));
    for (int mode = 0; mode < 3; mode++) {
        hipGraph_t g; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < CH; i++) {
            const half_t *w = W + (size_t)(i % NW) * N * K;
            half_t *xi = (i & 1) ? xb : xa, *xo = (i & 1) ? xa : xb;
            unsigned long long *sp = stamps + (size_t)i * tiles * NST;
            if (mode == 0) hipLaunchKernelGGL(gemv_trace<0>, dim3(tiles), dim3(256), 0, st, w, xi, resid, xo, bias, K, N, R, sp);
            else if (mode == 1) hipLaunchKernelGGL(gemv_trace<1>, dim3(tiles), dim3(256), 0, st, w, xi, resid, xo, bias, K, N, R, sp);
            else hipLaunchKernelGGL(gemv_trace<2>, dim3(tiles), dim3(256), 0, st, w, xi, resid, xo, bias, K, N, R, sp);
        }
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int wu = 0; wu < 5; wu++) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
        hipEventRecord(ea, st); for (int wu = 0; wu < 10; wu++) CK(hipGraphLaunch(ex, st)); hipEventRecord(eb, st); CK(hipEventSynchronize(eb));
        float ms; hipEventElapsedTime(&ms, ea, eb);
        std::vector<unsigned long long> h((size_t)CH * tiles * NST);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %.2f us per kernel in the chain\n", mode, mode == 0 ? "full" : mode == 1 ? "no x loads" : "no weight loads", ms * 1e3 / (10.0 * CH));
        const char *names[7] = {"start", "addr", "issued", "loads landed", "mfma+lds+barrier", "stores issued", "stores done"};
        for (int i = 20; i < 23; i++) {
            unsigned long long prev_end = 0;
            for (int b = 0; b < tiles; b++) prev_end = std::max(prev_end, h[((size_t)(i - 1) * tiles + b) * NST + 6]);
            printf("  launch %d (times in us after the last stamp of launch %d):\n", i, i - 1);
            for (int s = 0; s < 7; s++) {
                double mn = 1e30, mx = -1e30, sum = 0;
                for (int b = 0; b < tiles; b++) {
                    double v = ((double)h[((size_t)i * tiles + b) * NST + s] - (double)prev_end) * 0.01;
                    mn = std::min(mn, v); mx = std::max(mx, v); sum += v;
                }
                printf("    %-18s min %6.2f  avg %6.2f  max %6.2f\n", names[s], mn, sum / tiles, mx);
            }
        }
        hipGraphExecDestroy(ex); hipGraphDestroy(g);
    }
    return 0;
}
