// Microbenchmark: what does one dependent kernel boundary cost on this box (eager vs hipGraph)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
__global__ void k_empty() {}
__global__ void k_touch(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.f; }
struct Big { void *a[8]; long b[6]; int c[12]; };
__global__ void k_bigargs(Big g, float *p) { if (threadIdx.x == 0 && blockIdx.x == 0 && g.c[0] == 12345) p[0] = 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> double time_us(hipStream_t st, int reps, F f) {
    hipStreamSynchronize(st);
    auto t0 = std::chrono::high_resolution_clock::now();
    for (int i = 0; i < reps; i++) f();
    hipStreamSynchronize(st);
    auto t1 = std::chrono::high_resolution_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    float *p; CK(hipMalloc(&p, 1 << 24)); CK(hipMemset(p, 0, 1 << 24));
    Big g{}; 
    for (int blocks : {1, 32, 256}) {
        double e = time_us(st, 2000, [&] { hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, st); });
        double t = time_us(st, 2000, [&] { hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, st, p, blocks * 256); });
        double b = time_us(st, 2000, [&] { hipLaunchKernelGGL(k_bigargs, dim3(blocks), dim3(256), 0, st, g, p); });
        printf("eager  blocks=%4d  empty %.2f us/kernel  touch %.2f  bigargs %.2f\n", blocks, e, t, b);
    }
    for (int nk : {26, 100}) {
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < nk; i++) hipLaunchKernelGGL(k_touch, dim3(32), dim3(256), 0, st, p, 32 * 256);
        CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        double t = time_us(st, 200, [&] { hipGraphLaunch(ex, st); });
        printf("graph  %3d touch kernels: %.2f us/replay = %.2f us/kernel\n", nk, t, t / nk);
        hipGraphExecDestroy(ex); hipGraphDestroy(gr);
    }
    // device-side duration of a chain measured with events (no host in the loop)
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(k_touch, dim3(32), dim3(256), 0, st, p, 32 * 256);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("events: 1000 eager touch kernels: %.2f us/kernel\n", ms);
    return 0;
}
