// gridbar: cost of a software grid barrier on this GPU (one workgroup per CU, all co-resident), the quantity that decides
// whether a persistent whole-decode-step kernel can beat ~20 graph-replayed launches.  Spins are bounded: a barrier that
// is not reached within ~50 ms sets a failure flag instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, int *fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { *fail = 1; ok = false; break; }
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void bar_loop(unsigned *counter, int iters, int *fail, float *sink) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; i++) {
        if (!grid_barrier(counter, (unsigned)(i + 1) * gridDim.x, fail)) break;
        v = v * 1.0001f + 1.0f;
    }
    if (v == 12345.678f) sink[0] = v;
}

int main() {
    unsigned *counter; int *fail; float *sink;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&sink, 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {64, 128, 256}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(counter, 0, 4)); CK(hipMemset(fail, 0, 4));
            const int iters = 2000;
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(bar_loop, dim3(blocks), dim3(256), 0, 0, counter, iters, fail, sink);
            hipEventRecord(b, 0); CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b);
            int hf; CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
            if (rep) printf("%3d workgroups: %.2f us per grid barrier%s\n", blocks, ms * 1e3 / iters, hf ? "  (TIMED OUT)" : "");
        }
    }
    return 0;
}
