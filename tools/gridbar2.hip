// gridbar2: grid-barrier designs for a persistent decode-step kernel (one workgroup per CU, all co-resident), each carrying
// a real payload hand-off (every workgroup writes a 1 KiB slice of a buffer before the barrier and reads another
// workgroup's slice after it, checked), against the cost of the same hand-off through a kernel boundary in a hipGraph.
//   v0  one agent-scope counter: atomic add + spin on it                                  (tools/gridbar.hip: 10.8 us / 256)
//   v1  flag array: workgroup i release-stores the epoch to flags[i]; everyone polls all flags with one coalesced load
//   v2  eight counters (one per XCD, own cache lines): atomic add on counter[wg & 7], poll the eight
//   v3  v1 with the flags on separate 64-byte lines
// All spins are bounded (a barrier not reached within ~20 ms raises a failure flag and every workgroup leaves).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct Bar { unsigned *counter; unsigned *flags; unsigned *xcd; int *fail; };

template <int V>
__device__ __forceinline__ bool barrier(const Bar &b, unsigned epoch, int nwg) {
    __shared__ int sh_ok;
    __syncthreads();   // every thread's payload stores are issued
    bool ok = true;
    if (V == 0) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(b.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(b.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < epoch * (unsigned)nwg) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 400000) { *b.fail = 1; break; }
            }
        }
        __syncthreads();
        return *b.fail == 0;
    }
    if (V == 2) {
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(b.xcd + 32 * (blockIdx.x & 7), 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const unsigned want = epoch * (unsigned)(nwg >> 3);
            int spins = 0;
            for (;;) {
                unsigned v = lane < 8 ? __hip_atomic_load(b.xcd + 32 * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : want;
                if (__all(v >= want)) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 400000) { if (lane == 0) *b.fail = 1; break; }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        return *b.fail == 0;
    }
    // V == 1 / 3: flags
    const int stride = V == 3 ? 16 : 1;
    if (threadIdx.x == 0) __hip_atomic_store(b.flags + (long)blockIdx.x * stride, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    {
        int spins = 0;
        for (;;) {
            bool mine = true;
            for (int i = threadIdx.x; i < nwg; i += blockDim.x)
                mine = mine && (__hip_atomic_load(b.flags + (long)i * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch);
            if (__syncthreads_and(mine)) break;
            if (++spins > 200000) { if (threadIdx.x == 0) *b.fail = 1; ok = false; }
            if (!__syncthreads_and(ok)) break;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok && *b.fail == 0;
}

template <int V>
__global__ __launch_bounds__(256) void bar_loop(Bar b, float *buf, int iters, unsigned *bad) {
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    unsigned wrong = 0;
    for (int i = 0; i < iters; i++) {
        buf[((long)(i & 1) * nwg + wg) * 256 + tid] = (float)(i * 1000 + wg);   // plain stores: the release publishes them
        if (!barrier<V>(b, (unsigned)(i + 1), nwg)) break;
        const int src = (wg + 37 + i) % nwg;                                    // somebody else's slice (usually another XCD)
        const float v = buf[((long)(i & 1) * nwg + src) * 256 + tid];
        if (v != (float)(i * 1000 + src)) wrong++;
    }
    if (wrong) atomicAdd(bad, wrong);
}

__global__ __launch_bounds__(256) void step_kernel(float *buf, int i, unsigned *bad) {
    const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
    if (i > 0) {
        const int src = (wg + 37 + i - 1) % nwg;
        const float v = buf[((long)((i - 1) & 1) * nwg + src) * 256 + tid];
        if (v != (float)((i - 1) * 1000 + src)) atomicAdd(bad, 1u);
    }
    buf[((long)(i & 1) * nwg + wg) * 256 + tid] = (float)(i * 1000 + wg);
}

template <int V>
static void run(const char *name, int blocks, Bar b, float *buf, unsigned *bad, hipEvent_t ea, hipEvent_t eb) {
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(b.counter, 0, 4)); CK(hipMemset(b.flags, 0, 4096 * 64)); CK(hipMemset(b.xcd, 0, 8 * 128)); CK(hipMemset(b.fail, 0, 4));
        CK(hipMemset(bad, 0, 4));
        const int iters = 2000;
        hipEventRecord(ea, 0);
        hipLaunchKernelGGL(bar_loop<V>, dim3(blocks), dim3(256), 0, 0, b, buf, iters, bad);
        hipEventRecord(eb, 0); CK(hipEventSynchronize(eb));
        float ms; hipEventElapsedTime(&ms, ea, eb);
        int hf; unsigned hb; CK(hipMemcpy(&hf, b.fail, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        if (rep) printf("%-28s %3d workgroups: %6.2f us per barrier + hand-off   wrong reads %u%s\n", name, blocks, ms * 1e3 / iters, hb, hf ? "  (TIMED OUT)" : "");
    }
}

int main() {
    Bar b; float *buf; unsigned *bad;
    CK(hipMalloc(&b.counter, 4)); CK(hipMalloc(&b.flags, 4096 * 64)); CK(hipMalloc(&b.xcd, 8 * 128)); CK(hipMalloc(&b.fail, 4));
    CK(hipMalloc(&buf, 2L * 1024 * 256 * 4)); CK(hipMalloc(&bad, 4));
    hipEvent_t ea, eb; hipEventCreate(&ea); hipEventCreate(&eb);
    for (int blocks : {80, 256, 512}) {
        run<0>("v0 single counter", blocks, b, buf, bad, ea, eb);
        run<1>("v1 flag array (packed)", blocks, b, buf, bad, ea, eb);
        run<2>("v2 eight per-XCD counters", blocks, b, buf, bad, ea, eb);
        run<3>("v3 flag array (64 B apart)", blocks, b, buf, bad, ea, eb);
    }
    // the same hand-off through kernel boundaries, replayed from a graph
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int blocks : {80, 256, 512}) {
        CK(hipMemset(bad, 0, 4));
        const int nk = 100;
        hipGraph_t gr; hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < nk; i++) hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(256), 0, st, buf, i, bad);
        CK(hipStreamEndCapture(st, &gr)); CK(hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0));
        for (int w = 0; w < 3; w++) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        hipEventRecord(ea, st);
        for (int w = 0; w < 20; w++) CK(hipGraphLaunch(ex, st));
        hipEventRecord(eb, st); CK(hipEventSynchronize(eb));
        float ms; hipEventElapsedTime(&ms, ea, eb);
        unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("graph kernel boundary          %3d workgroups: %6.2f us per kernel (write 1 KiB, next kernel reads it)   wrong reads %u\n", blocks, ms * 1e3 / (20.0 * nk), hb);
        hipGraphExecDestroy(ex); hipGraphDestroy(gr);
    }
    return 0;
}
