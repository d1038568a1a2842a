// ldsprobe: do two workgroups of DIFFERENT kernels that share a CU keep their LDS allocations apart when one of them uses a
// large dynamic allocation?  Victim: 256 threads, `vbytes` of static-sized (here dynamic, same effect) LDS filled with a pattern
// derived from its own id, held for a while, verified.  Aggressor: 512 threads, `abytes` of dynamic LDS, written in full (in
// bounds only) over and over.  Both on their own stream, many rounds.  Any word the victim reads back wrong is counted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void victim(unsigned *errs, unsigned *first, int words, int spins) {
    extern __shared__ unsigned vs[];
    const unsigned tag = 0x51000000u ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < words; i += 256) vs[i] = tag + i;
    __syncthreads();
    unsigned bad = 0;
    for (int s = 0; s < spins; s++) {
        for (int i = threadIdx.x; i < words; i += 256) {
            unsigned v = vs[i];
            if (v != tag + i) { bad++; if (atomicAdd(errs, 1u) == 0) { first[0] = blockIdx.x; first[1] = i; first[2] = v; first[3] = tag + i; } vs[i] = tag + i; }
        }
        __syncthreads();
    }
}

// second victim: the waves of a workgroup hand data to each other across a barrier, every round with a new pattern -- a wave
// that is let through the barrier early reads the previous round's words
__global__ __launch_bounds__(256) void victim_xwave(unsigned *errs, unsigned *first, int words, int spins) {
    extern __shared__ unsigned vs[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = words / 4;
    for (int s = 0; s < spins; s++) {
        const unsigned tag = 0x77000000u ^ (blockIdx.x * 2654435761u) ^ (s << 20);
        for (int i = lane; i < per; i += 64) vs[w * per + i] = tag + w * per + i;     // my quarter
        __syncthreads();
        const int o = (w + 1 + (s & 1)) & 3;                                            // somebody else's quarter
        for (int i = lane; i < per; i += 64) {
            unsigned v = vs[o * per + i];
            if (v != tag + o * per + i) { if (atomicAdd(errs, 1u) == 0) { first[0] = blockIdx.x; first[1] = o * per + i; first[2] = v; first[3] = tag + o * per + i; } }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512) void aggressor(unsigned *sink, int words, int spins) {
    extern __shared__ unsigned as[];
    unsigned acc = 0;
    for (int s = 0; s < spins; s++) {
        for (int i = threadIdx.x; i < words; i += 512) as[i] = 0xA6000000u + (s << 16) + i;
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += 512) acc += as[i];
        __syncthreads();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipSetDevice(0));
    unsigned *errs, *first, *sink;
    CK(hipMalloc(&errs, 4)); CK(hipMalloc(&first, 16)); CK(hipMalloc(&sink, 4));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&victim), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int vsizes[] = {37012, 16384, 4096};
    const int asizes[] = {40960, 65536, 66560, 81920, 122880};
    for (int vb : vsizes)
        for (int ab : asizes) {
            if (vb + ab > 160 * 1024) continue;
            CK(hipMemset(errs, 0, 4)); CK(hipMemset(first, 0, 16));
            for (int r = 0; r < rounds; r++) {
                hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), ab, sa, sink, ab / 4, 40);
                hipLaunchKernelGGL(victim, dim3(3000), dim3(256), vb, sv, errs, first, vb / 4, 6);
            }
            CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
            unsigned e = 0, f[4];
            CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 16, hipMemcpyDeviceToHost));
            printf("victim %6d B beside aggressor %6d B, %d rounds: %u wrong words", vb, ab, rounds, e);
            if (e) printf("  (first: victim workgroup %u word %u read 0x%08x expected 0x%08x)", f[0], f[1], f[2], f[3]);
            printf("\n"); fflush(stdout);
        }
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&victim_xwave), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int ab : {81920, 122880}) {
        const int vb = 36864;
        CK(hipMemset(errs, 0, 4)); CK(hipMemset(first, 0, 16));
        for (int r = 0; r < rounds; r++) {
            hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), ab, sa, sink, ab / 4, 40);
            hipLaunchKernelGGL(victim_xwave, dim3(3000), dim3(256), vb, sv, errs, first, vb / 4, 6);
        }
        CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
        unsigned e = 0, f[4];
        CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 16, hipMemcpyDeviceToHost));
        printf("cross-wave victim %6d B beside aggressor %6d B (80 barriers per workgroup), %d rounds: %u wrong words", vb, ab, rounds, e);
        if (e) printf("  (first: victim workgroup %u word %u read 0x%08x expected 0x%08x)", f[0], f[1], f[2], f[3]);
        printf("\n"); fflush(stdout);
    }
    return 0;
}
