// dmastream: how many bytes per microsecond can ONE CU pull from HBM, and how does the chip's read rate scale with the number
// of CUs doing the pulling?  (the decode kernels need >= 150 CUs to reach 6 TB/s with plain loads; could
// a loader built on LDS-DMA rings do it from 64-96?)  Pure streaming, no compute: every wave walks its own contiguous slice
// of a 2 GiB buffer in 1 KiB pieces,
//   lds   global_load_lds_dwordx4 into a per-wave ring of R KiB, vmcnt(R-1) before a slot is reused (R KiB in flight per wave)
//   reg   global_load_dwordx4 into R registers-quads per lane, xor-folded (R KiB in flight per wave)
// on a stream confined to the first n CUs (hipExtStreamCreateWithCUMask), one workgroup of W waves per CU.
// usage: dmastream            -> table over n = 256, 128, 96, 64; W = 8, 16; R = 4, 8, 16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <int R, int NT>
__global__ __launch_bounds__(1024) void stream_lds(const char *src, size_t bytes_per_wave, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const char *p = src + ((size_t)blockIdx.x * nw + w) * bytes_per_wave + lane * 16;
    char *ring = smem + __builtin_amdgcn_readfirstlane(w) * (R * 1024);
    const int pieces = (int)(bytes_per_wave >> 10);
    for (int i = 0; i < pieces; i += R) {
#pragma unroll
        for (int r = 0; r < R; r++) {
            // slot r was filled R pieces ago: all but the R - 1 youngest loads have landed
            if (i) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(R - 1) : "memory");
            __builtin_amdgcn_global_load_lds((gbl_void *)(p + (size_t)(i + r) * 1024), (lds_void *)(ring + r * 1024), 16, 0, NT ? 2 : 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (*reinterpret_cast<volatile unsigned *>(ring + lane * 16) == 0x12345678u) sink[0] = 1;
}

template <int R>
__global__ __launch_bounds__(1024) void stream_reg(const char *src, size_t bytes_per_wave, unsigned *sink) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const u32x4 *p = reinterpret_cast<const u32x4 *>(src + ((size_t)blockIdx.x * nw + w) * bytes_per_wave) + lane;
    const int pieces = (int)(bytes_per_wave >> 10);
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < pieces; i += R) {
        u32x4 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = p[(size_t)(i + r) * 64];
#pragma unroll
        for (int r = 0; r < R; r++) acc ^= v[r];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

static hipStream_t masked(int n) {
    hipStream_t s;
    if (n >= 256) { CK(hipStreamCreate(&s)); return s; }
    uint32_t mask[8] = {0};
    for (int i = 0; i < n; i++) mask[i >> 5] |= 1u << (i & 31);
    CK(hipExtStreamCreateWithCUMask(&s, 8, mask));  // never destroyed (hipStreamDestroy of a masked stream hangs, ROCm 7.2)
    return s;
}

template <typename F>
static double time_us(hipStream_t st, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); CK(hipStreamSynchronize(st));
    hipEventRecord(a, st); launch(); hipEventRecord(b, st); CK(hipEventSynchronize(b));
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    const size_t total = (size_t)2 << 30;
    char *buf; unsigned *sink;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&sink, 4));
    CK(hipDeviceSynchronize());
    printf("%-5s %-3s %-6s %-3s %10s %9s %14s\n", "CUs", "W", "path", "R", "us", "TB/s", "KB/us per CU");
    for (int n : {256, 128, 96, 64}) {
        hipStream_t st = masked(n);
        for (int W : {8, 16}) {
            const size_t per_wave = (total / ((size_t)n * W)) & ~(size_t)16383;
            const double bytes = (double)per_wave * n * W;
            auto row = [&](const char *path, int R, double us) {
                printf("%-5d %-3d %-6s %-3d %10.1f %9.2f %14.1f\n", n, W, path, R, us, bytes / us * 1e-6, bytes / us / n * 1e-3);
            };
#define LDS_ROW(R, NT, NAME)                                                                                             \
    CK(hipFuncSetAttribute((const void *)stream_lds<R, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, W * R * 1024));  \
    row(NAME, R, time_us(st, [&] { hipLaunchKernelGGL((stream_lds<R, NT>), dim3(n), dim3(W * 64), W * R * 1024, st, buf, per_wave, sink); }));
            LDS_ROW(4, 0, "lds") LDS_ROW(8, 0, "lds")
            if (W * 16 <= 160) { LDS_ROW(16, 0, "lds") }
            LDS_ROW(8, 1, "lds-nt")
#define REG_ROW(R) row("reg", R, time_us(st, [&] { hipLaunchKernelGGL((stream_reg<R>), dim3(n), dim3(W * 64), 0, st, buf, per_wave, sink); }));
            REG_ROW(4) REG_ROW(8) REG_ROW(16)
        }
    }
    return 0;
}
