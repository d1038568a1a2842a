// gskel: ceiling of a GEMM main loop WITHOUT any global traffic -- LDS fragment reads + MFMAs + one barrier per 64-deep K-tile --
// for two shapes of the same 256 x 256 x 64 workgroup tile:
//   w8: 8 waves (two per SIMD), wave tile 64 x 128 (2 x 4 blocks of 32 x 32), 6 ds_read_b128 per 8 MFMAs     (k_gemm.hip today)
//   w4: 4 waves (one per SIMD), wave tile 128 x 128 (4 x 4 blocks, 256 accumulator registers), 8 reads per 16 MFMAs
//       ("one wave per SIMD with a 128 x 128 wave tile, accumulators in AGPRs")
// Both compiler-scheduled from the same source; the LDS image is the product's conflict-free swizzle.  What this answers: does the
// one-wave-per-SIMD shape raise the loop's own ceiling enough to be worth a rewrite, given that the product loop is also fetch-bound?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void skel(float *out, int ktiles, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char lds[];   // 2 stages x (A 32 KB + B 32 KB)
    constexpr int MB = WAVES == 4 ? 4 : 2;    // 32-row blocks of A per wave
    constexpr int NB = 4;                     // 32-column blocks of B per wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = WAVES == 4 ? (w >> 1) : (w >> 1), wn = w & 1;   // w8: 4 x 2 waves of 64 x 128; w4: 2 x 2 waves of 128 x 128
    for (int i = tid * 16; i < 131072; i += 64 * WAVES * 16) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0x3c003c00u, 0x38003800u, 0x3c003c00u, 0x34003400u);
    __syncthreads();
    const int r = lane & 31, hh = lane >> 5;
    int offA[MB], offB[NB];
#pragma unroll
    for (int i = 0; i < MB; i++) { const int row = (wm * MB + i) * 32 + r; offA[i] = row * 128 + ((hh ^ (row & 7)) << 4); }
#pragma unroll
    for (int j = 0; j < NB; j++) { const int row = (wn * NB + j) * 32 + r; offB[j] = 32768 + row * 128 + ((hh ^ (row & 7)) << 4); }
    float total = 0.f;
    for (int t = 0; t < tiles; t++) {
        f32x16 acc[MB][NB];
#pragma unroll
        for (int i = 0; i < MB; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
        for (int kt = 0; kt < ktiles; kt++) {
            const char *st = lds + (kt & 1) * 65536;
#pragma unroll
            for (int ks = 0; ks < 4; ks++) {   // chunk 2 ks + hh: xor (ks << 5) on the byte offset
                half8 a[MB], b[NB];
#pragma unroll
                for (int i = 0; i < MB; i++) a[i] = *reinterpret_cast<const half8 *>(st + (offA[i] ^ (ks << 5)));
#pragma unroll
                for (int j = 0; j < NB; j++) b[j] = *reinterpret_cast<const half8 *>(st + (offB[j] ^ (ks << 5)));
#pragma unroll
                for (int i = 0; i < MB; i++)
#pragma unroll
                    for (int j = 0; j < NB; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < MB; i++)
#pragma unroll
            for (int j = 0; j < NB; j++)
#pragma unroll
                for (int e = 0; e < 16; e++) total += acc[i][j][e];
    }
    if (total == 12345.678f) out[blockIdx.x] = total;
}

template <int WAVES>
static void run(const char *name, float *out, int ktiles, int tiles) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&skel<WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ms;
    for (int it = 0; it < 7; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(skel<WAVES>, dim3(256), dim3(64 * WAVES), 131072, 0, out, ktiles, tiles);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    const double flops = 256.0 * tiles * ktiles * 2.0 * 256 * 256 * 64;
    printf("%-4s K-tiles per tile %3d: median %8.1f us  %7.0f TFLOP/s  (%.2f us per K-tile)\n", name, ktiles, ms[3] * 1e3, flops / (ms[3] * 1e-3) / 1e12,
           ms[3] * 1e3 / (tiles * ktiles));
}

int main() {
    CK(hipSetDevice(0));
    float *out; CK(hipMalloc(&out, 4096));
    for (int kt : {20, 80}) {
        run<8>("w8", out, kt, 11);
        run<4>("w4", out, kt, 11);
    }
    return 0;
}
