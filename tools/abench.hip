// abench: the encoder attention kernel alone, several builds of k_attn_enc.hip linked side by side (tools/Makefile: each with
// its own -DENC_ATTN_KERNEL / -DENC_ATTN_LAUNCH names and variant switches), interleaved rounds in ONE process on random
// data -- box-to-box spread on this pool is +-5 %, larger than most of what is being compared.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "../norma_amd/csrc/nh_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef void (*attn_fn)(const half_t *, const half_t *, long, const half_t *, half_t *, long, int, int, int, hipStream_t);
#define DECL(n) void n(const half_t *, const half_t *, long, const half_t *, half_t *, long, int, int, int, hipStream_t);
ABENCH_DECLS
__global__ void fill_rand(half_t *p, size_t n, unsigned seed, float amp) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (half_t)(((int)(x & 0xffff) - 32768) * (amp / 32768.f));
    }
}
int main() {
    const int B = 32, S = 1500, H = 20, d = 1280;
    hipStream_t st; CK(hipStreamCreate(&st));
    half_t *q, *k, *vt, *out;
    CK(hipMalloc(&q, (size_t)B * S * d * 2)); CK(hipMalloc(&k, (size_t)B * S * d * 2)); CK(hipMalloc(&vt, (size_t)B * d * NH_SP * 2)); CK(hipMalloc(&out, (size_t)B * S * d * 2));
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, q, (size_t)B * S * d, 1u, 1.0f);   // scores of a few units, like LayerNorm-fed projections
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, k, (size_t)B * S * d, 7u, 1.0f);
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, st, vt, (size_t)B * d * NH_SP, 9u, 1.0f);
    CK(hipStreamSynchronize(st));
    struct V { const char *name; attn_fn f; std::vector<float> us; };
    std::vector<V> vs = { ABENCH_LIST };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (auto &v : vs) for (int i = 0; i < 3; i++) v.f(q, k, d, vt, out, d, B, S, H, st);
    CK(hipStreamSynchronize(st));
    for (int round = 0; round < 6; round++)
        for (auto &v : vs) {
            hipEventRecord(a, st);
            for (int i = 0; i < 8; i++) v.f(q, k, d, vt, out, d, B, S, H, st);
            hipEventRecord(b, st); CK(hipEventSynchronize(b));
            float ms; hipEventElapsedTime(&ms, a, b); v.us.push_back(ms * 1e3f / 8);
        }
    const double flop = 4.0 * S * (double)S * 64 * H * B;
    for (auto &v : vs) {
        std::sort(v.us.begin(), v.us.end());
        printf("%-28s median %7.1f us  min %7.1f us   %6.0f TFLOP/s (algorithmic, at the median)\n", v.name, v.us[v.us.size() / 2], v.us[0], flop / (v.us[v.us.size() / 2] * 1e-6) / 1e12);
    }
    return 0;
}
