// xabs_bench: the one-pass "absorbed" cross-attention kernels of k_decode.hip (NH_OPT_ABSORBED_XATTN = 2) on random data, next to
// the K/V kernel they replace: time per launch group, and (built with -DXA_STAMPS) the cycle counter at eight points of one tile.
#include "../norma_amd/csrc/nh_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static void fill(half_t *p, size_t n, float s, unsigned seed) {
    std::vector<half_t> h(n); unsigned x = seed * 2654435761u + 1;
    for (size_t i = 0; i < n; i++) { x = x * 1664525u + 1013904223u; h[i] = (half_t)(s * ((float)(x >> 8) / 8388608.0f - 1.0f)); }
    CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice));
}
int main(int argc, char **argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 64, d = 1280, H = 20, S = 1500;
    CK(hipSetDevice(0));
    half_t *q, *WkT, *Wkv, *xa, *U, *out, *kc, *vc, *out2; float *bkv, *zpart, *mlpart;
    CK(hipMalloc(&q, (size_t)R * d * 2)); CK(hipMalloc(&WkT, (size_t)d * d * 2)); CK(hipMalloc(&Wkv, (size_t)2 * d * d * 2));
    CK(hipMalloc(&xa, (size_t)R * S * d * 2)); CK(hipMalloc(&U, (size_t)R * 32 * d * 2)); CK(hipMalloc(&out, (size_t)R * d * 2)); CK(hipMalloc(&out2, (size_t)R * d * 2));
    CK(hipMalloc(&kc, (size_t)R * S * d * 2)); CK(hipMalloc(&vc, (size_t)R * S * d * 2));
    CK(hipMalloc(&bkv, 2 * d * 4)); CK(hipMalloc(&zpart, (size_t)R * 4 * H * d * 4)); CK(hipMalloc(&mlpart, (size_t)R * 4 * 32 * 2 * 4 + 4096));
    CK(hipMemset(U, 0, (size_t)R * 32 * d * 2)); CK(hipMemset(bkv, 0, 2 * d * 4)); CK(hipMemset(mlpart, 0, (size_t)R * 4 * 32 * 2 * 4 + 4096));
    fill(q, (size_t)R * d, 1.0f, 1); fill(Wkv, (size_t)2 * d * d, 0.03f, 2); fill(xa, (size_t)R * S * d, 1.0f, 3);
    fill(kc, (size_t)R * S * d, 1.0f, 4); fill(vc, (size_t)R * S * d, 1.0f, 5);
    hipStream_t st; CK(hipStreamCreate(&st));
    launch_transpose_sq(Wkv, WkT, d, st);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn) {
        std::vector<float> ms;
        for (int it = 0; it < 12; it++) { CK(hipEventRecord(e0, st)); fn(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float m; CK(hipEventElapsedTime(&m, e0, e1)); ms.push_back(m); }
        std::sort(ms.begin(), ms.end());
        printf("%-52s rows %3d: median %7.1f us  min %7.1f us\n", name, R, ms[6] * 1e3, ms[0] * 1e3); fflush(stdout);
    };
    timeit("K/V cross-attention (dec_attn_kernel, head-major)", [&] { launch_dec_attention(q, kc, vc, out2, R, 1, H, d, S, S, nullptr, st, 1, nullptr); });
    timeit("absorbed: u + main + z-merge + o-projection", [&] { launch_xabs_attention_fast(q, WkT, Wkv, bkv, xa, U, zpart, mlpart, out, R, H, d, S, nullptr, st); });
    CK(hipStreamSynchronize(st));
    std::vector<float> stamps(64);
    CK(hipMemcpy(stamps.data(), mlpart + (size_t)R * 4 * 32 * 2, 64 * 4, hipMemcpyDeviceToHost));
    if (stamps[1] > 0.f) {
        printf("cycle counter of tile 5 per wave: top | stored+barrier | S partial done | barrier | reduced+barrier | softmax done | z updated | barrier\n");
        for (int w = 0; w < 8; w++) { printf("wave %d:", w); for (int k = 0; k < 8; k++) printf(" %7.0f", stamps[w * 8 + k]); printf("\n"); }
    }
    return 0;
}
