// lnfuse_check: fused LayerNorm + skinny GEMM against the stand-alone sliced LayerNorm followed by the plain skinny GEMM,
// bit for bit, on random rows (development check; tests/test_gpu_parity.py holds the end-to-end form of it).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../norma_amd/csrc/nh_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static float frand(unsigned &s) { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    unsigned seed = 12345;
    const int shapes[][3] = {{32, 1280, 1280}, {32, 1280, 1280}, {32, 1024, 1024}, {16, 5120, 1280}, {17, 1280, 1280}, {32, 1280, 1280}, {32, 3840, 1280}, {32, 1024, 1024}, {32, 1024, 768}, {32, 512, 512}, {32, 768, 384}, {32, 5120, 1280}};
    for (auto &sh : shapes) {
        const int R = sh[0], N = sh[1], K = sh[2];
        std::vector<float> x((size_t)R * K), g(K), b(K), bias(N);
        std::vector<half_t> W((size_t)N * K);
        for (auto &v : x) v = 3.0f * frand(seed) + 0.5f;
        for (auto &v : g) v = 1.0f + 0.3f * frand(seed);
        for (auto &v : b) v = 0.2f * frand(seed);
        for (auto &v : bias) v = 0.1f * frand(seed);
        for (auto &v : W) v = (half_t)(0.05f * frand(seed));
        const bool ident = (N == K);  // W = I, bias = 0: the outputs are the fp16 LayerNorm activations themselves
        if (ident) { for (size_t i = 0; i < W.size(); i++) W[i] = (half_t)((i / K) == (i % K) ? 1.0f : 0.0f); for (auto &v : bias) v = 0.f; }
        float *dx, *dg, *db, *dbias; half_t *dW, *dxn, *o1, *o2;
        CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dg, K * 4)); CK(hipMalloc(&db, K * 4)); CK(hipMalloc(&dbias, N * 4));
        CK(hipMalloc(&dW, W.size() * 2)); CK(hipMalloc(&dxn, (size_t)R * K * 2)); CK(hipMalloc(&o1, (size_t)R * N * 2)); CK(hipMalloc(&o2, (size_t)R * N * 2));
        CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, g.data(), K * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), K * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dbias, bias.data(), N * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
        for (int epi : {SK_F16, SK_GELU_F16}) {
            SkinnyParams p{};
            p.W = dW; p.bias = dbias; p.R = R; p.N = N; p.K = K; p.epi = epi; p.ldo = N; p.d = K; p.Tn = 1; p.ctx = 448;
            // (a) stand-alone LN, plain GEMM
            if (!launch_layernorm_sliced(dx, dg, db, dxn, nullptr, R, K, st)) { printf("sliced LN does not cover K=%d\n", K); return 1; }
            p.x = dxn; p.ldx = K; p.out[0] = o1;
            launch_skinny(p, st);
            // (b) fused
            if (!skinny_ln_supported(R, N, K)) { printf("R=%d N=%d K=%d: fusion not supported\n", R, N, K); continue; }
            p.x = nullptr; p.ln_x = dx; p.ln_w = dg; p.ln_b = db; p.out[0] = o2;
            launch_skinny(p, st);
            CK(hipStreamSynchronize(st));
            std::vector<half_t> h1((size_t)R * N), h2((size_t)R * N);
            CK(hipMemcpy(h1.data(), o1, h1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), o2, h2.size() * 2, hipMemcpyDeviceToHost));
            size_t bad = 0; double maxd = 0;
            for (size_t i = 0; i < h1.size(); i++) {
                if (memcmp(&h1[i], &h2[i], 2) != 0) { bad++; double d = fabs((double)h1[i] - (double)h2[i]); if (d > maxd) maxd = d; }
            }
            printf("R=%2d N=%4d K=%4d epi=%d: %zu of %zu outputs differ (max %.3g)\n", R, N, K, epi, bad, h1.size(), maxd);
            if (ident && epi == SK_F16) {
                std::vector<half_t> hx((size_t)R * K);
                CK(hipMemcpy(hx.data(), dxn, hx.size() * 2, hipMemcpyDeviceToHost));
                size_t b1 = 0, b2 = 0;
                for (size_t i = 0; i < hx.size(); i++) { b1 += memcmp(&hx[i], &h1[i], 2) != 0; b2 += memcmp(&hx[i], &h2[i], 2) != 0; }
                printf("    vs stand-alone LN activations: plain GEMM path %zu differ, fused path %zu differ\n", b1, b2);
                for (size_t i = 0, shown = 0; i < hx.size() && shown < 6; i++)
                    if (memcmp(&hx[i], &h2[i], 2) != 0) {
                        int r = i / K, k = i % K;
                        double m = 0, v = 0; for (int q = 0; q < K; q++) m += x[(size_t)r * K + q]; m /= K;
                        for (int q = 0; q < K; q++) v += (x[(size_t)r * K + q] - m) * (x[(size_t)r * K + q] - m); v /= K;
                        double ref = (x[i] - m) / sqrt(v + 1e-5) * g[k] + b[k];
                        printf("    row %d k %d: stand-alone %.6f fused %.6f  exact %.8f\n", r, k, (float)hx[i], (float)h2[i], ref); shown++;
                    }
            }
        }
    }
    return 0;
}
