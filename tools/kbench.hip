// kbench: per-kernel timing of the decode-step kernels at distil-large-v3 b32 shapes, each measured as
// the average of many back-to-back launches replayed from a hipGraph (so the ~2 us dependent-launch
// floor is included exactly as in the real decode loop).  Usage: tools/bin/kbench [B]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <functional>
#include "../norma_amd/csrc/nh_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static hipStream_t st;
template <typename T> T *dmalloc(size_t n, int fill = 0) { void *p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, fill, n * sizeof(T))); return (T *)p; }
static double bench(const char *name, int reps, std::function<void()> f, double bytes = 0) {
    hipGraph_t g; hipGraphExec_t ex;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; i++) f();
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    CK(hipGraphLaunch(ex, st)); CK(hipStreamSynchronize(st));
    double best = 1e30;
    for (int it = 0; it < 5; it++) {
        hipEventRecord(a, st); CK(hipGraphLaunch(ex, st)); hipEventRecord(b, st); CK(hipEventSynchronize(b));
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    double us = best * 1e3 / reps;
    if (bytes > 0) printf("%-44s %8.2f us   %7.2f TB/s\n", name, us, bytes / us * 1e-6);
    else printf("%-44s %8.2f us\n", name, us);
    hipGraphExecDestroy(ex); hipGraphDestroy(g);
    return us;
}
int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, d = 1280, V = 51866, VP = (V + 63) & ~63, C = 448, S = 1500, H = 20;
    // KBENCH_CUS=n: run everything on a stream confined to the first n CUs (n / 8 per XCD) -- how the decode kernels scale
    // down; the stream is not destroyed (hipStreamDestroy of a masked stream hangs on ROCm 7.2)
    if (getenv("KBENCH_CUS") && atoi(getenv("KBENCH_CUS")) > 0) {
        const int n = atoi(getenv("KBENCH_CUS"));
        uint32_t mask[8] = {0};
        for (int i = 0; i < n && i < 256; i++) mask[i >> 5] |= 1u << (i & 31);
        CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        printf("stream confined to %d CUs\n", n);
    } else CK(hipStreamCreate(&st));
    const size_t ARENA = (size_t)300 << 20;  // halfs (600 MB): larger than the 256 MB Infinity Cache
    half_t *arena = dmalloc<half_t>(ARENA, 0x11), *w_e = arena;
    size_t rot = 0;
    auto W = [&](size_t elems) { if (rot + elems > ARENA) rot = 0; half_t *p = arena + rot; rot += (elems + 127) & ~(size_t)127; return p; };
    half_t *xn = dmalloc<half_t>((size_t)B * 4 * d, 0x11), *q = dmalloc<half_t>((size_t)B * d, 0x11), *att = dmalloc<half_t>((size_t)B * d);
    half_t *hid = dmalloc<half_t>((size_t)B * 4 * d);
    half_t *kc = dmalloc<half_t>((size_t)3 * B * S * d, 0x11), *vc = dmalloc<half_t>((size_t)3 * B * S * d, 0x11); int kvr = 0;
    half_t *sk = dmalloc<half_t>((size_t)B * C * d, 0x11), *sv = dmalloc<half_t>((size_t)B * C * d, 0x11);
    float *x = dmalloc<float>((size_t)B * d), *bias = dmalloc<float>(4 * d), *lnw = dmalloc<float>(d), *lnb = dmalloc<float>(d);
    float *logits = dmalloc<float>((size_t)B * VP), *lpart = dmalloc<float>(B * 64);
    unsigned *ltick = dmalloc<unsigned>(B);
    int32_t *pos = dmalloc<int32_t>(4); (void)pos;
    DecodeState ds{};
    ds.tokens = dmalloc<int32_t>((size_t)B * 4096); ds.n_tokens = dmalloc<int32_t>(B); ds.done = dmalloc<int32_t>(B);
    ds.have_last = dmalloc<int32_t>(B); ds.last_ts = dmalloc<int32_t>(B); ds.sum_logprob = dmalloc<double>(B);
    ds.no_speech = dmalloc<double>(B); ds.n_active = dmalloc<int32_t>(1); ds.suppress = dmalloc<uint8_t>(V);
    std::vector<int32_t> nt(B, 3); CK(hipMemcpy(ds.n_tokens, nt.data(), B * 4, hipMemcpyHostToDevice));
    RuleTokens tk{50258, 50257, 50259, 50360, 50363, 50364, 50365, 50415};
    auto sk_call = [&](const half_t *xin, long ldx, half_t *W, int N, int K, int epi, void *o0, long ldo) {
        SkinnyParams p{}; p.x = xin; p.ldx = ldx; p.W = W; p.bias = bias; p.R = B; p.N = N; p.K = K; p.epi = epi;
        p.out[0] = o0; p.out[1] = sk; p.out[2] = sv; p.ldo = ldo; p.d = d; p.t0 = 5; p.Tn = 1; p.ctx = C;
        launch_skinny(p, st);
    };
    auto ln_call = [&](half_t *W, int N, int K, int epi, void *o0, long ldo) {  // LayerNorm fused into the GEMV
        SkinnyParams p{}; p.x = nullptr; p.ldx = K; p.W = W; p.bias = bias; p.R = B; p.N = N; p.K = K; p.epi = epi;
        p.out[0] = o0; p.out[1] = sk; p.out[2] = sv; p.ldo = ldo; p.d = d; p.t0 = 5; p.Tn = 1; p.ctx = C;
        p.ln_x = x; p.ln_w = lnw; p.ln_b = lnb;
        launch_skinny(p, st);
    };
    const int R = 100;
    if (skinny_ln_supported(B, 3 * d, d)) {
        bench("LN+qkv   N=3840 K=1280 (fused)", R, [&] { ln_call(W(3ul * d * d), 3 * d, d, SK_QKV, q, d); }, 3.0 * d * d * 2);
        bench("LN+cq    N=1280 K=1280 (fused)", R, [&] { ln_call(W(1ul * d * d), d, d, SK_F16, q, d); }, 1.0 * d * d * 2);
        bench("LN+fc1   N=5120 K=1280 (fused)", R, [&] { ln_call(W(4ul * d * d), 4 * d, d, SK_GELU_F16, hid, 4 * d); }, 4.0 * d * d * 2);
        bench("LN+logits N=51866 K=1280 (fused)", 9, [&] { SkinnyParams p{}; p.ldx = d; p.W = W((size_t)V * d); p.R = B; p.N = V; p.K = d; p.epi = SK_F32; p.out[0] = logits; p.ldo = VP; p.ln_x = x; p.ln_w = lnw; p.ln_b = lnb; launch_skinny(p, st); }, (double)V * d * 2);
    }
    {   // the same GEMVs on the tile-major repack of their weights
        half_t *wt = dmalloc<half_t>((size_t)((V + 15) / 16) * 16 * d);
        auto tiled = [&](const char *name, int reps, int N, int K, int epi, void *o0, long ldo, bool ln) {
            half_t *Wr = W((size_t)N * K);
            launch_repack_tiles(Wr, wt, N, K, st);
            bench(name, reps, [&] {
                SkinnyParams p{}; p.x = ln ? nullptr : (K == d ? xn : hid); p.ldx = K; p.W = Wr; p.Wt = wt; p.bias = epi == SK_F32 ? nullptr : bias; p.R = B; p.N = N; p.K = K; p.epi = epi;
                p.out[0] = o0; p.out[1] = sk; p.out[2] = sv; p.ldo = ldo; p.d = d; p.t0 = 5; p.Tn = 1; p.ctx = C;
                if (ln) { p.ln_x = x; p.ln_w = lnw; p.ln_b = lnb; }
                launch_skinny(p, st);
            }, (double)N * K * 2);
        };
        tiled("tiled LN+qkv   N=3840 K=1280", R, 3 * d, d, SK_QKV, q, d, true);
        tiled("tiled o        N=1280 K=1280 (RESID)", R, d, d, SK_RESID_F32, x, d, false);
        tiled("tiled LN+fc1   N=5120 K=1280", R, 4 * d, d, SK_GELU_F16, hid, 4 * d, true);
        tiled("tiled fc2      N=1280 K=5120 (RESID)", R, d, 4 * d, SK_RESID_F32, x, d, false);
        tiled("tiled LN+logits N=51866 K=1280", 9, V, d, SK_F32, logits, VP, true);
    }
    bench("layernorm (sliced) B rows", R, [&] { launch_layernorm_sliced(x, lnw, lnb, xn, nullptr, B, d, st); });
    bench("layernorm B rows", R, [&] { launch_layernorm(x, lnw, lnb, xn, nullptr, B, d, st); });
    bench("embed", R, [&] { launch_embed(ds.tokens, 4096, w_e, w_e, x, B, 1, 5, nullptr, d, st); });
    bench("skinny qkv   N=3840 K=1280 (SK_QKV)", R, [&] { sk_call(xn, d, W(3ul * d * d), 3 * d, d, SK_QKV, q, d); }, 3.0 * d * d * 2);
    bench("skinny o     N=1280 K=1280 (RESID)", R, [&] { sk_call(att, d, W(1ul * d * d), d, d, SK_RESID_F32, x, d); }, 1.0 * d * d * 2);
    bench("skinny cq    N=1280 K=1280 (F16)", R, [&] { sk_call(xn, d, W(1ul * d * d), d, d, SK_F16, q, d); }, 1.0 * d * d * 2);
    bench("skinny fc1   N=5120 K=1280 (GELU)", R, [&] { sk_call(xn, d, W(4ul * d * d), 4 * d, d, SK_GELU_F16, hid, 4 * d); }, 4.0 * d * d * 2);
    bench("skinny fc2   N=1280 K=5120 (RESID)", R, [&] { sk_call(hid, 4 * d, W(4ul * d * d), d, 4 * d, SK_RESID_F32, x, d); }, 4.0 * d * d * 2);
    bench("skinny logits N=51866 K=1280 (F32)", 9, [&] { SkinnyParams p{}; p.x = xn; p.ldx = d; p.W = W((size_t)V * d); p.R = B; p.N = V; p.K = d; p.epi = SK_F32; p.out[0] = logits; p.ldo = VP; launch_skinny(p, st); }, (double)V * d * 2);
    bench("dec_attn self Tk=200", R, [&] { launch_dec_attention(q, sk, sv, att, B, 1, H, d, C, 200, nullptr, st); }, 2.0 * B * 200 * d * 2);
    bench("dec_attn cross Tk=1500", 9, [&] { size_t o = (size_t)(kvr++ % 3) * B * S * d; launch_dec_attention(q, kc + o, vc + o, att, B, 1, H, d, S, S, nullptr, st); }, 2.0 * B * S * d * 2);
    bench("dec_attn cross Tk=1500, head-major K/V", 9, [&] { size_t o = (size_t)(kvr++ % 3) * B * S * d; launch_dec_attention(q, kc + o, vc + o, att, B, 1, H, d, S, S, nullptr, st, 1); }, 2.0 * B * S * d * 2);
    bench("logit_step mode 1", R, [&] { launch_logit_step(logits, V, ds, tk, B, 4096, 1 << 30, 0, 3, 1, lpart, ltick, nullptr, st); }, (double)B * V * 4);
    return 0;
}
