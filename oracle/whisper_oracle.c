/*
 * whisper_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT A PRODUCT PATH)
 *
 * A plain-C, f32, batch-1 restatement of the reference's Whisper hot path:
 *
 *     log-mel spectrogram -> Whisper encoder -> greedy decode loop with norma's logit rules
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * shared object.  The product path (norma_amd/csrc, libnorma_hip.so) never links, imports or
 * calls it and has no CPU fallback.
 *
 * PARITY STATUS: **parity unpinned**.  The reference (MikeIvanichev/norma @ 2024_10_08) is
 * Rust and cannot be built here (no cargo/rustc), and its tensor arithmetic lives in
 * un-vendored third-party crates that are absent from /root/reference:
 *     candle-core / candle-nn / candle-transformers 0.7.2   (Cargo.lock:215-216,275-276,293-294)
 * The reference's own tests never execute Whisper (tests/transcriber.rs uses a mock and is
 * #[ignore]d; monolingual.rs:454-535 only parses JSON), so there is no golden vector, known
 * answer or fixture to pin this oracle against.  What pins it instead:
 *   - norma's decode policy is restated line by line from files that ARE in the reference
 *     (citations below, all relative to /root/reference);
 *   - candle's published algorithm (models::whisper::{audio,model} at 0.7.2) is restated from
 *     its public source as summarised in SURVEY.md 3.3 [A]-[D];
 *   - the transformer stack is cross-checked against transformers' Whisper (tanh-GELU) and the
 *     mel front end against an independent numpy restatement (tests/test_oracle.py).
 *
 * Reference call sites followed:
 *   src/models/whisper/model.rs:55-159    Model::transcribe            -> wo_transcribe
 *   src/models/whisper/model.rs:164-191   decode_with_fallback (t = 0) -> wo_transcribe
 *   src/models/whisper/model.rs:212-277   supress_* logit rules        -> wo_apply_rules
 *   src/models/whisper/model.rs:279-389   Model::decode                -> wo_decode
 *   src/models/whisper/model.rs:447-491   Type::{encoder_forward, decoder_forward,
 *                                         decoder_final_linear, reset_kv_cache}
 *   src/models/whisper/model.rs:74        audio::pcm_to_mel [candle]   -> wo_pcm_to_mel
 *   src/models/whisper/monolingual.rs:386-430  the four vocab masks    -> wo_set_tokens
 *   src/utils.rs:29-48                    inclusive_boxed_by           -> boxed_next
 *
 * Determinism: every output element of every contraction is accumulated in plain increasing-k
 * order by exactly one thread, so results do not depend on the OpenMP thread count, and the
 * optional self-attention KV cache (WO_USE_KV_CACHE) is bit-identical to the reference's
 * recompute-the-whole-prefix structure (tests assert this).
 */
#define _GNU_SOURCE
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef float v8 __attribute__((vector_size(32), aligned(4)));

/* ------------------------------------------------------------------------------------------ */
/* public types                                                                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int n_mel;       /* num_mel_bins            */
    int n_audio_ctx; /* max_source_positions    */
    int d;           /* d_model                 */
    int n_head;      /* encoder/decoder heads   */
    int n_enc;       /* encoder_layers          */
    int n_vocab;     /* vocab_size              */
    int n_text_ctx;  /* max_target_positions    */
    int n_dec;       /* decoder_layers          */
} wo_config;

typedef struct {
    int sot, eot, lang /* <0: none yet */, task, no_speech, no_timestamps, zero_sec, one_sec;
} wo_tokens;

#define WO_USE_KV_CACHE 1 /* decode flag: cache decoder self-attn K/V (bit-identical, faster) */

typedef struct { float *wt /* [n_in][n_out] */, *b; int n_out, n_in; } lin_t;
typedef struct { float *w, *b; } ln_t;
typedef struct {
    lin_t q, k, v, o;
    float *kc, *vc; /* cross-attn cache [S][d] (model.rs:485-490 resets it) */
    int kv_len;
    float *sk, *sv; /* optional self-attn cache [n_text_ctx][d] (oracle-only fast mode) */
} mha_t;
typedef struct {
    ln_t attn_ln; mha_t attn;
    ln_t cross_ln; mha_t cross; int has_cross;
    ln_t mlp_ln; lin_t fc1, fc2;
} block_t;

typedef struct wo_model {
    wo_config c;
    lin_t conv1, conv2; /* stored as linear over im2col: n_in = 3*c_in, index k = ci*3 + kk */
    float *enc_pos;     /* sinusoids [n_audio_ctx][d] */
    block_t *enc; ln_t ln_post;
    float *tok_emb;     /* [V][d] */
    float *tok_emb_t;   /* [d][V] */
    float *dec_pos;     /* [n_text_ctx][d] */
    block_t *dec; ln_t dec_ln;
    /* norma Model fields (model.rs:16-42) */
    wo_tokens tk;
    float *suppress_tokens, *supress_non_timestamps, *supress_timestamps, *first_token_supress;
    int self_cache_len;
    /* decode_with_fallback's sampled attempts (model.rs:175-188): off = return the t = 0 result (needed_fallback is
     * reported); on = temperatures 0.2 .. 1.0 under the seeded sampling contract; `slices` numbers the decoded slices */
    int fallback; uint64_t sample_seed; uint32_t slices;
} wo_model;

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */
static void *xcalloc(size_t n, size_t sz) {
    void *p = NULL;
    if (n == 0) n = 1;
    if (posix_memalign(&p, 64, n * sz)) { fprintf(stderr, "oracle: out of memory\n"); abort(); }
    memset(p, 0, n * sz);
    return p;
}

/* ------------------------------------------------------------------------------------------ */
/* GEMM: C[M][N] = A[M][K] * Bt[K][N] (+ bias[N]); every C element sums k = 0..K-1 in order.  */
/* ------------------------------------------------------------------------------------------ */
#define MR 6
#define NR 16
/* contractions use fused multiply-add explicitly (candle's CPU backend, the `gemm` crate, does the
 * same on x86); everything else is compiled with -ffp-contract=off like rustc would. */
#define FMA8(a, b, c) ((v8)_mm256_fmadd_ps((__m256)(a), (__m256)(b), (__m256)(c)))
static inline v8 bcast8(float x) { return (v8){x, x, x, x, x, x, x, x}; }

/* rows [mlo,mhi) x columns [nlo,nhi): 6x16 register tiles over a packed 16-column panel of Bt */
static void gemm_block(int mlo, int mhi, int nlo, int nhi, int K, const float *A, long lda,
                       const float *Bt, long ldb, const float *bias, float *C, long ldc, float *pack) {
    int n0 = nlo;
    for (; n0 + NR <= nhi; n0 += NR) {
        for (int k = 0; k < K; k++) memcpy(pack + (long)k * NR, Bt + (long)k * ldb + n0, NR * sizeof(float));
        v8 bz0 = {0, 0, 0, 0, 0, 0, 0, 0}, bz1 = bz0;
        if (bias) { memcpy(&bz0, bias + n0, 32); memcpy(&bz1, bias + n0 + 8, 32); }
        int m0 = mlo;
        for (; m0 + MR <= mhi; m0 += MR) {
            const float *a0 = A + (long)m0 * lda, *a1 = a0 + lda, *a2 = a1 + lda, *a3 = a2 + lda, *a4 = a3 + lda,
                        *a5 = a4 + lda;
            v8 c00 = bz0, c01 = bz1, c10 = bz0, c11 = bz1, c20 = bz0, c21 = bz1, c30 = bz0, c31 = bz1, c40 = bz0,
               c41 = bz1, c50 = bz0, c51 = bz1;
            const float *bp = pack;
            for (int k = 0; k < K; k++, bp += NR) {
                v8 b0, b1;
                memcpy(&b0, bp, 32); memcpy(&b1, bp + 8, 32);
                v8 s;
                s = bcast8(a0[k]); c00 = FMA8(s, b0, c00); c01 = FMA8(s, b1, c01);
                s = bcast8(a1[k]); c10 = FMA8(s, b0, c10); c11 = FMA8(s, b1, c11);
                s = bcast8(a2[k]); c20 = FMA8(s, b0, c20); c21 = FMA8(s, b1, c21);
                s = bcast8(a3[k]); c30 = FMA8(s, b0, c30); c31 = FMA8(s, b1, c31);
                s = bcast8(a4[k]); c40 = FMA8(s, b0, c40); c41 = FMA8(s, b1, c41);
                s = bcast8(a5[k]); c50 = FMA8(s, b0, c50); c51 = FMA8(s, b1, c51);
            }
            float *c = C + (long)m0 * ldc + n0;
            memcpy(c, &c00, 32); memcpy(c + 8, &c01, 32); c += ldc;
            memcpy(c, &c10, 32); memcpy(c + 8, &c11, 32); c += ldc;
            memcpy(c, &c20, 32); memcpy(c + 8, &c21, 32); c += ldc;
            memcpy(c, &c30, 32); memcpy(c + 8, &c31, 32); c += ldc;
            memcpy(c, &c40, 32); memcpy(c + 8, &c41, 32); c += ldc;
            memcpy(c, &c50, 32); memcpy(c + 8, &c51, 32);
        }
        for (; m0 < mhi; m0++) { /* row tail, same k order */
            const float *a0 = A + (long)m0 * lda;
            v8 c0 = bz0, c1 = bz1;
            const float *bp = pack;
            for (int k = 0; k < K; k++, bp += NR) {
                v8 b0, b1;
                memcpy(&b0, bp, 32); memcpy(&b1, bp + 8, 32);
                v8 s = bcast8(a0[k]);
                c0 = FMA8(s, b0, c0); c1 = FMA8(s, b1, c1);
            }
            memcpy(C + (long)m0 * ldc + n0, &c0, 32); memcpy(C + (long)m0 * ldc + n0 + 8, &c1, 32);
        }
    }
    for (; n0 < nhi; n0++) { /* column tail, same k order and the same fused multiply-add */
        for (int m = mlo; m < mhi; m++) {
            float acc = bias ? bias[n0] : 0.f;
            for (int k = 0; k < K; k++) acc = __builtin_fmaf(A[(long)m * lda + k], Bt[(long)k * ldb + n0], acc);
            C[(long)m * ldc + n0] = acc;
        }
    }
}

static void gemm(int M, int N, int K, const float *A, long lda, const float *Bt, long ldb,
                 const float *bias, float *C, long ldc, int parallel) {
    const int MB = 192, NB = 64;
    int mblocks = (M + MB - 1) / MB, nblocks = (N + NB - 1) / NB;
    if (!parallel || (long)M * N * K < 200000) {
        float *pack = (float *)xcalloc((size_t)K * NR, sizeof(float));
        gemm_block(0, M, 0, N, K, A, lda, Bt, ldb, bias, C, ldc, pack);
        free(pack);
        return;
    }
#pragma omp parallel
    {
        float *pack = (float *)xcalloc((size_t)K * NR, sizeof(float));
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (int mb = 0; mb < mblocks; mb++)
            for (int nb = 0; nb < nblocks; nb++) {
                int mlo = mb * MB, mhi = mlo + MB < M ? mlo + MB : M;
                int nlo = nb * NB, nhi = nlo + NB < N ? nlo + NB : N;
                gemm_block(mlo, mhi, nlo, nhi, K, A, lda, Bt, ldb, bias, C, ldc, pack);
            }
        free(pack);
    }
}

static void linear(const lin_t *l, const float *x, int M, float *y) {
    gemm(M, l->n_out, l->n_in, x, l->n_in, l->wt, l->n_out, l->b, y, l->n_out, 1);
}

/* candle-nn LayerNorm (fused CPU path, eps 1e-5): mean = sum/n, var = sum2/n - mean^2 */
static void layer_norm(const ln_t *ln, const float *x, int M, int d, float *y) {
#pragma omp parallel for if (M > 16)
    for (int m = 0; m < M; m++) {
        const float *r = x + (long)m * d;
        float s = 0.f, s2 = 0.f;
        for (int i = 0; i < d; i++) { s += r[i]; s2 += r[i] * r[i]; }
        float mean = s / d, var = s2 / d - mean * mean;
        float inv = 1.0f / sqrtf(var + 1e-5f);
        float *o = y + (long)m * d;
        for (int i = 0; i < d; i++) o[i] = (r[i] - mean) * inv * ln->w[i] + ln->b[i];
    }
}

/* candle Tensor::gelu = tanh approximation, NOT erf-GELU */
static inline float gelu_tanh(float v) {
    return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * v * (1.0f + 0.044715f * v * v)));
}

/* candle_nn::ops::softmax / softmax_last_dim: max, exp(x - max), sum, divide */
static void softmax_row(float *r, int n) {
    float mx = -INFINITY;
    for (int i = 0; i < n; i++) if (r[i] > mx) mx = r[i];
    float s = 0.f;
    for (int i = 0; i < n; i++) { r[i] = expf(r[i] - mx); s += r[i]; }
    for (int i = 0; i < n; i++) r[i] = r[i] / s;
}

/* ------------------------------------------------------------------------------------------ */
/* [A] log-mel: candle_transformers 0.7.2 models::whisper::audio (called at model.rs:74)      */
/* ------------------------------------------------------------------------------------------ */
#define WO_N_FFT 400
#define WO_HOP 160
#define WO_CHUNK_LENGTH 30
#define WO_N_SAMPLES 480000
#define WO_N_FRAMES 3000

static void wo_dft(const float *in, int n, float *out) {
    const float two_pi = (float)M_PI + (float)M_PI;
    float n_t = (float)n;
    for (int k = 0; k < n; k++) {
        float re = 0.f, im = 0.f;
        for (int j = 0; j < n; j++) {
            float angle = two_pi * (float)k * (float)j / n_t;
            re += in[j] * cosf(angle);
            im -= in[j] * sinf(angle);
        }
        out[2 * k] = re; out[2 * k + 1] = im;
    }
}

/* recursive radix-2 FFT that falls back to the O(n^2) DFT at odd lengths (400->...->25) */
static void wo_fft(const float *in, int n, float *out) {
    if (n == 1) { out[0] = in[0]; out[1] = 0.f; return; }
    if (n % 2 == 1) { wo_dft(in, n, out); return; }
    int h = n / 2;
    float *even = (float *)malloc(sizeof(float) * (size_t)(2 * h + 4 * h));
    float *odd = even + h, *ef = odd + h, *of = ef + 2 * h;
    for (int i = 0; i < n; i++) { if (i % 2 == 0) even[i / 2] = in[i]; else odd[i / 2] = in[i]; }
    wo_fft(even, h, ef);
    wo_fft(odd, h, of);
    const float two_pi = (float)M_PI + (float)M_PI;
    float n_t = (float)n;
    for (int k = 0; k < h; k++) {
        float theta = two_pi * (float)k / n_t;
        float re = cosf(theta), im = -sinf(theta);
        float re_odd = of[2 * k], im_odd = of[2 * k + 1];
        out[2 * k] = ef[2 * k] + re * re_odd - im * im_odd;
        out[2 * k + 1] = ef[2 * k + 1] + re * im_odd + im * re_odd;
        out[2 * (k + h)] = ef[2 * k] - re * re_odd + im * im_odd;
        out[2 * (k + h) + 1] = ef[2 * k + 1] - re * im_odd - im * re_odd;
    }
    free(even);
}

/* number of mel frames pcm_to_mel produces for n samples (SURVEY 3.3[A]-2) */
long wo_mel_frames(long n_samples) {
    long n_len = n_samples / WO_HOP;
    long pad = 100 * WO_CHUNK_LENGTH / 2;
    if (n_len % pad != 0) n_len = (n_len / pad + 1) * pad;
    return n_len + pad;
}

/* out: [n_mel][n_len] row-major, n_len = wo_mel_frames(n) */
void wo_pcm_to_mel(int n_mel, const float *samples_in, long n, const float *filters, float *mel) {
    const int fft_size = WO_N_FFT, fft_step = WO_HOP, n_fft = 1 + fft_size / 2;
    float hann[WO_N_FFT];
    const float two_pi = (float)M_PI + (float)M_PI;
    for (int i = 0; i < fft_size; i++)
        hann[i] = 0.5f * (1.0f - cosf((two_pi * (float)i) / (float)fft_size));
    long n_len = wo_mel_frames(n);
    long n_samples = n_len * fft_step;
    float *samples = (float *)xcalloc((size_t)n_samples + fft_size, sizeof(float));
    memcpy(samples, samples_in, sizeof(float) * (size_t)n);
    long end = n_samples / fft_step + 1 < n_len ? n_samples / fft_step + 1 : n_len;
    /* candle strides frames over 2..12 std threads and sums the disjoint partial results;
     * that is thread-count independent, so a parallel-for over frames is equivalent. */
#pragma omp parallel
    {
        float fft_in[WO_N_FFT], fft_out[2 * WO_N_FFT];
#pragma omp for schedule(static)
        for (long i = 0; i < end; i++) {
            long offset = i * fft_step;
            long lim = n_samples - offset < fft_size ? n_samples - offset : fft_size;
            for (long j = 0; j < lim; j++) fft_in[j] = hann[j] * samples[offset + j];
            for (long j = lim; j < fft_size; j++) fft_in[j] = 0.f;
            wo_fft(fft_in, fft_size, fft_out);
            for (int j = 0; j < fft_size; j++)
                fft_out[j] = fft_out[2 * j] * fft_out[2 * j] + fft_out[2 * j + 1] * fft_out[2 * j + 1];
            for (int j = 1; j < fft_size / 2; j++) fft_out[j] += fft_out[fft_size - j];
            for (int j = 0; j < n_mel; j++) {
                float sum = 0.f;
                int k = 0;
                const float *f = filters + (long)j * n_fft;
                while (k + 3 < n_fft) { /* k < n_fft.saturating_sub(3) */
                    sum += fft_out[k] * f[k] + fft_out[k + 1] * f[k + 1] + fft_out[k + 2] * f[k + 2] +
                           fft_out[k + 3] * f[k + 3];
                    k += 4;
                }
                while (k < n_fft) { sum += fft_out[k] * f[k]; k++; }
                mel[(long)j * n_len + i] = log10f(sum > 1e-10f ? sum : 1e-10f);
            }
        }
    }
    float mmax = -INFINITY;
    for (long i = 0; i < (long)n_mel * n_len; i++) if (mel[i] > mmax) mmax = mel[i];
    mmax -= 8.0f;
    for (long i = 0; i < (long)n_mel * n_len; i++) {
        float v = mel[i] > mmax ? mel[i] : mmax;
        mel[i] = v / 4.0f + 1.0f;
    }
    free(samples);
}

/* ------------------------------------------------------------------------------------------ */
/* model construction / weight loading by HF tensor name                       */
/* ------------------------------------------------------------------------------------------ */
static void lin_alloc(lin_t *l, int n_out, int n_in, int bias) {
    l->n_out = n_out; l->n_in = n_in;
    l->wt = (float *)xcalloc((size_t)n_out * n_in, sizeof(float));
    l->b = bias ? (float *)xcalloc((size_t)n_out, sizeof(float)) : NULL;
}
static void ln_alloc(ln_t *l, int d) {
    l->w = (float *)xcalloc((size_t)d, sizeof(float));
    l->b = (float *)xcalloc((size_t)d, sizeof(float));
}
static void mha_alloc(mha_t *a, int d) {
    lin_alloc(&a->q, d, d, 1); lin_alloc(&a->k, d, d, 0); /* k_proj has no bias */
    lin_alloc(&a->v, d, d, 1); lin_alloc(&a->o, d, d, 1);
    a->kc = a->vc = a->sk = a->sv = NULL; a->kv_len = 0;
}
static void block_alloc(block_t *b, int d, int cross) {
    ln_alloc(&b->attn_ln, d); mha_alloc(&b->attn, d);
    b->has_cross = cross;
    if (cross) { ln_alloc(&b->cross_ln, d); mha_alloc(&b->cross, d); }
    ln_alloc(&b->mlp_ln, d);
    lin_alloc(&b->fc1, 4 * d, d, 1); lin_alloc(&b->fc2, d, 4 * d, 1);
}

wo_model *wo_create(const wo_config *cfg) {
    wo_model *m = (wo_model *)xcalloc(1, sizeof(wo_model));
    m->c = *cfg;
    int d = cfg->d;
    lin_alloc(&m->conv1, d, 3 * cfg->n_mel, 1);
    lin_alloc(&m->conv2, d, 3 * d, 1);
    /* sinusoids(): recomputed in f32, [sin | cos] */
    m->enc_pos = (float *)xcalloc((size_t)cfg->n_audio_ctx * d, sizeof(float));
    {
        int half = d / 2;
        float inc = logf(10000.0f) / (float)(half - 1);
        for (int p = 0; p < cfg->n_audio_ctx; p++)
            for (int i = 0; i < half; i++) {
                float inv = expf((float)i * (-inc));
                float st = (float)p * inv;
                m->enc_pos[(long)p * d + i] = sinf(st);
                m->enc_pos[(long)p * d + half + i] = cosf(st);
            }
    }
    m->enc = (block_t *)xcalloc((size_t)cfg->n_enc, sizeof(block_t));
    for (int i = 0; i < cfg->n_enc; i++) block_alloc(&m->enc[i], d, 0);
    ln_alloc(&m->ln_post, d);
    m->tok_emb = (float *)xcalloc((size_t)cfg->n_vocab * d, sizeof(float));
    m->tok_emb_t = (float *)xcalloc((size_t)cfg->n_vocab * d, sizeof(float));
    m->dec_pos = (float *)xcalloc((size_t)cfg->n_text_ctx * d, sizeof(float));
    m->dec = (block_t *)xcalloc((size_t)cfg->n_dec, sizeof(block_t));
    for (int i = 0; i < cfg->n_dec; i++) block_alloc(&m->dec[i], d, 1);
    ln_alloc(&m->dec_ln, d);
    m->tk.lang = -1;
    return m;
}

static void lin_free(lin_t *l) { free(l->wt); free(l->b); }
static void ln_free(ln_t *l) { free(l->w); free(l->b); }
static void mha_free(mha_t *a) {
    lin_free(&a->q); lin_free(&a->k); lin_free(&a->v); lin_free(&a->o);
    free(a->kc); free(a->vc); free(a->sk); free(a->sv);
}
static void block_free(block_t *b) {
    ln_free(&b->attn_ln); mha_free(&b->attn);
    if (b->has_cross) { ln_free(&b->cross_ln); mha_free(&b->cross); }
    ln_free(&b->mlp_ln); lin_free(&b->fc1); lin_free(&b->fc2);
}
void wo_free(wo_model *m) {
    if (!m) return;
    lin_free(&m->conv1); lin_free(&m->conv2); free(m->enc_pos);
    for (int i = 0; i < m->c.n_enc; i++) block_free(&m->enc[i]);
    for (int i = 0; i < m->c.n_dec; i++) block_free(&m->dec[i]);
    free(m->enc); free(m->dec); ln_free(&m->ln_post); ln_free(&m->dec_ln);
    free(m->tok_emb); free(m->tok_emb_t); free(m->dec_pos);
    free(m->suppress_tokens); free(m->supress_non_timestamps); free(m->supress_timestamps);
    free(m->first_token_supress);
    free(m);
}

/* store W[n_out][n_in] transposed (blocked, so both sides stay cache friendly) */
static void transpose_into(float *dst, const float *src, long rows, long cols) { /* dst[c][r] = src[r][c] */
#pragma omp parallel for collapse(2) schedule(static)
    for (long r0 = 0; r0 < rows; r0 += 64)
        for (long c0 = 0; c0 < cols; c0 += 64) {
            long r1 = r0 + 64 < rows ? r0 + 64 : rows, c1 = c0 + 64 < cols ? c0 + 64 : cols;
            for (long r = r0; r < r1; r++)
                for (long c = c0; c < c1; c++) dst[c * rows + r] = src[r * cols + c];
        }
}
static int set_lin_w(lin_t *l, const float *w, long n) {
    if (n != (long)l->n_out * l->n_in) return -2;
    transpose_into(l->wt, w, l->n_out, l->n_in);
    return 0;
}
static int set_vec(float *dst, long want, const float *src, long n) {
    if (!dst) return -1;
    if (n != want) return -2;
    memcpy(dst, src, sizeof(float) * (size_t)n);
    return 0;
}
static int set_lin(lin_t *l, const char *leaf, const float *data, long n) {
    if (!strcmp(leaf, "weight")) return set_lin_w(l, data, n);
    if (!strcmp(leaf, "bias")) return set_vec(l->b, l->n_out, data, n);
    return -1;
}
static int set_ln(ln_t *l, int d, const char *leaf, const float *data, long n) {
    if (!strcmp(leaf, "weight")) return set_vec(l->w, d, data, n);
    if (!strcmp(leaf, "bias")) return set_vec(l->b, d, data, n);
    return -1;
}
static int set_mha(mha_t *a, const char *rest, const float *data, long n) {
    if (!strncmp(rest, "q_proj.", 7)) return set_lin(&a->q, rest + 7, data, n);
    if (!strncmp(rest, "k_proj.", 7)) return set_lin(&a->k, rest + 7, data, n);
    if (!strncmp(rest, "v_proj.", 7)) return set_lin(&a->v, rest + 7, data, n);
    if (!strncmp(rest, "out_proj.", 9)) return set_lin(&a->o, rest + 9, data, n);
    return -1;
}
static int set_block(block_t *b, int d, const char *rest, const float *data, long n) {
    if (!strncmp(rest, "self_attn.", 10)) return set_mha(&b->attn, rest + 10, data, n);
    if (!strncmp(rest, "self_attn_layer_norm.", 21)) return set_ln(&b->attn_ln, d, rest + 21, data, n);
    if (b->has_cross && !strncmp(rest, "encoder_attn.", 13)) return set_mha(&b->cross, rest + 13, data, n);
    if (b->has_cross && !strncmp(rest, "encoder_attn_layer_norm.", 24))
        return set_ln(&b->cross_ln, d, rest + 24, data, n);
    if (!strncmp(rest, "fc1.", 4)) return set_lin(&b->fc1, rest + 4, data, n);
    if (!strncmp(rest, "fc2.", 4)) return set_lin(&b->fc2, rest + 4, data, n);
    if (!strncmp(rest, "final_layer_norm.", 17)) return set_ln(&b->mlp_ln, d, rest + 17, data, n);
    return -1;
}

/* returns 0 ok, 1 ignored (tensor candle does not read), -1 unknown name, -2 wrong size */
int wo_set_tensor(wo_model *m, const char *name, const float *data, long n) {
    int d = m->c.d;
    const char *p;
    if (!strcmp(name, "model.encoder.embed_positions.weight") || !strcmp(name, "proj_out.weight"))
        return 1;
    if ((p = "model.encoder.conv1.", !strncmp(name, p, strlen(p)))) {
        /* conv weight [co][ci][3] is already W[co][k = ci*3+kk] */
        return set_lin(&m->conv1, name + strlen(p), data, n);
    }
    if ((p = "model.encoder.conv2.", !strncmp(name, p, strlen(p))))
        return set_lin(&m->conv2, name + strlen(p), data, n);
    if ((p = "model.encoder.layer_norm.", !strncmp(name, p, strlen(p))))
        return set_ln(&m->ln_post, d, name + strlen(p), data, n);
    if ((p = "model.decoder.layer_norm.", !strncmp(name, p, strlen(p))))
        return set_ln(&m->dec_ln, d, name + strlen(p), data, n);
    if (!strcmp(name, "model.decoder.embed_tokens.weight")) {
        if (n != (long)m->c.n_vocab * d) return -2;
        memcpy(m->tok_emb, data, sizeof(float) * (size_t)n);
        transpose_into(m->tok_emb_t, data, m->c.n_vocab, d);
        return 0;
    }
    if (!strcmp(name, "model.decoder.embed_positions.weight"))
        return set_vec(m->dec_pos, (long)m->c.n_text_ctx * d, data, n);
    int enc = !strncmp(name, "model.encoder.layers.", 21);
    int dec = !strncmp(name, "model.decoder.layers.", 21);
    if (enc || dec) {
        char *end;
        long idx = strtol(name + 21, &end, 10);
        if (*end != '.' || idx < 0 || idx >= (enc ? m->c.n_enc : m->c.n_dec)) return -1;
        return set_block(enc ? &m->enc[idx] : &m->dec[idx], d, end + 1, data, n);
    }
    return -1;
}

/* monolingual.rs:376-430: special-token ids and the four -inf/0 vocab masks */
void wo_set_tokens(wo_model *m, const wo_tokens *tk, const int *suppress, int n_suppress) {
    m->tk = *tk;
    int V = m->c.n_vocab;
    free(m->suppress_tokens); free(m->supress_non_timestamps); free(m->supress_timestamps);
    free(m->first_token_supress);
    m->suppress_tokens = (float *)xcalloc((size_t)V, sizeof(float));
    m->supress_non_timestamps = (float *)xcalloc((size_t)V, sizeof(float));
    m->supress_timestamps = (float *)xcalloc((size_t)V, sizeof(float));
    m->first_token_supress = (float *)xcalloc((size_t)V, sizeof(float));
    for (int i = 0; i < V; i++) {
        int sup = (i == tk->no_timestamps);
        for (int j = 0; j < n_suppress && !sup; j++) sup = (suppress[j] == i);
        m->suppress_tokens[i] = sup ? -INFINITY : 0.f;                          /* :386-395 */
        m->supress_non_timestamps[i] = i > tk->no_timestamps ? 0.f : -INFINITY; /* :397-406 */
        m->supress_timestamps[i] = i > tk->no_timestamps ? -INFINITY : 0.f;     /* :408-417 */
        m->first_token_supress[i] = (i < tk->zero_sec || i > tk->one_sec) ? -INFINITY : 0.f; /* :419-430 */
    }
}
void wo_get_mask(const wo_model *m, int which, float *out) {
    const float *src = which == 0 ? m->suppress_tokens : which == 1 ? m->supress_non_timestamps
                     : which == 2 ? m->supress_timestamps : m->first_token_supress;
    memcpy(out, src, sizeof(float) * (size_t)m->c.n_vocab);
}

/* ------------------------------------------------------------------------------------------ */
/* attention (candle MultiHeadAttention::qkv_attention)                                       */
/* ------------------------------------------------------------------------------------------ */
/* q [Tq][d], k/v [Tk][d]; causal_off >= 0: query row i may see keys j <= i + causal_off.     */
static void attention(const float *q, int Tq, const float *k, const float *v, int Tk, int d,
                      int n_head, int causal_off, float *out) {
    int dh = d / n_head;
    float scale = (float)pow((double)dh, -0.25);
    if (Tq >= 64) {
        /* encoder: per head, scores via GEMM */
#pragma omp parallel
        {
            float *qs = (float *)xcalloc((size_t)Tq * dh, sizeof(float));
            float *kst = (float *)xcalloc((size_t)Tk * dh, sizeof(float));
            float *sc = (float *)xcalloc((size_t)Tq * Tk, sizeof(float));
#pragma omp for schedule(dynamic, 1)
            for (int h = 0; h < n_head; h++) {
                for (int i = 0; i < Tq; i++)
                    for (int c = 0; c < dh; c++) qs[(long)i * dh + c] = q[(long)i * d + h * dh + c] * scale;
                for (int j = 0; j < Tk; j++)
                    for (int c = 0; c < dh; c++) kst[(long)c * Tk + j] = k[(long)j * d + h * dh + c] * scale;
                gemm(Tq, Tk, dh, qs, dh, kst, Tk, NULL, sc, Tk, 0);
                for (int i = 0; i < Tq; i++) {
                    float *r = sc + (long)i * Tk;
                    if (causal_off >= 0)
                        for (int j = i + causal_off + 1; j < Tk; j++) r[j] += -INFINITY;
                    softmax_row(r, Tk);
                }
                gemm(Tq, dh, Tk, sc, Tk, v + h * dh, d, NULL, out + h * dh, d, 0);
            }
            free(qs); free(kst); free(sc);
        }
        return;
    }
    /* decoder: few query rows; same arithmetic (k order 0..dh-1, j order 0..Tk-1) */
#pragma omp parallel
    {
        float *sc = (float *)xcalloc((size_t)Tk, sizeof(float));
        float *qs = (float *)xcalloc((size_t)dh, sizeof(float));
#pragma omp for collapse(2) schedule(static)
        for (int h = 0; h < n_head; h++)
            for (int i = 0; i < Tq; i++) {
                for (int c = 0; c < dh; c++) qs[c] = q[(long)i * d + h * dh + c] * scale;
                for (int j = 0; j < Tk; j++) {
                    const float *kr = k + (long)j * d + h * dh;
                    float acc = 0.f;
                    for (int c = 0; c < dh; c++) acc = __builtin_fmaf(qs[c], kr[c] * scale, acc);
                    if (causal_off >= 0 && j > i + causal_off) acc += -INFINITY;
                    sc[j] = acc;
                }
                softmax_row(sc, Tk);
                float *o = out + (long)i * d + h * dh;
                for (int c = 0; c < dh; c++) o[c] = 0.f;
                for (int j = 0; j < Tk; j++) {
                    const float *vr = v + (long)j * d + h * dh;
                    float w = sc[j];
                    for (int c = 0; c < dh; c++) o[c] = __builtin_fmaf(w, vr[c], o[c]);
                }
            }
        free(sc); free(qs);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* [B] encoder (candle AudioEncoder::forward; model.rs:455-464)                               */
/* ------------------------------------------------------------------------------------------ */
/* mel: [n_mel][frames] with row stride mel_stride; out: [frames/2][d] */
void wo_encoder_forward(wo_model *m, const float *mel, int frames, long mel_stride, float *out) {
    int d = m->c.d, n_mel = m->c.n_mel, S = (frames + 2 - 3) / 2 + 1;
    /* conv1 k=3 s=1 p=1 as im2col + gemm, then gelu */
    float *col1 = (float *)xcalloc((size_t)frames * 3 * n_mel, sizeof(float));
    for (int t = 0; t < frames; t++)
        for (int ci = 0; ci < n_mel; ci++)
            for (int kk = 0; kk < 3; kk++) {
                int s = t + kk - 1;
                col1[(long)t * 3 * n_mel + ci * 3 + kk] = (s >= 0 && s < frames) ? mel[(long)ci * mel_stride + s] : 0.f;
            }
    float *h1 = (float *)xcalloc((size_t)frames * d, sizeof(float));
    linear(&m->conv1, col1, frames, h1);
    free(col1);
    for (long i = 0; i < (long)frames * d; i++) h1[i] = gelu_tanh(h1[i]);
    /* conv2 k=3 s=2 p=1 */
    float *col2 = (float *)xcalloc((size_t)S * 3 * d, sizeof(float));
#pragma omp parallel for
    for (int t = 0; t < S; t++)
        for (int ci = 0; ci < d; ci++)
            for (int kk = 0; kk < 3; kk++) {
                int s = 2 * t + kk - 1;
                col2[(long)t * 3 * d + ci * 3 + kk] = (s >= 0 && s < frames) ? h1[(long)s * d + ci] : 0.f;
            }
    float *x = (float *)xcalloc((size_t)S * d, sizeof(float));
    linear(&m->conv2, col2, S, x);
    free(col2); free(h1);
    for (long i = 0; i < (long)S * d; i++) x[i] = gelu_tanh(x[i]) + m->enc_pos[i]; /* transpose + pos */
    float *ln = (float *)xcalloc((size_t)S * d, sizeof(float));
    float *q = (float *)xcalloc((size_t)S * d, sizeof(float));
    float *k = (float *)xcalloc((size_t)S * d, sizeof(float));
    float *v = (float *)xcalloc((size_t)S * d, sizeof(float));
    float *wv = (float *)xcalloc((size_t)S * d, sizeof(float));
    float *hid = (float *)xcalloc((size_t)S * 4 * d, sizeof(float));
    for (int l = 0; l < m->c.n_enc; l++) {
        block_t *b = &m->enc[l];
        layer_norm(&b->attn_ln, x, S, d, ln);
        linear(&b->attn.q, ln, S, q); linear(&b->attn.k, ln, S, k); linear(&b->attn.v, ln, S, v);
        attention(q, S, k, v, S, d, m->c.n_head, -1, wv);
        linear(&b->attn.o, wv, S, q);
        for (long i = 0; i < (long)S * d; i++) x[i] = x[i] + q[i];
        layer_norm(&b->mlp_ln, x, S, d, ln);
        linear(&b->fc1, ln, S, hid);
#pragma omp parallel for
        for (long i = 0; i < (long)S * 4 * d; i++) hid[i] = gelu_tanh(hid[i]);
        linear(&b->fc2, hid, S, q);
        for (long i = 0; i < (long)S * d; i++) x[i] = x[i] + q[i];
    }
    layer_norm(&m->ln_post, x, S, d, out);
    free(x); free(ln); free(q); free(k); free(v); free(wv); free(hid);
}

/* ------------------------------------------------------------------------------------------ */
/* [D] decoder (candle TextDecoder::forward / final_linear; model.rs:466-483)                 */
/* ------------------------------------------------------------------------------------------ */
void wo_reset_kv_cache(wo_model *m) { /* model.rs:485-490 */
    for (int l = 0; l < m->c.n_dec; l++) {
        free(m->dec[l].cross.kc); free(m->dec[l].cross.vc);
        m->dec[l].cross.kc = m->dec[l].cross.vc = NULL; m->dec[l].cross.kv_len = 0;
    }
    m->self_cache_len = 0;
}

/* Forward rows [t0, T) of the prefix `tokens[0..T)`.  t0 == 0 is the reference structure
 * (whole prefix recomputed, no self-attention cache).  t0 > 0 is the oracle-only fast mode: rows
 * < t0 come from the self K/V cache filled by earlier calls.  out: [T - t0][d]. */
static void decoder_rows(wo_model *m, const int *tokens, int T, int t0, const float *xa, int S, int flush,
                         float *out) {
    int d = m->c.d, R = T - t0;
    float *x = (float *)xcalloc((size_t)R * d, sizeof(float));
    for (int r = 0; r < R; r++)
        for (int i = 0; i < d; i++)
            x[(long)r * d + i] = m->tok_emb[(long)tokens[t0 + r] * d + i] + m->dec_pos[(long)(t0 + r) * d + i];
    float *ln = (float *)xcalloc((size_t)R * d, sizeof(float));
    float *q = (float *)xcalloc((size_t)R * d, sizeof(float));
    float *wv = (float *)xcalloc((size_t)R * d, sizeof(float));
    float *hid = (float *)xcalloc((size_t)R * 4 * d, sizeof(float));
    for (int l = 0; l < m->c.n_dec; l++) {
        block_t *b = &m->dec[l];
        if (!b->attn.sk) {
            b->attn.sk = (float *)xcalloc((size_t)m->c.n_text_ctx * d, sizeof(float));
            b->attn.sv = (float *)xcalloc((size_t)m->c.n_text_ctx * d, sizeof(float));
        }
        layer_norm(&b->attn_ln, x, R, d, ln);
        linear(&b->attn.q, ln, R, q);
        linear(&b->attn.k, ln, R, b->attn.sk + (long)t0 * d);
        linear(&b->attn.v, ln, R, b->attn.sv + (long)t0 * d);
        /* causal mask [448,448] sliced to [T,T]: row i sees j <= i (absolute positions) */
        attention(q, R, b->attn.sk, b->attn.sv, T, d, m->c.n_head, t0, wv);
        linear(&b->attn.o, wv, R, q);
        for (long i = 0; i < (long)R * d; i++) x[i] = x[i] + q[i];
        /* cross attention with K/V cached until the next flush */
        if (flush) { free(b->cross.kc); free(b->cross.vc); b->cross.kc = b->cross.vc = NULL; b->cross.kv_len = 0; }
        if (!b->cross.kc) {
            b->cross.kc = (float *)xcalloc((size_t)S * d, sizeof(float));
            b->cross.vc = (float *)xcalloc((size_t)S * d, sizeof(float));
            linear(&b->cross.k, xa, S, b->cross.kc);
            linear(&b->cross.v, xa, S, b->cross.vc);
            b->cross.kv_len = S;
        }
        layer_norm(&b->cross_ln, x, R, d, ln);
        linear(&b->cross.q, ln, R, q);
        attention(q, R, b->cross.kc, b->cross.vc, b->cross.kv_len, d, m->c.n_head, -1, wv);
        linear(&b->cross.o, wv, R, q);
        for (long i = 0; i < (long)R * d; i++) x[i] = x[i] + q[i];
        layer_norm(&b->mlp_ln, x, R, d, ln);
        linear(&b->fc1, ln, R, hid);
        for (long i = 0; i < (long)R * 4 * d; i++) hid[i] = gelu_tanh(hid[i]);
        linear(&b->fc2, hid, R, q);
        for (long i = 0; i < (long)R * d; i++) x[i] = x[i] + q[i];
    }
    layer_norm(&m->dec_ln, x, R, d, out);
    free(x); free(ln); free(q); free(wv); free(hid);
}

void wo_decoder_forward(wo_model *m, const int *tokens, int T, const float *xa, int S, int flush, float *out) {
    decoder_rows(m, tokens, T, 0, xa, S, flush, out);
}

/* logits[T][V] = x[T][d] . E^T (tied embedding, no bias) */
void wo_final_linear(wo_model *m, const float *x, int T, float *logits) {
    gemm(T, m->c.n_vocab, m->c.d, x, m->c.d, m->tok_emb_t, m->c.n_vocab, NULL, logits, m->c.n_vocab, 1);
}

/* ------------------------------------------------------------------------------------------ */
/* logit rules on masked PROBABILITIES (model.rs:212-277)                                     */
/* ------------------------------------------------------------------------------------------ */
static void add_mask(float *p, const float *mask, int V) { for (int i = 0; i < V; i++) p[i] += mask[i]; }

static void supress_past_timestamps(const wo_model *m, float *p, int last_timestep) { /* :225-243 */
    for (int i = 0; i < m->c.n_vocab; i++)
        p[i] += (i > m->tk.no_timestamps && i <= last_timestep) ? -INFINITY : 0.f;
}
static void supress_non_timestamps(const wo_model *m, float *p, int last_timestep) { /* :216-223 */
    supress_past_timestamps(m, p, last_timestep);
    add_mask(p, m->supress_non_timestamps, m->c.n_vocab);
}
/* model.rs:245-277 */
void wo_apply_rules(const wo_model *m, float *p, const int *tokens, int n_tokens, int last_timestep) {
    int V = m->c.n_vocab, NT = m->tk.no_timestamps;
    add_mask(p, m->suppress_tokens, V);
    int l_token = tokens[n_tokens - 1];
    if (l_token > NT) {
        if (n_tokens >= 2 && tokens[n_tokens - 2] >= m->tk.eot) { add_mask(p, m->supress_timestamps, V); return; }
        supress_non_timestamps(m, p, last_timestep);
        return;
    }
    /* candle fast_sum / fast_max over contiguous slices: plain f32 accumulation */
    float sum_prob_timestamp = 0.f;
    for (int i = NT + 1; i < V; i++) sum_prob_timestamp += p[i];
    float prob_non_timestamp = -INFINITY;
    for (int i = 0; i < NT; i++) if (p[i] > prob_non_timestamp) prob_non_timestamp = p[i];
    if (sum_prob_timestamp >= prob_non_timestamp) supress_non_timestamps(m, p, last_timestep);
    else supress_past_timestamps(m, p, last_timestep);
}

/* f32::total_cmp key */
static inline int32_t total_key(float f);
static inline int32_t total_key(float f) {
    int32_t b; memcpy(&b, &f, 4);
    return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1);
}
/* Iterator::max_by(total_cmp): the LAST maximum wins (model.rs:350-356) */
int wo_argmax_total(const float *p, int n) {
    int best = 0; int32_t bk = total_key(p[0]);
    for (int i = 1; i < n; i++) { int32_t k = total_key(p[i]); if (k >= bk) { bk = k; best = i; } }
    return best;
}

/* ------------------------------------------------------------------------------------------ */
/* Model::detect_language (model.rs:194-210): one decoder step on [sot], softmax over the       */
/* language-token logits, stable descending sort by total_cmp -> the FIRST maximum wins          */
/* ------------------------------------------------------------------------------------------ */
int wo_detect_language(wo_model *m, const float *xa, int S, const int *lang_tokens, int n_lang, float *probs_out) {
    int V = m->c.n_vocab, d = m->c.d;
    int tok = m->tk.sot;
    float *ys = (float *)xcalloc((size_t)d, sizeof(float));
    float *logits = (float *)xcalloc((size_t)V, sizeof(float));
    float *pl = (float *)xcalloc((size_t)n_lang, sizeof(float));
    decoder_rows(m, &tok, 1, 0, xa, S, 1, ys);
    wo_final_linear(m, ys, 1, logits);
    for (int i = 0; i < n_lang; i++) pl[i] = logits[lang_tokens[i]]; /* index_select */
    softmax_row(pl, n_lang);
    int best = 0; int32_t bk = total_key(pl[0]);
    for (int i = 1; i < n_lang; i++) { int32_t k = total_key(pl[i]); if (k > bk) { bk = k; best = i; } }
    if (probs_out) memcpy(probs_out, pl, sizeof(float) * (size_t)n_lang);
    int res = lang_tokens[best];
    free(ys); free(logits); free(pl);
    return res;
}

void wo_set_language(wo_model *m, int lang_token) { m->tk.lang = lang_token; } /* LanguageState::set_language_token */

/* ------------------------------------------------------------------------------------------ */
/* Sampled decoding at t > 0 (model.rs:340-348).  The reference draws from rand's WeightedIndex */
/* with an entropy-seeded StdRng (model.rs:30, monolingual.rs:433), so only the DISTRIBUTION   */
/* can be matched; what is restated here is the distribution, under this build's seeded        */
/* contract (include/norma_hip.h, "sampling contract"), which the HIP path follows bit for bit: */
/*   weights  w_i = sexp((q_i - max q) * inv_t), q = the rule-masked probabilities (:333-338),  */
/*            i.e. softmax(q / t) of :341 up to its normalisation, which WeightedIndex ignores; */
/*   uniform  u = (philox4x32-10(key = seed, ctr = {step, clip, attempt, "norm"})[0] >> 8) 2^-24 */
/*   choice   first j whose cumulative weight exceeds u * total  (WeightedIndex::sample's       */
/*            partition_point), cumulated in f64 over chunks of ceil(V / 1024) tokens.           */
/* ------------------------------------------------------------------------------------------ */
static void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3], k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
void wo_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }

/* exp(y) for y <= 0 from IEEE f32 operations only (Cephes expf polynomial), so that the C and the HIP side round alike */
static float sexp(float y) {
    if (!(y >= -87.0f)) return 0.0f; /* also -inf and NaN: a masked token has weight 0 */
    float kf = floorf(fmaf(y, 1.44269504088896341f, 0.5f));
    float r = fmaf(kf, -0.693359375f, y);
    r = fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float z = r * r;
    float res = fmaf(p, z, r) + 1.0f;
    union { uint32_t u; float f; } sc;
    sc.u = (uint32_t)((int)kf + 127) << 23;
    return res * sc.f;
}
float wo_sexp(float y) { return sexp(y); }

/* q: rule-masked probabilities [V].  Returns the sampled token, or -1 when every entry is masked (the reference pushes
 * eot and stops, model.rs:343-346). */
int wo_sample_token(const float *q, int V, float inv_t, uint64_t seed, uint32_t clip, uint32_t step, uint32_t attempt) {
    float qmax = -INFINITY;
    for (int i = 0; i < V; i++) if (q[i] > qmax) qmax = q[i];
    if (!(qmax > -INFINITY)) return -1;
    const int CH = (V + 1023) / 1024;
    double chunk[1024];
    for (int c = 0; c < 1024; c++) {
        double sc = 0.0;
        for (int i = c * CH; i < (c + 1) * CH && i < V; i++) sc += (double)sexp((q[i] - qmax) * inv_t);
        chunk[c] = sc;
    }
    double total = 0.0;
    for (int c = 0; c < 1024; c++) total += chunk[c];
    uint32_t ctr[4] = {step, clip, attempt, 0x6e6f726du}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)}, r[4];
    philox4x32_10(ctr, key, r);
    const float u = (float)(r[0] >> 8) * (1.0f / 16777216.0f);
    const double x = (double)u * total;
    double run = 0.0;
    int c = 0;
    for (; c < 1023; c++) { if (run + chunk[c] > x) break; run += chunk[c]; }
    int last_pos = -1;
    for (int i = c * CH; i < V; i++) { /* walks on past the chunk only if rounding left x >= the chunk's end */
        const float w = sexp((q[i] - qmax) * inv_t);
        if (w > 0.0f) last_pos = i;
        run += (double)w;
        if (run > x) return i;
    }
    return last_pos; /* x == total exactly: the last token with weight */
}

/* ------------------------------------------------------------------------------------------ */
/* Model::decode at t = 0 (model.rs:279-389)                                                  */
/* ------------------------------------------------------------------------------------------ */
/* tokens_out must hold n_text_ctx + 1 ints.  max_new_tokens <= 0: reference behaviour (cap at
 * n_text_ctx - 1 tokens).  step_probs (optional, [cap][4]): per generated token
 * {p(next), best competing masked prob, sum_ts, max_text} for margin analysis in tests. */
int wo_decode_t(wo_model *m, const float *xa, int S, int flags, int max_new_tokens, int *tokens_out,
                double *avg_logprob, double *no_speech_prob, float *step_probs, double temperature, uint64_t seed,
                int clip, int attempt);
int wo_decode(wo_model *m, const float *xa, int S, int flags, int max_new_tokens, int *tokens_out,
              double *avg_logprob, double *no_speech_prob, float *step_probs) {
    return wo_decode_t(m, xa, S, flags, max_new_tokens, tokens_out, avg_logprob, no_speech_prob, step_probs, 0.0, 0, 0, 0);
}
/* temperature > 0: the token is sampled (model.rs:340-348) under the seeded contract above */
int wo_decode_t(wo_model *m, const float *xa, int S, int flags, int max_new_tokens, int *tokens_out,
                double *avg_logprob, double *no_speech_prob, float *step_probs, double temperature, uint64_t seed,
                int clip, int attempt) {
    int V = m->c.n_vocab, d = m->c.d, cap = m->c.n_text_ctx - 1;
    int *tokens = tokens_out, n = 0;
    double sum_logprob = 0.0;
    tokens[n++] = m->tk.sot;
    if (m->tk.lang >= 0) tokens[n++] = m->tk.lang;
    tokens[n++] = m->tk.task;
    int have_last = 0, last_timestamp = 0;
    float *ys = (float *)xcalloc((size_t)(cap + 2) * d, sizeof(float));
    float *logits = (float *)xcalloc((size_t)V, sizeof(float));
    /* :293-305 -- prompt forward with flush = true, logits at position 0 */
    decoder_rows(m, tokens, n, 0, xa, S, 1, ys);
    wo_final_linear(m, ys, 1, logits);
    softmax_row(logits, V);
    *no_speech_prob = (double)logits[m->tk.no_speech];
    if (*no_speech_prob > 0.6) { /* NO_SPEECH_THRESHOLD, :308-315 */
        *avg_logprob = 0.0;
        free(ys); free(logits);
        return n;
    }
    int cached = 0, new_tokens = 0;
    if (flags & WO_USE_KV_CACHE) cached = n; /* rows 0..n-1 are in the self cache from the prompt pass */
    while (tokens[n - 1] != m->tk.eot) { /* :317 */
        const float *last_row;
        if (flags & WO_USE_KV_CACHE) {
            if (cached < n) { decoder_rows(m, tokens, n, cached, xa, S, 0, ys + (long)cached * d); cached = n; }
            last_row = ys + (long)(n - 1) * d;
        } else {
            decoder_rows(m, tokens, n, 0, xa, S, 0, ys);
            last_row = ys + (long)(n - 1) * d;
        }
        wo_final_linear(m, last_row, 1, logits);
        softmax_row(logits, V); /* :331 -- masks are applied to probabilities from here on */
        if (have_last) wo_apply_rules(m, logits, tokens, n, last_timestamp);
        else add_mask(logits, m->first_token_supress, V); /* :336-337 */
        int next;
        if (temperature > 0.0) { /* :340-348 */
            next = wo_sample_token(logits, V, 1.0f / (float)temperature, seed, (uint32_t)clip, (uint32_t)n, (uint32_t)attempt);
            if (next < 0) { tokens[n++] = m->tk.eot; break; } /* all NaN: push eot, stop */
        } else {
            next = wo_argmax_total(logits, V);            /* :350-356 */
        }
        if (step_probs) {
            float second = -INFINITY, sum_ts = 0.f, max_text = -INFINITY;
            for (int i = 0; i < V; i++) {
                if (i != next && logits[i] > second) second = logits[i];
                if (i > m->tk.no_timestamps) { if (logits[i] > -INFINITY) sum_ts += logits[i]; }
                else if (logits[i] > max_text) max_text = logits[i];
            }
            float *sp = step_probs + (long)new_tokens * 4;
            sp[0] = logits[next]; sp[1] = second; sp[2] = sum_ts; sp[3] = max_text;
        }
        if (next > m->tk.no_timestamps) { last_timestamp = next; have_last = 1; } /* :359-361 */
        tokens[n++] = next;
        new_tokens++;
        sum_logprob += log((double)logits[next]); /* :364-365 */
        if (n >= cap) { tokens[n++] = m->tk.eot; break; } /* :367-370 */
        if (max_new_tokens > 0 && new_tokens >= max_new_tokens && tokens[n - 1] != m->tk.eot) {
            tokens[n++] = m->tk.eot; /* bench knob, not reference behaviour */
            break;
        }
    }
    *avg_logprob = sum_logprob / (double)n; /* :373 */
    while (n >= 2 && tokens[n - 2] > m->tk.no_timestamps) { /* :375-381 */
        memmove(tokens + n - 2, tokens + n - 1, sizeof(int));
        n--;
    }
    free(ys); free(logits);
    return n;
}

/* ------------------------------------------------------------------------------------------ */
/* src/utils.rs:29-48 inclusive_boxed_by                                                      */
/* ------------------------------------------------------------------------------------------ */
typedef struct { const int *v; int len; int finished; } boxed_iter;
static int boxed_pred(const wo_model *m, int tok) { return tok > m->tk.no_timestamps || tok == m->tk.eot; }
static int boxed_next(const wo_model *m, boxed_iter *it, const int **seg, int *seg_len) {
    if (it->finished) return 0;
    int s = -1;
    for (int i = 0; i < it->len; i++) if (boxed_pred(m, it->v[i])) { s = i; break; }
    if (s >= 0) {
        for (int i = s + 1; i < it->len; i++)
            if (boxed_pred(m, it->v[i])) {
                int e = i + 1;
                *seg = it->v + s; *seg_len = e - s;
                it->v += e; it->len -= e;
                return 1;
            }
    }
    it->finished = 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* Model::transcribe + decode_with_fallback restricted to the t = 0 pass (model.rs:55-191)    */
/* ------------------------------------------------------------------------------------------ */
/* buf/buf_len: the model's carried-over PCM buffer (caller-owned storage of capacity buf_cap).
 * Emits, instead of text (no tokenizer offline), the token ids that the reference would hand to
 * tokenizer.decode (model.rs:147), segments separated by -1, into text_tokens.
 * Deviation H1: a no-speech early return (tokens without timestamps) drains the
 * slice instead of spinning forever.  Returns number of ints written to text_tokens, or < 0. */
int wo_transcribe(wo_model *m, const float *filters, float *buf, long *buf_len, int final_chunk, int flags,
                  int max_new_tokens, int *text_tokens, int text_cap, int *n_slices, double *last_avg_logprob,
                  double *last_no_speech) {
    int n_out = 0, d = m->c.d;
    *n_slices = 0;
    int *tokens = (int *)xcalloc((size_t)m->c.n_text_ctx + 2, sizeof(int));
    while (*buf_len > 0) { /* 'new_chunk, :68 */
        long slice_len = *buf_len < WO_N_SAMPLES ? *buf_len : WO_N_SAMPLES;
        long n_len = wo_mel_frames(slice_len);
        float *mel = (float *)xcalloc((size_t)m->c.n_mel * n_len, sizeof(float));
        wo_pcm_to_mel(m->c.n_mel, buf, slice_len, filters, mel);
        int frames = n_len < WO_N_FRAMES ? (int)n_len : WO_N_FRAMES; /* narrow, :88 */
        int S = (frames + 2 - 3) / 2 + 1;
        float *xa = (float *)xcalloc((size_t)S * d, sizeof(float));
        wo_encoder_forward(m, mel, frames, n_len, xa); /* decode_with_fallback :168 */
        free(mel);
        double alp = 0.0, nsp = 0.0;
        int n = 0, have = 0;
        static const double temps[6] = {0.0, 0.2, 0.4, 0.6, 0.8, 1.0}; /* candle m::TEMPERATURES */
        for (int a = 0; a < 6 && !have; a++) { /* decode_with_fallback, :175-188 */
            n = wo_decode_t(m, xa, S, flags, max_new_tokens, tokens, &alp, &nsp, NULL, temps[a], m->sample_seed, (int)m->slices, a);
            int needs_fallback = alp < -1.0; /* compression_ratio is NaN: that comparison is always false (:177) */
            if (!needs_fallback || nsp > 0.6 || !m->fallback) have = 1;
        }
        m->slices++;
        free(xa);
        (*n_slices)++;
        *last_avg_logprob = alp; *last_no_speech = nsp;
        int drained = 0;
        if (!have) { /* :89-92: no attempt was acceptable */
            memmove(buf, buf + slice_len, sizeof(float) * (size_t)(*buf_len - slice_len));
            *buf_len -= slice_len;
            continue;
        }
        if (nsp > 0.6 && alp < -1.0) { /* :95-98 */
            memmove(buf, buf + slice_len, sizeof(float) * (size_t)(*buf_len - slice_len));
            *buf_len -= slice_len;
            continue;
        }
        boxed_iter it = {tokens, n, n == 0};
        const int *seg; int seg_len, stop_all = 0, any_seg = 0;
        while (boxed_next(m, &it, &seg, &seg_len)) { /* :100-150 */
            any_seg = 1;
            int s_timestamp = seg[0] - m->tk.no_timestamps - 1;
            int e_tok = seg[seg_len - 1];
            if (e_tok == m->tk.eot) {
                if (s_timestamp == 0 || final_chunk) {
                    if (slice_len == WO_N_SAMPLES || final_chunk) {
                        memmove(buf, buf + slice_len, sizeof(float) * (size_t)(*buf_len - slice_len));
                        *buf_len -= slice_len; drained = 1;
                    } else { stop_all = 1; break; }
                } else {
                    long pre = *buf_len;
                    long dr = (long)s_timestamp * 320 < slice_len ? (long)s_timestamp * 320 : slice_len;
                    memmove(buf, buf + dr, sizeof(float) * (size_t)(*buf_len - dr));
                    *buf_len -= dr; drained = 1;
                    if (pre > slice_len) break;
                    stop_all = 1; break;
                }
            }
            if (n_out + seg_len - 2 + 1 > text_cap) { free(tokens); return -1; }
            for (int i = 1; i < seg_len - 1; i++) text_tokens[n_out++] = seg[i]; /* tokens[1..len-1], :147 */
            text_tokens[n_out++] = -1;
        }
        if (stop_all) break;
        if (!any_seg && !drained) { /* H1: reference would loop forever here; drain instead */
            memmove(buf, buf + slice_len, sizeof(float) * (size_t)(*buf_len - slice_len));
            *buf_len -= slice_len;
        } else if (!drained) {
            /* segments ended on timestamps only (no eot segment): reference leaves buf untouched and
             * re-enters the loop on the same data forever; treat like H1. */
            memmove(buf, buf + slice_len, sizeof(float) * (size_t)(*buf_len - slice_len));
            *buf_len -= slice_len;
        }
    }
    if (final_chunk) { wo_reset_kv_cache(m); } /* :153-156 (lang.clear() is a no-op for ConstLang) */
    free(tokens);
    return n_out;
}

void wo_set_sampling(wo_model *m, int enable_fallback, uint64_t seed) { m->fallback = enable_fallback; m->sample_seed = seed; m->slices = 0; }

void wo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int wo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
