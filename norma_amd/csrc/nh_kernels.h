// nh_kernels.h -- internal launch interface between the C-ABI layer (nh_api.hip) and the gfx950
// kernels.  Not part of the public ABI (that is include/norma_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define NH_MELP 128   // mel channels padded to 128 in the fp16 conv1 input (K = 3*128)
#define NH_SP 1536    // encoder sequence padded to a multiple of 64 for the V^T image
#define NH_DH 64      // head dim of every Whisper size
#define NH_MAX_DEVICES 64  // per-device caches of the launchers (one process may hold contexts on several GPUs)

// ---- big MFMA GEMM: C[M][N] = A[M][K] . W[N][K]^T (+bias), fp16 in, fp32 accumulate ------------
enum GemmEpi {
    EPI_F16 = 0,        // fp16 out = acc + bias, up to 3 column segments with own destinations
    EPI_GELU_F16 = 1,   // fp16 out = gelu_tanh(acc + bias)
    EPI_RESID_F32 = 2,  // f32 x[m][n] += acc + bias
    EPI_CONV2_F32 = 3,  // f32 x[m][n] = gelu_tanh(acc + bias) + pos[m % S][n]
};

struct GemmParams {
    const half_t *A; long lda; int a_rpb; long a_bstride;  // row m at A + (m/a_rpb)*a_bstride + (m%a_rpb)*lda
    const half_t *W;    // [N][K]
    const float *bias;  // [N] or nullptr
    int M, N, K;
    int epi;
    void *out[3]; int seg_n; long ldo;          // EPI_F16*: column segment s = n / seg_n goes to out[s]
    int o_rpb; long o_bstride; long o_off;      // output row = (m/o_rpb)*o_bstride + (m%o_rpb) + o_off
    int vt_seg;                                 // segment written as V^T [b][h][64][SP] (or -1)
    int head_major;                             // EPI_F16: every segment is written [b][h][S][64] (clip b = m / S, 64-wide head h)
                                                // instead of [m][seg_n]: the decoder streams cross K/V per (clip, head)
    float seg0_scale;                           // EPI_F16: segment 0 is written as (acc + bias) * seg0_scale (0 = no scaling): the encoder's
                                                // q carries the softmax scale dh^-1/2 * log2(e) into the attention kernel
    int S, H;                                   // rows per clip / heads (V^T and conv2 epilogues)
    const float *pos;                           // conv2: positional embedding [S][N]
    int cus;                                    // workgroups of the persistent grid; 0 = one per CU (tools/cumask runs it on CU-masked streams)
    unsigned long long *dbg;                    // -DG2_STAMPS diagnostic builds only (tools/gstamps): per-workgroup time stamps; else unused
    // filled in by launch_gemm: multipliers for the kernels' integer divisions by a_rpb, o_rpb and S (nh_magic / nh_div below)
    unsigned a_magic, o_magic, s_magic;
};
// m / d for 0 <= m <= max_m as one multiply-high: magic = ceil(2^32 / d), exact while m * d < 2^32.  magic == 0 encodes
// "d > every m" (quotient 0: a GEMM whose rows are one batch), magic == ~0u "outside the exact range: divide" (no shape of
// this library: d is rows per clip, <= 3000).  The kernels divide row indices by run-time constants (rows per clip, S);
// hipcc's own expansion computes the reciprocal per lane with v_rcp_iflag and keeps it live across the tile loop, where it
// does not fit beside 128 accumulators + 64 fragment registers: it is spilled, and on gfx950 the reload (a scratch load,
// counted in vmcnt) drains the LDS-DMA pipeline in front of every tile's first fetches.
static inline unsigned nh_magic(int d, long max_m) {
    if (d <= 0 || (long)d > max_m) return 0u;
    if ((unsigned long long)max_m * (unsigned long long)d >= 0x100000000ull) return ~0u;
    return (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d);
}
#ifdef __HIPCC__
__device__ __forceinline__ int nh_div(int m, int d, unsigned magic) {
    return magic == 0u ? 0 : magic == ~0u ? m / d : (int)__umulhi((unsigned)m, magic);
}
#endif
void launch_gemm(const GemmParams &p, hipStream_t st);
void launch_gemm_128(const GemmParams &p, hipStream_t st);  // always the 128 x 128 kernel

// ---- skinny GEMM for the decoder: y[R][N] = x[R][K] . W[N][K]^T, R <= 96 rows -------------------
enum SkinnyEpi {
    SK_F16 = 0,         // fp16 out (row stride ldo, per-row base offsets)
    SK_GELU_F16 = 1,
    SK_RESID_F32 = 2,   // f32 x[r][n] += acc + bias
    SK_F32 = 3,         // f32 out = acc (+bias)  (logits)
    SK_QKV = 4,         // self-attn fused q|k|v: q -> out[0] [R][d]; k,v -> caches at position t
};
struct SkinnyParams {
    const half_t *x; long ldx;  // [R][K] fp16
    const half_t *W; const float *bias;
    const half_t *Wt;           // optional tile-major repack of W (launch_repack_tiles); the kernels prefer it
    int R, N, K;
    int epi;
    void *out[3]; long ldo;
    // SK_QKV: row r = b*Tn + i  (Tn new positions per sequence); k, v go to the head-major caches [b][h][ctx][64] at t0 + i
    int d, t0, Tn, ctx;
    const int32_t *pos_ptr;  // when set: i32 [B], sequence b writes its K/V at pos_ptr[b] (hipGraph replay of the decode step)
    // LayerNorm fused into the activation load (skinny_ln_supported): x is ignored, the activations are
    // LN(ln_x[r][0..K)) * ln_w + ln_b (eps 1e-5, two-pass f32 statistics), rounded to fp16 like layernorm_kernel does
    const float *ln_x, *ln_w, *ln_b;
    float ln_rk;  // 1.0f / K, filled in by launch_skinny
};
// true when launch_skinny can take its activations through the fused LayerNorm for this shape
bool skinny_ln_supported(int R, int N, int K);

// ---- the decoder's LayerNorm arithmetic ("sliced") ------------------------------------------------
// One fixed summation tree for every LayerNorm of a decode step, whether it runs fused inside a skinny GEMM, in the
// logits kernel's staging pass or as the stand-alone kernel (rows > 32, teacher-forced views): a row of K = 128 STEPS
// values is cut into 4 slices of 32 STEPS; in slice w lane fq sums the 8 values at 32 s + 8 fq (s ascending), the four
// lanes meet as (l0 + l1) + (l2 + l3), the slices as (p0 + p1) + (p2 + p3).  Results are therefore bit-identical across
// batch sizes and across the fused / unfused paths.
#ifdef __HIPCC__
// tanh-GELU (candle's Activation::Gelu is the tanh form): 0.5 v (1 + tanh(u)) == v / (1 + exp(-2u)),
// u = sqrt(2/pi) v (1 + 0.044715 v^2), as 3 full-rate VALU ops + v_exp_f32 + add + v_rcp_f32 + mul.  The IEEE division
// and expf expansions cost ~3x that (the GELU epilogue of a 256^2 GEMM tile is 128 of these per lane with no MFMA to
// hide under), and the library expf carries fast-math flags that let the compiler fold it differently into different
// kernels; this form is explicit, so every kernel gives the same bits.  v_exp_f32 / v_rcp_f32 are 1 ulp; the result is
// then rounded to fp16 (or added to an O(1) positional embedding).  v -> -inf gives v * rcp(inf) = -0, v -> +inf gives v.
// (hipcc's default -ffp-contract=fast lets the backend fuse any mul + add it meets after inlining, differently from
// kernel to kernel; the helpers below switch that off and spell their FMAs out, so they round the same everywhere.)
__device__ __forceinline__ float gelu_tanh_fast(float v) {
#pragma clang fp contract(off)
    const float k0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f;  // -2 sqrt(2/pi) log2(e)
    const float k1 = k0 * 0.044715f;
    const float m = v * __builtin_fmaf(v * v, k1, k0);
    float r = v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(m));
    asm("" : "+v"(r));  // keep the f32 product: no v_fma_mix*_f16 fusion with a following fp16 conversion (see ln_apply)
    return r;
}
__device__ __forceinline__ float ln_sum8(float acc, const f32x4 &a, const f32x4 &b) {
#pragma clang fp contract(off)
    acc += (a[0] + a[1]) + (a[2] + a[3]);
    acc += (b[0] + b[1]) + (b[2] + b[3]);
    return acc;
}
__device__ __forceinline__ float ln_sq8(float acc, const f32x4 &a, const f32x4 &b, float mean) {
#pragma clang fp contract(off)
    const f32x4 t = a - mean, u = b - mean;
    acc += __builtin_fmaf(t[1], t[1], t[0] * t[0]) + __builtin_fmaf(t[3], t[3], t[2] * t[2]);
    acc += __builtin_fmaf(u[1], u[1], u[0] * u[0]) + __builtin_fmaf(u[3], u[3], u[2] * u[2]);
    return acc;
}
// rk = 1.0f / K, computed by the host (or at compile time): hipcc turns a division by a compile-time constant into a
// multiplication by its reciprocal but divides at run time, which rounds differently
__device__ __forceinline__ float ln_mean(float sum, float rk) {
#pragma clang fp contract(off)
    return sum * rk;
}
__device__ __forceinline__ float ln_inv(float sumsq, float rk) {
#pragma clang fp contract(off)
    const float var = sumsq * rk;
    return 1.0f / sqrtf(var + 1e-5f);
}
__device__ __forceinline__ f32x4 ln_apply(const f32x4 &x, float mean, float inv, const f32x4 &g, const f32x4 &b) {
#pragma clang fp contract(off)
    const f32x4 t = (x - mean) * inv;
    f32x4 o = {__builtin_fmaf(t[0], g[0], b[0]), __builtin_fmaf(t[1], g[1], b[1]), __builtin_fmaf(t[2], g[2], b[2]),
               __builtin_fmaf(t[3], g[3], b[3])};
    // the f32 value has to exist: otherwise hipcc folds the FMA and the fp16 conversion that follows into v_fma_mixlo_f16
    // (ONE rounding, to fp16) in some kernels and keeps v_cvt_pk_f16_f32 (two roundings) in others -- different bits at ties
    asm("" : "+v"(o));
    return o;
}
#endif
// stand-alone kernel with that arithmetic (K % 128 == 0, K <= 1280); returns false when the shape is not covered
bool launch_layernorm_sliced(const float *x, const float *w, const float *b, half_t *y, float *y32, int M, int K, hipStream_t st);
void launch_skinny(const SkinnyParams &p, hipStream_t st);
// out: ceil(N/16) * 16 * K halfs.  Tile-major: the MFMA A fragment of (16-row tile, 32-deep k-step) is 1 KiB contiguous, so a
// GEMV streams its weights like a memcpy (the row-major form reads 16 x 64 B per wave instruction: 3.7 vs 5.1 TB/s).
void launch_repack_tiles(const half_t *W, half_t *out, int N, int K, hipStream_t st);

// ---- elementwise / normalisation -------------------------------------------------------------------
// LayerNorm over rows of f32 x[M][d] -> fp16 y[M][d] (and optionally f32 y32[M][d])
void launch_layernorm(const float *x, const float *w, const float *b, half_t *y, float *y32, int M, int d,
                      hipStream_t st);
// decoder input: x[(b*Tn+i)][:] = E[tokens[b*tok_stride + t0 + i]][:] + P[t0+i][:]
void launch_embed(const int32_t *tokens, int tok_stride, const half_t *E, const half_t *P, float *x, int B,
                  int Tn, int t0, const int32_t *pos_ptr, int d, hipStream_t st);

// ---- sample conversion (src/dtype.rs / dasp_sample): native capture samples -> f32 PCM -------------------------------
// src: `count` samples of NH_SAMPLE_* type `dtype` (device memory); dst: f32
void launch_convert_samples(const void *src, float *dst, long count, int dtype, hipStream_t st);

// ---- log-mel -------------------------------------------------------------------------------------------
struct MelTables {      // device pointers, built once on the host with libm (bit-identical twiddles)
    const float *hann;      // [400]
    const float *dft_cos;   // [25][25]  cos(2pi*k*j/25) evaluated as candle's dft() does in f32
    const float *dft_sin;   // [25][25]
    const float *tw_cos;    // radix-2 twiddles for n = 50,100,200,400: offsets 0,25,75,175 (k < n/2)
    const float *tw_sin;
    const float *filters;   // [n_mel][201]
};
// pcm [B][stride] (device), n_samples[B] (device) -> mel32 f32 [B][n_mel][frames] (unnormalised
// log10) and chunk_max[B] (order-preserving uint encoding of the per-clip max, atomicMax).
// grp: [n_mel][2] first / last+1 group-of-4 index with a non-zero filter tap.
void launch_logmel_grp(const float *pcm, const int32_t *n_samples, long stride, const MelTables &t,
                       const int32_t *grp, int n_mel, int frames, float *mel32, unsigned *chunk_max, int B,
                       hipStream_t st);
// normalise=1: v = max(v, max-8)/4+1 in place; always writes the fp16 conv1 image [B][frames+2][128]
void launch_mel_finish_ex(float *mel32, const unsigned *chunk_max, half_t *img, int B, int n_mel, int frames,
                          int normalise, hipStream_t st);

// ---- encoder attention ---------------------------------------------------------------------------------
// q,k: fp16 [B*S][ld] (head h at column h*64), q PRE-SCALED by NH_ENC_Q_SCALE (GemmParams::seg0_scale);
// vt: fp16 [B][H][64][SP]; out: fp16 [B*S][ldo]
#define NH_ENC_Q_SCALE (0.125f * 1.4426950408889634f)   // dh^-1/2 * log2(e), dh = 64
void launch_enc_attention(const half_t *q, const half_t *k, long ld, const half_t *vt, half_t *out, long ldo,
                          int B, int S, int H, hipStream_t st);

// ---- decoder attention (one query row per (b, h, i)) -------------------------------------------------
// q: fp16 [B*Tn][d]; kc,vc: fp16 [B][ctx][d]; keys visible to new row i: t0 + i + 1 (causal) or Tk (cross)
// pos_ptr != nullptr: i32 [B]; sequence b attends causally over pos_ptr[b] + 1 keys (device-side positions)
// kv_head_major: K/V are [b][h][ctx][64] (the cross K/V written with GemmParams::head_major) instead of [b][ctx][d]
// done: optional i32 [B]; sequences with done[b] != 0 are skipped (their output row is left untouched)
void launch_dec_attention(const half_t *q, const half_t *kc, const half_t *vc, half_t *out, int B, int Tn,
                          int H, int d, int ctx, int Tk, const int32_t *pos_ptr, hipStream_t st, int kv_head_major = 0,
                          const int32_t *done = nullptr);

// numerics prototype of cross-attention on the encoder output itself (NH_OPT_ABSORBED_XATTN; k_decode.hip): Wkv = the fused
// [2 d][d] cross K/V projection (K rows first), bkv its bias, xa fp16 [B][S][d], U scratch fp16 [B][H][d]
void launch_xabs_attention(const half_t *q, const half_t *Wkv, const float *bkv, const half_t *xa, half_t *U, half_t *out, int B, int H, int d, int S,
                           const int32_t *done, hipStream_t st);

// the one-pass form (xa streamed once per layer).  U fp16 [B][32][d] with the rows of heads >= H zero, zpart f32 [B][4][H][d],
// mlpart f32 [B][4][32][2]
bool xabs_fast_supported(int d, int H);
void launch_transpose_sq(const half_t *in, half_t *out, int d, hipStream_t st);   // out[c][r] = in[r][c], d x d, d % 32 == 0
void launch_xabs_attention_fast(const half_t *q, const half_t *WkT, const half_t *Wkv, const float *bkv, const half_t *xa, half_t *U, float *zpart, float *mlpart,
                                half_t *out, int B, int H, int d, int S, const int32_t *done, hipStream_t st);

// ---- logit processor: softmax + norma rules + argmax + bookkeeping -----------------------------------
struct DecodeState {        // all device pointers
    int32_t *tokens;        // [B][ctx]
    int32_t *n_tokens;      // [B]
    int32_t *done;          // [B] 0 running, 1 finished, 2 finished by no-speech early exit
    int32_t *have_last;     // [B]
    int32_t *last_ts;       // [B]
    double *sum_logprob;    // [B]
    double *no_speech;      // [B]
    int32_t *n_active;      // [1] running sequences after the last step
    const uint8_t *suppress;  // [V] 1 = suppressed (suppress_tokens U {no_timestamps})
};
struct RuleTokens { int sot, eot, lang, task, no_speech, no_timestamps, zero_sec, one_sec; };
// mode 0: no-speech probe at prompt position 0; mode 1: generate a token from logits [B][V]; mode 2 (decode pool): every
// sequence is in the phase its own position pos_ptr[b] says -- 0: no-speech probe, < prompt_len - 1: nothing (the next prompt
// token is given), otherwise generate
// partials: f32 [B][8][8] scratch, tickets: u32 [B] zero-initialised (the kernel re-zeroes them)
// pos_ptr != nullptr: i32 [B], the position of every sequence; the kernel advances those it stepped
void launch_logit_step(const float *logits, int V, DecodeState s, RuleTokens tk, int B, int ctx, int cap,
                       int max_new, int prompt_len, int mode, float *partials, unsigned *tickets, int32_t *pos_ptr,
                       hipStream_t st);
// decode pool: sequence `row` restarts at position 0 with the prompt [t0, t1] (P = 2) or [t0, t1, t2] (P = 3)
void launch_pool_admit(DecodeState s, int32_t *pos, unsigned *tickets, int row, int ctx, int t0, int t1, int t2, int P, hipStream_t st);
// t > 0: one SAMPLED token per sequence (model.rs:340-348) under the seeded contract of include/norma_hip.h;
// sequence b draws with clip id clip0 + b, step = its current token count
void launch_sample_step(const float *logits, int V, DecodeState s, RuleTokens tk, int B, int ctx, int cap, int max_new,
                        int prompt_len, float inv_t, unsigned long long seed, unsigned clip0, unsigned attempt, hipStream_t st);
// parity view: rules + one draw on an already soft-maxed probability vector (token -1: everything masked)
void launch_sample_rules(const float *probs_in, int32_t *token_out, const int32_t *tokens, int n_tokens, int last_ts,
                         const uint8_t *suppress, RuleTokens tk, int V, float inv_t, unsigned long long seed, unsigned clip,
                         unsigned attempt, hipStream_t st);
// detect_language: logits [B][ldl] at prompt position 0 -> per-sequence language token (first maximum), optional probs [B][n]
void launch_lang_detect(const float *logits, int V, const int32_t *lang_tokens, int n, float *probs_out, int32_t *lang_out,
                        int B, hipStream_t st);
// parity helper: apply rules to one already soft-maxed probability vector
void launch_rules_only(const float *probs_in, float *masked_out, int32_t *argmax_out, const int32_t *tokens,
                       int n_tokens, int last_ts, const uint8_t *suppress, RuleTokens tk, int V, hipStream_t st);
