// k_decode.hip -- the per-token decoder kernels (gfx950).  All HBM-bound: weights, the tied
// embedding and the cross-attention K/V are streamed once per generated token for the whole batch.
//
// Replaces, per decode step of Model::decode (src/models/whisper/model.rs:317-371):
//   skinny_gemm_kernel   candle Linear matmuls of TextDecoder::forward and final_linear for the
//                        newest position only (the reference recomputes the whole prefix: it has
//                        no self-attention KV cache; a cache is mathematically identical)
//   dec_attn_kernel      qkv_attention of the decoder blocks (causal self-attention over the cache,
//                        cross-attention over the K/V cached at flush time)
//   logit_step_kernel    softmax over V, the suppression rules on PROBABILITIES (model.rs:212-277,
//                        :331-338), greedy argmax with Iterator::max_by(total_cmp) semantics
//                        (:350-356, last maximum wins), log-prob bookkeeping (:359-370) -- replacing
//                        a 207 KB D2H + a fresh [V] mask H2D + three sync scalar reads per token.
#include "nh_kernels.h"

__device__ __forceinline__ float gelu_tanh_d(float v) {
    float u = 0.7978845608028654f * v * (1.0f + 0.044715f * v * v);
    return v / (1.0f + __expf(-2.0f * u));
}

// ---------------------------------------------------------------------------------------------------
// skinny GEMM: y[R][N] = x[R][K] . W[N][K]^T, R <= 64.  One workgroup = 16 output features; its 4
// waves split K; weights are the MFMA A operand (each lane streams 16 B of one weight row per step,
// 4 lanes cover a 64 B run), activations (L2-resident) the B operand; fp32 partials meet in LDS.
// ---------------------------------------------------------------------------------------------------
template <int NCB>
__global__ __launch_bounds__(256) void skinny_gemm_kernel(SkinnyParams p) {
    __shared__ f32x4 red[4][NCB][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16;
    int wrow = n0 + fr; if (wrow >= p.N) wrow = p.N - 1;
    const int kslice = p.K >> 2, kbeg = w * kslice;
    const half_t *wp = p.W + (long)wrow * p.K + kbeg + 8 * fq;
    const half_t *xp[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        int r = 16 * cb + fr; if (r >= p.R) r = p.R - 1;
        xp[cb] = p.x + (long)r * p.ldx + kbeg + 8 * fq;
    }
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int steps = kslice >> 5;
    int s = 0;
    for (; s + 4 <= steps; s += 4) {  // 4 weight loads in flight per lane
        half8 a0 = *reinterpret_cast<const half8 *>(wp + 32 * (s + 0));
        half8 a1 = *reinterpret_cast<const half8 *>(wp + 32 * (s + 1));
        half8 a2 = *reinterpret_cast<const half8 *>(wp + 32 * (s + 2));
        half8 a3 = *reinterpret_cast<const half8 *>(wp + 32 * (s + 3));
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            half8 b0 = *reinterpret_cast<const half8 *>(xp[cb] + 32 * (s + 0));
            half8 b1 = *reinterpret_cast<const half8 *>(xp[cb] + 32 * (s + 1));
            half8 b2 = *reinterpret_cast<const half8 *>(xp[cb] + 32 * (s + 2));
            half8 b3 = *reinterpret_cast<const half8 *>(xp[cb] + 32 * (s + 3));
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a3, b3, acc[cb], 0, 0, 0);
        }
    }
    for (; s < steps; s++) {
        half8 a0 = *reinterpret_cast<const half8 *>(wp + 32 * s);
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            half8 b0 = *reinterpret_cast<const half8 *>(xp[cb] + 32 * s);
            acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0, acc[cb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) red[w][cb][lane] = acc[cb];
    __syncthreads();
    // thread t owns row r = t / 4 and the 4 consecutive features n0 + 4 (t % 4) + i:
    // D[n = 4 fq + i][r = fr] lives in lane 16 fq + fr of column block r / 16
    const int r = tid >> 2, nq = tid & 3;
    if (r >= 16 * NCB || r >= p.R) return;
    const int src_lane = 16 * nq + (r & 15), cb = r >> 4;
    f32x4 v = red[0][cb][src_lane];
    v += red[1][cb][src_lane];
    v += red[2][cb][src_lane];
    v += red[3][cb][src_lane];
    const int n = n0 + 4 * nq;
    if (n >= p.N) return;
    if (p.bias) {
        if (n + 3 < p.N) v += *reinterpret_cast<const f32x4 *>(p.bias + n);
        else for (int i = 0; i < 4 && n + i < p.N; i++) v[i] += p.bias[n + i];
    }
    if (p.epi == SK_F32) {
        float *dst = reinterpret_cast<float *>(p.out[0]) + (long)r * p.ldo + n;
        if (n + 3 < p.N) *reinterpret_cast<f32x4 *>(dst) = v;
        else for (int i = 0; i < 4 && n + i < p.N; i++) dst[i] = v[i];
        return;
    }
    // the remaining epilogues have N % 4 == 0
    if (p.epi == SK_RESID_F32) {
        float *dst = reinterpret_cast<float *>(p.out[0]) + (long)r * p.ldo + n;
        f32x4 x = *reinterpret_cast<const f32x4 *>(dst);
        *reinterpret_cast<f32x4 *>(dst) = x + v;
        return;
    }
    if (p.epi == SK_GELU_F16) { v[0] = gelu_tanh_d(v[0]); v[1] = gelu_tanh_d(v[1]); v[2] = gelu_tanh_d(v[2]); v[3] = gelu_tanh_d(v[3]); }
    half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
    if (p.epi == SK_QKV) {
        const int sg = n / p.d, nl = n - sg * p.d;
        const int b = r / p.Tn, i = r - b * p.Tn;
        half_t *dst;
        if (sg == 0) dst = reinterpret_cast<half_t *>(p.out[0]) + (long)r * p.d + nl;
        else dst = reinterpret_cast<half_t *>(sg == 1 ? p.out[1] : p.out[2]) + ((long)b * p.ctx + p.t0 + i) * p.d + nl;
        *reinterpret_cast<half4 *>(dst) = hv;
        return;
    }
    *reinterpret_cast<half4 *>(reinterpret_cast<half_t *>(p.out[0]) + (long)r * p.ldo + n) = hv;
}

void launch_skinny(const SkinnyParams &p, hipStream_t st) {
    dim3 grid((p.N + 15) / 16), block(256);
    int ncb = (p.R + 15) / 16;
    if (ncb <= 1) hipLaunchKernelGGL(skinny_gemm_kernel<1>, grid, block, 0, st, p);
    else if (ncb == 2) hipLaunchKernelGGL(skinny_gemm_kernel<2>, grid, block, 0, st, p);
    else if (ncb == 3) hipLaunchKernelGGL(skinny_gemm_kernel<3>, grid, block, 0, st, p);
    else hipLaunchKernelGGL(skinny_gemm_kernel<4>, grid, block, 0, st, p);
}

// ---------------------------------------------------------------------------------------------------
// decoder attention, one query row per (b, h): 4 waves split the keys, inside a wave 8 key slots x
// 8 lanes (16 B of the 64-wide head each); each slot runs its own online softmax, merged at the end.
// ---------------------------------------------------------------------------------------------------
struct AttnPart { float m, l; float acc[8]; };

__device__ __forceinline__ void merge_part(AttnPart &a, float mo, float lo, const float (&ao)[8]) {
    float mn = fmaxf(a.m, mo);
    float sa = (a.m == -INFINITY) ? 0.f : __expf(a.m - mn);
    float so = (mo == -INFINITY) ? 0.f : __expf(mo - mn);
    a.l = a.l * sa + lo * so;
#pragma unroll
    for (int c = 0; c < 8; c++) a.acc[c] = a.acc[c] * sa + ao[c] * so;
    a.m = mn;
}

__global__ __launch_bounds__(256) void dec_attn_kernel(const half_t *__restrict__ q, const half_t *__restrict__ kc,
                                                       const half_t *__restrict__ vc, half_t *__restrict__ out,
                                                       int d, int ctx, int Tk) {
    __shared__ float part[4][8][10];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int slot = lane >> 3, pp = lane & 7;
    const int h = blockIdx.x, b = blockIdx.y;
    float qv[8];
    {
        half8 qh = *reinterpret_cast<const half8 *>(q + (long)b * d + h * NH_DH + 8 * pp);
#pragma unroll
        for (int c = 0; c < 8; c++) qv[c] = (float)qh[c];
    }
    // candle scales q and k by dh^-1/4 each; the product of the two scalings is exactly 1/8
    const int per = (((Tk + 3) >> 2) + 7) & ~7;  // keys per wave, multiple of 8
    const int kbeg = w * per, kend = min(Tk, kbeg + per);
    const half_t *kb = kc + (long)b * ctx * d + h * NH_DH + 8 * pp;
    const half_t *vb = vc + (long)b * ctx * d + h * NH_DH + 8 * pp;
    AttnPart st; st.m = -INFINITY; st.l = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) st.acc[c] = 0.f;
    for (int j0 = kbeg; j0 < kend; j0 += 32) {
        half8 kk[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            int j = j0 + 8 * u + slot; if (j >= kend) j = kend - 1;
            kk[u] = *reinterpret_cast<const half8 *>(kb + (long)j * d);
            vv[u] = *reinterpret_cast<const half8 *>(vb + (long)j * d);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 8; c++) s += qv[c] * (float)kk[u][c];
            s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
            s *= 0.125f;
            if (j0 + 8 * u + slot < kend) {
                float mn = fmaxf(st.m, s);
                float al = __expf(st.m - mn);  // exp(-inf) = 0 on the first key
                float pr = __expf(s - mn);
                st.l = st.l * al + pr;
#pragma unroll
                for (int c = 0; c < 8; c++) st.acc[c] = st.acc[c] * al + pr * (float)vv[u][c];
                st.m = mn;
            }
        }
    }
    // merge the 8 key slots of the wave (lane bits 3..5)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        float mo = __shfl_xor(st.m, o), lo = __shfl_xor(st.l, o);
        float ao[8];
#pragma unroll
        for (int c = 0; c < 8; c++) ao[c] = __shfl_xor(st.acc[c], o);
        merge_part(st, mo, lo, ao);
    }
    if (slot == 0) {
        part[w][pp][0] = st.m; part[w][pp][1] = st.l;
#pragma unroll
        for (int c = 0; c < 8; c++) part[w][pp][2 + c] = st.acc[c];
    }
    __syncthreads();
    if (tid < 8) {
        AttnPart a; a.m = part[0][tid][0]; a.l = part[0][tid][1];
#pragma unroll
        for (int c = 0; c < 8; c++) a.acc[c] = part[0][tid][2 + c];
        for (int ww = 1; ww < 4; ww++) {
            float ao[8];
#pragma unroll
            for (int c = 0; c < 8; c++) ao[c] = part[ww][tid][2 + c];
            merge_part(a, part[ww][tid][0], part[ww][tid][1], ao);
        }
        float inv = 1.0f / a.l;
        half8 o;
#pragma unroll
        for (int c = 0; c < 8; c++) o[c] = (half_t)(a.acc[c] * inv);
        *reinterpret_cast<half8 *>(out + (long)b * d + h * NH_DH + 8 * tid) = o;
    }
}

void launch_dec_attention(const half_t *q, const half_t *kc, const half_t *vc, half_t *out, int B, int Tn,
                          int H, int d, int ctx, int Tk, int causal_t0, hipStream_t st) {
    (void)Tn; (void)causal_t0;  // one new position per sequence; its visible keys are exactly Tk
    hipLaunchKernelGGL(dec_attn_kernel, dim3(H, B), dim3(256), 0, st, q, kc, vc, out, d, ctx, Tk);
}

// ---------------------------------------------------------------------------------------------------
// logit processor
// ---------------------------------------------------------------------------------------------------
// f32::total_cmp key (Rust std): flip the magnitude bits of negative numbers
__device__ __forceinline__ int total_key(float f) {
    int b = __float_as_int(f);
    return b ^ (int)(((unsigned)(b >> 31)) >> 1);
}

struct BlockRed {
    float fa[16], fb[16]; int ia[16], ib[16];
};

__device__ __forceinline__ float block_max(float v, BlockRed &sm) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm.fa[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm.fa[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) r = fmaxf(r, sm.fa[i]);
    return r;
}
__device__ __forceinline__ float block_sum(float v, BlockRed &sm) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm.fb[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm.fb[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) r += sm.fb[i];
    return r;
}
// argmax under total_cmp, last maximum wins
__device__ __forceinline__ void block_argmax(int key, int idx, BlockRed &sm, int &okey, int &oidx) {
    for (int o = 32; o > 0; o >>= 1) {
        int k2 = __shfl_xor(key, o), i2 = __shfl_xor(idx, o);
        if (k2 > key || (k2 == key && i2 > idx)) { key = k2; idx = i2; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sm.ia[threadIdx.x >> 6] = key; sm.ib[threadIdx.x >> 6] = idx; }
    __syncthreads();
    key = sm.ia[0]; idx = sm.ib[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); i++) {
        int k2 = sm.ia[i], i2 = sm.ib[i];
        if (k2 > key || (k2 == key && i2 > idx)) { key = k2; idx = i2; }
    }
    okey = key; oidx = idx;
}

enum { RULE_FIRST = 0, RULE_SUP_TS = 1, RULE_NON_TS = 2, RULE_PAST = 3 };

// masked probability of index i: p + (0 | -inf) exactly as the chain of broadcast_adds produces it
__device__ __forceinline__ float masked_value(float p, int i, int rule, const uint8_t *sup, const RuleTokens &tk,
                                              int last_ts) {
    bool m;
    if (rule == RULE_FIRST) m = (i < tk.zero_sec || i > tk.one_sec);                    // model.rs:336-337
    else if (rule == RULE_SUP_TS) m = sup[i] || i > tk.no_timestamps;                   // :256-259
    else if (rule == RULE_NON_TS) m = sup[i] || i <= tk.no_timestamps || i <= last_ts;  // :216-223
    else m = sup[i] || (i > tk.no_timestamps && i <= last_ts);                          // :225-243
    return m ? p + (-INFINITY) : p;
}

// shared by the decode step and the parity helper.  probs(i) gives the soft-maxed probability.
template <typename ProbFn>
__device__ __forceinline__ void rules_argmax(ProbFn probs, int V, const int32_t *tokens, int n, int have_last,
                                             int last_ts, const uint8_t *sup, const RuleTokens &tk, BlockRed &sm,
                                             int &rule_out, int &next_out) {
    int rule;
    if (!have_last) rule = RULE_FIRST;
    else {
        int l = tokens[n - 1];
        if (l > tk.no_timestamps) {
            int sl = n >= 2 ? tokens[n - 2] : -1;
            rule = (n >= 2 && sl >= tk.eot) ? RULE_SUP_TS : RULE_NON_TS;
        } else {
            float ps = 0.f, pm = -INFINITY;  // model.rs:263-270 on the suppress-masked probabilities
            for (int i = threadIdx.x; i < V; i += blockDim.x) {
                float p = probs(i);
                float pv = sup[i] ? p + (-INFINITY) : p;
                if (i > tk.no_timestamps) ps += pv;
                else if (i < tk.no_timestamps) pm = fmaxf(pm, pv);
            }
            float sum_ts = block_sum(ps, sm);
            float max_text = block_max(pm, sm);
            rule = (sum_ts >= max_text) ? RULE_NON_TS : RULE_PAST;
        }
    }
    int bk = INT_MIN, bi = -1;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        float v = masked_value(probs(i), i, rule, sup, tk, last_ts);
        int k = total_key(v);
        if (k > bk || (k == bk && i > bi)) { bk = k; bi = i; }
    }
    int ok, oi;
    block_argmax(bk, bi, sm, ok, oi);
    rule_out = rule; next_out = oi;
}

__global__ __launch_bounds__(1024) void logit_step_kernel(const float *__restrict__ logits, int V, int ldl,
                                                          DecodeState s, RuleTokens tk, int ctx, int cap,
                                                          int max_new, int prompt_len, int mode) {
    __shared__ BlockRed sm;
    const int b = blockIdx.x;
    if (s.done[b]) return;
    const float *lg = logits + (long)b * ldl;
    // candle_nn::ops::softmax: max, exp(x - max), sum, divide
    float mx = -INFINITY;
    for (int i = threadIdx.x; i < V; i += blockDim.x) mx = fmaxf(mx, lg[i]);
    mx = block_max(mx, sm);
    float se = 0.f;
    for (int i = threadIdx.x; i < V; i += blockDim.x) se += expf(lg[i] - mx);
    se = block_sum(se, sm);
    if (mode == 0) {  // model.rs:293-315: no-speech probability at prompt position 0
        if (threadIdx.x == 0) {
            float p = expf(lg[tk.no_speech] - mx) / se;
            s.no_speech[b] = (double)p;
            if ((double)p > 0.6) s.done[b] = 2;
        }
        return;
    }
    const int n = s.n_tokens[b];
    const int have_last = s.have_last[b], last_ts = s.last_ts[b];
    auto probs = [&](int i) { return expf(lg[i] - mx) / se; };
    int rule, next;
    rules_argmax(probs, V, s.tokens + (long)b * ctx, n, have_last, last_ts, s.suppress, tk, sm, rule, next);
    if (threadIdx.x == 0) {
        float pv = masked_value(probs(next), next, rule, s.suppress, tk, last_ts);
        int32_t *toks = s.tokens + (long)b * ctx;
        int nn = n;
        if (next > tk.no_timestamps) { s.last_ts[b] = next; s.have_last[b] = 1; }  // :359-361
        toks[nn++] = next;
        s.sum_logprob[b] += log((double)pv);                                        // :364-365
        int fin = 0;
        if (nn >= cap) { toks[nn++] = tk.eot; fin = 1; }                            // :367-370
        else if (next == tk.eot) fin = 1;                                           // :317
        else if (max_new > 0 && nn - prompt_len >= max_new) { toks[nn++] = tk.eot; fin = 1; }  // bench knob
        s.n_tokens[b] = nn;
        if (fin) s.done[b] = 1;
    }
}

void launch_logit_step(const float *logits, int V, DecodeState s, RuleTokens tk, int B, int ctx, int cap,
                       int max_new, int prompt_len, int mode, hipStream_t st) {
    int ldl = (V + 63) & ~63;
    hipLaunchKernelGGL(logit_step_kernel, dim3(B), dim3(1024), 0, st, logits, V, ldl, s, tk, ctx, cap, max_new,
                       prompt_len, mode);
}

__global__ __launch_bounds__(1024) void rules_only_kernel(const float *__restrict__ probs_in, float *masked_out,
                                                          int32_t *argmax_out, const int32_t *tokens, int n,
                                                          int last_ts, const uint8_t *sup, RuleTokens tk, int V) {
    __shared__ BlockRed sm;
    auto probs = [&](int i) { return probs_in[i]; };
    int rule, next;
    rules_argmax(probs, V, tokens, n, last_ts >= 0, last_ts, sup, tk, sm, rule, next);
    for (int i = threadIdx.x; i < V; i += blockDim.x) masked_out[i] = masked_value(probs_in[i], i, rule, sup, tk, last_ts);
    if (threadIdx.x == 0) *argmax_out = next;
}

void launch_rules_only(const float *probs_in, float *masked_out, int32_t *argmax_out, const int32_t *tokens,
                       int n_tokens, int last_ts, const uint8_t *suppress, RuleTokens tk, int V, hipStream_t st) {
    hipLaunchKernelGGL(rules_only_kernel, dim3(1), dim3(1024), 0, st, probs_in, masked_out, argmax_out, tokens,
                       n_tokens, last_ts, suppress, tk, V);
}
