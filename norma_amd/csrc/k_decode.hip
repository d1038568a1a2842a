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
#include <stdlib.h>

#include <atomic>

#include "nh_kernels.h"
#include <cstdlib>

__device__ __forceinline__ float gelu_tanh_d(float v) { return gelu_tanh_fast(v); }

// ---------------------------------------------------------------------------------------------------
// skinny GEMM: y[R][N] = x[R][K] . W[N][K]^T, R <= 96.  Weights are the MFMA A operand (16 rows per
// tile; each lane streams 16 B of one weight row per k-step, 4 lanes cover a 64 B run), activations
// (L2-resident) the B operand.  Two shapes of the same kernel:
//   KSPLIT = 2 .. 16 : one 16-row tile per workgroup, its KSPLIT waves split K, fp32 partials meet in
//                    LDS (small N: many waves in flight instead of few long ones; the long-K fc2 layer
//                    takes 16 waves so that every wave still has ONE group of <= 10 k-steps in flight);
//   KSPLIT = 1     : every wave owns NT 16-row tiles over the full K, no LDS (the 51866-row logits).
// Up to 8 k-steps of loads are in flight per wave before the first MFMA of a group.
// ---------------------------------------------------------------------------------------------------
#define SK_U 10

// has_pre: bias (and, for SK_RESID_F32, the residual) were fetched at kernel start into pre0, pre1
__device__ __forceinline__ void skinny_store(const SkinnyParams &p, f32x4 v, int r, int n, bool has_pre = false,
                                             f32x4 pre0 = (f32x4){0.f, 0.f, 0.f, 0.f}, f32x4 pre1 = (f32x4){0.f, 0.f, 0.f, 0.f}) {
    if (n >= p.N) return;
    if (has_pre) v += pre0;
    else if (p.bias) {
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
        f32x4 x = has_pre ? pre1 : *reinterpret_cast<const f32x4 *>(dst);
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
        else {
            const int t0 = p.pos_ptr ? p.pos_ptr[b] : p.t0;  // per-sequence position (decode pool / hipGraph replay)
            // self-attention K/V cache, head-major [b][h][ctx][64]: the keys of one (clip, head) are contiguous for dec_attn_kernel
            dst = reinterpret_cast<half_t *>(sg == 1 ? p.out[1] : p.out[2]) +
                  (((long)b * (p.d >> 6) + (nl >> 6)) * p.ctx + t0 + i) * NH_DH + (nl & 63);
        }
        *reinterpret_cast<half4 *>(dst) = hv;
        return;
    }
    *reinterpret_cast<half4 *>(reinterpret_cast<half_t *>(p.out[0]) + (long)r * p.ldo + n) = hv;
}

// U k-steps of one wave: all loads issued back to back (never guarded: a guarded load makes hipcc drain
// vmcnt(0) per element), then the MFMAs.  The callers cover `steps` with groups of 10, 5, 2 and 1.
template <int U, int NT, int NCB>
__device__ __forceinline__ void skinny_group(const half_t *const (&wp)[NT], int wstep, const half_t *const (&xp)[NCB], int s0,
                                             f32x4 (&acc)[NT][NCB]) {
    half8 a[NT][U], b[NCB][U];
#pragma unroll
    for (int u = 0; u < U; u++) {
#pragma unroll
        for (int t = 0; t < NT; t++) a[t][u] = *reinterpret_cast<const half8 *>(wp[t] + (long)wstep * (s0 + u));
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) b[cb][u] = *reinterpret_cast<const half8 *>(xp[cb] + 32 * (s0 + u));
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int cb = 0; cb < NCB; cb++)
                acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][u], b[cb][u], acc[t][cb], 0, 0, 0);
}

template <int NCB, int KSPLIT, int NT>
__global__ __launch_bounds__(KSPLIT == 1 ? 128 : 64 * KSPLIT) void skinny_gemm_kernel(SkinnyParams p) {
    constexpr int NW = KSPLIT == 1 ? 2 : KSPLIT;  // waves per workgroup
    __shared__ f32x4 red[KSPLIT == 1 ? 1 : KSPLIT][NCB][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // tile base row of this wave
    const int n0 = KSPLIT == 1 ? (blockIdx.x * NW + w) * 16 * NT : blockIdx.x * 16;
    // gridDim.y > 1: workgroup y handles activation rows rb .. rb + 16 NCB only (every CU must fetch the activation rows it
    // multiplies, and that fetch -- 35-47 KB/us per CU -- is what these kernels wait for: with few weight tiles it pays to
    // spread the ROWS over more CUs too; per-row arithmetic is unchanged, so results are)
    const int rb = blockIdx.y * 16 * NCB;
    // epilogue operands (bias, residual) of the element this thread will own: fetched now, so their latency
    // hides under the weight stream instead of extending the dependent chain of this latency-bound kernel
    f32x4 pre[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const bool can_pre = KSPLIT != 1 && (p.N & 3) == 0;
    if (can_pre) {
        const int er = rb + (tid >> 2), en = n0 + 4 * (tid & 3);
        if (tid < 64 * NCB && er < p.R && en < p.N) {
            if (p.bias) pre[0] = *reinterpret_cast<const f32x4 *>(p.bias + en);
            if (p.epi == SK_RESID_F32) pre[1] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p.out[0]) + (long)er * p.ldo + en);
        }
    }
    const int kslice = p.K / KSPLIT, kbeg = KSPLIT == 1 ? 0 : w * kslice;
    // weights: row-major [N][K] (a wave instruction = 16 rows x 64 B) or the tile-major repack (1 KiB contiguous)
    const half_t *wp[NT];
    const int wstep = p.Wt ? 512 : 32;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        int wrow = n0 + 16 * t + fr; if (wrow >= p.N) wrow = p.N - 1;
        wp[t] = p.Wt ? p.Wt + ((long)((n0 >> 4) + t) * (p.K >> 5) + (kbeg >> 5)) * 512 + lane * 8
                     : p.W + (long)wrow * p.K + kbeg + 8 * fq;
    }
    const half_t *xp[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        int r = rb + 16 * cb + fr; if (r >= p.R) r = p.R - 1;
        xp[cb] = p.x + (long)r * p.ldx + kbeg + 8 * fq;
    }
    f32x4 acc[NT][NCB];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[t][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int steps = kslice >> 5;
    {
        int s0 = 0;
        for (; s0 + 10 <= steps; s0 += 10) skinny_group<10, NT, NCB>(wp, wstep, xp, s0, acc);
        if (s0 + 5 <= steps) { skinny_group<5, NT, NCB>(wp, wstep, xp, s0, acc); s0 += 5; }
        for (; s0 + 2 <= steps; s0 += 2) skinny_group<2, NT, NCB>(wp, wstep, xp, s0, acc);
        if (s0 < steps) skinny_group<1, NT, NCB>(wp, wstep, xp, s0, acc);
    }
    if (KSPLIT == 1) {
        // D[n = 4 fq + i][r = 16 cb + fr]: each lane already holds 4 consecutive features of one row
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                int r = rb + 16 * cb + fr;
                if (r < p.R) skinny_store(p, acc[t][cb], r, n0 + 16 * t + 4 * fq);
            }
        return;
    }
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) red[w][cb][lane] = acc[0][cb];
    __syncthreads();
    // thread t owns row r = t / 4 and the 4 consecutive features n0 + 4 (t % 4) + i:
    // D[n = 4 fq + i][r = fr] lives in lane 16 fq + fr of column block r / 16
    const int rl = tid >> 2, nq = tid & 3, r = rb + rl;
    const bool owner = (tid < 64 * NCB) && (r < p.R);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (owner) {
        const int src_lane = 16 * nq + (rl & 15), cb = rl >> 4;
        v = red[0][cb][src_lane];
#pragma unroll
        for (int ww = 1; ww < KSPLIT; ww++) v += red[ww][cb][src_lane];
    }
    if (owner) skinny_store(p, v, r, n0 + 4 * nq, can_pre, pre[0], pre[1]);
}

// ---------------------------------------------------------------------------------------------------
// LayerNorm fused into the skinny GEMM (K = 128 STEPS, R <= 32): the 5.4 us LayerNorm launch in front of every
// q|k|v, cross-q and fc1 projection of a decode step is pure latency (160 KB in, 80 KB out), so each workgroup
// normalises the rows itself while its weight tile is in flight.  Wave w owns K-slice w: its lanes already load
// exactly the x elements of their B fragments (row 16 cb + fr, columns kbeg + 32 s + 8 fq .. + 8), as f32 from the
// residual stream; the row statistics meet through LDS (two passes over the register-resident values, the "sliced"
// summation tree of nh_kernels.h), gamma/beta are staged in LDS once per workgroup.
// ---------------------------------------------------------------------------------------------------
template <int NCB, int STEPS, int NT>
__global__ __launch_bounds__(256) void skinny_ln_kernel(SkinnyParams p) {
    constexpr int K = 128 * STEPS;
    __shared__ f32x4 red[4][NT][NCB][64];
    __shared__ float part[2][4][NCB][16];
    __shared__ __attribute__((aligned(16))) float gb[2][K];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    // A workgroup owns NT consecutive 16-row weight tiles and the activation rows rb .. rb + 16 NCB (gridDim.y row blocks).
    // What it waits for is its own fetch (35-47 KB/us per CU): NT x 41 KB of weights + 16 NCB rows x K f32 of activations,
    // so the launcher shapes (NT, NCB, grid) to keep that sum small while every CU has work.
    const int tiles = (p.N + 15) >> 4, tile0 = blockIdx.x * NT;
    const int rb = blockIdx.y * 16 * NCB;
    const int kbeg = w * 32 * STEPS;
    // weights first: the only HBM stream of the kernel
    const int wstep = p.Wt ? 512 : 32;
    half8 a[NT][STEPS];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int tile = tile0 + t < tiles ? tile0 + t : tiles - 1;   // a tail workgroup re-reads the last tile; its stores are skipped
        int wrow = tile * 16 + fr; if (wrow >= p.N) wrow = p.N - 1;
        const half_t *wp = p.Wt ? p.Wt + ((long)tile * (K >> 5) + (kbeg >> 5)) * 512 + lane * 8 : p.W + (long)wrow * K + kbeg + 8 * fq;
#pragma unroll
        for (int s = 0; s < STEPS; s++) a[t][s] = *reinterpret_cast<const half8 *>(wp + wstep * s);
    }
    // epilogue operands of the elements this thread will own (see skinny_gemm_kernel)
    f32x4 pre[NT][2];
    const bool can_pre = (p.N & 3) == 0;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        pre[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; pre[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (can_pre) {
            const int er = rb + (tid >> 2), en = (tile0 + t) * 16 + 4 * (tid & 3);
            if (tid < 64 * NCB && er < p.R && en < p.N) {
                if (p.bias) pre[t][0] = *reinterpret_cast<const f32x4 *>(p.bias + en);
                if (p.epi == SK_RESID_F32) pre[t][1] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p.out[0]) + (long)er * p.ldo + en);
            }
        }
    }
    // the rows, f32
    f32x4 xv[NCB][STEPS][2];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        int r = rb + 16 * cb + fr; if (r >= p.R) r = p.R - 1;
        const float *xr = p.ln_x + (long)r * K + kbeg + 8 * fq;
#pragma unroll
        for (int s = 0; s < STEPS; s++) {
            xv[cb][s][0] = *reinterpret_cast<const f32x4 *>(xr + 32 * s);
            xv[cb][s][1] = *reinterpret_cast<const f32x4 *>(xr + 32 * s + 4);
        }
    }
    for (int c = tid; c < K / 4; c += 256) {
        reinterpret_cast<f32x4 *>(gb[0])[c] = reinterpret_cast<const f32x4 *>(p.ln_w)[c];
        reinterpret_cast<f32x4 *>(gb[1])[c] = reinterpret_cast<const f32x4 *>(p.ln_b)[c];
    }
    float mean[NCB], inv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        float s1 = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; s++) s1 = ln_sum8(s1, xv[cb][s][0], xv[cb][s][1]);
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        if (fq == 0) part[0][w][cb][fr] = s1;
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        mean[cb] = ln_mean((part[0][0][cb][fr] + part[0][1][cb][fr]) + (part[0][2][cb][fr] + part[0][3][cb][fr]), p.ln_rk);
        float s2 = 0.f;
#pragma unroll
        for (int s = 0; s < STEPS; s++) s2 = ln_sq8(s2, xv[cb][s][0], xv[cb][s][1], mean[cb]);
        s2 += __shfl_xor(s2, 16); s2 += __shfl_xor(s2, 32);
        if (fq == 0) part[1][w][cb][fr] = s2;
    }
    __syncthreads();
    f32x4 acc[NT][NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
        inv[cb] = ln_inv((part[1][0][cb][fr] + part[1][1][cb][fr]) + (part[1][2][cb][fr] + part[1][3][cb][fr]), p.ln_rk);
#pragma unroll
        for (int t = 0; t < NT; t++) acc[t][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
        const int k = kbeg + 32 * s + 8 * fq;
        const f32x4 g0 = *reinterpret_cast<const f32x4 *>(&gb[0][k]), g1 = *reinterpret_cast<const f32x4 *>(&gb[0][k + 4]);
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(&gb[1][k]), b1 = *reinterpret_cast<const f32x4 *>(&gb[1][k + 4]);
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
            const f32x4 o0 = ln_apply(xv[cb][s][0], mean[cb], inv[cb], g0, b0);
            const f32x4 o1 = ln_apply(xv[cb][s][1], mean[cb], inv[cb], g1, b1);
            const half8 b = {(half_t)o0[0], (half_t)o0[1], (half_t)o0[2], (half_t)o0[3],
                             (half_t)o1[0], (half_t)o1[1], (half_t)o1[2], (half_t)o1[3]};
#pragma unroll
            for (int t = 0; t < NT; t++) acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t][s], b, acc[t][cb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) red[w][t][cb][lane] = acc[t][cb];
    __syncthreads();
    const int rl = tid >> 2, nq = tid & 3, r = rb + rl;
    if (tid < 64 * NCB && r < p.R) {
        const int src_lane = 16 * nq + (rl & 15), cb = rl >> 4;
#pragma unroll
        for (int t = 0; t < NT; t++) {
            if (tile0 + t >= tiles) break;
            f32x4 v = red[0][t][cb][src_lane];  // same association as skinny_gemm_kernel: the two forms give identical bits
#pragma unroll
            for (int ww = 1; ww < 4; ww++) v += red[ww][t][cb][src_lane];
            skinny_store(p, v, r, (tile0 + t) * 16 + 4 * nq, can_pre, pre[t][0], pre[t][1]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The same LayerNorm with 16 consecutive lanes per row (lane t = 4 w + fq of the sliced tree): stand-alone kernel and
// the staging pass of the logits kernel.  K = 128 steps, steps <= 10; loads are unconditional and clamped.
// ---------------------------------------------------------------------------------------------------
#define LN_MAX_STEPS 10
struct SlicedRow { f32x4 v[LN_MAX_STEPS][2]; float mean, inv; };

__device__ __forceinline__ void sliced_row_stats(SlicedRow &sr, const float *__restrict__ xrow, int K, float rk, int t16) {
    const int steps = K >> 7, w = t16 >> 2, fq = t16 & 3;
    const float *xr = xrow + w * 32 * steps + 8 * fq;
    float s1 = 0.f;
#pragma unroll
    for (int s = 0; s < LN_MAX_STEPS; s++) {
        const int sc = s < steps ? s : steps - 1;
        sr.v[s][0] = *reinterpret_cast<const f32x4 *>(xr + 32 * sc);
        sr.v[s][1] = *reinterpret_cast<const f32x4 *>(xr + 32 * sc + 4);
    }
#pragma unroll
    for (int s = 0; s < LN_MAX_STEPS; s++)
        if (s < steps) s1 = ln_sum8(s1, sr.v[s][0], sr.v[s][1]);
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);   // the four lanes of a slice: (l0 + l1) + (l2 + l3)
    s1 += __shfl_xor(s1, 4); s1 += __shfl_xor(s1, 8);   // the four slices: (p0 + p1) + (p2 + p3)
    sr.mean = ln_mean(s1, rk);
    float s2 = 0.f;
#pragma unroll
    for (int s = 0; s < LN_MAX_STEPS; s++)
        if (s < steps) s2 = ln_sq8(s2, sr.v[s][0], sr.v[s][1], sr.mean);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    s2 += __shfl_xor(s2, 4); s2 += __shfl_xor(s2, 8);
    sr.inv = ln_inv(s2, rk);
}

__global__ __launch_bounds__(256) void layernorm_sliced_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                               const float *__restrict__ b, half_t *__restrict__ y,
                                                               float *__restrict__ y32, int M, int K, float rk) {
    const int t16 = threadIdx.x & 15;
    int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row < M;
    if (!live) row = M - 1;  // keep the 16-lane groups whole for the shuffles
    SlicedRow sr;
    sliced_row_stats(sr, x + (long)row * K, K, rk, t16);
    if (!live) return;
    const int steps = K >> 7, k0 = (t16 >> 2) * 32 * steps + 8 * (t16 & 3);
#pragma unroll
    for (int s = 0; s < LN_MAX_STEPS; s++) {
        if (s < steps) {
            const int k = k0 + 32 * s;
            const f32x4 o0 = ln_apply(sr.v[s][0], sr.mean, sr.inv, *reinterpret_cast<const f32x4 *>(w + k), *reinterpret_cast<const f32x4 *>(b + k));
            const f32x4 o1 = ln_apply(sr.v[s][1], sr.mean, sr.inv, *reinterpret_cast<const f32x4 *>(w + k + 4), *reinterpret_cast<const f32x4 *>(b + k + 4));
            const half8 h = {(half_t)o0[0], (half_t)o0[1], (half_t)o0[2], (half_t)o0[3], (half_t)o1[0], (half_t)o1[1], (half_t)o1[2], (half_t)o1[3]};
            *reinterpret_cast<half8 *>(y + (long)row * K + k) = h;
            if (y32) {
                *reinterpret_cast<f32x4 *>(y32 + (long)row * K + k) = o0;
                *reinterpret_cast<f32x4 *>(y32 + (long)row * K + k + 4) = o1;
            }
        }
    }
}

bool launch_layernorm_sliced(const float *x, const float *w, const float *b, half_t *y, float *y32, int M, int K, hipStream_t st) {
    if (K % 128 != 0 || K > 128 * LN_MAX_STEPS || M < 1) return false;
    hipLaunchKernelGGL(layernorm_sliced_kernel, dim3((M + 15) / 16), dim3(256), 0, st, x, w, b, y, y32, M, K, 1.0f / (float)K);
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Large-N variant (the 51866-row tied-embedding logits): the activations are staged ONCE per workgroup
// into LDS as [k-step][row][4 chunks of 16 B] with chunk' = chunk ^ (-(row >> 2) & 3) (conflict-free
// ds_read_b128 B fragments), so the only global traffic of the main loop is the weight stream:
// every wave walks 16-row weight tiles (grid-stride), up to 10 row-segment loads in flight.
// ---------------------------------------------------------------------------------------------------
template <int NCB>
__global__ __launch_bounds__(512) void skinny_lds_kernel(SkinnyParams p) {
    extern __shared__ __attribute__((aligned(16))) char xs[];  // (K / 32) * (16 NCB) * 64 bytes
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int rows = 16 * NCB, steps = p.K >> 5;
    // Work split: every wave of the grid owns a contiguous range of weight rows, 4-row units dealt out evenly (24 or 28 rows
    // at V = 51866 over 2048 waves; whole 16-row tiles per wave left 42 % of the waves with half the work of the others).
    // A tile that sticks out of the range clamps its rows to the last one (cache hits, not HBM) and masks the stores.
    // With the tile-major repack (p.Wt: 1 KiB contiguous per wave instruction, 5.1 vs 3.7 TB/s for this stream) the unit is
    // a whole 16-row tile, dealt out round-robin.
    const int gw = blockIdx.x * 8 + w, nwav = gridDim.x * 8;
    const bool tiled = p.Wt != nullptr;
    const int units = (p.N + 3) >> 2, upw = units / nwav, uex = units % nwav;
    const int ntiles = (p.N + 15) >> 4;
    const int lo = tiled ? 16 * gw : 4 * (gw * upw + (gw < uex ? gw : uex));
    int hi = tiled ? p.N : lo + 4 * (upw + (gw < uex ? 1 : 0)); if (hi > p.N) hi = p.N;
    const int tstride = tiled ? nwav : 1;  // tile t of this wave starts at row lo + 16 t tstride
    const int my_tiles = tiled ? (gw < ntiles ? (ntiles - gw + nwav - 1) / nwav : 0) : ((hi - lo + 15) >> 4);
    const int ngrp = (steps + SK_U - 1) / SK_U;
    const int G = my_tiles * ngrp;  // (tile, k-group) pairs of this wave, k-groups innermost
    // one k-group of weight-row segments: unconditional clamped loads, all in flight together
    auto issue = [&](int g, half8 (&a)[SK_U]) {
        const int tile = g / ngrp, s0 = (g - tile * ngrp) * SK_U;
        const int row0 = lo + 16 * tile * tstride;
        int wrow = row0 + fr; if (wrow > hi - 1) wrow = hi - 1;
        const half_t *wp = tiled ? p.Wt + (long)(row0 >> 4) * steps * 512 + lane * 8 : p.W + (long)wrow * p.K + 8 * fq;
        const int wstep = tiled ? 512 : 32;
#pragma unroll
        for (int u = 0; u < SK_U; u++) {
            const int sc = s0 + u < steps ? s0 + u : steps - 1;
            a[u] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(wp + (long)wstep * sc));  // 133 MB read once per token
        }
    };
    // the first group is requested BEFORE the activations are staged (and normalised): the weights do not depend on them
    half8 a0[SK_U], a1[SK_U];
    if (G > 0) issue(0, a0);
    if (p.ln_x) {
        // fused final LayerNorm (sliced tree, 16 lanes per row, 32 rows per pass), written as the swizzled fp16 image
        for (int r = tid >> 4; r < rows; r += 32) {
            const int t16 = tid & 15, rr = r < p.R ? r : p.R - 1;
            SlicedRow sr;
            sliced_row_stats(sr, p.ln_x + (long)rr * p.K, p.K, p.ln_rk, t16);
            const int nst = p.K >> 7, k0 = (t16 >> 2) * 32 * nst + 8 * (t16 & 3);
#pragma unroll
            for (int s = 0; s < LN_MAX_STEPS; s++) {
                if (s < nst) {
                    const int k = k0 + 32 * s;
                    const f32x4 o0 = ln_apply(sr.v[s][0], sr.mean, sr.inv, *reinterpret_cast<const f32x4 *>(p.ln_w + k), *reinterpret_cast<const f32x4 *>(p.ln_b + k));
                    const f32x4 o1 = ln_apply(sr.v[s][1], sr.mean, sr.inv, *reinterpret_cast<const f32x4 *>(p.ln_w + k + 4), *reinterpret_cast<const f32x4 *>(p.ln_b + k + 4));
                    const half8 hv = {(half_t)o0[0], (half_t)o0[1], (half_t)o0[2], (half_t)o0[3], (half_t)o1[0], (half_t)o1[1], (half_t)o1[2], (half_t)o1[3]};
                    const int st = k >> 5, q = (k >> 3) & 3;
                    *reinterpret_cast<half8 *>(xs + ((long)st * rows + r) * 64 + ((q ^ ((-(r >> 2)) & 3)) << 4)) = hv;
                }
            }
        }
    } else {
        for (int c = tid; c < steps * rows * 4; c += 512) {
            const int q = c & 3, r = (c >> 2) % rows, st = (c >> 2) / rows;
            const int rr = r < p.R ? r : p.R - 1;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(p.x + (long)rr * p.ldx + 32 * st + 8 * q);
            *reinterpret_cast<u32x4 *>(xs + ((long)st * rows + r) * 64 + ((q ^ ((-(r >> 2)) & 3)) << 4)) = v;
        }
    }
    __syncthreads();
    int boff[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) boff[cb] = (16 * cb + fr) * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
    f32x4 acc[NCB];
    auto compute = [&](int g, const half8 (&a)[SK_U]) {
        const int tile = g / ngrp, gi = g - tile * ngrp, s0 = gi * SK_U;
        if (gi == 0) {
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < SK_U; u++) {
            if (s0 + u < steps) {
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    half8 b = *reinterpret_cast<const half8 *>(xs + (long)(s0 + u) * rows * 64 + boff[cb]);
                    acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b, acc[cb], 0, 0, 0);
                }
            }
        }
        if (gi == ngrp - 1) {
            const int n = lo + 16 * tile * tstride + 4 * fq;  // lo and hi are multiples of 4 (hi may be N itself)
            if (n < hi) {
#pragma unroll
                for (int cb = 0; cb < NCB; cb++) {
                    int r = 16 * cb + fr;
                    if (r < p.R) skinny_store(p, acc[cb], r, n);
                }
            }
        }
    };
    // two register sets: the next group is in flight while the current one is multiplied
    for (int g = 0; g < G; g += 2) {
        if (g + 1 < G) issue(g + 1, a1);
        compute(g, a0);
        if (g + 2 < G) issue(g + 2, a0);
        if (g + 1 < G) compute(g + 1, a1);
    }
}

// ---------------------------------------------------------------------------------------------------
// The same logits kernel for 33 .. 96 rows (r03: several encoder batches decoded together, nh_encode_rows): the fp16 image of
// all rows no longer fits the LDS (96 rows x 1280 x 2 B = 240 KB), so K is cut into PHASES of `sp` k-steps; the image of one
// phase is staged, every wave multiplies its (<= 2) weight tiles over that k-range into accumulators it keeps across the
// phases, barrier, next phase.  The weights are still streamed exactly once per token, and every output element is still
// accumulated over k in ascending order in one f32 accumulator: bit-identical to skinny_lds_kernel's result for the same row.
// Activations come as fp16 (LayerNorm as its own launch: the fused form would have to keep every row's statistics).
// Requires the tile-major weights and at most LP_MT tiles per wave.
// ---------------------------------------------------------------------------------------------------
#define LP_MT 2
template <int NCB>
__global__ __launch_bounds__(512) void skinny_ldsp_kernel(SkinnyParams p, int sp) {
    extern __shared__ __attribute__((aligned(16))) char xs[];  // sp * (16 NCB) * 64 bytes
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    constexpr int rows = 16 * NCB;
    const int steps = p.K >> 5;
    const int gw = blockIdx.x * 8 + w, nwav = gridDim.x * 8;
    const int ntiles = (p.N + 15) >> 4;
    const int my_tiles = gw < ntiles ? min(LP_MT, (ntiles - gw + nwav - 1) / nwav) : 0;
    int boff[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) boff[cb] = (16 * cb + fr) * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
    f32x4 acc[LP_MT][NCB];
#pragma unroll
    for (int t = 0; t < LP_MT; t++)
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) acc[t][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ph0 = 0; ph0 < steps; ph0 += sp) {
        const int nst = min(sp, steps - ph0);
        const int ngrp = (nst + SK_U - 1) / SK_U;
        const int G = my_tiles * ngrp;                 // (tile, k-group) pairs of this wave in this phase
        auto issue = [&](int g, half8 (&a)[SK_U]) {    // unconditional clamped loads, all in flight together
            const int tile = g / ngrp, s0 = ph0 + (g - tile * ngrp) * SK_U;
            const half_t *wp = p.Wt + (long)(gw + tile * nwav) * steps * 512 + lane * 8;
#pragma unroll
            for (int u = 0; u < SK_U; u++) {
                const int sc = s0 + u < steps ? s0 + u : steps - 1;
                a[u] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(wp + (long)512 * sc));
            }
        };
        half8 a0[SK_U], a1[SK_U];
        if (G > 0) issue(0, a0);                       // in flight while the image is staged
        __syncthreads();                               // every wave is done with the previous phase's image
        for (int c = tid; c < nst * rows * 4; c += 512) {
            const int q = c & 3, r = (c >> 2) % rows, st = (c >> 2) / rows;
            const int rr = r < p.R ? r : p.R - 1;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(p.x + (long)rr * p.ldx + 32 * (ph0 + st) + 8 * q);
            *reinterpret_cast<u32x4 *>(xs + ((long)st * rows + r) * 64 + ((q ^ ((-(r >> 2)) & 3)) << 4)) = v;
        }
        __syncthreads();
        auto compute = [&](int g, const half8 (&a)[SK_U]) {
            const int tile = g / ngrp, s0 = (g - tile * ngrp) * SK_U;   // phase-local k-step of the group's first step
#pragma unroll
            for (int u = 0; u < SK_U; u++) {
                if (s0 + u < nst) {
#pragma unroll
                    for (int cb = 0; cb < NCB; cb++) {
                        const half8 b = *reinterpret_cast<const half8 *>(xs + (long)(s0 + u) * rows * 64 + boff[cb]);
                        if (tile == 0) acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b, acc[0][cb], 0, 0, 0);
                        else acc[1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b, acc[1][cb], 0, 0, 0);
                    }
                }
            }
        };
        for (int g = 0; g < G; g += 2) {
            if (g + 1 < G) issue(g + 1, a1);
            compute(g, a0);
            if (g + 2 < G) issue(g + 2, a0);
            if (g + 1 < G) compute(g + 1, a1);
        }
    }
#pragma unroll
    for (int t = 0; t < LP_MT; t++) {
        if (t < my_tiles) {
            const int n = 16 * (gw + t * nwav) + 4 * fq;
#pragma unroll
            for (int cb = 0; cb < NCB; cb++) {
                const int r = 16 * cb + fr;
                if (r < p.R) skinny_store(p, acc[t][cb], r, n);
            }
        }
    }
}

// tile-major repack of a row-major [N][K] fp16 weight: out[(tile * K/32 + s) * 512 + lane * 8 + j] =
// W[16 tile + (lane & 15)][32 s + 8 (lane >> 4) + j], rows >= N zero: the MFMA A fragment of (tile, k-step s) is 1 KiB contiguous
__global__ __launch_bounds__(256) void repack_tiles_kernel(const half_t *__restrict__ W, half_t *__restrict__ out, int N, int K) {
    const long chunk = blockIdx.x * 256L + threadIdx.x;  // one 16-byte chunk per thread
    const int steps = K >> 5;
    const long total = (long)((N + 15) >> 4) * steps * 64;
    if (chunk >= total) return;
    const int lane = (int)(chunk & 63);
    const long ts = chunk >> 6;
    const int s = (int)(ts % steps);
    const long tile = ts / steps;
    const long row = tile * 16 + (lane & 15);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < N) v = *reinterpret_cast<const u32x4 *>(W + row * K + 32 * s + 8 * (lane >> 4));
    *reinterpret_cast<u32x4 *>(out + chunk * 8) = v;
}

void launch_repack_tiles(const half_t *W, half_t *out, int N, int K, hipStream_t st) {
    const long total = (long)((N + 15) >> 4) * (K >> 5) * 64;
    hipLaunchKernelGGL(repack_tiles_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, out, N, K);
}

// The two logits kernels above keep their activations in LDS and feed every MFMA from a `ds_read_b128` -- 512 threads, 288-358
// of a SIMD's 512 registers, 40-120 KB of LDS: other kernels' workgroups fit beside them on a CU.  r03 found that they must not:
// the log-mel kernel of ANOTHER context (37 KB of LDS, 4 waves) gave wrong spectra in 1-250 frames of a clip whenever one of these
// workgroups ran beside it (tools/dbg/stress_mel.py: 150-190 wrong clip-mels in 3200 next to 64-row decodes, 6 in 9600 next to
// 32-row ones, none next to 16-row ones or with nothing running; `tools/ldsprobe` shows LDS allocations and barriers of
// co-resident workgroups do stay apart).  Stripping the kernel showed what it takes: the LDS read feeding the MFMA -- without the
// MFMAs (LDS reads into VALU adds), or with the MFMA's B operand taken from registers instead, the neighbour's results are right;
// stores, weight loads and the staging writes do not matter.  No software contract covers that, so these kernels do not share
// their CU's LDS: they ask for all of it, which keeps every LDS-using workgroup off the CU while they run (30 us per token).
#define NH_LDS_EXCLUSIVE (160 * 1024)
static size_t lds_exclusive_bytes(size_t needed) {
    static const bool share = getenv("NH_DBG_SHARE_LDS") != nullptr;   // tools/dbg/stress_mel.py's A/B switch: only what the kernel uses
    return share ? needed : (size_t)NH_LDS_EXCLUSIVE;
}

static bool ln_steps_ok(int K) {
    const int s = K / 128;
    return K % 128 == 0 && (s == 1 || s == 2 || s == 3 || s == 4 || s == 6 || s == 8 || s == 10);
}
static bool logits_lds_ok(int R, int N, int K) {
    return (N + 15) / 16 >= 2048 && R <= 32 && (size_t)(K >> 5) * 16 * ((R + 15) / 16) * 64 <= 96 * 1024;
}
bool skinny_ln_supported(int R, int N, int K) {
    if ((N + 15) / 16 >= 2048) return R <= 32 && logits_lds_ok(R, N, K) && K <= 128 * LN_MAX_STEPS && K % 128 == 0;
    return R <= 96 && ln_steps_ok(K);   // one 16-row block per workgroup: any number of row blocks
}

template <int NCB, int NT>
static void launch_skinny_ln_grid(const SkinnyParams &p, dim3 grid, hipStream_t st) {
    const dim3 block(256);
    switch (p.K / 128) {
        case 1: hipLaunchKernelGGL((skinny_ln_kernel<NCB, 1, NT>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((skinny_ln_kernel<NCB, 2, NT>), grid, block, 0, st, p); break;
        case 3: hipLaunchKernelGGL((skinny_ln_kernel<NCB, 3, NT>), grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((skinny_ln_kernel<NCB, 4, NT>), grid, block, 0, st, p); break;
        case 6: hipLaunchKernelGGL((skinny_ln_kernel<NCB, 6, NT>), grid, block, 0, st, p); break;
        case 8: hipLaunchKernelGGL((skinny_ln_kernel<NCB, 8, NT>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((skinny_ln_kernel<NCB, 10, NT>), grid, block, 0, st, p); break;
    }
}
template <int NCB>
static void launch_skinny_ln(const SkinnyParams &p, hipStream_t st) {
    const int tiles = (p.N + 15) / 16;
    // per-CU fetch = NT x (16 rows of W) + (16 NCB rows of x, f32): few tiles -> one tile x one 16-row block per workgroup;
    // many tiles -> two tiles x one 16-row block (as many workgroups as tiles, a fifth fewer bytes each than 1 tile x 32 rows)
    if (NCB > 1 && tiles <= 160) launch_skinny_ln_grid<1, 1>(p, dim3(tiles, NCB), st);
    else if (NCB > 1) launch_skinny_ln_grid<1, 2>(p, dim3((tiles + 1) / 2, NCB), st);
    else launch_skinny_ln_grid<1, 1>(p, dim3(tiles), st);
}

template <int NCB>
static void launch_skinny_ncb(const SkinnyParams &p, hipStream_t st) {
    const int tiles = (p.N + 15) / 16;
    if (p.ln_x && tiles < 2048) {  // the caller checked skinny_ln_supported
        launch_skinny_ln<NCB>(p, st);
        return;
    }
    if (tiles >= 2048) {  // the tied-embedding logits: plenty of tiles, stream full rows
        const size_t lds = (size_t)(p.K >> 5) * 16 * NCB * 64;
        if (NCB <= 2 && lds <= 96 * 1024) {
            // hipFuncSetAttribute acts on the CURRENT device: remember it per device (contexts on several GPUs of one
            // process, each driven by its own host thread)
            static std::atomic<bool> attr_set[NH_MAX_DEVICES];
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev < 0 || dev >= NH_MAX_DEVICES || !attr_set[dev].load(std::memory_order_acquire)) {
                hipFuncSetAttribute(reinterpret_cast<const void *>(&skinny_lds_kernel<NCB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, NH_LDS_EXCLUSIVE);
                if (dev >= 0 && dev < NH_MAX_DEVICES) attr_set[dev].store(true, std::memory_order_release);
            }
            // `lds` is what the kernel uses; it is given the whole LDS of the CU (NH_LDS_EXCLUSIVE, see there)
            hipLaunchKernelGGL((skinny_lds_kernel<NCB>), dim3(256), dim3(512), lds_exclusive_bytes(lds), st, p);
        } else if (NCB >= 3 && p.Wt && !p.ln_x && tiles <= LP_MT * 2048) {
            // 33 .. 96 rows: K in phases through the LDS (skinny_ldsp_kernel); as few phases as 144 KiB of LDS allow, balanced
            const int steps = p.K >> 5, spmax = (144 * 1024) / (16 * NCB * 64);
            const int phases = (steps + spmax - 1) / spmax, sp = (steps + phases - 1) / phases;
            static std::atomic<bool> attr_set[NH_MAX_DEVICES];
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (dev < 0 || dev >= NH_MAX_DEVICES || !attr_set[dev].load(std::memory_order_acquire)) {
                hipFuncSetAttribute(reinterpret_cast<const void *>(&skinny_ldsp_kernel<NCB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, NH_LDS_EXCLUSIVE);
                if (dev >= 0 && dev < NH_MAX_DEVICES) attr_set[dev].store(true, std::memory_order_release);
            }
            // the kernel uses sp * 16 NCB * 64 bytes; it is given the whole LDS of the CU (NH_LDS_EXCLUSIVE, see there)
            hipLaunchKernelGGL((skinny_ldsp_kernel<NCB>), dim3(256), dim3(512), lds_exclusive_bytes((size_t)sp * 16 * NCB * 64), st, p, sp);
        } else if constexpr (NCB <= 4) {
            int waves = (tiles + 1) / 2;
            hipLaunchKernelGGL((skinny_gemm_kernel<NCB, 1, 2>), dim3((waves + 1) / 2), dim3(128), 0, st, p);
        }
        return;
    }
    // waves per workgroup: every wave keeps a whole number of 32-deep k-steps, and at most ~10 of them (one group of
    // loads in flight = one memory round trip per wave).  r01 split the long-K layer (fc2, K = 4 d) across workgroups
    // instead (slab stores + ticket + slab loads: three more dependent round trips and cross-workgroup atomics for the
    // same ~11 us); 16 waves of one workgroup meet in LDS.
    auto fits = [&](int nw) { return p.K % (nw * 32) == 0; };
    // (the slicing must not depend on the batch: one summation order for every NCB = bit-exact batch invariance)
    // Few weight tiles (N <= 2560): one workgroup per (tile, 16-row block of the activations) -- the single-block
    // instantiation on a tiles x NCB grid -- so that a CU fetches 16 rows of activations instead of all of them
    // (skinny_gemm_kernel, rb).  Same per-row arithmetic as the NCB-block form.
    constexpr int RBN = NCB;   // row blocks when split
    // (more than 64 rows -- several encoder batches decoded together -- always take the split form: no NCB = 5, 6 instantiation
    // of the unsplit kernel exists, its LDS reduction buffer would not fit)
    const bool split_rows = NCB > 4 || (NCB > 1 && tiles * NCB <= 640 && tiles <= 160);
    const dim3 grid = split_rows ? dim3(tiles, RBN) : dim3(tiles);
#define SKG(KS, THR)                                                                                               \
    do {                                                                                                            \
        if (split_rows) hipLaunchKernelGGL((skinny_gemm_kernel<1, KS, 1>), grid, dim3(THR), 0, st, p);             \
        else if constexpr (NCB <= 4) hipLaunchKernelGGL((skinny_gemm_kernel<NCB, KS, 1>), grid, dim3(THR), 0, st, p); \
    } while (0)
    if (p.K >= 2560 && fits(16)) SKG(16, 1024);
    else if (p.K >= 2560 && fits(8)) SKG(8, 512);
    else if (fits(4)) SKG(4, 256);
    else SKG(2, 128);
#undef SKG
}

void launch_skinny(const SkinnyParams &p_in, hipStream_t st) {
    SkinnyParams p = p_in;
    p.ln_rk = 1.0f / (float)p.K;
    int ncb = (p.R + 15) / 16;
    if (ncb <= 1) launch_skinny_ncb<1>(p, st);
    else if (ncb == 2) launch_skinny_ncb<2>(p, st);
    else if (ncb == 3) launch_skinny_ncb<3>(p, st);
    else if (ncb == 4) launch_skinny_ncb<4>(p, st);
    else if (ncb == 5) launch_skinny_ncb<5>(p, st);
    else launch_skinny_ncb<6>(p, st);   // R <= 96 (nh_create rejects a larger max_batch)
}

// ---------------------------------------------------------------------------------------------------
// decoder attention, one query row per (b, h): 4 waves split the keys, inside a wave 8 key slots x
// 8 lanes (16 B of the 64-wide head each); each slot runs its own online softmax, merged at the end.
// ---------------------------------------------------------------------------------------------------
#ifndef DA_U
#define DA_U 4  // key groups (8 keys each) whose K and V rows are in flight together per wave: 2 x DA_U KiB (8: cross-attention 43 -> 46 us)
#endif
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
                                                       int d, int ctx, int Tk, const int32_t *__restrict__ pos_ptr,
                                                       int kv_head_major, const int32_t *__restrict__ done) {
    // a finished sequence (eot, cap, or the no-speech exit) no longer streams its K/V: the reference's loop ends per
    // sequence at eot (model.rs:317); in a batch the others go on, and 0.49 GB of the 0.72 GB a step streams is per-sequence
    // cross K/V.  Its attention row is left as it was; every later product is row-wise, nothing of it reaches another row.
    if (done && done[blockIdx.y]) return;
    if (pos_ptr) Tk = pos_ptr[blockIdx.y] + 1;  // causal self-attention at this sequence's own position
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
    // rows of one (clip, head): [b][ctx][d] (the self-attention cache, 128 B out of every d * 2) or head-major
    // [b][h][ctx][64] (cross K/V: the keys of a head are contiguous, a wave instruction reads 1 KiB in one piece)
    const long rs = kv_head_major ? NH_DH : d;
    const long base = kv_head_major ? ((long)b * gridDim.x + h) * ctx * NH_DH : (long)b * ctx * d + h * NH_DH;
    const half_t *kb = kc + base + 8 * pp;
    const half_t *vb = vc + base + 8 * pp;
    AttnPart st; st.m = -INFINITY; st.l = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) st.acc[c] = 0.f;
    for (int j0 = kbeg; j0 < kend; j0 += 8 * DA_U) {
        half8 kk[DA_U], vv[DA_U];
#pragma unroll
        for (int u = 0; u < DA_U; u++) {
            int j = j0 + 8 * u + slot; if (j >= kend) j = kend - 1;
            // streamed once per token: non-temporal, so the K/V stream (491 MB per step at b32) does not evict the
            // decoder weights from the Infinity Cache between tokens
            kk[u] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(kb + j * rs));
            vv[u] = __builtin_nontemporal_load(reinterpret_cast<const half8 *>(vb + j * rs));
        }
#pragma unroll
        for (int u = 0; u < DA_U; u++) {
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

// ---------------------------------------------------------------------------------------------------
// NUMERICS PROTOTYPE (NH_OPT_ABSORBED_XATTN, off by default): cross-attention computed on the encoder output
// xa itself instead of on K = xa Wk^T and V = xa Wv^T + bv -- with u_h = Wk_h^T q_h the scores are xa . u_h, and
// sum_s p_h[s] V[s] = Wv_h (sum_s p_h[s] xa[s]) + bv_h -- so that a decoder layer would stream xa (3.84 MB per row, shared by all
// layers) once instead of K and V (7.68 MB per row and layer).  These two kernels exist to answer ONE question on the GPU before
// the real (MFMA, one-pass) kernel is written: do the decoder tolerance tests survive moving the fp16 roundings from K and V to u
// and z?  They place the roundings where that kernel would (u, p and z in fp16; sums in f32) and are not fast: every (row, head)
// workgroup re-reads the whole xa of its row.
__global__ __launch_bounds__(256) void xabs_u_kernel(const half_t *__restrict__ q, const half_t *__restrict__ Wk, half_t *__restrict__ U, int d) {
    __shared__ float qs[NH_DH];
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid < NH_DH) qs[tid] = (float)q[(long)b * d + h * NH_DH + tid];
    __syncthreads();
    for (int c = tid; c < d; c += 256) {
        float acc = 0.f;
#pragma unroll 8
        for (int j = 0; j < NH_DH; j++) acc += (float)Wk[(long)(h * NH_DH + j) * d + c] * qs[j];
        U[((long)b * gridDim.x + h) * d + c] = (half_t)(0.125f * acc);   // candle scales q and k by dh^-1/4 each: exactly 1/8
    }
}

__global__ __launch_bounds__(256) void xabs_attn_kernel(const half_t *__restrict__ U, const half_t *__restrict__ xa, const half_t *__restrict__ Wv,
                                                        const float *__restrict__ bv, half_t *__restrict__ out, int d, int S,
                                                        const int32_t *__restrict__ done) {
    extern __shared__ float xsm[];          // u[d], then per wave: m, l, z[d]
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (done && done[b]) return;
    float *us = xsm, *zs = xsm + d;        // zs: [4][d + 2]
    for (int c = tid; c < d; c += 256) us[c] = (float)U[((long)b * gridDim.x + h) * d + c];
    __syncthreads();
    const int nper = d / 64;               // features per lane (d % 64 == 0)
    float z[20];                           // d <= 1280
    for (int i = 0; i < nper; i++) z[i] = 0.f;
    float m = -INFINITY, l = 0.f;
    const half_t *xr = xa + (long)b * S * d;
    for (int s = w; s < S; s += 4) {
        float xv[20], dot = 0.f;
        for (int i = 0; i < nper; i++) { xv[i] = (float)xr[(long)s * d + 64 * i + lane]; dot += xv[i] * us[64 * i + lane]; }
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        const float mn = fmaxf(m, dot);
        const float al = __expf(m - mn);
        const float pr = (float)(half_t)__expf(dot - mn);     // P goes to the matrix pipe as fp16 in the real kernel
        l = l * al + pr;
        for (int i = 0; i < nper; i++) z[i] = z[i] * al + pr * xv[i];
        m = mn;
    }
    for (int i = 0; i < nper; i++) zs[w * (d + 2) + 64 * i + lane] = z[i];
    if (lane == 0) { zs[w * (d + 2) + d] = m; zs[w * (d + 2) + d + 1] = l; }
    __syncthreads();
    // merge the four waves in one fixed order, normalise, round z to fp16 (the B operand of the value projection)
    float mm = zs[d], ll = zs[d + 1];
    float f[4]; f[0] = 1.f;
    for (int ww = 1; ww < 4; ww++) {
        const float m2 = zs[ww * (d + 2) + d], l2 = zs[ww * (d + 2) + d + 1];
        const float mn = fmaxf(mm, m2);
        const float a1 = (mm == -INFINITY) ? 0.f : __expf(mm - mn), a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
        for (int k = 0; k < ww; k++) f[k] *= a1;
        f[ww] = a2; ll = ll * a1 + l2 * a2; mm = mn;
    }
    __syncthreads();
    const float inv = 1.0f / ll;
    for (int c = tid; c < d; c += 256) {
        float v = ((zs[c] * f[0] + zs[(d + 2) + c] * f[1]) + zs[2 * (d + 2) + c] * f[2]) + zs[3 * (d + 2) + c] * f[3];
        us[c] = (float)(half_t)(v * inv);
    }
    __syncthreads();
    {   // o_h[j] = Wv[64 h + j][:] . z + bv: four threads per output
        const int j = tid >> 2, part = tid & 3;
        const half_t *wr = Wv + (long)(h * NH_DH + j) * d;
        float acc = 0.f;
        for (int c = part; c < d; c += 4) acc += (float)wr[c] * us[c];
        acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
        if (part == 0) out[(long)b * d + h * NH_DH + j] = (half_t)(acc + bv[h * NH_DH + j]);
    }
}

void launch_xabs_attention(const half_t *q, const half_t *Wkv, const float *bkv, const half_t *xa, half_t *U, half_t *out, int B, int H, int d, int S,
                           const int32_t *done, hipStream_t st) {
    hipLaunchKernelGGL(xabs_u_kernel, dim3(H, B), dim3(256), 0, st, q, Wkv, U, d);
    hipLaunchKernelGGL(xabs_attn_kernel, dim3(H, B), dim3(256), sizeof(float) * (d + 4 * (d + 2)), st, U, xa, Wkv + (long)d * d, bkv + d, out, d, S, done);
}

// ---------------------------------------------------------------------------------------------------
// The one-pass form of the same arithmetic (NH_OPT_ABSORBED_XATTN = 2): xa is streamed ONCE per decoder layer.
//   xabs_u_fast_kernel U[b][h][:] = 1/8 Wk_h^T q_bh (MFMA, A = the transposed weight), heads padded to 32
//   xabs_main_kernel   one workgroup per (row, key range of 384): 32-key tiles of xa through LDS; per tile
//                        S = xa_tile U^T           (MFMA 32x32x16, M = keys, N = heads, K = features split over the 8 waves,
//                                                   partial sums met in LDS in wave order 0..7)
//                        online softmax per head   (a head per lane, like the encoder attention kernel's query per lane)
//                        z^T += xa_tile^T P        (M = features, N = heads, K = keys; the SAME LDS tile read column-wise with
//                                                   ds_read_b64_tr_b16, P straight from the accumulator registers)
//                      and writes (m, l, z) of its key range
//   xabs_zmerge_kernel merges the four key ranges of a (row, head) in the order 0,1,2,3 and rounds z to fp16
//   xabs_oproj_kernel  o_h = Wv_h z_h + bv_h
// The key ranges are the same for every batch size (one merge tree: a clip alone == the clip in a batch, bit for bit).
// LDS image of a tile: ten [32 rows][128 features] panels with 256-byte rows, chunk' = chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))
// (playbook T10 image (b): conflict-free for the row reads and the transposed reads); LDS row r holds key swap23(r) of the tile,
// which makes accumulator registers 0-7 / 8-15 of a lane the tile's keys 8 hh + 0..7 / 16 + 8 hh + 0..7, i.e. P is the B operand.
#define XA_KT 32                 // keys per tile
#define XA_LEAVES 4              // key ranges per row
__device__ __forceinline__ int xa_swap23(int x) { return (x & ~12) | ((x & 4) << 1) | ((x & 8) >> 1); }
__device__ __forceinline__ int xa_off(int panel, int row, int ch) { return panel * 8192 + 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// WkT[c][r] = Wk[r][c] (d x d): built once per model beside the tile-major repacks, so that u = Wk_h^T q reads 128 contiguous
// bytes per feature
__global__ __launch_bounds__(256) void transpose_sq_kernel(const half_t *__restrict__ in, half_t *__restrict__ out, int d) {
    __shared__ half_t t[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) t[r][tx] = in[(long)(by + r) * d + bx + tx];
    __syncthreads();
    for (int r = ty; r < 32; r += 8) out[(long)(bx + r) * d + by + tx] = t[tx][r];
}
void launch_transpose_sq(const half_t *in, half_t *out, int d, hipStream_t st) {
    hipLaunchKernelGGL(transpose_sq_kernel, dim3(d / 32, d / 32), dim3(256), 0, st, in, out, d);
}

// U[b][h][c] = 1/8 sum_j WkT[c][64 h + j] q[b][64 h + j] on the matrix pipe: per head a [d x 64] x [64 x rows] product.  One wave per
// (head, 32 features, 32 rows): four k-steps of v_mfma_f32_32x32x16_f16, A = WkT rows (features), B = q rows (j ascending: one
// summation order for every batch size; rows beyond R are clamped and not stored)
__global__ __launch_bounds__(256) void xabs_u_fast_kernel(const half_t *__restrict__ q, const half_t *__restrict__ WkT, half_t *__restrict__ U, int d, int R) {
    const int h = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, n = lane & 31, hh = lane >> 5;
    const int c0 = (blockIdx.y * 4 + w) * 32, b0 = blockIdx.z * 32;
    const int brow = min(b0 + n, R - 1);
    f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    half8 a[4], bq[4];
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
        a[ks] = *reinterpret_cast<const half8 *>(WkT + (long)(c0 + n) * d + h * NH_DH + 16 * ks + 8 * hh);
        bq[ks] = *reinterpret_cast<const half8 *>(q + (long)brow * d + h * NH_DH + 16 * ks + 8 * hh);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[ks], bq[ks], acc, 0, 0, 0);
    if (b0 + n < R) {   // register j is feature c0 + 8 (j >> 2) + 4 hh + (j & 3), column n is the row
        half_t *up = U + ((long)(b0 + n) * 32 + h) * d + c0 + 4 * hh;
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
            const half4 o = {(half_t)(0.125f * acc[4 * g4]), (half_t)(0.125f * acc[4 * g4 + 1]), (half_t)(0.125f * acc[4 * g4 + 2]), (half_t)(0.125f * acc[4 * g4 + 3])};
            *reinterpret_cast<half4 *>(up + 8 * g4) = o;
        }
    }
}

template <int D>   // D = d_model, a multiple of 256
__global__ __launch_bounds__(512) void xabs_main_kernel(const half_t *__restrict__ U, const half_t *__restrict__ xa, float *__restrict__ zpart,
                                                        float *__restrict__ mlpart, int S, int H, const int32_t *__restrict__ done) {
    constexpr int FW = D / 8;          // features per wave
    constexpr int KS1 = FW / 16;       // k-steps of the score product per wave
    constexpr int MB2 = FW / 32;       // 32-feature blocks of z^T per wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *tile = lds;                                            // XA_KT * D * 2 bytes
    float *part = reinterpret_cast<float *>(lds + XA_KT * D * 2);   // [8][16][64]
    float *red = part + 8 * 16 * 64;                             // [16][64]
    const int leaf = blockIdx.x, b = blockIdx.y;
    if (done && done[b]) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int KL = (((S + XA_LEAVES - 1) / XA_LEAVES + XA_KT - 1) / XA_KT) * XA_KT;   // keys per range, a multiple of the tile
    const int k0 = leaf * KL, k1 = min(S, k0 + KL);
    const int ntile = k1 > k0 ? (k1 - k0 + XA_KT - 1) / XA_KT : 0;
    const half_t *xr = xa + (long)b * S * D;
    // B operand of the score product: U[head n][features of this wave], resident
    half8 uf[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ks++) uf[ks] = *reinterpret_cast<const half8 *>(U + ((long)b * 32 + n) * D + w * FW + 16 * ks + 8 * hh);
    // staging: thread t moves chunk (t & 15) of every panel for key t >> 4 of the tile: one global base and one LDS base per tile,
    // everything else immediate offsets (the generic piece mapping kept ten addresses live and spilled them -- and a scratch
    // reload drains vmcnt, i.e. waits for the tile prefetch it sits in)
    constexpr int NPAN = D / 128;
    u32x4 stg[NPAN];
    const int skk = tid >> 4, sch = tid & 15;
    const int lds_st = xa_off(0, xa_swap23(skk), sch);
    auto gload = [&](int t) {
        int key = k0 + t * XA_KT + skk; if (key >= S) key = S - 1;
        const half_t *gp = xr + (long)key * D + 8 * sch;
#pragma unroll
        for (int i = 0; i < NPAN; i++) stg[i] = *reinterpret_cast<const u32x4 *>(gp + 128 * i);
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < NPAN; i++) *reinterpret_cast<u32x4 *>(tile + lds_st + 8192 * i) = stg[i];
    };
    f32x16 z[MB2];
#pragma unroll
    for (int i = 0; i < MB2; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) z[i][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifdef XA_STAMPS
    long long stamp[8]; int nst = 0;
#define XA_STAMP() do { if (t == 5 && nst < 8) stamp[nst++] = __builtin_readcyclecounter(); } while (0)
#else
#define XA_STAMP() do { } while (0)
#endif
    if (ntile > 0) gload(0);
    for (int t = 0; t < ntile; t++) {
        XA_STAMP();
        lstore();
        __syncthreads();
        XA_STAMP();
        if (t + 1 < ntile) gload(t + 1);
        // ---- S partial over this wave's features: A = tile rows (keys), row m = n
        f32x16 sp = zero, sq = zero;     // two chains (even / odd k-steps), summed once
#pragma unroll
        for (int ks = 0; ks < KS1; ks++) {
            const int f = w * FW + 16 * ks + 8 * hh;
            const half8 a = *reinterpret_cast<const half8 *>(tile + xa_off(f >> 7, n, (f & 127) >> 3));
            if (ks & 1) sq = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, uf[ks], sq, 0, 0, 0);
            else sp = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, uf[ks], sp, 0, 0, 0);
        }
        XA_STAMP();
#pragma unroll
        for (int j = 0; j < 16; j++) part[(w * 16 + j) * 64 + lane] = sp[j] + sq[j];
        __syncthreads();
        XA_STAMP();
        {   // wave w sums accumulator registers 2 w, 2 w + 1 of all eight partials, in wave order
#pragma unroll
            for (int jj = 0; jj < 2; jj++) {
                const int j = 2 * w + jj;
                float v = part[(0 * 16 + j) * 64 + lane];
#pragma unroll
                for (int ww = 1; ww < 8; ww++) v += part[(ww * 16 + j) * 64 + lane];
                red[j * 64 + lane] = v;
            }
        }
        __syncthreads();
        XA_STAMP();
        float sv[16];
#pragma unroll
        for (int j = 0; j < 16; j++) sv[j] = red[j * 64 + lane];
        // ---- online softmax of head n over the tile's keys: register j is key (j & 7) + 8 hh + 16 (j >> 3) of the tile
        const int kbase = k0 + t * XA_KT;
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const int key = kbase + (j & 7) + 8 * hh + 16 * (j >> 3);
            if (key >= k1) sv[j] = -INFINITY;
            mx = fmaxf(mx, sv[j]);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float mn = fmaxf(m_run, mx);                 // finite: every tile holds at least one key < k1
        const float al = __expf(m_run - mn);               // exp(-inf) = 0 on the first tile
        float ps = 0.f;
        half8 p0, p1;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const half_t a = (half_t)__expf(sv[j] - mn), c = (half_t)__expf(sv[8 + j] - mn);
            p0[j] = a; p1[j] = c; ps += (float)a + (float)c;
        }
        ps += __shfl_xor(ps, 32);
        l_run = l_run * al + ps;
        m_run = mn;
        // ---- z^T[features of this wave][heads] = al z^T + tile^T P: A by transposed reads, lane group G = lane >> 4
        XA_STAMP();
        const int G = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
        if (__builtin_amdgcn_ballot_w64(al != 1.0f) != 0) {   // the running maxima rarely move after the first tiles
#pragma unroll
            for (int fb = 0; fb < MB2; fb++)
#pragma unroll
                for (int e = 0; e < 16; e++) z[fb][e] *= al;
        }
        // address of this lane's part of a transposed read: row = key 8 (G >> 1) + 4 half + q4 (+ 16 ts) of the tile, chunk pair c0.
        // swap23 only moves bits 2 and 3 of the key: key = 16 ts + 8 g + 4 half + q4 -> row = 16 ts + 4 g + 8 half + q4
        const int trow0 = 4 * (G >> 1) + q4;
#pragma unroll
        for (int fb = 0; fb < MB2; fb++) {
            const int f0 = w * FW + 32 * fb + 16 * (G & 1);          // first feature of this group's 16 columns
            const int panel = f0 >> 7, c0 = (f0 & 127) >> 3;          // chunk of 8 features; the group covers chunks c0, c0 + 1
#pragma unroll
            for (int ts = 0; ts < 2; ts++) {                          // k-step: keys 16 ts + 8 (G >> 1) + 0..7 of the tile
                half8 a;
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int row = 16 * ts + 8 * half + trow0;
                    typedef __fp16 fp16x4 __attribute__((ext_vector_type(4)));
                    // (the tile is the start of the dynamic LDS; the builtin wants an LDS-qualified pointer)
                    const fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16(
                        (__attribute__((address_space(3))) fp16x4 *)(&lds[xa_off(panel, row, c0 + (p4 >> 1)) + 8 * (p4 & 1)]));
                    a[4 * half + 0] = (half_t)v[0]; a[4 * half + 1] = (half_t)v[1]; a[4 * half + 2] = (half_t)v[2]; a[4 * half + 3] = (half_t)v[3];
                }
                z[fb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, ts == 0 ? p0 : p1, z[fb], 0, 0, 0);
            }
        }
        XA_STAMP();
        __syncthreads();   // every wave is done with the tile (and with red) before the next one is stored
        XA_STAMP();
    }
#ifdef XA_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && lane == 0) {   // per wave: cycle counter at the eight points of tile 5
        float *sp_ = mlpart + (long)gridDim.y * XA_LEAVES * 32 * 2 + w * 8;
        for (int k = 0; k < 8; k++) sp_[k] = k < nst ? (float)(stamp[k] - stamp[0]) : -1.f;
    }
#endif
    // ---- this key range's state: m, l per head; z[head][feature] f32.  Accumulator register j of block fb is feature
    // 32 fb + 8 (j >> 2) + 4 hh + (j & 3) of this wave's range, column n is the head
    if (n < H) {
        float *zp = zpart + (((long)b * XA_LEAVES + leaf) * H + n) * D + w * FW;
#pragma unroll
        for (int fb = 0; fb < MB2; fb++)
#pragma unroll
            for (int g4 = 0; g4 < 4; g4++) {
                const f32x4 v = {z[fb][4 * g4], z[fb][4 * g4 + 1], z[fb][4 * g4 + 2], z[fb][4 * g4 + 3]};
                *reinterpret_cast<f32x4 *>(zp + 32 * fb + 8 * g4 + 4 * hh) = v;
            }
        if (w == 0 && hh == 0) {
            float *ml = mlpart + (((long)b * XA_LEAVES + leaf) * 32 + n) * 2;
            ml[0] = m_run; ml[1] = l_run;
        }
    }
}

// merge the key ranges of (row, head) in the order 0..3 and round z to fp16: one workgroup per (head, row), one round trip
__global__ __launch_bounds__(256) void xabs_zmerge_kernel(const float *__restrict__ zpart, const float *__restrict__ mlpart, half_t *__restrict__ z16,
                                                          int d, int H, const int32_t *__restrict__ done) {
    const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (done && done[b]) return;
    float f[XA_LEAVES], mm = -INFINITY, ll = 0.f;
#pragma unroll
    for (int lf = 0; lf < XA_LEAVES; lf++) {
        const float m2 = mlpart[(((long)b * XA_LEAVES + lf) * 32 + h) * 2], l2 = mlpart[(((long)b * XA_LEAVES + lf) * 32 + h) * 2 + 1];
        const float mn = fmaxf(mm, m2);
        const float a1 = (mm == -INFINITY) ? 0.f : __expf(mm - mn), a2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
#pragma unroll
        for (int k = 0; k < lf; k++) f[k] *= a1;
        f[lf] = a2; ll = ll * a1 + l2 * a2; mm = mn;
    }
    const float inv = 1.0f / ll;
    for (int c = 4 * tid; c < d; c += 1024) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int lf = 0; lf < XA_LEAVES; lf++) v += *reinterpret_cast<const f32x4 *>(zpart + (((long)b * XA_LEAVES + lf) * H + h) * d + c) * f[lf];
        const half4 o = {(half_t)(v[0] * inv), (half_t)(v[1] * inv), (half_t)(v[2] * inv), (half_t)(v[3] * inv)};
        *reinterpret_cast<half4 *>(z16 + ((long)b * 32 + h) * d + c) = o;
    }
}

// o_h = Wv_h z_h + bv_h on the matrix pipe: per head a [64 x d] x [d x rows] product.  One workgroup per (head, 32 rows); the eight
// waves split d (k ascending inside a wave), their partial sums meet in LDS in wave order 0..7
template <int D>
__global__ __launch_bounds__(512) void xabs_oproj_kernel(const half_t *__restrict__ z16, const half_t *__restrict__ Wv, const float *__restrict__ bv,
                                                         half_t *__restrict__ out, int R, const int32_t *__restrict__ done) {
    constexpr int FW = D / 8, KS = FW / 16;
    __shared__ float part[8][2][16][64];
    const int h = blockIdx.x, b0 = blockIdx.y * 32, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, n = lane & 31, hh = lane >> 5;
    const int brow = min(b0 + n, R - 1);
    half8 a0[KS], a1[KS], bz[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        const int c = w * FW + 16 * ks + 8 * hh;
        a0[ks] = *reinterpret_cast<const half8 *>(Wv + (long)(h * NH_DH + n) * D + c);
        a1[ks] = *reinterpret_cast<const half8 *>(Wv + (long)(h * NH_DH + 32 + n) * D + c);
        bz[ks] = *reinterpret_cast<const half8 *>(z16 + ((long)brow * 32 + h) * D + c);
    }
    f32x16 o0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0[ks], bz[ks], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1[ks], bz[ks], o1, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 16; j++) { part[w][0][j][lane] = o0[j]; part[w][1][j][lane] = o1[j]; }
    __syncthreads();
    // 2 x 16 x 64 sums: thread t takes (block, register, lane) = (t >> 10 .. ) in four passes of 512
#pragma unroll
    for (int pss = 0; pss < 4; pss++) {
        const int e = tid + 512 * pss, blk = e >> 10, jj = (e >> 6) & 15, ln = e & 63;
        float v = part[0][blk][jj][ln];
#pragma unroll
        for (int ww = 1; ww < 8; ww++) v += part[ww][blk][jj][ln];
        // register jj of lane ln: output 32 blk + 8 (jj >> 2) + 4 (ln >> 5) + (jj & 3), row b0 + (ln & 31)
        const int jo = 32 * blk + 8 * (jj >> 2) + 4 * (ln >> 5) + (jj & 3), b = b0 + (ln & 31);
        if (b < R && !(done && done[b])) out[(long)b * D + h * NH_DH + jo] = (half_t)(v + bv[h * NH_DH + jo]);
    }
}

bool xabs_fast_supported(int d, int H) { return (d == 1280 || d == 1024 || d == 768 || d == 512) && H <= 32; }

// scratch: U fp16 [B][32][d] (rows of heads >= H must be zero), zpart f32 [B][4][H][d], mlpart f32 [B][4][32][2]
void launch_xabs_attention_fast(const half_t *q, const half_t *WkT, const half_t *Wkv, const float *bkv, const half_t *xa, half_t *U, float *zpart, float *mlpart,
                                half_t *out, int B, int H, int d, int S, const int32_t *done, hipStream_t st) {
    hipLaunchKernelGGL(xabs_u_fast_kernel, dim3(H, d / 128, (B + 31) / 32), dim3(256), 0, st, q, WkT, U, d, B);
    const size_t lds = (size_t)XA_KT * d * 2 + (8 * 16 * 64 + 16 * 64) * sizeof(float);
    static std::atomic<bool> attr_set[NH_MAX_DEVICES];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= NH_MAX_DEVICES || !attr_set[dev].load(std::memory_order_acquire)) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&xabs_main_kernel<1280>), hipFuncAttributeMaxDynamicSharedMemorySize, NH_LDS_EXCLUSIVE);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&xabs_main_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, NH_LDS_EXCLUSIVE);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&xabs_main_kernel<768>), hipFuncAttributeMaxDynamicSharedMemorySize, NH_LDS_EXCLUSIVE);
        hipFuncSetAttribute(reinterpret_cast<const void *>(&xabs_main_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, NH_LDS_EXCLUSIVE);
        if (dev >= 0 && dev < NH_MAX_DEVICES) attr_set[dev].store(true, std::memory_order_release);
    }
    // MFMAs fed from LDS reads: the workgroup takes its CU's whole LDS (NH_LDS_EXCLUSIVE, see there); it uses `lds` bytes
    const dim3 grid(XA_LEAVES, B);
    if (d == 1280) hipLaunchKernelGGL((xabs_main_kernel<1280>), grid, dim3(512), lds_exclusive_bytes(lds), st, U, xa, zpart, mlpart, S, H, done);
    else if (d == 1024) hipLaunchKernelGGL((xabs_main_kernel<1024>), grid, dim3(512), lds_exclusive_bytes(lds), st, U, xa, zpart, mlpart, S, H, done);
    else if (d == 768) hipLaunchKernelGGL((xabs_main_kernel<768>), grid, dim3(512), lds_exclusive_bytes(lds), st, U, xa, zpart, mlpart, S, H, done);
    else hipLaunchKernelGGL((xabs_main_kernel<512>), grid, dim3(512), lds_exclusive_bytes(lds), st, U, xa, zpart, mlpart, S, H, done);
    // the merged z (fp16 [B][32][d]) takes U's place: U was consumed by the main kernel
    hipLaunchKernelGGL(xabs_zmerge_kernel, dim3(H, B), dim3(256), 0, st, zpart, mlpart, U, d, H, done);
    const dim3 og(H, (B + 31) / 32);
    const half_t *Wv = Wkv + (long)d * d; const float *bvp = bkv + d;
    if (d == 1280) hipLaunchKernelGGL((xabs_oproj_kernel<1280>), og, dim3(512), 0, st, U, Wv, bvp, out, B, done);
    else if (d == 1024) hipLaunchKernelGGL((xabs_oproj_kernel<1024>), og, dim3(512), 0, st, U, Wv, bvp, out, B, done);
    else if (d == 768) hipLaunchKernelGGL((xabs_oproj_kernel<768>), og, dim3(512), 0, st, U, Wv, bvp, out, B, done);
    else hipLaunchKernelGGL((xabs_oproj_kernel<512>), og, dim3(512), 0, st, U, Wv, bvp, out, B, done);
}

void launch_dec_attention(const half_t *q, const half_t *kc, const half_t *vc, half_t *out, int B, int Tn,
                          int H, int d, int ctx, int Tk, const int32_t *pos_ptr, hipStream_t st, int kv_head_major,
                          const int32_t *done) {
    (void)Tn;  // one new position per sequence; its visible keys are exactly Tk (or pos_ptr[b] + 1)
    hipLaunchKernelGGL(dec_attn_kernel, dim3(H, B), dim3(256), 0, st, q, kc, vc, out, d, ctx, Tk, pos_ptr, kv_head_major, done);
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

// which of the four mask chains of model.rs:331-338 / :245-277 applies.  probs(i) gives the soft-maxed probability.
template <typename ProbFn>
__device__ __forceinline__ int rules_decide(ProbFn probs, int V, const int32_t *tokens, int n, int have_last,
                                            const uint8_t *sup, const RuleTokens &tk, BlockRed &sm) {
    if (!have_last) return RULE_FIRST;
    int l = tokens[n - 1];
    if (l > tk.no_timestamps) {
        int sl = n >= 2 ? tokens[n - 2] : -1;
        return (n >= 2 && sl >= tk.eot) ? RULE_SUP_TS : RULE_NON_TS;
    }
    float ps = 0.f, pm = -INFINITY;  // model.rs:263-270 on the suppress-masked probabilities
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        float p = probs(i);
        float pv = sup[i] ? p + (-INFINITY) : p;
        if (i > tk.no_timestamps) ps += pv;
        else if (i < tk.no_timestamps) pm = fmaxf(pm, pv);
    }
    float sum_ts = block_sum(ps, sm);
    float max_text = block_max(pm, sm);
    return (sum_ts >= max_text) ? RULE_NON_TS : RULE_PAST;
}

// shared by the parity helper and the sampled step: rules, then the greedy arg max
template <typename ProbFn>
__device__ __forceinline__ void rules_argmax(ProbFn probs, int V, const int32_t *tokens, int n, int have_last,
                                             int last_ts, const uint8_t *sup, const RuleTokens &tk, BlockRed &sm,
                                             int &rule_out, int &next_out) {
    const int rule = rules_decide(probs, V, tokens, n, have_last, sup, tk, sm);
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

// ---- sampled decoding, t > 0 (model.rs:340-348) ------------------------------------------------------------
// The reference draws from rand's WeightedIndex over softmax(q / t), q = the rule-masked PROBABILITIES, with an
// entropy-seeded StdRng, so only its distribution can be matched.  The seeded contract (include/norma_hip.h):
//   w_i = sexp((q_i - max q) * inv_t)        sexp: exp from IEEE f32 operations only, identical in the C oracle
//   u   = (philox4x32-10(key = seed, ctr = {step, clip, attempt, "norm"})[0] >> 8) * 2^-24
//   token = first j whose cumulative weight exceeds u * total, cumulated in f64 over 1024 chunks of ceil(V / 1024)
// One 1024-thread workgroup per sequence; thread c owns chunk c, thread 0 then walks the chunk sums.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned &o0) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o0 = c0;
}

__device__ __forceinline__ float sexp(float y) {
#pragma clang fp contract(off)
    if (!(y >= -87.0f)) return 0.0f;  // also -inf and NaN: a masked token has weight 0
    const float kf = floorf(__builtin_fmaf(y, 1.44269504088896341f, 0.5f));
    float r = __builtin_fmaf(kf, -0.693359375f, y);
    r = __builtin_fmaf(kf, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float z = r * r;
    const float res = __builtin_fmaf(p, z, r) + 1.0f;
    return res * __uint_as_float((unsigned)((int)kf + 127) << 23);
}

struct SampleShared { BlockRed red; double chunk[1024]; int result; float qres; };

// q(i): rule-masked probability.  Returns (all threads) the sampled token or -1 when everything is masked; qout = q(token).
template <typename QFn>
__device__ __forceinline__ int sample_masked(QFn q, int V, float inv_t, unsigned long long seed, unsigned clip, unsigned step,
                                             unsigned attempt, SampleShared &sh, float &qout) {
#pragma clang fp contract(off)
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, q(i));
    const float qmax = block_max(mx, sh.red);
    if (!(qmax > -INFINITY)) { qout = 0.f; return -1; }
    const int CH = (V + 1023) / 1024;
    double sc = 0.0;
    for (int i = tid * CH; i < (tid + 1) * CH && i < V; i++) sc += (double)sexp((q(i) - qmax) * inv_t);
    sh.chunk[tid] = sc;
    __syncthreads();
    if (tid == 0) {
        double total = 0.0;
        for (int c = 0; c < 1024; c++) total += sh.chunk[c];
        unsigned r0;
        philox4x32_10(step, clip, attempt, 0x6e6f726du, (unsigned)seed, (unsigned)(seed >> 32), r0);
        const float u = (float)(r0 >> 8) * (1.0f / 16777216.0f);
        const double x = (double)u * total;
        double run = 0.0;
        int c = 0;
        for (; c < 1023; c++) { if (run + sh.chunk[c] > x) break; run += sh.chunk[c]; }
        int res = -1, last_pos = -1;
        for (int i = c * CH; i < V; i++) {  // walks on past the chunk only if rounding left x >= the chunk's end
            const float w = sexp((q(i) - qmax) * inv_t);
            if (w > 0.0f) last_pos = i;
            run += (double)w;
            if (run > x) { res = i; break; }
        }
        if (res < 0) res = last_pos;
        sh.result = res; sh.qres = q(res);
    }
    __syncthreads();
    qout = sh.qres;
    return sh.result;
}

// one generated token per sequence at temperature 1 / inv_t: softmax, rules, sampling, the bookkeeping of
// model.rs:359-370.  Same state as logit_step_kernel (which stays the t = 0 path and the no-speech probe).
__global__ __launch_bounds__(1024) void sample_step_kernel(const float *__restrict__ logits, int V, int ldl, DecodeState s,
                                                           RuleTokens tk, int ctx, int cap, int max_new, int prompt_len,
                                                           float inv_t, unsigned long long seed, unsigned clip0, unsigned attempt) {
    __shared__ SampleShared sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (s.done[b]) return;
    const float *lg = logits + (long)b * ldl;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 1024) mx = fmaxf(mx, lg[i]);
    const float m = block_max(mx, sh.red);
    float se = 0.f;
    for (int i = tid; i < V; i += 1024) se += expf(lg[i] - m);
    se = block_sum(se, sh.red);
    auto probs = [&](int i) { return expf(lg[i] - m) / se; };  // model.rs:331
    int32_t *toks = s.tokens + (long)b * ctx;
    const int n = s.n_tokens[b], have_last = s.have_last[b], last_ts = s.last_ts[b];
    const int rule = rules_decide(probs, V, toks, n, have_last, s.suppress, tk, sh.red);
    auto q = [&](int i) { return masked_value(probs(i), i, rule, s.suppress, tk, last_ts); };
    float qv;
    const int next = sample_masked(q, V, inv_t, seed, clip0 + b, (unsigned)n, attempt, sh, qv);
    if (tid != 0) return;
    int nn = n, fin = 0;
    if (next < 0) { toks[nn++] = tk.eot; fin = 1; }                // :343-346 all NaN: push eot, stop (no log-prob)
    else {
        if (next > tk.no_timestamps) { s.last_ts[b] = next; s.have_last[b] = 1; }  // :359-361
        toks[nn++] = next;
        s.sum_logprob[b] += log((double)qv);                       // :364-365
        if (nn >= cap) { toks[nn++] = tk.eot; fin = 1; }           // :367-370
        else if (next == tk.eot) fin = 1;                          // :317
        else if (max_new > 0 && nn - prompt_len >= max_new) { toks[nn++] = tk.eot; fin = 1; }
    }
    s.n_tokens[b] = nn;
    if (fin) s.done[b] = 1;
}

void launch_sample_step(const float *logits, int V, DecodeState s, RuleTokens tk, int B, int ctx, int cap, int max_new,
                        int prompt_len, float inv_t, unsigned long long seed, unsigned clip0, unsigned attempt, hipStream_t st) {
    const int ldl = (V + 63) & ~63;
    hipLaunchKernelGGL(sample_step_kernel, dim3(B), dim3(1024), 0, st, logits, V, ldl, s, tk, ctx, cap, max_new, prompt_len,
                       inv_t, seed, clip0, attempt);
}

// parity view of the sampler: rules + one draw on an already soft-maxed probability vector
__global__ __launch_bounds__(1024) void sample_rules_kernel(const float *__restrict__ probs_in, int32_t *token_out,
                                                            const int32_t *tokens, int n, int last_ts, const uint8_t *sup,
                                                            RuleTokens tk, int V, float inv_t, unsigned long long seed,
                                                            unsigned clip, unsigned attempt) {
    __shared__ SampleShared sh;
    auto probs = [&](int i) { return probs_in[i]; };
    const int rule = rules_decide(probs, V, tokens, n, last_ts >= 0, sup, tk, sh.red);
    auto q = [&](int i) { return masked_value(probs_in[i], i, rule, sup, tk, last_ts); };
    float qv;
    const int next = sample_masked(q, V, inv_t, seed, clip, (unsigned)n, attempt, sh, qv);
    if (threadIdx.x == 0) *token_out = next;
}

void launch_sample_rules(const float *probs_in, int32_t *token_out, const int32_t *tokens, int n_tokens, int last_ts,
                         const uint8_t *suppress, RuleTokens tk, int V, float inv_t, unsigned long long seed, unsigned clip,
                         unsigned attempt, hipStream_t st) {
    hipLaunchKernelGGL(sample_rules_kernel, dim3(1), dim3(1024), 0, st, probs_in, token_out, tokens, n_tokens, last_ts, suppress,
                       tk, V, inv_t, seed, clip, attempt);
}

// ---- the fused decode-step version: ONE sweep over the logits, LSPLIT workgroups per sequence -----------
// softmax is monotonic, so the arg max over an allowed set can be taken on the logits; which set is
// allowed is known before the sweep except for the "last token is text" case, where both candidates
// (best timestamp after last_ts, best allowed text token) are tracked and the choice
// sum_ts >= max_text is made by the last workgroup to arrive.  (Two DISTINCT logits whose f32
// probabilities round to the same value would tie in the reference and resolve to the higher index;
// here the larger logit wins.  That needs a relative gap < 6e-8 and is far below the fp16 noise floor.)
#define LSPLIT 8

__device__ __forceinline__ void merge_ms(float &m, float &s, float &ts, float m2, float s2, float ts2) {
    float mn = fmaxf(m, m2);
    float f1 = (m == -INFINITY) ? 0.f : __expf(m - mn), f2 = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
    s = s * f1 + s2 * f2; ts = ts * f1 + ts2 * f2; m = mn;
}
__device__ __forceinline__ void better(float &v, int &i, float v2, int i2) {  // larger value, then larger index
    if (i2 >= 0 && (i < 0 || v2 > v || (v2 == v && i2 > i))) { v = v2; i = i2; }
}

__global__ __launch_bounds__(256) void logit_step_kernel(const float *__restrict__ logits, int V, int ldl,
                                                         DecodeState s, RuleTokens tk, int ctx, int cap, int max_new,
                                                         int prompt_len, int mode, float *partials, unsigned *tickets,
                                                         int32_t *pos_ptr) {
    __shared__ float sh_f[4][6];
    __shared__ int sh_i[4][2];
    __shared__ int sh_last;
    const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *lg = logits + (long)b * ldl;
    const int per = (V + LSPLIT - 1) / LSPLIT, lo = part * per, hi = min(V, lo + per);
    // fetch the whole slice first: the statistics below are a dependent chain, the loads are not
    // (they are issued before the per-sequence state is even looked at -- one memory round trip for both)
    constexpr int LMAX = 32;  // ceil(51866 / 8 / 256) = 26 elements per thread
    float lv[LMAX]; unsigned char sv[LMAX];
#pragma unroll
    for (int u = 0; u < LMAX; u++) {  // unconditional (clamped) loads: a guarded load makes hipcc wait vmcnt(0) per element
        int i = lo + tid + 256 * u;
        int ic = i < hi ? i : hi - 1;
        lv[u] = lg[ic];
        sv[u] = s.suppress[ic];
    }
    const int32_t *toks = s.tokens + (long)b * ctx;
    const int done = s.done[b];
    const int my_pos = pos_ptr ? pos_ptr[b] : 0;  // this sequence's position; advanced below by whoever finishes its step
    const int n = s.n_tokens[b];
    const int have_last = s.have_last[b], last_ts = s.last_ts[b];
    const double sum_lp_in = s.sum_logprob[b];  // prefetched for the bookkeeping at the end
    const int l1 = toks[n >= 1 ? n - 1 : 0], l2 = toks[n >= 2 ? n - 2 : 0];  // unconditional: one round trip for both
    if (done) return;
    // mode 2 (decode pool): sequences join a running decode, so each is in its own phase -- position 0 of its prompt is the
    // no-speech probe, the other prompt positions only feed the caches (their next token is given), then it generates
    if (mode == 2) {
        mode = my_pos == 0 ? 0 : (my_pos < prompt_len - 1 ? 3 : 1);
        if (mode == 3) {  // the position moves only when all LSPLIT workgroups of the sequence have read it: same ticket as below
            if (tid == 0 && __hip_atomic_fetch_add(tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == LSPLIT - 1) {
                __hip_atomic_store(tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pos_ptr[b] = my_pos + 1;
            }
            return;
        }
    }
    const int NT = tk.no_timestamps;
    // candidate sets: A = allowed non-timestamp tokens (or the first-token window), B = allowed timestamps
    int kind;  // 0 FIRST, 1 SUP_TS, 2 NON_TS, 3 TEXT (NON_TS vs PAST decided at the end), 4 no-speech probe
    if (mode == 0) kind = 4;
    else if (!have_last) kind = 0;
    else if (l1 > NT) kind = (n >= 2 && l2 >= tk.eot) ? 1 : 2;
    else kind = 3;
    float m = -INFINITY, se = 0.f, ts = 0.f, tsinf = 0.f, av = -INFINITY, bv = -INFINITY;
    int ai = -1, bi = -1;
    for (int i0 = lo + 256 * LMAX; i0 < hi; i0 += 256) {  // vocabularies beyond 8 * 256 * LMAX tokens (none today)
        int i = i0 + tid;
        if (i < hi) { float l = lg[i]; if (l > m) { float f = __expf(m - l); se = se * f + 1.f; ts = ts * f; m = l; if (i > NT) ts += 1.f; } else { float e = __expf(l - m); se += e; if (i > NT) ts += e; } }
    }
    // slice maximum first, then one exp per element against it (the slice lives in registers)
#pragma unroll
    for (int u = 0; u < LMAX; u++) m = fmaxf(m, lv[u]);
#pragma unroll
    for (int u = 0; u < LMAX; u++) {
        const int i = lo + tid + 256 * u;
        if (i >= hi) continue;
        const float l = lv[u];
        const bool is_ts = i > NT;
        const bool sup = sv[u] != 0;
        const float e = __expf(l - m);   // softmax mass, and the timestamp mass of model.rs:263-266
        se += e;
        if (is_ts) { ts += e; if (sup) tsinf = 1.f; }  // p + (-inf) inside the summed slice -> the sum is -inf
        if (kind == 0) { if (i >= tk.zero_sec && i <= tk.one_sec) better(av, ai, l, i); }
        else if (kind == 1) { if (!is_ts && !sup) better(av, ai, l, i); }
        else if (kind == 2) { if (is_ts && i > last_ts && !sup) better(bv, bi, l, i); }
        else if (kind == 3) {
            if (!is_ts && !sup) better(av, ai, l, i);
            else if (is_ts && i > last_ts && !sup) better(bv, bi, l, i);
        }
    }
    // wave reduction
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        merge_ms(m, se, ts, __shfl_xor(m, o), __shfl_xor(se, o), __shfl_xor(ts, o));
        tsinf = fmaxf(tsinf, __shfl_xor(tsinf, o));
        better(av, ai, __shfl_xor(av, o), __shfl_xor(ai, o));
        better(bv, bi, __shfl_xor(bv, o), __shfl_xor(bi, o));
    }
    if (lane == 0) { sh_f[w][0] = m; sh_f[w][1] = se; sh_f[w][2] = ts; sh_f[w][3] = tsinf; sh_f[w][4] = av; sh_f[w][5] = bv; sh_i[w][0] = ai; sh_i[w][1] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int ww = 1; ww < 4; ww++) {
            merge_ms(m, se, ts, sh_f[ww][0], sh_f[ww][1], sh_f[ww][2]);
            tsinf = fmaxf(tsinf, sh_f[ww][3]);
            better(av, ai, sh_f[ww][4], sh_i[ww][0]);
            better(bv, bi, sh_f[ww][5], sh_i[ww][1]);
        }
        // publish the partial (agent-scope atomics: other workgroups may sit on another XCD/L2), then take a ticket
        float *pp = partials + ((long)b * LSPLIT + part) * 8;
        __hip_atomic_store(pp + 0, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 1, se, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 2, ts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 3, tsinf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 4, av, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pp + 5, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<int *>(pp) + 6, ai, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<int *>(pp) + 7, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Hand-off to the last arriver.  The payload above went out as agent-scope (sc1, write-through) atomic stores; on gfx950
        // `s_waitcnt vmcnt(0)` returns only when the memory side has acknowledged them, so a relaxed ticket RMW issued after
        // it cannot be observed before them, and the last arriver reads the payload with sc1 loads that bypass its own L2.
        // That is the ISA-level contract this library (gfx950 only) relies on.  The portable spelling -- a RELEASE ticket +
        // an ACQUIRE fence in the last arriver, -DNH_STRICT_MEMORY_MODEL -- makes every arrival write back its whole L2
        // (buffer_wbl2): measured -2 % end-to-end with three batches in flight (5940 vs 6060 audio-s/s), same results.
#if defined(NH_STRICT_MEMORY_MODEL)
        unsigned t = __hip_atomic_fetch_add(tickets + b, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (t == LSPLIT - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned t = __hip_atomic_fetch_add(tickets + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        sh_last = (t == LSPLIT - 1);
    }
    __syncthreads();
    if (!sh_last || w != 0) return;
    // ---- last workgroup of this sequence: combine and do the bookkeeping of model.rs:331-370 ----
    // one L2 round trip: lane 8 q + f fetches field f of partial q, thread 0 then walks them by shuffle
    const unsigned raw = __hip_atomic_load(reinterpret_cast<const unsigned *>(partials) + (long)b * LSPLIT * 8 + lane,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    m = -INFINITY; se = 0.f; ts = 0.f; tsinf = 0.f; av = -INFINITY; bv = -INFINITY; ai = -1; bi = -1;
#pragma unroll
    for (int q = 0; q < LSPLIT; q++) {
        float m2 = __uint_as_float(__shfl(raw, 8 * q + 0)), s2 = __uint_as_float(__shfl(raw, 8 * q + 1));
        float t2 = __uint_as_float(__shfl(raw, 8 * q + 2)), i2 = __uint_as_float(__shfl(raw, 8 * q + 3));
        float a2 = __uint_as_float(__shfl(raw, 8 * q + 4)), b2 = __uint_as_float(__shfl(raw, 8 * q + 5));
        int ai2 = (int)__shfl(raw, 8 * q + 6), bi2 = (int)__shfl(raw, 8 * q + 7);
        merge_ms(m, se, ts, m2, s2, t2);
        tsinf = fmaxf(tsinf, i2);
        better(av, ai, a2, ai2);
        better(bv, bi, b2, bi2);
    }
    if (tid != 0) return;
    __hip_atomic_store(tickets + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next token
    if (mode == 0) {  // model.rs:293-315
        float p = expf(lg[tk.no_speech] - m) / se;
        s.no_speech[b] = (double)p;
        if ((double)p > 0.6) s.done[b] = 2;
        if (pos_ptr) pos_ptr[b] = my_pos + 1;
        return;
    }
    if (pos_ptr) pos_ptr[b] = my_pos + 1;  // every workgroup of this sequence read it before taking its ticket
    int next = -1; float lnext = 0.f;
    if (kind == 0 || kind == 1) { next = ai; lnext = av; }
    else if (kind == 2) { next = bi; lnext = bv; }
    else {
        float sum_ts = tsinf > 0.f ? -INFINITY : ts / se;          // probabilities, as the reference compares them
        float max_text = ai >= 0 ? expf(av - m) / se : -INFINITY;
        if (sum_ts >= max_text) { next = bi; lnext = bv; }          // supress_non_timestamps
        else { next = ai; lnext = av; better(lnext, next, bv, bi); }  // supress_past_timestamps only
    }
    float pv;
    if (next < 0) { next = V - 1; pv = -INFINITY; }  // every candidate masked: all -inf, last index wins (H3)
    else pv = expf(lnext - m) / se;
    int32_t *wt = s.tokens + (long)b * ctx;
    int nn = n;
    if (next > NT) { s.last_ts[b] = next; s.have_last[b] = 1; }  // :359-361
    wt[nn++] = next;
    s.sum_logprob[b] = sum_lp_in + log((double)pv);               // :364-365
    int fin = 0;
    if (nn >= cap) { wt[nn++] = tk.eot; fin = 1; }                // :367-370
    else if (next == tk.eot) fin = 1;                             // :317
    else if (max_new > 0 && nn - prompt_len >= max_new) { wt[nn++] = tk.eot; fin = 1; }  // bench knob
    s.n_tokens[b] = nn;
    if (fin) s.done[b] = 1;
}

void launch_logit_step(const float *logits, int V, DecodeState s, RuleTokens tk, int B, int ctx, int cap,
                       int max_new, int prompt_len, int mode, float *partials, unsigned *tickets, int32_t *pos_ptr,
                       hipStream_t st) {
    int ldl = (V + 63) & ~63;
    hipLaunchKernelGGL(logit_step_kernel, dim3(LSPLIT, B), dim3(256), 0, st, logits, V, ldl, s, tk, ctx, cap, max_new,
                       prompt_len, mode, partials, tickets, pos_ptr);
}

// decode pool: sequence `row` starts over with the prompt [t0, t1, (t2)] (model.rs:285-289) at position 0
__global__ void pool_admit_kernel(DecodeState s, int32_t *pos, unsigned *tickets, int row, int ctx, int t0, int t1, int t2, int P) {
    int32_t *t = s.tokens + (long)row * ctx;
    t[0] = t0; t[1] = t1; if (P == 3) t[2] = t2;
    s.n_tokens[row] = P; s.done[row] = 0; s.have_last[row] = 0; s.last_ts[row] = 0;
    s.sum_logprob[row] = 0.0; s.no_speech[row] = 0.0;
    pos[row] = 0; tickets[row] = 0u;
}
void launch_pool_admit(DecodeState s, int32_t *pos, unsigned *tickets, int row, int ctx, int t0, int t1, int t2, int P, hipStream_t st) {
    hipLaunchKernelGGL(pool_admit_kernel, dim3(1), dim3(1), 0, st, s, pos, tickets, row, ctx, t0, t1, t2, P);
}

// Model::detect_language (model.rs:194-210) on the position-0 logits of a [sot] prompt: softmax over the language
// tokens and the FIRST maximum (the reference sorts descending with a stable sort).  One wave per sequence.
__global__ __launch_bounds__(64) void lang_detect_kernel(const float *__restrict__ logits, int ldl,
                                                         const int32_t *__restrict__ lang_tokens, int n,
                                                         float *__restrict__ probs_out, int32_t *__restrict__ lang_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *lg = logits + (long)b * ldl;
    float v[4]; int idx[4];
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        idx[u] = lane + 64 * u;
        v[u] = idx[u] < n ? lg[lang_tokens[idx[u] < n ? idx[u] : 0]] : -INFINITY;
        mx = fmaxf(mx, v[u]);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f, e[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { e[u] = idx[u] < n ? expf(v[u] - mx) : 0.f; se += e[u]; }
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    int bk = INT_MIN, bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (idx[u] < n) {
            float p = e[u] / se;
            if (probs_out) probs_out[(long)b * n + idx[u]] = p;
            int key = total_key(p);
            if (key > bk || (key == bk && idx[u] < bi)) { bk = key; bi = idx[u]; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        int k2 = __shfl_xor(bk, o), i2 = __shfl_xor(bi, o);
        if (k2 > bk || (k2 == bk && i2 < bi)) { bk = k2; bi = i2; }
    }
    if (lane == 0) lang_out[b] = lang_tokens[bi];
}

void launch_lang_detect(const float *logits, int V, const int32_t *lang_tokens, int n, float *probs_out, int32_t *lang_out,
                        int B, hipStream_t st) {
    int ldl = (V + 63) & ~63;
    hipLaunchKernelGGL(lang_detect_kernel, dim3(B), dim3(64), 0, st, logits, ldl, lang_tokens, n, probs_out, lang_out);
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
