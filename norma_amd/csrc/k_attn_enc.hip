// k_attn_enc.hip -- encoder self-attention, flash style (gfx950 MFMA + LDS tiles).
//
// Replaces candle's MultiHeadAttention::qkv_attention for the AudioEncoder blocks (two batched
// matmuls + softmax_last_dim materialising [B,H,1500,1500]; reached from Type::encoder_forward,
// src/models/whisper/model.rs:455-464).  Scores never leave the chip: per 64-key tile
// S^T = K Q^T (v_mfma_f32_32x32x16_f16), online softmax in registers (fp32, one query per lane,
// max/sum finished with one cross-half shuffle), O^T += V^T P^T.
//
// Layout tricks (CDNA4 fragment maps, cdna_hip_programming.md 3):
//  * "swapped" QK^T puts a query on the lane and the tile's keys in the 16 accumulator registers,
//    so softmax is lane-local and the P registers are directly the B operand of the PV MFMA.
//  * the accumulator's k order is row = 16s + 8(j>>2) + 4h + (j&3); storing K rows in LDS with
//    bits 2 and 3 of the row index swapped makes that order "8 consecutive keys per (s,h)", so the
//    V^T fragment is one aligned 16-byte LDS read.  V^T ([b][h][64][1536]) is written directly by
//    the QKV GEMM epilogue (k_gemm.hip), never transposed here.
//  * K and V^T tiles are [64 rows][128 B] images with chunk' = chunk ^ ((row >> 1) & 7):
//    conflict-free ds_read_b128 for both fragment shapes.
// The q/k pre-scaling by dh^-1/4 each and the change of base to exp2 are carried by q itself
// (q arrives multiplied by dh^-1/2 log2(e), GemmParams::seg0_scale).  1500 keys are not a multiple of 64: the last
// tile masks keys >= S; V^T pad is zero.
#include "nh_kernels.h"

// tools/abench links several builds of this file in one process (A/B of kernel variants on one box): the symbol names are macros
#ifndef ENC_ATTN_KERNEL
#define ENC_ATTN_KERNEL enc_attn_kernel
#define ENC_ATTN_LAUNCH launch_enc_attention
#endif
#define KT 64              // keys per tile
#define QW 32              // queries per wave
#ifndef ENC_ATTN_WAVES
#define ENC_ATTN_WAVES 8   // waves per workgroup
#endif
#define QB (QW * ENC_ATTN_WAVES)   // queries per workgroup
#define NTHR (64 * ENC_ATTN_WAVES)
#define NSLOT (NTHR >= 512 ? 1 : 512 / NTHR)   // 16-byte staging slots per thread per tile (a tile is 512 slots; threads >= 512 stage nothing)
#define TILE_B 8192        // bytes per K or V^T tile

__device__ __forceinline__ int swap23(int x) { return (x & ~12) | ((x & 4) << 1) | ((x & 8) >> 1); }

#ifndef ENC_ATTN_MINWAVES
#ifdef ENC_ATTN_PIPE
#define ENC_ATTN_MINWAVES 2   // one 8-wave workgroup per CU: 256 VGPRs for two tiles of scores in flight
#else
#define ENC_ATTN_MINWAVES 4   // two 8-wave workgroups per CU need <= 128 VGPRs
#endif
#endif
__global__ __launch_bounds__(NTHR, (NTHR <= 512 ? ENC_ATTN_MINWAVES : 1)) void ENC_ATTN_KERNEL(const half_t *__restrict__ q, const half_t *__restrict__ k,
                                                         long ld, const half_t *__restrict__ vt,
                                                         half_t *__restrict__ out, long ldo, int S, int H) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE_B];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int b = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * QB + w * QW;
    const int nT = (S + KT - 1) / KT;

    // Q fragments (B operand of S^T = K Q^T): lane (query r, half hh) holds Q[query][16 ks + 8 hh + j]
    half8 qf[4];
    {
        int qrow = q0 + r; if (qrow >= S) qrow = S - 1;
        const half_t *qp = q + ((long)b * S + qrow) * ld + h * NH_DH + 8 * hh;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) qf[ks] = *reinterpret_cast<const half8 *>(qp + 16 * ks);
    }
    // staging: slot s = tid + NTHR i (i < NSLOT): LDS row = s >> 3, chunk' = s & 7
    const half_t *kg[NSLOT]; const half_t *vg[NSLOT];
    int krow_off[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; i++) {
        int row = (tid >> 3) + (NTHR / 8) * i;
        int c = (tid & 7) ^ ((row >> 1) & 7);
        krow_off[i] = swap23(row);  // LDS row `row` holds key key0 + swap23(row)
        kg[i] = k + (long)b * S * ld + h * NH_DH + c * 8;
        vg[i] = vt + ((long)(b * H + h) * NH_DH + row) * NH_SP + c * 8;
    }
    u32x4 rk[NSLOT], rv[NSLOT];
    const bool stager = NTHR <= 512 || tid < 512;
    auto load_k = [&](int t) {
        if (!stager) return;
#pragma unroll
        for (int i = 0; i < NSLOT; i++) {
            int key = t * KT + krow_off[i]; if (key >= S) key = S - 1;
            rk[i] = *reinterpret_cast<const u32x4 *>(kg[i] + (long)key * ld);
        }
    };
    auto load_v = [&](int t) {
        if (!stager) return;
#pragma unroll
        for (int i = 0; i < NSLOT; i++) rv[i] = *reinterpret_cast<const u32x4 *>(vg[i] + t * KT);
    };
    // LDS: K tiles of buffer 0 / 1 at 0 / 2 TILE_B, V^T tiles at TILE_B / 3 TILE_B
    auto store_k = [&](int buf) {
        if (!stager) return;
        u32x4 *lk = reinterpret_cast<u32x4 *>(smem + buf * 2 * TILE_B);
#pragma unroll
        for (int i = 0; i < NSLOT; i++) lk[tid + NTHR * i] = rk[i];
    };
    auto store_v = [&](int buf) {
        if (!stager) return;
        u32x4 *lv = reinterpret_cast<u32x4 *>(smem + buf * 2 * TILE_B + TILE_B);
#pragma unroll
        for (int i = 0; i < NSLOT; i++) lv[tid + NTHR * i] = rv[i];
    };
    auto load_tile = [&](int t) { load_k(t); load_v(t); };
    auto store_tile = [&](int buf) { store_k(buf); store_v(buf); };
    load_tile(0);
    store_tile(0);
#ifdef ENC_ATTN_PIPE
    if (nT > 1) { load_k(1); store_k(1); }
#endif
    __syncthreads();

    // fragment byte offsets inside a tile (before adding the per-step chunk xor)
    const int g = (r >> 1) & 7;                       // same for rows r and r + 32
    const int offK0 = r * 128 + ((hh ^ g) << 4);      // K block 0 (rows 0..31); block 1 at + 4096
    const int offV0 = r * 128 + ((hh ^ g) << 4);      // V^T dh block 0; block 1 at + 4096

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; i++) { o0[i] = 0.f; o1[i] = 0.f; }
    // Softmax bookkeeping (r03).  This kernel is bound by VALU + transcendental issue, not by the matrix pipe (per 64-key tile a
    // lane owns 32 scores: r02 spent 32 v_fma (scale, subtract the maximum) + 32 v_max + 32 v_exp + 32 v_add + 16 v_cvt_pk on
    // them against 16 MFMAs), so the per-score VALU work is cut to exp + add + half a max3 + half a cvt:
    //  * q arrives pre-scaled by dh^-1/2 * log2(e) (QKV GEMM epilogue, one rounding), so a score is already the exp2 argument;
    //  * the subtraction of the reference maximum rides in the MATRIX product: one more 16-deep k-step whose key-side operand
    //    is the constant (1, 1, 0, ...) and whose query-side operand is (hi, lo, 0, ...) with hi + lo = -m_ref split into two
    //    fp16 (residual < 2^-22 |m_ref|): S' = K (c Q)^T - m_ref leaves the accumulator ready for v_exp_f32, for 2 MFMAs per tile;
    //  * the reference moves only when it has to ("defer-max"): m_ref is set to the running maximum on the first tile and
    //    afterwards only when a tile's maximum exceeds it by more than 2^8; in between p = exp2(S') may exceed 1 (<= 256: exact
    //    power-of-two headroom in fp16 P and f32 sums), and the rescale of O and l happens in that rare branch only.
    // Same softmax (the shift cancels in O / l); rounding differs from r02's in the last bit of p.
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const half_t one_h = hh == 0 ? (half_t)1.0f : (half_t)0.0f;
    const half8 kx = {one_h, one_h, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};   // A operand of the extra k-step
    half8 qx = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};   // B operand: (hi, lo, 0 ..) in the hh == 0 lanes
    float m_ref = 0.f, l_run = 0.f;
    const float REBASE = 8.0f;

    f32x16 s0, s1;   // scores of the tile in flight: written by qk_phase, consumed by sv_phase
    // first half of a tile: S'^T = K (cQ)^T - m_ref (10 MFMAs)
    auto qk_phase = [&](const char *tk, f32x16 &s0, f32x16 &s1) {
        {
            // the first k-step takes a literal zero accumulator (an inline constant of the MFMA, not 32 v_mov per tile)
            half8 k0 = *reinterpret_cast<const half8 *>(tk + offK0);
            half8 k1 = *reinterpret_cast<const half8 *>(tk + 4096 + offK0);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[0], zero, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[0], zero, 0, 0, 0);
        }
#pragma unroll
        for (int ks = 1; ks < 4; ks++) {
            // chunk 2 ks + hh: (2ks + hh) ^ g == (hh ^ g) ^ (2 ks)  -> byte offset ^ (ks << 5)
            half8 k0 = *reinterpret_cast<const half8 *>(tk + (offK0 ^ (ks << 5)));
            half8 k1 = *reinterpret_cast<const half8 *>(tk + 4096 + (offK0 ^ (ks << 5)));
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0, qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1, qf[ks], s1, 0, 0, 0);
        }
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kx, qx, s0, 0, 0, 0);   // - m_ref for every key of the tile
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kx, qx, s1, 0, 0, 0);
    };
    // second half: mask, maximum / rare rebase, exp2, row sum, P -> fp16, O^T += V^T P^T (8 MFMAs)
    auto sm_head = [&](int t) {
        if (t * KT + KT > S) {  // last, partial tile: mask keys >= S
#pragma unroll
            for (int i = 0; i < 16; i++) {
                int rho = (i & 3) + 8 * (i >> 2) + 4 * hh;
                int key0 = t * KT + swap23(rho), key1 = t * KT + 32 + swap23(rho);
                if (key0 >= S) s0[i] = -INFINITY;
                if (key1 >= S) s1[i] = -INFINITY;
            }
        }
        // tile maximum of S' with three-input maxima (16 instructions for 32 values), both key halves of the query
        float mx = fmaxf(fmaxf(s0[0], s0[1]), s0[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) mx = fmaxf(fmaxf(mx, s0[i]), s0[i + 1]);
        mx = fmaxf(fmaxf(mx, s0[15]), s1[0]);
#pragma unroll
        for (int i = 1; i < 15; i += 2) mx = fmaxf(fmaxf(mx, s1[i]), s1[i + 1]);
        mx = fmaxf(mx, s1[15]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const bool reb = (t == 0) || (mx > REBASE);
        if (__builtin_amdgcn_ballot_w64(reb) != 0) {   // rare after the first tile: move the reference of the queries that need it
            const float dlt = reb ? mx : 0.f;          // tile 0: m_ref = the tile's maximum (any sign); later: only upwards
            const float alpha = __builtin_amdgcn_exp2f(-dlt);
#pragma unroll
            for (int i = 0; i < 16; i++) { o0[i] *= alpha; o1[i] *= alpha; s0[i] -= dlt; s1[i] -= dlt; }
            l_run *= alpha;
            m_ref += dlt;
            const float nm = fminf(fmaxf(-m_ref, -60000.f), 60000.f);
            const half_t hi = (half_t)nm;
            const half_t lo = (half_t)(nm - (float)hi);
            qx[0] = hh == 0 ? hi : (half_t)0.f;
            qx[1] = hh == 0 ? lo : (half_t)0.f;
        }
    };
    auto sm_body = [&](const char *tv) {
        float ps = 0.f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            s0[i] = __builtin_amdgcn_exp2f(s0[i]); s1[i] = __builtin_amdgcn_exp2f(s1[i]);
            ps += s0[i]; ps += s1[i];
        }
        l_run += ps;
        // P^T fragments (B operand): k-step (blk, s2) takes accumulator registers 8 s2 .. 8 s2 + 7
        half8 p00, p01, p10, p11;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            p00[j] = (half_t)s0[j]; p01[j] = (half_t)s0[8 + j];
            p10[j] = (half_t)s1[j]; p11[j] = (half_t)s1[8 + j];
        }
        // O^T[dh][q] += V^T[dh][keys] P^T[keys][q]; V^T chunk = 4 blk + 2 s2 + hh -> xor ((4blk+2s2) << 4)
        {
            half8 v;
            v = *reinterpret_cast<const half8 *>(tv + (offV0 ^ (0 << 4)));          o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p00, o0, 0, 0, 0);
            v = *reinterpret_cast<const half8 *>(tv + 4096 + (offV0 ^ (0 << 4)));   o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p00, o1, 0, 0, 0);
            v = *reinterpret_cast<const half8 *>(tv + (offV0 ^ (2 << 4)));          o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p01, o0, 0, 0, 0);
            v = *reinterpret_cast<const half8 *>(tv + 4096 + (offV0 ^ (2 << 4)));   o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p01, o1, 0, 0, 0);
            v = *reinterpret_cast<const half8 *>(tv + (offV0 ^ (4 << 4)));          o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p10, o0, 0, 0, 0);
            v = *reinterpret_cast<const half8 *>(tv + 4096 + (offV0 ^ (4 << 4)));   o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p10, o1, 0, 0, 0);
            v = *reinterpret_cast<const half8 *>(tv + (offV0 ^ (6 << 4)));          o0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p11, o0, 0, 0, 0);
            v = *reinterpret_cast<const half8 *>(tv + 4096 + (offV0 ^ (6 << 4)));   o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p11, o1, 0, 0, 0);
        }
    };
    // (r03, tools/abench, one box: a schedule in which waves 4-7 run half a tile behind waves 0-3 -- two barriers per tile, one
    // half multiplying K Q^T while the other does softmax + P V -- measured 640-670 us against 446-454 for this loop; four
    // independent max / sum chains, v_dot2_f32_f16 row sums and s_setprio around the MFMA clusters all within +-1.5 %.  Later in
    // r03: workgroups of 4 and 16 waves 856 / 501 us; the second workgroup of a CU started 1000-3000 cycles late: no change; the
    // software-pipelined form below (ENC_ATTN_PIPE: one 8-wave workgroup per CU, 161 VGPRs, K Q^T of tile t + 1 issued between
    // the exponentials of tile t) 577-609 us: two waves per SIMD do not cover the dependent-MFMA and LDS latencies that four do.)
#ifndef ENC_ATTN_PIPE
    int cur = 0;
    for (int t = 0; t < nT; t++) {
        const bool more = (t + 1 < nT);
        if (more) load_tile(t + 1);
        qk_phase(smem + cur * 2 * TILE_B, s0, s1);
        sm_head(t);
        sm_body(smem + cur * 2 * TILE_B + TILE_B);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
#else
    // Software-pipelined form: the matrix pipe and the vector ALU of a SIMD work at the same time only when ONE instruction
    // stream offers both (waves do not fill each other's gaps here: 576 MFMA + 540 VALU cycles per wave-tile took 1090), so
    // the K Q^T products of tile t + 1 are issued between the exponentials of tile t.  At the top of iteration t the LDS holds
    // K(t + 1) and V^T(t); K(t + 2) and V^T(t + 1) are fetched during the iteration into the buffers of K(t) and V^T(t - 1).
    f32x16 n0, n1;
    qk_phase(smem, s0, s1);
    for (int t = 0; t < nT; t++) {
        if (t + 2 < nT) load_k(t + 2);
        if (t + 1 < nT) load_v(t + 1);
        sm_head(t);                                                   // mask, maximum, (rare) rebase: moves qx before it is used below
        qk_phase(smem + ((t + 1) & 1) * 2 * TILE_B, n0, n1);         // tile t + 1 (after the last tile: a stale buffer, result unused)
        sm_body(smem + (t & 1) * 2 * TILE_B + TILE_B);               // exp2, row sums, P, O^T += V^T P^T of tile t
#pragma unroll
        for (int i = 0; i < 18; i++) {                                // one MFMA, then its share of the 100-odd vector instructions
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, ENC_ATTN_PIPE, 0);
        }
        if (t + 2 < nT) store_k(t & 1);
        if (t + 1 < nT) store_v((t + 1) & 1);
        __syncthreads();
        s0 = n0; s1 = n1;
    }
#endif
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    const int qrow = q0 + r;
    if (qrow < S) {
        half_t *op = out + ((long)b * S + qrow) * ldo + h * NH_DH + 4 * hh;
#pragma unroll
        for (int gq = 0; gq < 4; gq++) {
            half4 a = {(half_t)(o0[4 * gq] * inv), (half_t)(o0[4 * gq + 1] * inv), (half_t)(o0[4 * gq + 2] * inv),
                       (half_t)(o0[4 * gq + 3] * inv)};
            half4 c = {(half_t)(o1[4 * gq] * inv), (half_t)(o1[4 * gq + 1] * inv), (half_t)(o1[4 * gq + 2] * inv),
                       (half_t)(o1[4 * gq + 3] * inv)};
            *reinterpret_cast<half4 *>(op + 8 * gq) = a;
            *reinterpret_cast<half4 *>(op + 32 + 8 * gq) = c;
        }
    }
}

void ENC_ATTN_LAUNCH(const half_t *q, const half_t *k, long ld, const half_t *vt, half_t *out, long ldo,
                          int B, int S, int H, hipStream_t st) {
    dim3 grid((S + QB - 1) / QB, H, B);
    hipLaunchKernelGGL(ENC_ATTN_KERNEL, grid, dim3(NTHR), 0, st, q, k, ld, vt, out, ldo, S, H);
}
