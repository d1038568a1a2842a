// k_gemm.hip -- MFMA GEMM for the encoder and the cross-K/V projection (gfx950).
//
//   C[M][N] = A[M][K] . W[N][K]^T (+ bias) with fused epilogues; fp16 operands, fp32 accumulate.
//
// Replaces, on the reference path, candle's Linear/Conv1d matmuls inside AudioEncoder::forward and
// the cross-attention key/value projections (reached through Type::encoder_forward /
// Type::decoder_forward, src/models/whisper/model.rs:455-476).  Conv1d k=3 is expressed as a GEMM
// whose A rows overlap (lda = stride * channels, K = 3 * channels) -- no im2col buffer.
//
// Tile: 128 (M) x 128 (N) x 64 (K) per 256-thread workgroup (4 waves as 2 x 2, 64 x 64 per wave,
// v_mfma_f32_16x16x32_f16).  Both operands are K-contiguous, so each lane's fragment is one 16-byte
// LDS read.  LDS image: [128 rows][8 chunks of 16 B], chunk' = chunk ^ ((row >> 1) & 7) (conflict-free
// ds_read_b128, MI355X LDS banking), double buffered (64 KiB).  Global->LDS is global_load_lds_dwordx4
// (LDS-DMA, no VGPR round trip, no ds_write): the loads of tile t+1 are issued before the MFMAs of tile t.
//
// Orientation: by default the weight rows are the MFMA "A" operand, so a lane's 4 accumulator
// registers are 4 consecutive output COLUMNS of one row -> 8-byte fp16 / 16-byte f32 row-major
// stores.  The V^T segment uses the other orientation (4 consecutive rows of one column).
#include <stdlib.h>

#include "nh_kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand tile

__device__ __forceinline__ float gelu_tanh_f(float v) {
    // 0.5 v (1 + tanh(u)) == v / (1 + exp(-2u)),  u = sqrt(2/pi) v (1 + 0.044715 v^2)
    float u = 0.7978845608028654f * v * (1.0f + 0.044715f * v * v);
    return v / (1.0f + __expf(-2.0f * u));
}

__device__ __forceinline__ const half_t *a_row_ptr(const GemmParams &p, int m) {
    if (m >= p.M) m = p.M - 1;  // clamp: tail rows are loaded but never stored
    int b = m / p.a_rpb, r = m - b * p.a_rpb;
    return p.A + (long)b * p.a_bstride + (long)r * p.lda;
}

#ifndef NH_GEMM_GLDS
#define NH_GEMM_GLDS 1  // 1: global_load_lds_dwordx4 straight into LDS; 0: stage through registers + ds_write_b128
#endif

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

template <bool SWAP>
__device__ __forceinline__ void gemm_mainloop(const GemmParams &p, char *smem, int m0, int n0,
                                              f32x4 (&acc)[4][4]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    // staging map: slot s = tid + 256 i (i = 0..3): row = s >> 3, chunk' = s & 7 (tid & 7 for all i).
    // One wave-instruction covers 64 consecutive slots = 1 KiB of the LDS image, in lane order -- exactly
    // the (wave-uniform base + lane * 16) destination of global_load_lds; the swizzle lives in the SOURCE.
    const int cq = tid & 7;
    const half_t *ag[4], *wg[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        int row = (tid >> 3) + 32 * i;
        int c = cq ^ ((row >> 1) & 7);
        ag[i] = a_row_ptr(p, m0 + row) + c * 8;
        int n = n0 + row;  // N is a multiple of BN
        wg[i] = p.W + (long)n * p.K + c * 8;
    }
    const int nt = p.K / BK;
    // fragment read offsets (bytes within a tile): row r, chunk 4*ks + (lane>>4), swizzled
    const int fr = lane & 15, fq = lane >> 4;
    int offA[4], offB[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int rowA = wm * 64 + 16 * j + fr;  // activation rows (m)
        int rowB = wn * 64 + 16 * j + fr;  // weight rows (n)
        offA[j] = rowA * 128 + ((fq ^ ((rowA >> 1) & 7)) << 4);
        offB[j] = rowB * 128 + ((fq ^ ((rowB >> 1) & 7)) << 4);
    }
#if NH_GEMM_GLDS
    const int wbase = __builtin_amdgcn_readfirstlane(w) * 1024;  // this wave's 1 KiB run inside each 4 KiB group
    auto stage = [&](int buf, int t) {
        char *la = smem + buf * 2 * TILE_BYTES + wbase, *lb = la + TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(ag[i] + (long)t * BK), (lds_void *)(la + 4096 * i), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(wg[i] + (long)t * BK), (lds_void *)(lb + 4096 * i), 16, 0, 0);
        }
    };
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#else
    u32x4 ra[4], rb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ra[i] = *reinterpret_cast<const u32x4 *>(ag[i]);
        rb[i] = *reinterpret_cast<const u32x4 *>(wg[i]);
    }
    {
        u32x4 *la = reinterpret_cast<u32x4 *>(smem), *lb = reinterpret_cast<u32x4 *>(smem + TILE_BYTES);
#pragma unroll
        for (int i = 0; i < 4; i++) { la[tid + 256 * i] = ra[i]; lb[tid + 256 * i] = rb[i]; }
    }
    __syncthreads();
#endif
    int cur = 0;
    for (int t = 0; t < nt; t++) {
        const bool more = (t + 1 < nt);
#if NH_GEMM_GLDS
        if (more) stage(cur ^ 1, t + 1);  // all waves left buf[cur^1] at the barrier that ended iteration t-1
#else
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ra[i] = *reinterpret_cast<const u32x4 *>(ag[i] + (long)(t + 1) * BK);
                rb[i] = *reinterpret_cast<const u32x4 *>(wg[i] + (long)(t + 1) * BK);
            }
        }
#endif
        const char *ta = smem + cur * 2 * TILE_BYTES, *tb = ta + TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ks++) {
            half8 fa[4], fb[4];
            // chunk index 4*ks + fq; the xor only touches the low 3 bits, and (4*ks) flips bit 2:
            // (4ks + fq) ^ g == (fq ^ g) ^ (4ks) because fq < 4 -> byte offset ^ (ks << 6)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                fa[j] = *reinterpret_cast<const half8 *>(ta + (offA[j] ^ (ks << 6)));
                fb[j] = *reinterpret_cast<const half8 *>(tb + (offB[j] ^ (ks << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    if (SWAP)  // D[row = n (weights)][col = m (activations)]
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[i], fa[j], acc[i][j], 0, 0, 0);
                    else       // D[row = m][col = n]
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[j], fb[i], acc[i][j], 0, 0, 0);
                }
        }
#if NH_GEMM_GLDS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA for tile t+1 has landed
#else
        if (more) {
            u32x4 *la = reinterpret_cast<u32x4 *>(smem + (cur ^ 1) * 2 * TILE_BYTES);
            u32x4 *lb = reinterpret_cast<u32x4 *>(smem + (cur ^ 1) * 2 * TILE_BYTES + TILE_BYTES);
#pragma unroll
            for (int i = 0; i < 4; i++) { la[tid + 256 * i] = ra[i]; lb[tid + 256 * i] = rb[i]; }
        }
#endif
        __syncthreads();
        cur ^= 1;
    }
}

__device__ __forceinline__ long out_row(const GemmParams &p, int m) {
    int b = m / p.o_rpb, r = m - b * p.o_rpb;
    return (long)b * p.o_bstride + r + p.o_off;
}

__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];
    // XCD-aware tile order: blocks b and b+8 share an XCD (L2); give each XCD a contiguous run of
    // tiles so the N-tiles of one M-panel (same A rows) are neighbours in one L2.
    const int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    const int nwg = ntn * ntm;
    int bid = blockIdx.x;
    {
        int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = bid / ntn, tn = bid - tm * ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int fr = lane & 15, fq = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int seg = (p.epi == EPI_F16 || p.epi == EPI_GELU_F16) ? n0 / p.seg_n : 0;
    const bool vt = (p.epi == EPI_F16 && seg == p.vt_seg);
    if (vt) gemm_mainloop<false>(p, smem, m0, n0, acc);
    else gemm_mainloop<true>(p, smem, m0, n0, acc);

    if (vt) {
        // lane holds rows m = mb + 4 fq + r (r = 0..3) of column n = nb + fr
        half_t *dst = reinterpret_cast<half_t *>(seg == 0 ? p.out[0] : seg == 1 ? p.out[1] : p.out[2]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int n = n0 + wn * 64 + 16 * i + fr;
            float bv = p.bias ? p.bias[n] : 0.f;
            int nl = n - seg * p.seg_n, h = nl >> 6, dh = nl & 63;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int m = m0 + wm * 64 + 16 * j + 4 * fq;
                if (m >= p.M) continue;
                int b = m / p.S, s = m - b * p.S;
                half_t *row = dst + ((long)(b * p.H + h) * NH_DH + dh) * NH_SP;
                if (s + 3 < p.S && m + 3 < p.M) {
                    half4 v = {(half_t)(acc[i][j][0] + bv), (half_t)(acc[i][j][1] + bv),
                               (half_t)(acc[i][j][2] + bv), (half_t)(acc[i][j][3] + bv)};
                    *reinterpret_cast<half4 *>(row + s) = v;
                } else {
                    for (int r = 0; r < 4; r++) {
                        int mm = m + r;
                        if (mm >= p.M) break;
                        int bb = mm / p.S, ss = mm - bb * p.S;
                        dst[((long)(bb * p.H + h) * NH_DH + dh) * NH_SP + ss] = (half_t)(acc[i][j][r] + bv);
                    }
                }
            }
        }
        return;
    }
    half_t *obase = reinterpret_cast<half_t *>(seg == 0 ? p.out[0] : seg == 1 ? p.out[1] : p.out[2]);
    // SWAP orientation: lane holds columns n = nb + 4 fq + r (r = 0..3) of row m = mb + fr
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int n = n0 + wn * 64 + 16 * i + 4 * fq;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int m = m0 + wm * 64 + 16 * j + fr;
            if (m >= p.M) continue;
            f32x4 v = acc[i][j] + bv;
            if (p.epi == EPI_F16 || p.epi == EPI_GELU_F16) {
                if (p.epi == EPI_GELU_F16) {
                    v[0] = gelu_tanh_f(v[0]); v[1] = gelu_tanh_f(v[1]);
                    v[2] = gelu_tanh_f(v[2]); v[3] = gelu_tanh_f(v[3]);
                }
                half_t *dst = obase + out_row(p, m) * p.ldo + (n - seg * p.seg_n);
                half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4 *>(dst) = hv;
            } else if (p.epi == EPI_RESID_F32) {
                float *dst = reinterpret_cast<float *>(p.out[0]) + (long)m * p.ldo + n;
                f32x4 x = *reinterpret_cast<const f32x4 *>(dst);
                *reinterpret_cast<f32x4 *>(dst) = x + v;
            } else {  // EPI_CONV2_F32
                int s = m % p.S;
                f32x4 pe = *reinterpret_cast<const f32x4 *>(p.pos + (long)s * p.N + n);
                f32x4 o = {gelu_tanh_f(v[0]) + pe[0], gelu_tanh_f(v[1]) + pe[1], gelu_tanh_f(v[2]) + pe[2],
                           gelu_tanh_f(v[3]) + pe[3]};
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out[0]) + (long)m * p.ldo + n) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 256 x 256 tile kernel for the large encoder shapes (N % 256 == 0).
//
// 512 threads = 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave (128 accumulator VGPRs).  K is
// consumed in 32-deep stages (one v_mfma_f32_16x16x32_f16 k-step); a 4-slot LDS ring (4 x 32 KiB) is
// filled by global_load_lds_dwordx4 THREE stages ahead of the MFMAs, so a tile has ~3 stages of matrix
// work (~3000 cycles at 2 waves/SIMD) to arrive from L2/HBM.  Synchronisation per stage:
//     s_waitcnt vmcnt(8)   this wave's DMA for stage s has landed (stages s+1, s+2 stay in flight)
//     s_barrier            => every wave's DMA for stage s has landed, and every wave is done reading
//                             stage s-1, whose slot the next DMA (stage s+3) overwrites
// (counted vmcnt + raw s_barrier: __syncthreads() would drain the DMA queue every stage).
// LDS image per stage and operand: [256 rows][4 chunks of 16 B], chunk' = chunk ^ (-(row >> 2) & 3):
// conflict-free ds_read_b128 for the 16x16x32 fragment shape on 64-byte rows; the permutation is applied
// to the per-lane global SOURCE address, the LDS destination of LDS-DMA stays lane-linear.
// ---------------------------------------------------------------------------------------------------
#define G2_BM 256
#define G2_BN 256
#define G2_BK 32
#define G2_STAGE_BYTES 32768  // A 16 KiB + B 16 KiB
#define G2_NSTAGE 4
#define G2_GM 4
#ifndef G2_PAIR
#define G2_PAIR 0
#endif
#ifndef G2_PIPE
#define G2_PIPE 0
#endif
#ifndef G2_ABL
#define G2_ABL 0
#endif
#ifndef G2_SPREAD
#define G2_SPREAD 0
#endif
#ifndef G2_STAGE_AFTER_READS
#define G2_STAGE_AFTER_READS 1
#endif
// fp16 epilogues go through LDS so that every global store instruction writes whole 128/256-byte rows:
// per wave a [128][64] image with 136-byte rows (row-major outputs) or a [64][128] image with 264-byte rows (V^T)
#define G2_EPI_ROW 136
#define G2_EPI_ROW_T 264
#define G2_EPI_WAVE 17408  // max(128 * 136, 64 * 264)
#define G2_EPI_BYTES (8 * G2_EPI_WAVE)

// WM = 2: 256 x 256 tile, 8 waves (2 x 4), 4-slot ring of 32 KiB, one workgroup per CU.
// WM = 1: 128 x 256 tile, 4 waves (1 x 4), 3-slot ring of 24 KiB, TWO independent workgroups per CU: while one waits at
//         its barrier / for its DMA, the other one's MFMAs keep the matrix pipes busy (same 128 x 64 tile per wave).
template <bool SWAP, int WM>
__device__ __forceinline__ void gemm256_mainloop(const GemmParams &p, char *smem, int m0, int n0, f32x4 (&acc)[8][4]) {
    constexpr int NT = 256 * WM;                     // threads
    constexpr int ABYTES = 8192 * WM;                // A image per stage: (128 WM) rows x 64 B
    constexpr int STAGE_BYTES = ABYTES + 16384;      // + B image: 256 rows x 64 B
    constexpr int NSTAGE = WM == 2 ? 4 : 3;
    constexpr int NB = 4 / WM;                       // LDS-DMA instructions per thread for the B image (A: always 2)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = WM == 2 ? (w >> 2) : 0, wn = w & 3;
    const int fr = lane & 15, fq = lane >> 4;
    // staging: slot s = tid + NT i: row = s >> 2, chunk' = s & 3
    const half_t *ag[2], *wg[NB];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int row = (tid >> 2) + (NT / 4) * i;
        int c = (tid & 3) ^ ((-(row >> 2)) & 3);
        ag[i] = a_row_ptr(p, m0 + row) + c * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        int row = (tid >> 2) + (NT / 4) * i;
        int c = (tid & 3) ^ ((-(row >> 2)) & 3);
        wg[i] = p.W + (long)(n0 + row) * p.K + c * 8;
    }
    const int ns = p.K / G2_BK;
    const int wbase = __builtin_amdgcn_readfirstlane(w) * 1024;
    auto slot_of = [&](int s) { return WM == 2 ? (s & 3) : (s % 3); };
    auto stage = [&](int s) {
#if G2_ABL & 1  // ablation (tools/gbench only): no LDS-DMA -- s = 0 (stage once, stay in cache) keeps results meaningless
        if (s >= NSTAGE) return;
#endif
        char *la = smem + slot_of(s) * STAGE_BYTES + wbase, *lb = la + ABYTES;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((gbl_void *)(ag[i] + (long)s * G2_BK), (lds_void *)(la + NT * 16 * i), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NB; i++)
            __builtin_amdgcn_global_load_lds((gbl_void *)(wg[i] + (long)s * G2_BK), (lds_void *)(lb + NT * 16 * i), 16, 0, 0);
    };
#if G2_SPREAD
    auto stage_piece = [&](int s, int piece) {  // piece 0..3: A rows 0-127, B rows 0-127, A rows 128-255, B rows 128-255
        char *la = smem + (s & (G2_NSTAGE - 1)) * G2_STAGE_BYTES + wbase, *lb = la + 16384;
        const int i = piece >> 1;
        if (piece & 1) __builtin_amdgcn_global_load_lds((gbl_void *)(wg[i] + (long)s * G2_BK), (lds_void *)(lb + 8192 * i), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds((gbl_void *)(ag[i] + (long)s * G2_BK), (lds_void *)(la + 8192 * i), 16, 0, 0);
    };
#endif
    // fragment offsets inside a stage: row r, chunk fq swizzled; A rows wm*128 + 16 i + fr, B rows wn*64 + 16 j + fr
    int offA[8], offB[4];
    const int sw = (fq ^ ((-(fr >> 2)) & 3)) << 4;
#pragma unroll
    for (int i = 0; i < 8; i++) offA[i] = (wm * 128 + 16 * i + fr) * 64 + sw;
#pragma unroll
    for (int j = 0; j < 4; j++) offB[j] = ABYTES + (wn * 64 + 16 * j + fr) * 64 + sw;

#if G2_PAIR
    auto compute = [&](int s) {
        const char *ts = smem + (s & (G2_NSTAGE - 1)) * G2_STAGE_BYTES;
        half8 fa[8], fb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const half8 *>(ts + offB[j]);
#pragma unroll
        for (int i = 0; i < 8; i++) fa[i] = *reinterpret_cast<const half8 *>(ts + offA[i]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };
#endif
#if G2_PIPE == 2
    // Partial software pipelining of the fragments: the 4 B fragments and the first 2 A fragments of stage s+1 are
    // read from LDS during stage s (24 extra VGPRs), so the first 8 MFMAs after a barrier need no LDS data; the
    // other 6 A fragments are read behind them.  (PMC: ~30 % of wave time was parked in s_waitcnt/s_barrier; the
    // fully double-buffered variant G2_PIPE=1 needs 48 extra VGPRs and spills.)
    constexpr int NPRE = 2;
    auto mfma_row = [&](int i, const half8 &a, const half8 (&fb)[4]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], a, acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, fb[j], acc[i][j], 0, 0, 0);
        }
    };
    half8 pb0[4], pa0[NPRE], pb1[4], pa1[NPRE];
    auto pre_read = [&](int s, half8 (&pa)[NPRE], half8 (&pb)[4]) {
        const char *ts = smem + (s & (G2_NSTAGE - 1)) * G2_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; j++) pb[j] = *reinterpret_cast<const half8 *>(ts + offB[j]);
#pragma unroll
        for (int i = 0; i < NPRE; i++) pa[i] = *reinterpret_cast<const half8 *>(ts + offA[i]);
    };
    stage(0);
    if (ns > 1) stage(1);
    if (ns > 2) stage(2);
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    pre_read(0, pa0, pb0);
    auto body = [&](int s, half8 (&pa)[NPRE], half8 (&pb)[4], half8 (&na)[NPRE], half8 (&nb)[4]) {
        const int rem = ns - 1 - s;
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (rem >= 1) __builtin_amdgcn_s_barrier();
        const char *ts = smem + (s & (G2_NSTAGE - 1)) * G2_STAGE_BYTES;
        half8 fa[8 - NPRE];
#pragma unroll
        for (int i = NPRE; i < 8; i++) fa[i - NPRE] = *reinterpret_cast<const half8 *>(ts + offA[i]);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 3 < ns) stage(s + 3);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < NPRE; i++) mfma_row(i, pa[i], pb);
        if (rem >= 1) pre_read(s + 1, na, nb);
#pragma unroll
        for (int i = NPRE; i < 8; i++) mfma_row(i, fa[i - NPRE], pb);
        __builtin_amdgcn_s_setprio(0);
    };
    for (int s = 0; s < ns; s += 2) {  // ns is even (K % 64 == 0)
        body(s, pa0, pb0, pa1, pb1);
        body(s + 1, pa1, pb1, pa0, pb0);
    }
#elif G2_PIPE
    // Software-pipelined fragments: the MFMAs of stage s run on registers that were read from LDS during stage s-1,
    // while the fragments of stage s+1 are being read -- no LDS latency bubble after the barrier (PMC: ~30 % of
    // wave time was parked in s_waitcnt/s_barrier).  Two named register sets, loop unrolled by two (static indexing).
    auto read_frags = [&](int s, half8 (&fa)[8], half8 (&fb)[4]) {
        const char *ts = smem + (s & (G2_NSTAGE - 1)) * G2_STAGE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const half8 *>(ts + offB[j]);
#pragma unroll
        for (int i = 0; i < 8; i++) fa[i] = *reinterpret_cast<const half8 *>(ts + offA[i]);
    };
    auto mfmas = [&](const half8 (&fa)[8], const half8 (&fb)[4]) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
    };
    half8 fa0[8], fb0[4], fa1[8], fb1[4];
    stage(0);
    if (ns > 1) stage(1);
    if (ns > 2) stage(2);
    if (ns > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ns > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, fa0, fb0);
    auto body = [&](int s, half8 (&fa)[8], half8 (&fb)[4], half8 (&na)[8], half8 (&nb)[4]) {
        const int rem = ns - 1 - s;  // stages after s; s+1 must have landed, s+2 may stay in flight
        if (rem >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (rem == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (rem >= 1) __builtin_amdgcn_s_barrier();  // every wave's stage s+1 landed; every wave finished READING stage s-1
        if (s + 3 < ns) stage(s + 3);                 // overwrites the slot of stage s-1
        if (rem >= 1) read_frags(s + 1, na, nb);
        mfmas(fa, fb);
    };
    for (int s = 0; s < ns; s += 2) {  // ns is even (K % 64 == 0)
        body(s, fa0, fb0, fa1, fb1);
        body(s + 1, fa1, fb1, fa0, fb0);
    }
#elif G2_PAIR
    // two 32-deep stages per barrier: classic double buffering with 64-deep K-tiles built from the 4 ring slots
    stage(0); stage(1);
    for (int s = 0; s < ns; s += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (s + 2 < ns) { stage(s + 2); stage(s + 3); }
        compute(s);
        compute(s + 1);
    }
#else
    stage(0);
    if (ns > 1) stage(1);
    if (WM == 2 && ns > 2) stage(2);
    half8 fa[8], fb[4];
    for (int s = 0; s < ns; s++) {
        const int rem = ns - 1 - s;  // stages issued beyond s; WM=2 keeps 2 of them in flight (4 DMA each), WM=1 one (6 DMA)
        if (WM == 2) {
            if (rem >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (rem >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#if !(G2_ABL & 4)
        __builtin_amdgcn_s_barrier();
#endif
        const char *ts = smem + slot_of(s) * STAGE_BYTES;
#if G2_ABL & 2  // ablation: no LDS fragment reads after the first stage
        if (s == 0) {
#endif
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const half8 *>(ts + offB[j]);
#pragma unroll
        for (int i = 0; i < 8; i++) fa[i] = *reinterpret_cast<const half8 *>(ts + offA[i]);
#if G2_ABL & 2
        }
#endif
#if G2_SPREAD
        // the four LDS-DMA issues of stage s+3 are spread over the MFMA rows (one per 8 MFMAs) instead of a burst
        // that stalls both waves of a SIMD at the same time right after the barrier
        const bool more = s + 3 < ns;
#pragma unroll
        for (int i = 0; i < 8; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
            if (i & 1) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_piece(s + 3, i >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#else
#if G2_STAGE_AFTER_READS
        __builtin_amdgcn_sched_barrier(0);
#endif
        if (s + NSTAGE - 1 < ns) stage(s + NSTAGE - 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        __builtin_amdgcn_s_setprio(0);
#endif
    }
#endif
}

// ---------------------------------------------------------------------------------------------------
// Ping-pong main loop of the 256 x 256 tile (K % 128 == 0): the 8-phase schedule of the CDNA4 playbook
// (cdna_hip_programming.md, "The 256^2 8-phase template"), restated for this kernel's operand layout.
//
//  * BK = 64 K-tiles in two 64 KiB LDS buffers, each cut into four 16 KiB half-tiles A0 A1 B0 B1.  Half-tile Ah holds,
//    for BOTH wave rows, the h-th 64 rows of the wave's 128 (LDS row l <-> tile row (l >> 6) * 128 + 64 h + (l & 63));
//    Bh the h-th 32 columns of each wave column's 64 (l <-> (l >> 5) * 64 + 32 h + (l & 31)).  The permutation lives
//    in the per-lane SOURCE address of the LDS-DMA, so the wave -> output mapping (and every epilogue) is unchanged.
//  * a K-tile is four phases = the four 64 x 32 quadrants of the wave's 128 x 64 output, 16 MFMAs each:
//        q0 (m0,n0): reads A0 (8 ds_read_b128) + B0 (4)    q1 (m0,n1): reads B1 (4)
//        q2 (m1,n1): reads A1 (8)                          q3 (m1,n0): no reads (B0 fragments are kept)
//    so the half-tiles of a buffer die one after the other (A0, B0 after q0; B1 after q1; A1 after q2) and each is
//    re-staged for the K-tile after next two or three phases after its last read, FIVE phases before its first use:
//        q2: A0(t+2)   q3: B0(t+2)   q0: B1(t+1)   q1: A1(t+1)         (one half-tile = 2 LDS-DMA per thread per phase)
//  * the two waves of a SIMD (wave rows wr = 0 / 1) run half a phase apart (one extra s_barrier for wr = 1 up front):
//        phase = [ds_read, LDS-DMA issue, counted vmcnt] s_barrier [lgkmcnt(0), 16 MFMA] s_barrier
//    so in every barrier interval one wave of each SIMD issues MFMAs while the other one reads LDS: the matrix pipe is
//    fed by one of them at all times instead of both reading, then both multiplying.
//  * RAW: the vmcnt that retires a half-tile sits before the FIRST barrier of the phase before its first read (then the
//    lagging group's part has landed too before the leading group reads); WAR: >= 2 phases between last read and re-stage.
//    In steady state four half-tiles stay in flight across every wait: vmcnt(8), never 0 inside the loop.
// ---------------------------------------------------------------------------------------------------
#define PP_BUF 65536
#define PP_HT 16384

template <bool SWAP>
__device__ __forceinline__ void gemm256_pingpong(const GemmParams &p, char *smem, int m0, int n0, f32x4 (&acc)[8][4]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 2, wc = w & 3;
    const int fr = lane & 15, fq = lane >> 4;
    // staging: half-tile slot s = tid + 512 i (i = 0, 1): LDS row l = s >> 3 = (tid >> 3) + 64 i, chunk' = tid & 7,
    // source chunk = chunk' ^ ((l >> 1) & 7) (64 i leaves the swizzle unchanged)
    const int lrow = tid >> 3;
    const int csrc = ((tid & 7) ^ ((lrow >> 1) & 7)) * 8;
    // 32-bit element offsets from the (scalar) operand bases: 8 VGPRs instead of 16 for the eight source rows
    unsigned ag[2][2], wg[2][2];  // [h][i]
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            ag[h][i] = (unsigned)(a_row_ptr(p, m0 + i * 128 + h * 64 + lrow) - p.A) + csrc;
            const int l = lrow + 64 * i;
            wg[h][i] = (unsigned)(n0 + (l >> 5) * 64 + h * 32 + (l & 31)) * (unsigned)p.K + csrc;
        }
    const int nt = p.K >> 6;
    char *const wbase = smem + __builtin_amdgcn_readfirstlane(w) * 1024;
    // which: 0 A0, 1 B0, 2 B1, 3 A1 (the issue order of a K-tile)
    auto stage = [&](int t, int which) {
        const int h = (which >> 1) & 1;            // A0 0, B0 0, B1 1, A1 1
        const bool isB = which == 1 || which == 2;
        char *dst = wbase + (t & 1) * PP_BUF + (isB ? 2 * PP_HT : 0) + h * PP_HT;
        const unsigned ko = (unsigned)t * 64u;
        if (isB) {
            __builtin_amdgcn_global_load_lds((gbl_void *)(p.W + (size_t)(wg[h][0] + ko)), (lds_void *)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(p.W + (size_t)(wg[h][1] + ko)), (lds_void *)(dst + 8192), 16, 0, 0);
        } else {
            __builtin_amdgcn_global_load_lds((gbl_void *)(p.A + (size_t)(ag[h][0] + ko)), (lds_void *)dst, 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gbl_void *)(p.A + (size_t)(ag[h][1] + ko)), (lds_void *)(dst + 8192), 16, 0, 0);
        }
    };
    // fragment read offsets inside a half-tile: row r, chunk 4 ks + fq, swizzle (r >> 1) & 7 == (fr >> 1) for every fragment
    // (the ks = 1 chunk is the ks = 0 address with bit 6 flipped: one per-lane base per k-step, everything else is an
    // immediate offset of the ds_read)
    const int offA0 = (wr * 64 + fr) * 128 + ((fq ^ (fr >> 1)) << 4);  // + 2048 i (i = 0..3)
    const int offB0 = (wc * 32 + fr) * 128 + ((fq ^ (fr >> 1)) << 4);  // + 2048 j (j = 0..1)
    const char *const pa[2] = {smem + offA0, smem + (offA0 ^ 64)};
    const char *const pb[2] = {smem + offB0, smem + (offB0 ^ 64)};
    half8 fa[4][2], fb[4][2];  // fa[i][ks]: current m-half; fb[j][ks]: j < 2 n0, j >= 2 n1

    // one phase.  Q: quadrant; BUF: LDS buffer of the K-tile being multiplied; t: that K-tile; WAIT: vmcnt count
    // (< 0: no wait); issue: re-stage the half-tile this phase is responsible for (tile index st)
#define PP_PHASE(Q, BUF, WAIT, DO_ISSUE, ST)                                                                             \
    {                                                                                                                    \
        constexpr int tb_ = (BUF) * PP_BUF;                                                                              \
        if ((Q) == 0) {                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fb[j][ks] = *reinterpret_cast<const half8 *>(pb[ks] + (tb_ + 2 * PP_HT + 2048 * j));                     \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fa[i][ks] = *reinterpret_cast<const half8 *>(pa[ks] + (tb_ + 2048 * i));                                \
        } else if ((Q) == 1) {                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fb[2 + j][ks] = *reinterpret_cast<const half8 *>(pb[ks] + (tb_ + 3 * PP_HT + 2048 * j));                 \
        } else if ((Q) == 2) {                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fa[i][ks] = *reinterpret_cast<const half8 *>(pa[ks] + (tb_ + PP_HT + 2048 * i));                        \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (DO_ISSUE) stage((ST), (Q) == 2 ? 0 : (Q) == 3 ? 1 : (Q) == 0 ? 2 : 3);                                      \
        if ((WAIT) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                                \
        else if ((WAIT) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                           \
        else if ((WAIT) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                           \
        else if ((WAIT) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                           \
        __builtin_amdgcn_s_barrier();                                                                                    \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                   \
        {                                                                                                                \
            constexpr int ib_ = ((Q) >= 2) ? 4 : 0, jb_ = ((Q) == 1 || (Q) == 2) ? 2 : 0;                                \
            _Pragma("unroll") for (int ks = 0; ks < 2; ks++) _Pragma("unroll") for (int i = 0; i < 4; i++)             \
                _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                         \
                    if (SWAP) acc[ib_ + i][jb_ + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jb_ + j][ks], fa[i][ks], acc[ib_ + i][jb_ + j], 0, 0, 0); \
                    else acc[ib_ + i][jb_ + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i][ks], fb[jb_ + j][ks], acc[ib_ + i][jb_ + j], 0, 0, 0);      \
                }                                                                                                        \
        }                                                                                                                \
        __builtin_amdgcn_s_setprio(0);                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                    \
    }

    // prologue: K-tile 0 complete, A0 and B0 of K-tile 1 (what phases (-1, 2) and (-1, 3) would have issued)
    stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3);
    stage(1, 0); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // A0(0), B0(0) of this wave have landed
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();        // the stagger: wave row 1 runs half a phase behind
    // steady state: K-tile pairs (t, t + 1) with t + 3 < nt, i.e. every issue below is in range
    int t = 0;
    for (; t + 4 <= nt; t += 2) {
        PP_PHASE(0, 0, 8, true, t + 1)
        PP_PHASE(1, 0, 8, true, t + 1)
        PP_PHASE(2, 0, -1, true, t + 2)
        PP_PHASE(3, 0, 8, true, t + 2)
        PP_PHASE(0, 1, 8, true, t + 2)
        PP_PHASE(1, 1, 8, true, t + 2)
        PP_PHASE(2, 1, -1, true, t + 3)
        PP_PHASE(3, 1, 8, true, t + 3)
    }
    // last pair (nt - 2, nt - 1): nothing left to issue after A1(nt - 1); the waits count down 8 8 . 4 2 0
    PP_PHASE(0, 0, 8, true, t + 1)
    PP_PHASE(1, 0, 8, true, t + 1)
    PP_PHASE(2, 0, -1, false, 0)
    PP_PHASE(3, 0, 4, false, 0)
    PP_PHASE(0, 1, 2, false, 0)
    PP_PHASE(1, 1, 0, false, 0)
    PP_PHASE(2, 1, -1, false, 0)
    PP_PHASE(3, 1, -1, false, 0)
    if (wr == 0) __builtin_amdgcn_s_barrier();        // balance the barrier count of the stagger
#undef PP_PHASE
}

template <int EPI, int WM, bool PP>
__global__ __launch_bounds__(256 * WM, 2) void gemm256_f16_kernel(GemmParams p) {
    constexpr int TBM = 128 * WM;
    constexpr int RING = WM == 2 ? 4 * 32768 : 3 * 24576, EPIB = 4 * WM * G2_EPI_WAVE;
    __shared__ __attribute__((aligned(16))) char smem[EPIB > RING ? EPIB : RING];
    const int ntn = p.N / G2_BN, ntm = (p.M + TBM - 1) / TBM;
    const int nwg = ntn * ntm;
    int bid = blockIdx.x;
    {
        int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // grouped order inside the XCD's run of tiles: G2_GM M-panels x all N-tiles at a time, N outer / M inner, so the
    // ~32 tiles an XCD has in flight are 4 M-panels x 8 N-tiles: the A panels (4 x 256 x K) stay in that XCD's 4 MiB
    // L2 for the whole N sweep and each streamed W tile is shared by 4 workgroups (PMC: W was re-fetched once per
    // M-panel from the Infinity Cache with the plain M-major order)
    int tm, tn;
    {
        constexpr int GM = G2_GM * (2 / WM);          // 1024 rows of A per group either way
        const int per_group = GM * ntn;
        const int g = bid / per_group, r = bid - g * per_group;
        const int gm = min(GM, ntm - g * GM);         // the last group may hold fewer M-panels
        tn = r / gm; tm = g * GM + (r - tn * gm);
    }
    const int m0 = tm * TBM, n0 = tn * G2_BN;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = WM == 2 ? (w >> 2) : 0, wn = w & 3;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[8][4];
    if (EPI == EPI_RESID_F32) {
        // x += A.W + bias: the residual tile and the bias are the INITIAL accumulator, fetched while the first
        // stages are in flight, so the epilogue is store-only (fire and forget) instead of an HBM-bound
        // read-modify-write that nothing overlaps (one workgroup per CU)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + wn * 64 + 16 * j + 4 * fq;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                int m = m0 + wm * 128 + 16 * i + fr;
                if (m >= p.M) m = p.M - 1;
                acc[i][j] = *reinterpret_cast<const f32x4 *>(reinterpret_cast<const float *>(p.out[0]) + (long)m * p.ldo + n) + bv;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const int seg = (EPI == EPI_F16 || EPI == EPI_GELU_F16) ? n0 / p.seg_n : 0;
    const bool vt = (EPI == EPI_F16 && seg == p.vt_seg);
    if (PP) {
        if (vt) gemm256_pingpong<false>(p, smem, m0, n0, acc);
        else gemm256_pingpong<true>(p, smem, m0, n0, acc);
    } else {
        if (vt) gemm256_mainloop<false, WM>(p, smem, m0, n0, acc);
        else gemm256_mainloop<true, WM>(p, smem, m0, n0, acc);
    }

    __builtin_amdgcn_s_barrier();  // every wave has left the staging ring: LDS is free for the epilogue images
    char *img = smem + w * G2_EPI_WAVE;
    if (vt) {  // lane holds rows m = mb + 4 fq + r of column n = nb + fr: image [n = 64][m = 128] (V^T)
        half_t *dst = reinterpret_cast<half_t *>(seg == 0 ? p.out[0] : seg == 1 ? p.out[1] : p.out[2]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float bv = p.bias ? p.bias[n0 + wn * 64 + 16 * j + fr] : 0.f;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                half4 v = {(half_t)(acc[i][j][0] + bv), (half_t)(acc[i][j][1] + bv), (half_t)(acc[i][j][2] + bv),
                           (half_t)(acc[i][j][3] + bv)};
                *reinterpret_cast<half4 *>(img + (16 * j + fr) * G2_EPI_ROW_T + (16 * i + 4 * fq) * 2) = v;
            }
        }
        // wave-private image: same-wave LDS accesses are ordered, no barrier needed
        // read back rows: 16 lanes x 16 B = one 256-byte row of 128 consecutive m; 4 rows per instruction
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int nrow = 4 * it + (lane >> 4), mc = (lane & 15) * 8;
            const half4 lo = *reinterpret_cast<const half4 *>(img + nrow * G2_EPI_ROW_T + mc * 2);
            const half4 hi = *reinterpret_cast<const half4 *>(img + nrow * G2_EPI_ROW_T + mc * 2 + 8);
            const int n = n0 + wn * 64 + nrow, nl = n - seg * p.seg_n, h = nl >> 6, dh = nl & 63;
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int m = m0 + wm * 128 + mc + 4 * half;
                if (m >= p.M) continue;
                const half4 v = half ? hi : lo;
                int b = m / p.S, ss = m - b * p.S;
                if (ss + 3 < p.S && m + 3 < p.M) {
                    *reinterpret_cast<half4 *>(dst + ((long)(b * p.H + h) * NH_DH + dh) * NH_SP + ss) = v;
                } else {
                    for (int r = 0; r < 4; r++) {
                        int mm = m + r;
                        if (mm >= p.M) break;
                        int bb = mm / p.S, s2 = mm - bb * p.S;
                        dst[((long)(bb * p.H + h) * NH_DH + dh) * NH_SP + s2] = v[r];
                    }
                }
            }
        }
        return;
    }
    if (EPI == EPI_F16 || EPI == EPI_GELU_F16) {
        // lane holds columns n = nb + 4 fq + r of row m = mb + fr: image [m = 128][n = 64] fp16
        half_t *ob = reinterpret_cast<half_t *>(seg == 0 ? p.out[0] : seg == 1 ? p.out[1] : p.out[2]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) bv = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wn * 64 + 16 * j + 4 * fq);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                f32x4 v = acc[i][j] + bv;
                if (EPI == EPI_GELU_F16) {
                    v[0] = gelu_tanh_f(v[0]); v[1] = gelu_tanh_f(v[1]); v[2] = gelu_tanh_f(v[2]); v[3] = gelu_tanh_f(v[3]);
                }
                half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4 *>(img + (16 * i + fr) * G2_EPI_ROW + (16 * j + 4 * fq) * 2) = hv;
            }
        }
        // read back: 8 lanes x 16 B = one 128-byte output row segment; 8 rows per instruction
        const int ncol = n0 + wn * 64 - seg * p.seg_n + (lane & 7) * 8;
#pragma unroll 4
        for (int it = 0; it < 16; it++) {
            const int mr = 8 * it + (lane >> 3);
            const int m = m0 + wm * 128 + mr;
            const half4 lo = *reinterpret_cast<const half4 *>(img + mr * G2_EPI_ROW + (lane & 7) * 16);
            const half4 hi = *reinterpret_cast<const half4 *>(img + mr * G2_EPI_ROW + (lane & 7) * 16 + 8);
            if (m < p.M) {
                half8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                *reinterpret_cast<half8 *>(ob + out_row(p, m) * p.ldo + ncol) = o;
            }
        }
        return;
    }
    half_t *obase = reinterpret_cast<half_t *>(seg == 0 ? p.out[0] : seg == 1 ? p.out[1] : p.out[2]);
    // SWAP orientation: lane holds columns n = nb + 4 fq + r (r = 0..3) of row m = mb + fr
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = n0 + wn * 64 + 16 * j + 4 * fq;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int m = m0 + wm * 128 + 16 * i + fr;
            if (m >= p.M) continue;
            f32x4 v = acc[i][j] + bv;
            if (EPI == EPI_F16 || EPI == EPI_GELU_F16) {
                if (EPI == EPI_GELU_F16) {
                    v[0] = gelu_tanh_f(v[0]); v[1] = gelu_tanh_f(v[1]);
                    v[2] = gelu_tanh_f(v[2]); v[3] = gelu_tanh_f(v[3]);
                }
                half_t *dst = obase + out_row(p, m) * p.ldo + (n - seg * p.seg_n);
                half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4 *>(dst) = hv;
            } else if (EPI == EPI_RESID_F32) {
                // acc already holds x + bias + A.W (the accumulator was initialised with the residual tile)
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out[0]) + (long)m * p.ldo + n) = acc[i][j];
            } else {  // EPI_CONV2_F32
                int s = m % p.S;
                f32x4 pe = *reinterpret_cast<const f32x4 *>(p.pos + (long)s * p.N + n);
                f32x4 o = {gelu_tanh_f(v[0]) + pe[0], gelu_tanh_f(v[1]) + pe[1], gelu_tanh_f(v[2]) + pe[2],
                           gelu_tanh_f(v[3]) + pe[3]};
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out[0]) + (long)m * p.ldo + n) = o;
            }
        }
    }
}

static const bool g_gemm_small_only = getenv("NORMA_HIP_GEMM128") != nullptr;  // A/B switch: force the 128^2 kernel

void launch_gemm(const GemmParams &p, hipStream_t st) {
    const bool seg_ok = (p.epi != EPI_F16 && p.epi != EPI_GELU_F16) || (p.seg_n % G2_BN == 0);
    if (!g_gemm_small_only && p.N % G2_BN == 0 && p.K % G2_BK == 0 && seg_ok && p.M >= G2_BM) {
        // NORMA_HIP_GEMM_WM: A/B switch. 3 (default): ping-pong 256 x 256 when K % 128 == 0; 2: single-phase 256 x 256; 1: 128 x 256
        static const int wm_env = getenv("NORMA_HIP_GEMM_WM") ? atoi(getenv("NORMA_HIP_GEMM_WM")) : 3;
        const int ntn = p.N / G2_BN;
#define G2_LAUNCH(WM_, PP_, GRID, BLOCK)                                                                                  \
        switch (p.epi) { /* one instantiation per epilogue: a single accumulator-init / store path each (registers) */  \
            case EPI_F16: hipLaunchKernelGGL((gemm256_f16_kernel<EPI_F16, WM_, PP_>), GRID, BLOCK, 0, st, p); break;       \
            case EPI_GELU_F16: hipLaunchKernelGGL((gemm256_f16_kernel<EPI_GELU_F16, WM_, PP_>), GRID, BLOCK, 0, st, p); break; \
            case EPI_RESID_F32: hipLaunchKernelGGL((gemm256_f16_kernel<EPI_RESID_F32, WM_, PP_>), GRID, BLOCK, 0, st, p); break; \
            default: hipLaunchKernelGGL((gemm256_f16_kernel<EPI_CONV2_F32, WM_, PP_>), GRID, BLOCK, 0, st, p); break;      \
        }
        if (wm_env == 1) {
            const dim3 grid(ntn * ((p.M + 127) / 128)), block(256);
            G2_LAUNCH(1, false, grid, block)
            return;
        }
        const dim3 grid(ntn * ((p.M + G2_BM - 1) / G2_BM)), block(512);
        if (wm_env == 3 && p.K % 128 == 0 && p.K >= 256) { G2_LAUNCH(2, true, grid, block) }
        else { G2_LAUNCH(2, false, grid, block) }
#undef G2_LAUNCH
        return;
    }
    int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    hipLaunchKernelGGL(gemm_f16_kernel, dim3(ntn * ntm), dim3(256), 0, st, p);
}
