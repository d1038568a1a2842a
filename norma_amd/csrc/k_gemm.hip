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

#include <atomic>

#include "nh_kernels.h"

#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)  // 16 KiB per operand tile

__device__ __forceinline__ float gelu_tanh_f(float v) { return gelu_tanh_fast(v); }

__device__ __forceinline__ const half_t *a_row_ptr(const GemmParams &p, int m) {
    if (m >= p.M) m = p.M - 1;  // clamp: tail rows are loaded but never stored
    int b = nh_div(m, p.a_rpb, p.a_magic), r = m - b * p.a_rpb;
    return p.A + (long)b * p.a_bstride + (long)r * p.lda;
}

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
    int cur = 0;
    for (int t = 0; t < nt; t++) {
        const bool more = (t + 1 < nt);
        if (more) stage(cur ^ 1, t + 1);  // all waves left buf[cur^1] at the barrier that ended iteration t-1
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
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA for tile t+1 has landed
        __syncthreads();
        cur ^= 1;
    }
}

__device__ __forceinline__ long out_row(const GemmParams &p, int m) {
    int b = nh_div(m, p.o_rpb, p.o_magic), r = m - b * p.o_rpb;
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
                int b = nh_div(m, p.S, p.s_magic), s = m - b * p.S;
                half_t *row = dst + ((long)(b * p.H + h) * NH_DH + dh) * NH_SP;
                if (s + 3 < p.S && m + 3 < p.M) {
                    half4 v = {(half_t)(acc[i][j][0] + bv), (half_t)(acc[i][j][1] + bv),
                               (half_t)(acc[i][j][2] + bv), (half_t)(acc[i][j][3] + bv)};
                    *reinterpret_cast<half4 *>(row + s) = v;
                } else {
                    for (int r = 0; r < 4; r++) {
                        int mm = m + r;
                        if (mm >= p.M) break;
                        int bb = nh_div(mm, p.S, p.s_magic), ss = mm - bb * p.S;
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
                if (p.epi == EPI_F16 && seg == 0 && p.seg0_scale != 0.f) v *= p.seg0_scale;
                if (p.epi == EPI_GELU_F16) {
                    v[0] = gelu_tanh_f(v[0]); v[1] = gelu_tanh_f(v[1]);
                    v[2] = gelu_tanh_f(v[2]); v[3] = gelu_tanh_f(v[3]);
                }
                const int nl = n - seg * p.seg_n;
                half_t *dst = p.head_major ? obase + (((long)nh_div(m, p.S, p.s_magic) * p.H + (nl >> 6)) * p.S + (m - nh_div(m, p.S, p.s_magic) * p.S)) * NH_DH + (nl & 63)
                                           : obase + out_row(p, m) * p.ldo + nl;
                half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                *reinterpret_cast<half4 *>(dst) = hv;
            } else if (p.epi == EPI_RESID_F32) {
                float *dst = reinterpret_cast<float *>(p.out[0]) + (long)m * p.ldo + n;
                f32x4 x = *reinterpret_cast<const f32x4 *>(dst);
                *reinterpret_cast<f32x4 *>(dst) = x + v;
            } else {  // EPI_CONV2_F32
                int s = m - nh_div(m, p.S, p.s_magic) * p.S;
                f32x4 pe = *reinterpret_cast<const f32x4 *>(p.pos + (long)s * p.N + n);
                f32x4 o = {gelu_tanh_f(v[0]) + pe[0], gelu_tanh_f(v[1]) + pe[1], gelu_tanh_f(v[2]) + pe[2],
                           gelu_tanh_f(v[3]) + pe[3]};
                *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out[0]) + (long)m * p.ldo + n) = o;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// 256 x 256 tile kernel for the large encoder shapes (N % 256 == 0, K % 128 == 0, M >= 256): persistent, one
// 512-thread workgroup per CU = 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave (128 accumulator VGPRs).
//
// Main loop = the 8-phase "ping-pong" schedule of the CDNA4 playbook (cdna_hip_programming.md, "The 256^2 8-phase
// template"), restated for this kernel's operand layout:
//  * BK = 64 K-tiles in two 64 KiB LDS buffers, each cut into four 16 KiB half-tiles A0 A1 B0 B1.  Half-tile Ah holds,
//    for BOTH wave rows, the h-th 64 rows of the wave's 128 (LDS row l <-> tile row (l >> 6) * 128 + 64 h + (l & 63));
//    Bh the h-th 32 columns of each wave column's 64 (l <-> (l >> 5) * 64 + 32 h + (l & 31)).  The permutation lives
//    in the per-lane SOURCE address of the LDS-DMA (global_load_lds_dwordx4; its LDS destination is lane-linear), and so
//    does the bank swizzle chunk' = chunk ^ ((row >> 1) & 7) that makes the ds_read_b128 fragment reads conflict-free.
//  * a K-tile is four phases = the four 64 x 32 quadrants of the wave's 128 x 64 output, 16 MFMAs each:
//        q0 (m0,n0): reads A0 (8 ds_read_b128) + B0 (4)    q1 (m0,n1): reads B1 (4)
//        q2 (m1,n1): reads A1 (8)                          q3 (m1,n0): no reads (the B0 fragments are kept)
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
//
// Persistent tiles: the workgroup walks tiles blockIdx.x, + gridDim.x, ... (XCD-aware order below).  As soon as the
// main loop of a tile ends it issues the six prologue half-tiles of its NEXT tile, then runs the epilogue of the
// finished one: block launch, first-fetch latency and the store drain (14 of 47 us per K = 1280 tile before) hide
// under each other.  The epilogue therefore stays out of the ring: fp16 outputs go through a 2176-byte per-wave
// image (16 rows x 64 columns, 136-byte pitch) behind it, so that every global store writes whole 128-byte rows.
// ---------------------------------------------------------------------------------------------------
#define G2_BM 256
#define G2_BN 256
#ifndef G2_GM
#define G2_GM 4           // M-panels per group of the tile order (2 and 8 measured: within +-3 %, 4 best overall)
#endif
#define PP_BUF 65536
#define PP_HT 16384
#define PP_RING (2 * PP_BUF)
#define PP_EPI_PITCH 136  // bytes per image row (64 fp16 + 8 pad)
#define PP_EPI_WAVE (16 * PP_EPI_PITCH)

struct PPSource { unsigned ag[2][2], wg[2][2]; };  // [h][i]: 32-bit element offsets of this lane's eight source rows

// staging map: half-tile slot s = tid + 512 i (i = 0, 1): LDS row l = s >> 3 = (tid >> 3) + 64 i, chunk' = tid & 7,
// source chunk = chunk' ^ ((l >> 1) & 7) (64 i leaves the swizzle unchanged)
__device__ __forceinline__ PPSource pp_source(const GemmParams &p, int m0, int n0) {
    // the lane constants below are recomputed for every tile on purpose: hoisted out of the tile loop they do not fit beside
    // 128 accumulators + 64 fragment registers, get spilled, and their reloads (scratch loads count in vmcnt) make hipcc
    // drain vmcnt(0) in front of the prologue DMAs
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lrow = tid >> 3;
    const int csrc = ((tid & 7) ^ ((lrow >> 1) & 7)) * 8;
    PPSource sg;
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
            sg.ag[h][i] = (unsigned)(a_row_ptr(p, m0 + i * 128 + h * 64 + lrow) - p.A) + csrc;
            const int l = lrow + 64 * i;
            sg.wg[h][i] = (unsigned)(n0 + (l >> 5) * 64 + h * 32 + (l & 31)) * (unsigned)p.K + csrc;
        }
    return sg;
}

// which: 0 A0, 1 B0, 2 B1, 3 A1 (the issue order of a K-tile)
__device__ __forceinline__ void pp_stage(const GemmParams &p, char *wbase, const PPSource &sg, int t, int which) {
    const int h = (which >> 1) & 1;
    const bool isB = which == 1 || which == 2;
    char *dst = wbase + (t & 1) * PP_BUF + (isB ? 2 * PP_HT : 0) + h * PP_HT;
    const unsigned ko = (unsigned)t * 64u;
    if (isB) {
        __builtin_amdgcn_global_load_lds((gbl_void *)(p.W + (size_t)(sg.wg[h][0] + ko)), (lds_void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void *)(p.W + (size_t)(sg.wg[h][1] + ko)), (lds_void *)(dst + 8192), 16, 0, 0);
    } else {
        __builtin_amdgcn_global_load_lds((gbl_void *)(p.A + (size_t)(sg.ag[h][0] + ko)), (lds_void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void *)(p.A + (size_t)(sg.ag[h][1] + ko)), (lds_void *)(dst + 8192), 16, 0, 0);
    }
}

// K-tile 0 complete, A0 and B0 of K-tile 1: what phases (-1, 2) and (-1, 3) would have issued
__device__ __forceinline__ void pp_prologue(const GemmParams &p, char *wbase, const PPSource &sg) {
    pp_stage(p, wbase, sg, 0, 0); pp_stage(p, wbase, sg, 0, 1); pp_stage(p, wbase, sg, 0, 2); pp_stage(p, wbase, sg, 0, 3);
    pp_stage(p, wbase, sg, 1, 0); pp_stage(p, wbase, sg, 1, 1);
}

// the main loop proper; the prologue of this tile has been issued (any time) before
template <bool SWAP>
__device__ __forceinline__ void pp_main(const GemmParams &p, char *smem, char *wbase, const PPSource &sg, f32x4 (&acc)[8][4]) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wr = w >> 2, wc = w & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int nt = p.K >> 6;
    // fragment read addresses inside a half-tile: row r, chunk 4 ks + fq, swizzle (r >> 1) & 7 == fr >> 1 for every
    // fragment; the ks = 1 chunk is the ks = 0 address with bit 6 flipped: one per-lane base per k-step, everything
    // else is an immediate offset of the ds_read
    const int offA0 = (wr * 64 + fr) * 128 + ((fq ^ (fr >> 1)) << 4);  // + 2048 i (i = 0..3)
    const int offB0 = (wc * 32 + fr) * 128 + ((fq ^ (fr >> 1)) << 4);  // + 2048 j (j = 0..1)
    const char *const pa[2] = {smem + offA0, smem + (offA0 ^ 64)};
    const char *const pb[2] = {smem + offB0, smem + (offB0 ^ 64)};
    half8 fa[4][2], fb[4][2];  // fa[i][ks]: current m-half; fb[j][ks]: j < 2 n0, j >= 2 n1

    // one phase.  Q: quadrant; BUF: LDS buffer of the K-tile being multiplied; WAIT: vmcnt count (< 0: none);
    // DO_ISSUE: re-stage the half-tile this phase is responsible for, from K-tile ST
#define PP_PHASE(Q, BUF, WAIT, DO_ISSUE, ST)                                                                             \
    {                                                                                                                    \
        constexpr int tb_ = (BUF) * PP_BUF;                                                                              \
        if ((Q) == 0) {                                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fb[j][ks] = *reinterpret_cast<const half8 *>(pb[ks] + (tb_ + 2 * PP_HT + 2048 * j));                    \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fa[i][ks] = *reinterpret_cast<const half8 *>(pa[ks] + (tb_ + 2048 * i));                               \
        } else if ((Q) == 1) {                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fb[2 + j][ks] = *reinterpret_cast<const half8 *>(pb[ks] + (tb_ + 3 * PP_HT + 2048 * j));                \
        } else if ((Q) == 2) {                                                                                           \
            _Pragma("unroll") for (int i = 0; i < 4; i++) _Pragma("unroll") for (int ks = 0; ks < 2; ks++)             \
                fa[i][ks] = *reinterpret_cast<const half8 *>(pa[ks] + (tb_ + PP_HT + 2048 * i));                       \
        }                                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                               \
        if (DO_ISSUE) pp_stage(p, wbase, sg, (ST), (Q) == 2 ? 0 : (Q) == 3 ? 1 : (Q) == 0 ? 2 : 3);                     \
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

    // everything issued so far (prologue DMAs, the previous tile's stores, the residual tile) has landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

// XCD-aware tile order.  Workgroups b and b + 8 share an XCD (L2): every XCD gets a contiguous run of tiles, walked in
// groups of G2_GM M-panels x all N-tiles, N outer / M inner, so the ~32 tiles an XCD has in flight are 4 M-panels x 8
// N-tiles: the A panels (4 x 256 x K) stay in that XCD's 4 MiB L2 for the whole N sweep and each streamed W tile is
// shared by 4 workgroups (PMC: W was re-fetched once per M-panel from the Infinity Cache with a plain M-major order).
__device__ __forceinline__ void pp_tile(int vb, int ntn, int ntm, int &tm, int &tn) {
    const int nwg = ntn * ntm;
    const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
    const int bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
    const int per_group = G2_GM * ntn;
    const int g = bid / per_group, rr = bid - g * per_group;
    const int gm = min(G2_GM, ntm - g * G2_GM);  // the last group may hold fewer M-panels
    tn = rr / gm; tm = g * G2_GM + (rr - tn * gm);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_f16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char smem[PP_RING + 8 * PP_EPI_WAVE];
    const int ntn = p.N / G2_BN, ntm = (p.M + G2_BM - 1) / G2_BM, nwg = ntn * ntm;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int fr = lane & 15, fq = lane >> 4;
    char *const wbase = smem + __builtin_amdgcn_readfirstlane(w) * 1024;  // this wave's 1 KiB run of every 8 KiB DMA group
    char *const img = smem + PP_RING + w * PP_EPI_WAVE;

    int vb = blockIdx.x, tm, tn;
    pp_tile(vb, ntn, ntm, tm, tn);
    PPSource sg = pp_source(p, tm * G2_BM, tn * G2_BN);
    pp_prologue(p, wbase, sg);
#ifdef G2_STAMPS   // diagnostic build (tools/gstamps): when does every workgroup start, finish each tile, and on which XCD
    int stamp_n = 0;
    if (threadIdx.x == 0 && p.dbg) {
        p.dbg[blockIdx.x * 32 + 0] = __builtin_amdgcn_s_memrealtime();
        p.dbg[blockIdx.x * 32 + 1] = __builtin_amdgcn_s_getreg(6164);   // HW_REG_XCC_ID[3:0]
    }
#endif
    for (;;) {
        const int m0 = tm * G2_BM, n0 = tn * G2_BN;
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int seg = (EPI == EPI_F16 || EPI == EPI_GELU_F16) ? n0 / p.seg_n : 0;
        const bool vt = (EPI == EPI_F16 && seg == p.vt_seg);
        if (vt) pp_main<false>(p, smem, wbase, sg, acc);
        else pp_main<true>(p, smem, wbase, sg, acc);

#ifdef G2_STAMPS
        if (threadIdx.x == 0 && p.dbg) p.dbg[blockIdx.x * 32 + 3] += __builtin_amdgcn_s_memrealtime() - (stamp_n ? p.dbg[blockIdx.x * 32 + 4 + stamp_n - 1] : p.dbg[blockIdx.x * 32]);   // time from tile start to the end of its main loop, summed
#endif
        // EPI_RESID_F32: residual values of four 16-row blocks (64 VGPRs: the fragment registers are dead here), fetched through a
        // buffer descriptor over x[M][ldo]: rows >= M of the last, partial M-panel fall outside it, so their loads return 0 and
        // their stores are dropped by the hardware -- no per-row guard (a guard makes hipcc wrap each row block's stores in an
        // exec-masked region behind s_waitcnt vmcnt(0), and on gfx950 vmcnt counts STORES too: the row blocks would go out one
        // store round trip after the other)
        f32x4 rs[4][4];
        __amdgpu_buffer_rsrc_t xrs;
        unsigned xoff = 0;
        if (EPI == EPI_RESID_F32) {
            xrs = __builtin_amdgcn_make_buffer_rsrc(p.out[0], 0, (int)((long)p.M * p.ldo * 4), 0x00020000);
            int ln = threadIdx.x;
            asm volatile("" : "+v"(ln));   // recomputed per tile (a hoisted lane constant is spilled, and its reload drains vmcnt: pp_source)
            xoff = (unsigned)(((long)(m0 + (ln >> 8) * 128 + (ln & 15)) * p.ldo + n0 + ((ln >> 6) & 3) * 64 + 4 * ((ln >> 4) & 3)) * 4);   // row block i at + 64 i ldo, piece j at + 64 j
#pragma unroll
            for (int ii = 0; ii < 4; ii++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    rs[ii][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + (unsigned)(64 * ii * p.ldo + 64 * j), 0, 0));
        }
        // every wave is through its last phase: the ring is free.  Start the next tile's first fetches now.
        const int nvb = vb + gridDim.x;
        const bool more = nvb < nwg;
        if (more) {
            pp_tile(nvb, ntn, ntm, tm, tn);
            sg = pp_source(p, tm * G2_BM, tn * G2_BN);
            pp_prologue(p, wbase, sg);
        }

        // ---- epilogue of tile (m0, n0) ----
        if (vt) {
            // lane holds rows m = mb + 4 fq + r of column n = nb + fr: per 16-column block j and 64-row half, a
            // [16 n][64 m] image (V^T); read back as 128-byte runs of 64 consecutive m
            half_t *dst = reinterpret_cast<half_t *>(seg == 0 ? p.out[0] : seg == 1 ? p.out[1] : p.out[2]);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float bv = p.bias ? p.bias[n0 + wn * 64 + 16 * j + fr] : 0.f;
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
#pragma unroll
                    for (int ii = 0; ii < 4; ii++) {
                        const f32x4 a = acc[4 * hh + ii][j];
                        half4 v = {(half_t)(a[0] + bv), (half_t)(a[1] + bv), (half_t)(a[2] + bv), (half_t)(a[3] + bv)};
                        *reinterpret_cast<half4 *>(img + fr * PP_EPI_PITCH + (16 * ii + 4 * fq) * 2) = v;
                    }
#pragma unroll
                    for (int it = 0; it < 2; it++) {
                        const int nrow = 8 * it + (lane >> 3), mc = (lane & 7) * 8;
                        const half4 lo = *reinterpret_cast<const half4 *>(img + nrow * PP_EPI_PITCH + mc * 2);
                        const half4 hi = *reinterpret_cast<const half4 *>(img + nrow * PP_EPI_PITCH + mc * 2 + 8);
                        const int n = n0 + wn * 64 + 16 * j + nrow, nl = n - seg * p.seg_n, h = nl >> 6, dh = nl & 63;
                        const int m = m0 + wm * 128 + 64 * hh + mc;
                        if (m >= p.M) continue;
                        const int b = nh_div(m, p.S, p.s_magic), ss = m - b * p.S;
                        if (ss + 7 < p.S && m + 7 < p.M) {
                            half8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            *reinterpret_cast<half8 *>(dst + ((long)(b * p.H + h) * NH_DH + dh) * NH_SP + ss) = o;
                        } else {
                            for (int r = 0; r < 8; r++) {
                                const int mm = m + r;
                                if (mm >= p.M) break;
                                const int bb = nh_div(mm, p.S, p.s_magic), s2 = mm - bb * p.S;
                                dst[((long)(bb * p.H + h) * NH_DH + dh) * NH_SP + s2] = r < 4 ? lo[r] : hi[r - 4];
                            }
                        }
                    }
                }
            }
        } else if (EPI == EPI_F16 || EPI == EPI_GELU_F16) {
            // lane holds columns n = nb + 4 fq + r of row m = mb + fr: per 16-row block i a [16 m][64 n] fp16 image
            half_t *ob = reinterpret_cast<half_t *>(seg == 0 ? p.out[0] : seg == 1 ? p.out[1] : p.out[2]);
            f32x4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv[j] = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wn * 64 + 16 * j + 4 * fq);
            }
            const int ncol = n0 + wn * 64 - seg * p.seg_n + (lane & 7) * 8;
            const bool q_scaled = EPI == EPI_F16 && seg == 0 && p.seg0_scale != 0.f;   // tile-uniform
#pragma unroll
            for (int i = 0; i < 8; i++) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    f32x4 v = acc[i][j] + bv[j];
                    if (EPI == EPI_F16 && q_scaled) v *= p.seg0_scale;
                    if (EPI == EPI_GELU_F16) {
                        v[0] = gelu_tanh_f(v[0]); v[1] = gelu_tanh_f(v[1]); v[2] = gelu_tanh_f(v[2]); v[3] = gelu_tanh_f(v[3]);
                    }
                    half4 hv = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *reinterpret_cast<half4 *>(img + fr * PP_EPI_PITCH + (16 * j + 4 * fq) * 2) = hv;
                }
                // read back: 8 lanes x 16 B = one 128-byte output row segment; 8 rows per instruction
#pragma unroll
                for (int it = 0; it < 2; it++) {
                    const int mr = 8 * it + (lane >> 3);
                    const int m = m0 + wm * 128 + 16 * i + mr;
                    const half4 lo = *reinterpret_cast<const half4 *>(img + mr * PP_EPI_PITCH + (lane & 7) * 16);
                    const half4 hi = *reinterpret_cast<const half4 *>(img + mr * PP_EPI_PITCH + (lane & 7) * 16 + 8);
                    if (m < p.M) {
                        half8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        // head-major: the wave's 64 columns are one head (seg_n and the tile base are multiples of 64)
                        half_t *dst = p.head_major ? ob + (((long)nh_div(m, p.S, p.s_magic) * p.H + (ncol >> 6)) * p.S + (m - nh_div(m, p.S, p.s_magic) * p.S)) * NH_DH + (lane & 7) * 8
                                                   : ob + out_row(p, m) * p.ldo + ncol;
                        *reinterpret_cast<half8 *>(dst) = o;
                    }
                }
            }
        } else if (EPI == EPI_RESID_F32) {
            // x[m][n] += acc + bias.  r01/r02 took the residual tile as the INITIAL accumulator: 32 loads per lane queued behind the
            // twelve prologue DMAs and waited for before the first MFMA -- 12.8 us of every 57 us out-proj tile with the matrix
            // pipe idle (tools/gstamps, profiles/r03_gstamps.txt).  Now the tile is fetched here, after the main loop, 16 loads per
            // lane at a time into the registers the fragments no longer need (the first sixteen were issued BEFORE the next tile's
            // prologue DMAs, see above), added in the reference's order x + (A.W + bias), and stored: the residual traffic runs
            // under the next tile's first fetches instead of in front of this tile's first MFMA.
            f32x4 bv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv[j] = *reinterpret_cast<const f32x4 *>(p.bias + n0 + wn * 64 + 16 * j + 4 * fq);
            }
#pragma unroll
            for (int ii = 0; ii < 4; ii++)
#pragma unroll
                for (int j = 0; j < 4; j++)   // the four 64-byte pieces of one 256-byte row segment back to back
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs[ii][j] + (acc[ii][j] + bv[j])), xrs, xoff + (unsigned)(64 * ii * p.ldo + 64 * j), 0, 0);
#pragma unroll
            for (int ii = 0; ii < 4; ii++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    rs[ii][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff + (unsigned)(64 * (4 + ii) * p.ldo + 64 * j), 0, 0));
#pragma unroll
            for (int ii = 0; ii < 4; ii++)
#pragma unroll
                for (int j = 0; j < 4; j++)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, rs[ii][j] + (acc[4 + ii][j] + bv[j])), xrs, xoff + (unsigned)(64 * (4 + ii) * p.ldo + 64 * j), 0, 0);
        } else {
            // EPI_CONV2_F32: f32 outputs straight from the accumulators: lane holds columns n = nb + 4 fq + r (16 B) of row m = mb + fr
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = n0 + wn * 64 + 16 * j + 4 * fq;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) bv = *reinterpret_cast<const f32x4 *>(p.bias + n);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int m = m0 + wm * 128 + 16 * i + fr;
                    if (m >= p.M) continue;
                    const f32x4 v = acc[i][j] + bv;
                    const int s = m - nh_div(m, p.S, p.s_magic) * p.S;
                    const f32x4 pe = *reinterpret_cast<const f32x4 *>(p.pos + (long)s * p.N + n);
                    f32x4 o = {gelu_tanh_f(v[0]) + pe[0], gelu_tanh_f(v[1]) + pe[1], gelu_tanh_f(v[2]) + pe[2],
                               gelu_tanh_f(v[3]) + pe[3]};
                    *reinterpret_cast<f32x4 *>(reinterpret_cast<float *>(p.out[0]) + (long)m * p.ldo + n) = o;
                }
            }
        }
        if (!more) break;
        vb = nvb;
    }
}

// CU count of the CURRENT device, cached per device (one process may drive several GPUs from several threads)
static int device_cu_count() {
    static std::atomic<int> cus[NH_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NH_MAX_DEVICES) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (!n) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

static GemmParams with_magics(const GemmParams &p_in) {
    GemmParams p = p_in;
    const long max_m = (long)p.M - 1;     // the largest row index a kernel divides (tail rows are clamped or skipped first)
    p.a_magic = nh_magic(p.a_rpb, max_m); p.o_magic = nh_magic(p.o_rpb, max_m); p.s_magic = nh_magic(p.S, max_m);
    return p;
}

void launch_gemm(const GemmParams &p_in, hipStream_t st) {
    const GemmParams p = with_magics(p_in);
    const bool seg_ok = (p.epi != EPI_F16 && p.epi != EPI_GELU_F16) || (p.seg_n % G2_BN == 0);
    if (p.N % G2_BN == 0 && p.K % 128 == 0 && seg_ok && p.M >= G2_BM) {
        const int nwg = (p.N / G2_BN) * ((p.M + G2_BM - 1) / G2_BM);
        int cus = p.cus > 0 ? p.cus : device_cu_count();  // persistent grid: one workgroup per CU the stream can reach
        cus -= cus % 8;  // the tile order assumes workgroups b and b + gridDim.x sit on the same XCD
        const dim3 grid(nwg < cus ? nwg : cus), block(512);
        switch (p.epi) {  // one instantiation per epilogue: a single accumulator-init / store path each (register pressure)
            case EPI_F16: hipLaunchKernelGGL(gemm256_f16_kernel<EPI_F16>, grid, block, 0, st, p); break;
            case EPI_GELU_F16: hipLaunchKernelGGL(gemm256_f16_kernel<EPI_GELU_F16>, grid, block, 0, st, p); break;
            case EPI_RESID_F32: hipLaunchKernelGGL(gemm256_f16_kernel<EPI_RESID_F32>, grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL(gemm256_f16_kernel<EPI_CONV2_F32>, grid, block, 0, st, p); break;
        }
        return;
    }
    launch_gemm_128(p, st);
}

// the 128 x 128 kernel on any supported shape (N % 128 == 0, K % 64 == 0); also the reference of tools/gemm_check
void launch_gemm_128(const GemmParams &p_in, hipStream_t st) {
    const GemmParams p = with_magics(p_in);
    int ntn = p.N / BN, ntm = (p.M + BM - 1) / BM;
    hipLaunchKernelGGL(gemm_f16_kernel, dim3(ntn * ntm), dim3(256), 0, st, p);
}
