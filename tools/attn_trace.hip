// attn_trace: where does a 64-key tile of enc_attn_kernel spend its ~3100 cycles?  A copy of the kernel body with clock64
// stamps (shader clock) around the stages of one wave, run at 1, 2 and 3 workgroups per CU (LDS padding), distil-large-v3
// shapes (B = 32, H = 20, S = 1500).  Prints average cycles per stage over tiles 4..19 of one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../norma_amd/csrc/nh_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
#define KT 64              // keys per tile
#define QW 32              // queries per wave
#define QB 128             // queries per workgroup
#define TILE_B 8192        // bytes per K or V^T tile

__device__ __forceinline__ int swap23(int x) { return (x & ~12) | ((x & 4) << 1) | ((x & 8) >> 1); }

__global__ __launch_bounds__(256, 2) void enc_attn_trace(const half_t *__restrict__ q, const half_t *__restrict__ k,
                                                         long ld, const half_t *__restrict__ vt,
                                                         half_t *__restrict__ out, long ldo, int S, int H, unsigned long long *stamps) {
    unsigned long long T[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool rec = blockIdx.x == 3 && blockIdx.y == 5 && blockIdx.z == 1 && threadIdx.x == 0;
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
    // staging: slot s = tid + 256 i (i = 0,1): LDS row = s >> 3, chunk' = s & 7
    const half_t *kg[2]; const half_t *vg[2];
    int krow_off[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        int row = (tid >> 3) + 32 * i;
        int c = (tid & 7) ^ ((row >> 1) & 7);
        krow_off[i] = swap23(row);  // LDS row `row` holds key key0 + swap23(row)
        kg[i] = k + (long)b * S * ld + h * NH_DH + c * 8;
        vg[i] = vt + ((long)(b * H + h) * NH_DH + row) * NH_SP + c * 8;
    }
    u32x4 rk[2], rv[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            int key = t * KT + krow_off[i]; if (key >= S) key = S - 1;
            rk[i] = *reinterpret_cast<const u32x4 *>(kg[i] + (long)key * ld);
            rv[i] = *reinterpret_cast<const u32x4 *>(vg[i] + t * KT);
        }
    };
    auto store_tile = [&](int buf) {
        u32x4 *lk = reinterpret_cast<u32x4 *>(smem + buf * 2 * TILE_B);
        u32x4 *lv = reinterpret_cast<u32x4 *>(smem + buf * 2 * TILE_B + TILE_B);
#pragma unroll
        for (int i = 0; i < 2; i++) { lk[tid + 256 * i] = rk[i]; lv[tid + 256 * i] = rv[i]; }
    };
    load_tile(0);
    store_tile(0);
    __syncthreads();

    // fragment byte offsets inside a tile (before adding the per-step chunk xor)
    const int g = (r >> 1) & 7;                       // same for rows r and r + 32
    const int offK0 = r * 128 + ((hh ^ g) << 4);      // K block 0 (rows 0..31); block 1 at + 4096
    const int offV0 = r * 128 + ((hh ^ g) << 4);      // V^T dh block 0; block 1 at + 4096

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; i++) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float c_exp = 0.125f * 1.4426950408889634f;  // dh^-1/2 * log2(e)

    int cur = 0;
    for (int t = 0; t < nT; t++) {
        const bool more = (t + 1 < nT);
        const bool tr = t >= 4 && t < 20;
        unsigned long long k0_ = clock64();
        if (more) load_tile(t + 1);
        const char *tk = smem + cur * 2 * TILE_B, *tv = tk + TILE_B;
        f32x16 s0, s1;
        {
            // the first k-step takes a literal zero accumulator (an inline constant of the MFMA, not 32 v_mov per tile)
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
        asm volatile("s_nop 0" ::"v"(s0[0]), "v"(s1[0]));
        unsigned long long k1_ = clock64();   // QK MFMAs done (their results were just read)
        if (t * KT + KT > S) {  // last, partial tile: mask keys >= S
#pragma unroll
            for (int i = 0; i < 16; i++) {
                int rho = (i & 3) + 8 * (i >> 2) + 4 * hh;
                int key0 = t * KT + swap23(rho), key1 = t * KT + 32 + swap23(rho);
                if (key0 >= S) s0[i] = -INFINITY;
                if (key1 >= S) s1[i] = -INFINITY;
            }
        }
        float mx = s0[0];
#pragma unroll
        for (int i = 1; i < 16; i++) mx = fmaxf(mx, s0[i]);
#pragma unroll
        for (int i = 0; i < 16; i++) mx = fmaxf(mx, s1[i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        unsigned long long k2_ = clock64();   // max reduce + shuffle
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c_exp);
        const float mb = m_new * c_exp;
        // exponent arguments and the row sum two values per instruction (v_pk_fma_f32 / v_pk_add_f32): this kernel is bound
        // by VALU + transcendental issue, not by the matrix pipe (32 v_exp_f32 per tile are unavoidable, the rest is not)
        f32x2 ps2 = {0.f, 0.f};
        const f32x2 c2 = {c_exp, c_exp}, nmb2 = {-mb, -mb};
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            f32x2 a = {s0[i], s0[i + 1]}, b = {s1[i], s1[i + 1]};
            a = __builtin_elementwise_fma(a, c2, nmb2);
            b = __builtin_elementwise_fma(b, c2, nmb2);
            a[0] = __builtin_amdgcn_exp2f(a[0]); a[1] = __builtin_amdgcn_exp2f(a[1]);
            b[0] = __builtin_amdgcn_exp2f(b[0]); b[1] = __builtin_amdgcn_exp2f(b[1]);
            ps2 += a; ps2 += b;
            s0[i] = a[0]; s0[i + 1] = a[1]; s1[i] = b[0]; s1[i + 1] = b[1];
        }
        const float ps = ps2[0] + ps2[1];
        asm volatile("s_nop 0" ::"v"(ps));
        unsigned long long k3_ = clock64();   // exponentials + row sum
        l_run = l_run * alpha + ps;
        m_run = m_new;
        // the running maximum settles after the first tiles: when no query of the wave moved it, alpha is exactly 1 and the
        // 32 rescaling multiplies (VALU is what bounds this kernel) are skipped -- bit-identical either way
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
            for (int i = 0; i < 16; i++) { o0[i] *= alpha; o1[i] *= alpha; }
        }
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
        asm volatile("s_nop 0" ::"v"(o0[0]), "v"(o1[0]));
        unsigned long long k4_ = clock64();   // cvt + V reads + PV MFMAs done
        if (more) store_tile(cur ^ 1);
        unsigned long long k5_ = clock64();   // wait for the next tile's global loads + LDS store
        __syncthreads();
        unsigned long long k6_ = clock64();   // barrier
        if (tr) { T[0] += k1_ - k0_; T[1] += k2_ - k1_; T[2] += k3_ - k2_; T[3] += k4_ - k3_; T[4] += k5_ - k4_; T[5] += k6_ - k5_; T[6] += k6_ - k0_; T[7] += 1; }
        (void)tprev;
        cur ^= 1;
    }
    if (rec) for (int i = 0; i < 8; i++) stamps[i] = T[i];
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


int main() {
    const int B = 32, H = 20, S = 1500, d = 1280;
    half_t *q, *k, *vt, *out; unsigned long long *st;
    CK(hipMalloc(&q, (size_t)B * S * d * 2)); CK(hipMalloc(&k, (size_t)B * S * d * 2)); CK(hipMalloc(&vt, (size_t)B * d * NH_SP * 2));
    CK(hipMalloc(&out, (size_t)B * S * d * 2)); CK(hipMalloc(&st, 64));
    CK(hipMemset(q, 0x11, (size_t)B * S * d * 2)); CK(hipMemset(k, 0x12, (size_t)B * S * d * 2)); CK(hipMemset(vt, 0x13, (size_t)B * d * NH_SP * 2));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[7] = {"loads issue + K reads + QK MFMAs", "max + shuffle", "exp + sum", "cvt + V reads + PV MFMAs", "vmcnt wait + LDS store", "barrier", "whole tile"};
    for (int pad : {60000, 40000, 0}) {
        dim3 grid((S + QB - 1) / QB, H, B);
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(enc_attn_trace, grid, dim3(256), pad, 0, q, k, (long)d, vt, out, (long)d, S, H, st);
            hipEventRecord(b, 0); CK(hipEventSynchronize(b));
        }
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[8]; CK(hipMemcpy(h, st, 64, hipMemcpyDeviceToHost));
        printf("LDS pad %5d (%s workgroups per CU): kernel %.1f us\n", pad, pad == 0 ? "3" : pad == 40000 ? "2" : "1", ms * 1e3);
        for (int i = 0; i < 7; i++) printf("    %-34s %8.0f cycles\n", names[i], (double)h[i] / (double)h[7]);
    }
    return 0;
}
