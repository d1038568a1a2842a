// ldsprobe: do two workgroups of DIFFERENT kernels that share a CU keep their LDS allocations apart when one of them uses a
// large dynamic allocation?  Victim: 256 threads, `vbytes` of static-sized (here dynamic, same effect) LDS filled with a pattern
// derived from its own id, held for a while, verified.  Aggressor: 512 threads, `abytes` of dynamic LDS, written in full (in
// bounds only) over and over.  Both on their own stream, many rounds.  Any word the victim reads back wrong is counted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void victim(unsigned *errs, unsigned *first, int words, int spins) {
    extern __shared__ unsigned vs[];
    const unsigned tag = 0x51000000u ^ (blockIdx.x * 2654435761u);
    for (int i = threadIdx.x; i < words; i += 256) vs[i] = tag + i;
    __syncthreads();
    unsigned bad = 0;
    for (int s = 0; s < spins; s++) {
        for (int i = threadIdx.x; i < words; i += 256) {
            unsigned v = vs[i];
            if (v != tag + i) { bad++; if (atomicAdd(errs, 1u) == 0) { first[0] = blockIdx.x; first[1] = i; first[2] = v; first[3] = tag + i; } vs[i] = tag + i; }
        }
        __syncthreads();
    }
}

// second victim: the waves of a workgroup hand data to each other across a barrier, every round with a new pattern -- a wave
// that is let through the barrier early reads the previous round's words
__global__ __launch_bounds__(256) void victim_xwave(unsigned *errs, unsigned *first, int words, int spins) {
    extern __shared__ unsigned vs[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = words / 4;
    for (int s = 0; s < spins; s++) {
        const unsigned tag = 0x77000000u ^ (blockIdx.x * 2654435761u) ^ (s << 20);
        for (int i = lane; i < per; i += 64) vs[w * per + i] = tag + w * per + i;     // my quarter
        __syncthreads();
        const int o = (w + 1 + (s & 1)) & 3;                                            // somebody else's quarter
        for (int i = lane; i < per; i += 64) {
            unsigned v = vs[o * per + i];
            if (v != tag + o * per + i) { if (atomicAdd(errs, 1u) == 0) { first[0] = blockIdx.x; first[1] = o * per + i; first[2] = v; first[3] = tag + o * per + i; } }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(512) void aggressor(unsigned *sink, int words, int spins) {
    extern __shared__ unsigned as[];
    unsigned acc = 0;
    for (int s = 0; s < spins; s++) {
        for (int i = threadIdx.x; i < words; i += 512) as[i] = 0xA6000000u + (s << 16) + i;
        __syncthreads();
        for (int i = threadIdx.x; i < words; i += 512) acc += as[i];
        __syncthreads();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

// ---- third experiment: the aggressor as it is in the product (k_decode.hip, skinny_ldsp_kernel): every MFMA's B operand comes
// straight from a ds_read_b128, two waves per SIMD; the victims re-read LDS they filled themselves, in three access patterns
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void aggressor_mfma(float *sink, int bytes, int iters, int mode) {
    extern __shared__ __attribute__((aligned(16))) char xs[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid * 16; i < bytes; i += 512 * 16) *reinterpret_cast<uint4 *>(xs + i) = make_uint4(0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    __syncthreads();
    f4 acc[4];
    for (int c = 0; c < 4; c++) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    h8 a = {(_Float16)1.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const int rows = bytes / 64;
    for (int it = 0; it < iters; it++) {
        const int st = (it * 4) % (rows / 64);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int r = (st + c) * 64 + lane;
            h8 b = *reinterpret_cast<const h8 *>(xs + (long)(r % rows) * 64 + ((lane >> 4) << 4));
            if (mode == 1) {   // LDS reads without MFMA
                acc[c][0] += (float)b[0]; acc[c][1] += (float)b[1];
            } else if (mode == 2) {   // MFMA without the LDS read in front
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, acc[c], 0, 0, 0);
                acc[c][2] += (float)b[2] * 0.f;
            } else acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
        }
    }
    float t = 0.f;
    for (int c = 0; c < 4; c++) t += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (t == 123.456f) sink[0] = t;
}

// victim patterns: 0 = unit stride, 1 = stride 16 words (16-way bank conflicts), 2 = butterflies: read two words a long way apart,
// write sum / difference into the other buffer, barrier, verify against the closed form, like the FFT stages of logmel_kernel
__global__ __launch_bounds__(256) void victim_pat(unsigned *errs, unsigned *first, int pattern, int spins, const float *tw) {
    __shared__ float va[4][800], vb[4][800];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float base = (float)(blockIdx.x & 1023);
    for (int s = 0; s < spins; s++) {
        for (int i = lane; i < 800; i += 64) va[w][i] = base + (float)(i + s);
        __syncthreads();
        if (pattern == 2) {
            for (int o = lane; o < 400; o += 64) { float x = va[w][o], y = va[w][o + 400] * (tw ? tw[(o + s) & 1023] : 1.f); vb[w][2 * o] = x + y; vb[w][2 * o + 1] = y - x; }
            __syncthreads();
            for (int o = lane; o < 400; o += 64) {
                const float ex = 2.f * base + (float)(2 * o + 2 * s + 400), ey = 400.f;
                if (vb[w][2 * o] != ex || vb[w][2 * o + 1] != ey) { if (atomicAdd(errs, 1u) == 0) { first[0] = blockIdx.x; first[1] = o; first[2] = __float_as_uint(vb[w][2 * o]); first[3] = __float_as_uint(ex); } }
            }
        } else {
            const int stride = pattern == 1 ? 16 : 1;
            for (int k = 0; k < 12; k++) {
                const int i = (lane * stride + k * 64) % 800;
                const float v = va[w][i];
                if (v != base + (float)(i + s)) { if (atomicAdd(errs, 1u) == 0) { first[0] = blockIdx.x; first[1] = i; first[2] = __float_as_uint(v); first[3] = __float_as_uint(base + (float)(i + s)); } }
            }
        }
        __syncthreads();
    }
}

static void third_experiment(int rounds, unsigned *errs, unsigned *first, float *sinkf, hipStream_t sv, hipStream_t sa) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&aggressor_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const char *modes[] = {"ds_read_b128 -> MFMA", "ds_read_b128 -> VALU", "MFMA fed from registers"};
    const char *pats[] = {"unit-stride re-reads", "16-way conflicted re-reads", "butterflies across a barrier", "butterflies with twiddles loaded from global memory"};
    float *tw; CK(hipMalloc(&tw, 4096));
    { float ones[1024]; for (int i = 0; i < 1024; i++) ones[i] = 1.f; CK(hipMemcpy(tw, ones, 4096, hipMemcpyHostToDevice)); }
    for (int mode = 0; mode < 3; mode++)
        for (int pat = 0; pat < 4; pat++) {
            CK(hipMemset(errs, 0, 4)); CK(hipMemset(first, 0, 16));
            for (int r = 0; r < rounds; r++) {
                hipLaunchKernelGGL(aggressor_mfma, dim3(256), dim3(512), 81920, sa, sinkf, 81920, 3000, mode);
                hipLaunchKernelGGL(victim_pat, dim3(6000), dim3(256), 0, sv, errs, first, pat == 3 ? 2 : pat, 12, pat == 3 ? tw : nullptr);
            }
            CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
            unsigned e = 0, f[4];
            CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 16, hipMemcpyDeviceToHost));
            printf("victim (25.6 KB LDS, %s) beside 80 KB aggressor (%s), %d rounds: %u wrong values", pats[pat], modes[mode], rounds, e);
            if (e) printf("  (first: workgroup %u index %u got 0x%08x expected 0x%08x)", f[0], f[1], f[2], f[3]);
            printf("\n"); fflush(stdout);
        }
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    CK(hipSetDevice(0));
    unsigned *errs, *first, *sink;
    CK(hipMalloc(&errs, 4)); CK(hipMalloc(&first, 16)); CK(hipMalloc(&sink, 4));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&aggressor), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&victim), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (argc > 2) { third_experiment(rounds, errs, first, reinterpret_cast<float *>(sink), sv, sa); return 0; }
    const int vsizes[] = {37012, 16384, 4096};
    const int asizes[] = {40960, 65536, 66560, 81920, 122880};
    for (int vb : vsizes)
        for (int ab : asizes) {
            if (vb + ab > 160 * 1024) continue;
            CK(hipMemset(errs, 0, 4)); CK(hipMemset(first, 0, 16));
            for (int r = 0; r < rounds; r++) {
                hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), ab, sa, sink, ab / 4, 40);
                hipLaunchKernelGGL(victim, dim3(3000), dim3(256), vb, sv, errs, first, vb / 4, 6);
            }
            CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
            unsigned e = 0, f[4];
            CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 16, hipMemcpyDeviceToHost));
            printf("victim %6d B beside aggressor %6d B, %d rounds: %u wrong words", vb, ab, rounds, e);
            if (e) printf("  (first: victim workgroup %u word %u read 0x%08x expected 0x%08x)", f[0], f[1], f[2], f[3]);
            printf("\n"); fflush(stdout);
        }
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&victim_xwave), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int ab : {81920, 122880}) {
        const int vb = 36864;
        CK(hipMemset(errs, 0, 4)); CK(hipMemset(first, 0, 16));
        for (int r = 0; r < rounds; r++) {
            hipLaunchKernelGGL(aggressor, dim3(256), dim3(512), ab, sa, sink, ab / 4, 40);
            hipLaunchKernelGGL(victim_xwave, dim3(3000), dim3(256), vb, sv, errs, first, vb / 4, 6);
        }
        CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
        unsigned e = 0, f[4];
        CK(hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(f, first, 16, hipMemcpyDeviceToHost));
        printf("cross-wave victim %6d B beside aggressor %6d B (80 barriers per workgroup), %d rounds: %u wrong words", vb, ab, rounds, e);
        if (e) printf("  (first: victim workgroup %u word %u read 0x%08x expected 0x%08x)", f[0], f[1], f[2], f[3]);
        printf("\n"); fflush(stdout);
    }
    return 0;
}
