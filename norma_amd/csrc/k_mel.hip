// k_mel.hip -- log-mel spectrogram on device (gfx950), HBM/latency-bound f32 work.
//
// Replaces candle's host-side `audio::pcm_to_mel` + `Tensor::from_vec` H2D + `narrow`
// (src/models/whisper/model.rs:74-88).  The arithmetic follows candle-transformers 0.7.2
// models::whisper::audio step for step (SURVEY.md 3.3[A]) INCLUDING ITS OPERATION ORDER, so the
// result agrees with the f32 CPU path to the last bits instead of to an "f32 FFT noise" tolerance:
//   * periodic Hann window, frames of 400 samples every 160, no centring / reflect padding;
//   * the same recursive radix-2 decimation 400 -> 200 -> 100 -> 50 -> 25 with an O(n^2) DFT at
//     length 25, each output accumulated over j = 0..24 in order, twiddles taken from tables the
//     host builds with libm (cosf/sinf of the same f32 angle expressions);
//   * power |X|^2, P[j] += P[400-j] for j in 1..199 (interior bins doubled);
//   * mel = log10(max(sum_k P[k] f[m][k], 1e-10)) with the 4-way grouped accumulation;
//   * per-clip max - 8 clamp, /4 + 1.
// This file is compiled with -ffp-contract=off: the reference does not fuse multiply-adds here.
//
// One wavefront per frame, everything staged in LDS.  16 frames per 1024-thread workgroup, and the workgroup asks for the WHOLE LDS
// of its CU (it uses 133 KB of the 160): r03 found this kernel's FFT stages disturbed -- a few wrong spectra per clip, rarely --
// whenever a workgroup of another stream whose MFMAs are fed from LDS reads (the logits GEMVs, the encoder GEMM / attention of
// another context) ran on the same CU (DESIGN.md 5 "A neighbour on the CU", tests/test_gpu_load.py).  With four 256-thread
// workgroups of 37 KB each a CU had room for such a neighbour; one workgroup holding all of the LDS has the same sixteen waves per
// CU and no LDS-using neighbour, whoever it might be.
#include "nh_kernels.h"
#include <atomic>

#define N_FFT 400
#define HOP 160
#define FRAMES_PER_WG 16
#define MEL_LDS_BYTES (160 * 1024)

__device__ __forceinline__ unsigned f32_ordered(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float f32_from_ordered(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

struct MelArgs {
    const float *pcm; const int32_t *n_samples; long stride;
    MelTables t; const int32_t *grp; // [n_mel][2]: first/last+1 4-groups with a non-zero filter tap
    int n_mel; int frames; float *mel32; unsigned *chunk_max;
};

__global__ __launch_bounds__(64 * FRAMES_PER_WG) void logmel_kernel(MelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float mel_lds[];
    float (*s_in)[N_FFT] = reinterpret_cast<float (*)[N_FFT]>(mel_lds);
    float (*s_a)[2 * N_FFT] = reinterpret_cast<float (*)[2 * N_FFT]>(mel_lds + FRAMES_PER_WG * N_FFT);
    float (*s_b)[2 * N_FFT] = reinterpret_cast<float (*)[2 * N_FFT]>(mel_lds + FRAMES_PER_WG * 3 * N_FFT);
    float *s_dc = mel_lds + FRAMES_PER_WG * 5 * N_FFT, *s_ds = s_dc + 25 * 25;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int b = blockIdx.y;
    const int frame = blockIdx.x * FRAMES_PER_WG + w;
    const bool live = frame < a.frames;
    for (int i = tid; i < 625; i += 64 * FRAMES_PER_WG) { s_dc[i] = a.t.dft_cos[i]; s_ds[i] = a.t.dft_sin[i]; }
    // windowed frame (samples at or beyond the clip's length are the zero padding of pcm_to_mel)
    {
        const int nv = a.n_samples[b];
        const float *x = a.pcm + (long)b * a.stride;
        const long off = (long)frame * HOP;
        for (int j = lane; j < N_FFT; j += 64) {
            long idx = off + j;
            float v = (live && idx < nv) ? x[idx] : 0.f;
            s_in[w][j] = a.t.hann[j] * v;
        }
    }
    __syncthreads();
    // 16 leaf DFTs of length 25: leaf id = sample offset (0..15), element i = sample id + 16 i
    for (int o = lane; o < 400; o += 64) {
        const int leaf = o / 25, k = o - leaf * 25;
        float re = 0.f, im = 0.f;
        const float *xin = &s_in[w][leaf];
        const float *c = &s_dc[k * 25], *s = &s_ds[k * 25];
#pragma unroll 5
        for (int j = 0; j < 25; j++) {
            float v = xin[16 * j];
            re += v * c[j];
            im -= v * s[j];
        }
        s_a[w][2 * (leaf * 25 + k)] = re;
        s_a[w][2 * (leaf * 25 + k) + 1] = im;
    }
    __syncthreads();
    // four radix-2 combine levels: children of length h at depth D+1 -> nodes of length 2h at depth D
    float *src = s_a[w], *dst = s_b[w];
    int tw_off = 0;
#pragma unroll 1
    for (int D = 3; D >= 0; D--) {
        const int h = 25 << (3 - D);   // child length: 25, 50, 100, 200
        const int nodes = 1 << D;      // nodes at depth D: 8, 4, 2, 1
        for (int o = lane; o < 200; o += 64) {
            const int node = o / h, k = o - node * h;
            const float *ef = src + 2 * (node * h), *of = src + 2 * ((node + nodes) * h);
            float *out = dst + 2 * (node * 2 * h);
            const float re = a.t.tw_cos[tw_off + k], im = a.t.tw_sin[tw_off + k];  // im = -sin(theta)
            const float re_odd = of[2 * k], im_odd = of[2 * k + 1];
            const float e_re = ef[2 * k], e_im = ef[2 * k + 1];
            out[2 * k] = e_re + re * re_odd - im * im_odd;
            out[2 * k + 1] = e_im + re * im_odd + im * re_odd;
            out[2 * (k + h)] = e_re - re * re_odd + im * im_odd;
            out[2 * (k + h) + 1] = e_im - re * im_odd - im * re_odd;
        }
        __syncthreads();
        tw_off += h;
        float *t = src; src = dst; dst = t;
    }
    // src now holds the 400-point spectrum.  Power, then fold the mirrored bins.
    float *pw = dst;  // reuse the other buffer: pw[0..399]
    for (int j = lane; j < N_FFT; j += 64) pw[j] = src[2 * j] * src[2 * j] + src[2 * j + 1] * src[2 * j + 1];
    __syncthreads();
    float *pf = src;  // folded power pf[0..200]
    for (int j = lane; j <= 200; j += 64) {
        float v = pw[j];
        if (j >= 1 && j < 200) v += pw[N_FFT - j];
        pf[j] = v;
    }
    __syncthreads();
    float lmax = -INFINITY;
    for (int m = lane; m < a.n_mel; m += 64) {
        const float *f = a.t.filters + (long)m * 201;
        float sum = 0.f;
        const int g0 = a.grp[2 * m], g1 = a.grp[2 * m + 1];
        for (int g = g0; g < g1; g++) {
            int k = 4 * g;
            sum += pf[k] * f[k] + pf[k + 1] * f[k + 1] + pf[k + 2] * f[k + 2] + pf[k + 3] * f[k + 3];
        }
        sum += pf[200] * f[200];  // remainder term k = 200
        float v = sum > 1e-10f ? log10f(sum) : -10.0f;  // log10f(1e-10f) is exactly -10 in the reference's libm
        if (live) {
            a.mel32[((long)b * a.n_mel + m) * a.frames + frame] = v;
            lmax = fmaxf(lmax, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if (lane == 0 && live) atomicMax(a.chunk_max + b, f32_ordered(lmax));
}

// v = max(mel, mmax - 8) / 4 + 1; writes the f32 candle layout in place and the fp16 conv1 image
// [B][frames + 2][128] (row t + 1, zero rows 0 and frames+1 and zero columns >= n_mel stay untouched)
__global__ __launch_bounds__(256) void mel_finish_kernel(float *mel32, const unsigned *chunk_max, half_t *img,
                                                         int n_mel, int frames, int normalise) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    float mmax = 0.f;
    if (normalise) mmax = f32_from_ordered(chunk_max[b]) - 8.0f;
    for (int r = ty; r < 32; r += 8) {
        int c = c0 + r, t = t0 + tx;
        float v = 0.f;
        if (c < n_mel && t < frames) {
            long idx = ((long)b * n_mel + c) * frames + t;
            v = mel32[idx];
            if (normalise) {
                v = (v > mmax ? v : mmax) / 4.0f + 1.0f;
                mel32[idx] = v;
            }
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int t = t0 + r, c = c0 + tx;
        if (t < frames && c < n_mel) img[((long)b * (frames + 2) + 1 + t) * NH_MELP + c] = (half_t)tile[tx][r];
    }
}

void launch_logmel_grp(const float *pcm, const int32_t *n_samples, long stride, const MelTables &t,
                       const int32_t *grp, int n_mel, int frames, float *mel32, unsigned *chunk_max, int B,
                       hipStream_t st) {
    MelArgs a{pcm, n_samples, stride, t, grp, n_mel, frames, mel32, chunk_max};
    dim3 grid((frames + FRAMES_PER_WG - 1) / FRAMES_PER_WG, B);
    static std::atomic<bool> attr_set[NH_MAX_DEVICES];   // hipFuncSetAttribute acts on the current device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= NH_MAX_DEVICES || !attr_set[dev].load(std::memory_order_acquire)) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(&logmel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, MEL_LDS_BYTES);
        if (dev >= 0 && dev < NH_MAX_DEVICES) attr_set[dev].store(true, std::memory_order_release);
    }
    static_assert((FRAMES_PER_WG * 5 * N_FFT + 2 * 625) * 4 <= MEL_LDS_BYTES, "log-mel staging does not fit the LDS");
    hipLaunchKernelGGL(logmel_kernel, grid, dim3(64 * FRAMES_PER_WG), MEL_LDS_BYTES, st, a);
}

void launch_mel_finish_ex(float *mel32, const unsigned *chunk_max, half_t *img, int B, int n_mel, int frames,
                          int normalise, hipStream_t st) {
    dim3 grid((frames + 31) / 32, (n_mel + 31) / 32, B);
    hipLaunchKernelGGL(mel_finish_kernel, grid, dim3(256), 0, st, mel32, chunk_max, img, n_mel, frames, normalise);
}
