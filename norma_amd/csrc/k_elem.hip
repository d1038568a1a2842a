// k_elem.hip -- LayerNorm and decoder input embedding (HBM-bound row kernels, gfx950).
//
// LayerNorm replaces candle_nn::LayerNorm (eps 1e-5, f32 statistics) at every use inside
// AudioEncoder / TextDecoder / ResidualAttentionBlock::forward (reached from
// src/models/whisper/model.rs:455-476).  The residual stream stays f32 in HBM; LayerNorm is the
// single place where it is rounded to fp16 (the MFMA operand type).  One wavefront per row,
// 16-byte loads, wavefront-shuffle reductions, no LDS.
#include <stdlib.h>

#include "nh_kernels.h"

#define LN_MAXV 5  // d_model <= 1280 -> at most 5 float4 per lane

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One wave per row, grid-stride over the rows.  The affine parameters are fetched once per wave, together with its first
// row, and stay in registers (re-reading them per row tripled the bytes going through the vector L1: 99 -> 7x us at
// M = 48 000, d = 1280); with B rows (decode, teacher-forced views) every wave has exactly one row.
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                        const float *__restrict__ b, half_t *__restrict__ y,
                                                        float *__restrict__ y32, int M, int d) {
    const int lane = threadIdx.x & 63;
    const int nv = d >> 2, nwaves = gridDim.x * 4;
    const f32x4 *wr = reinterpret_cast<const f32x4 *>(w), *br = reinterpret_cast<const f32x4 *>(b);
    f32x4 wv[LN_MAXV], bv[LN_MAXV];
    int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
#pragma unroll
    for (int i = 0; i < LN_MAXV; i++) {
        const int c = lane + 64 * i;
        wv[i] = wr[c < nv ? c : nv - 1]; bv[i] = br[c < nv ? c : nv - 1];
    }
    for (; row < M; row += nwaves) {
        const f32x4 *xr = reinterpret_cast<const f32x4 *>(x + (long)row * d);
        f32x4 v[LN_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            v[i] = xr[c < nv ? c : nv - 1];  // unconditional, clamped
            if (c >= nv) v[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        const float mean = wave_sum(s) / (float)d;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            if (lane + 64 * i < nv) {
                f32x4 t = v[i] - mean;
                s2 += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
            }
        }
        const float inv = 1.0f / sqrtf(wave_sum(s2) / (float)d + 1e-5f);
#pragma unroll
        for (int i = 0; i < LN_MAXV; i++) {
            const int c = lane + 64 * i;
            if (c < nv) {
                f32x4 o = (v[i] - mean) * inv * wv[i] + bv[i];
                half4 h = {(half_t)o[0], (half_t)o[1], (half_t)o[2], (half_t)o[3]};
                *reinterpret_cast<half4 *>(y + (long)row * d + 4 * c) = h;
                if (y32) *reinterpret_cast<f32x4 *>(y32 + (long)row * d + 4 * c) = o;
            }
        }
    }
}

void launch_layernorm(const float *x, const float *w, const float *b, half_t *y, float *y32, int M, int d,
                      hipStream_t st) {
    const int max_blocks = 2048;  // grid-stride rows: the affine parameters are loaded once per workgroup
    int blocks = (M + 3) / 4;
    if (blocks > max_blocks) blocks = max_blocks;
    hipLaunchKernelGGL(layernorm_kernel, dim3(blocks), dim3(256), 0, st, x, w, b, y, y32, M, d);
}

// TextDecoder::forward input: token_embedding(x) + positional_embedding[0..T]  
__global__ __launch_bounds__(256) void embed_kernel(const int32_t *__restrict__ tokens, int tok_stride,
                                                    const half_t *__restrict__ E, const half_t *__restrict__ P,
                                                    float *__restrict__ x, int Tn, int t0,
                                                    const int32_t *__restrict__ pos_ptr, int d) {
    const int r = blockIdx.x;  // row = b * Tn + i
    const int b = r / Tn, i = r - b * Tn;
    if (pos_ptr) t0 = pos_ptr[b];  // per-sequence position
    const int tok = tokens[(long)b * tok_stride + t0 + i];
    const half_t *e = E + (long)tok * d, *p = P + (long)(t0 + i) * d;
    for (int c = threadIdx.x * 4; c < d; c += 256 * 4) {
        half4 ev = *reinterpret_cast<const half4 *>(e + c), pv = *reinterpret_cast<const half4 *>(p + c);
        f32x4 o = {(float)ev[0] + (float)pv[0], (float)ev[1] + (float)pv[1], (float)ev[2] + (float)pv[2],
                   (float)ev[3] + (float)pv[3]};
        *reinterpret_cast<f32x4 *>(x + (long)r * d + c) = o;
    }
}

void launch_embed(const int32_t *tokens, int tok_stride, const half_t *E, const half_t *P, float *x, int B,
                  int Tn, int t0, const int32_t *pos_ptr, int d, hipStream_t st) {
    hipLaunchKernelGGL(embed_kernel, dim3(B * Tn), dim3(256), 0, st, tokens, tok_stride, E, P, x, Tn, t0, pos_ptr, d);
}

// dasp_sample's conversions to f32 (the capture side of the reference converts every native sample type to Model::Data with
// Sample::to_sample, src/lib.rs:180,207; the admissible types are src/dtype.rs:37-45).  Signed: s / 2^(bits - 1); unsigned:
// (s - 2^(bits - 1)) / 2^(bits - 1), the subtraction in integers; f64: round to nearest.  Integer -> float conversions
// round to nearest even (v_cvt_f32_i32, and the compiler's i64 sequence), like Rust's `as f32`; the divisions are by
// powers of two, i.e. exact.
template <typename T> __device__ __forceinline__ float sample_to_f32(T v);
template <> __device__ __forceinline__ float sample_to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float sample_to_f32<double>(double v) { return (float)v; }
template <> __device__ __forceinline__ float sample_to_f32<int8_t>(int8_t v) { return (float)v * (1.0f / 128.0f); }
template <> __device__ __forceinline__ float sample_to_f32<int16_t>(int16_t v) { return (float)v * (1.0f / 32768.0f); }
template <> __device__ __forceinline__ float sample_to_f32<int32_t>(int32_t v) { return (float)v * (1.0f / 2147483648.0f); }
template <> __device__ __forceinline__ float sample_to_f32<int64_t>(int64_t v) { return (float)v * (1.0f / 9223372036854775808.0f); }
template <> __device__ __forceinline__ float sample_to_f32<uint8_t>(uint8_t v) { return (float)((int)v - 128) * (1.0f / 128.0f); }
template <> __device__ __forceinline__ float sample_to_f32<uint16_t>(uint16_t v) { return (float)((int)v - 32768) * (1.0f / 32768.0f); }
template <> __device__ __forceinline__ float sample_to_f32<uint32_t>(uint32_t v) { return (float)(int32_t)(v ^ 0x80000000u) * (1.0f / 2147483648.0f); }
template <> __device__ __forceinline__ float sample_to_f32<uint64_t>(uint64_t v) { return (float)(int64_t)(v ^ 0x8000000000000000ull) * (1.0f / 9223372036854775808.0f); }

template <typename T>
__global__ __launch_bounds__(256) void convert_samples_kernel(const T *__restrict__ src, float *__restrict__ dst, long count) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < count; i += (long)gridDim.x * 256L) dst[i] = sample_to_f32<T>(src[i]);
}

void launch_convert_samples(const void *src, float *dst, long count, int dtype, hipStream_t st) {
    if (count <= 0) return;
    long nb = (count + 255) / 256; if (nb > 4096) nb = 4096;
    const dim3 grid((unsigned)nb), block(256);
#define CONV(T) hipLaunchKernelGGL(convert_samples_kernel<T>, grid, block, 0, st, reinterpret_cast<const T *>(src), dst, count)
    switch (dtype) {
        case 0: CONV(float); break;     case 1: CONV(double); break;
        case 2: CONV(int8_t); break;    case 3: CONV(int16_t); break;  case 4: CONV(int32_t); break;  case 5: CONV(int64_t); break;
        case 6: CONV(uint8_t); break;   case 7: CONV(uint16_t); break; case 8: CONV(uint32_t); break; default: CONV(uint64_t); break;
    }
#undef CONV
}
