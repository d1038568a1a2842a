// cumask: can the encoder of one batch and the decode steps of others share the GPU in SPACE?  Encoder GEMMs are persistent
// 512-thread workgroups that own a CU (all its registers, 145 KiB of LDS) for a whole launch (~0.5 ms), so a decode step
// of another batch -- a chain of ~40 dependent kernels of a few microseconds -- advances one kernel per GEMM boundary.
// The tool runs the fc2-shaped GEMM back to back on a stream confined to the first E CUs (hipExtStreamCreateWithCUMask;
// bit i = CU i / 8 of XCD i % 8) and a dependent chain of small streaming kernels (160 workgroups x 256 threads reading
// 16 MB each) on a stream confined to the other 256 - E; GEMM and chain time alone and together, masked and unmasked.
// `cumask destroy` also destroys the masked streams.  r02 (gpurun_out/cumask_destroy.txt): with the timing events that had been
// recorded on them still alive, destroying `se` returned and destroying `sd` never did.  Nothing was queued (run() ends in
// hipDeviceSynchronize), the two masks are disjoint and non-empty on every XCD (bits [0, E) and [E, 256): CUs 0 .. E/8 - 1 and
// E/8 .. 31 of each XCD), the GEMM's persistent grid is E workgroups = the CUs its stream can reach.  What was left referring
// to `sd` when it was destroyed were the events c0 / c1 (last recorded on it) -- so the teardown now follows the order that
// needs no such reference: synchronise, destroy the events recorded on the masked streams, then the streams, last created
// first.  `cumask destroy <order>`: order 0 = that (default), 1 = events first but streams in creation order, 2 = r02's order.
// Outcome (profiles/r02_cumask.txt): the chain runs 4.7x slower beside an unmasked GEMM and only 1.3x slower
// beside a masked one that itself loses 8 % -- but with the real decode step the job's throughput did not move, with
// masks (decode on D CUs is per-CU fetch bound: 340 us per token on 128 CUs, 466 on 64, against 243 on 256) or with the
// GEMM grid alone cut to 256 - D workgroups (+-1 % for D = 16 .. 96): three batches in flight already keep the chip busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../norma_amd/csrc/nh_kernels.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void fill_rand(half_t *p, size_t n, unsigned seed) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (half_t)(((int)(x & 0xffff) - 32768) * (1.0f / 32768.f));
    }
}
__global__ __launch_bounds__(256) void stream_read(const u32x4 *src, size_t n16, unsigned *sink) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n16; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = src[i]; acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
    if (acc == 0x12345678u) sink[0] = acc;
}
static hipStream_t masked_stream(int first, int count) {
    uint32_t mask[8] = {0};
    for (int i = first; i < first + count; i++) mask[i >> 5] |= 1u << (i & 31);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
    return s;
}
int main(int argc, char **argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    const int M = 48000, N = 1280, K = 5120, gemm_reps = 40, chain = 400;
    half_t *A, *W; float *X, *bias; u32x4 *buf; unsigned *sink;
    CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&X, (size_t)M * N * 4));
    CK(hipMalloc(&bias, N * 4)); CK(hipMemset(bias, 0, N * 4)); CK(hipMemset(X, 0, (size_t)M * N * 4));
    const size_t chunk = 16u << 20, nchunks = 32;
    CK(hipMalloc(&buf, chunk * nchunks)); CK(hipMemset(buf, 1, chunk * nchunks)); CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, 0, A, (size_t)M * K, 1u);
    hipLaunchKernelGGL(fill_rand, dim3(2048), dim3(256), 0, 0, W, (size_t)N * K, 7u);
    CK(hipDeviceSynchronize());
    GemmParams p{};
    p.A = A; p.lda = K; p.a_rpb = M; p.W = W; p.bias = bias; p.M = M; p.N = N; p.K = K; p.epi = EPI_RESID_F32;
    p.out[0] = X; p.seg_n = N; p.ldo = N; p.o_rpb = M; p.vt_seg = -1; p.S = 1500; p.H = 20;
    hipEvent_t g0, g1, c0, c1;
    auto make_events = [&]() { hipEventCreate(&g0); hipEventCreate(&g1); hipEventCreate(&c0); hipEventCreate(&c1); };
    auto drop_events = [&]() { CK(hipEventDestroy(g0)); CK(hipEventDestroy(g1)); CK(hipEventDestroy(c0)); CK(hipEventDestroy(c1)); };
    make_events();
    const int order = argc > 2 ? atoi(argv[2]) : 0;
    auto run = [&](const char *name, hipStream_t sg, hipStream_t sc, int cus, bool do_gemm, bool do_chain) {
        p.cus = cus;
        if (do_gemm) { for (int i = 0; i < 3; i++) launch_gemm(p, sg); }
        CK(hipDeviceSynchronize());
        if (do_gemm) { hipEventRecord(g0, sg); for (int i = 0; i < gemm_reps; i++) launch_gemm(p, sg); hipEventRecord(g1, sg); }
        if (do_chain) {
            hipEventRecord(c0, sc);
            for (int i = 0; i < chain; i++)
                hipLaunchKernelGGL(stream_read, dim3(160), dim3(256), 0, sc, (const u32x4 *)((const char *)buf + (i % nchunks) * chunk), chunk / 16, sink);
            hipEventRecord(c1, sc);
        }
        CK(hipDeviceSynchronize());
        float gm = 0, cm = 0;
        if (do_gemm) hipEventElapsedTime(&gm, g0, g1);
        if (do_chain) hipEventElapsedTime(&cm, c0, c1);
        printf("%-52s gemm %7.1f us/launch   chain %6.2f us/kernel (%5.2f TB/s)\n", name, gm * 1e3 / gemm_reps, cm * 1e3 / chain,
               do_chain ? chunk * (double)chain / (cm * 1e-3) / 1e12 : 0.0);
    };
    hipStream_t a, b; CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    run("unmasked: gemm alone", a, b, 0, true, false);
    run("unmasked: chain alone", a, b, 0, false, true);
    run("unmasked: both", a, b, 0, true, true);
    for (int D : {32, 64, 96}) {
        const int E = 256 - D;
        hipStream_t se = masked_stream(0, E), sd = masked_stream(E, D);
        printf("streams for E=%d D=%d created\n", E, D);
        char nm[96];
        snprintf(nm, sizeof nm, "E=%d D=%d: gemm alone (masked)", E, D); run(nm, se, sd, E, true, false);
        snprintf(nm, sizeof nm, "E=%d D=%d: chain alone (masked)", E, D); run(nm, se, sd, E, false, true);
        snprintf(nm, sizeof nm, "E=%d D=%d: both", E, D); run(nm, se, sd, E, true, true);
        if (argc > 1) {
            CK(hipDeviceSynchronize());
            if (order != 2) { drop_events(); printf("destroyed the events recorded on se / sd\n"); }
            if (order == 0) { CK(hipStreamDestroy(sd)); printf("destroyed sd\n"); CK(hipStreamDestroy(se)); printf("destroyed se\n"); }
            else { CK(hipStreamDestroy(se)); printf("destroyed se\n"); CK(hipStreamDestroy(sd)); printf("destroyed sd\n"); }
            if (order != 2) make_events();
        }
    }
    return 0;
}
