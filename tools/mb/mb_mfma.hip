// Microbenchmark (GPU box only): the matrix phase of a task of k_train_fwd_steps_x3h in isolation -- 8 chunks x 24
// v_mfma_f32_16x16x32_f16 per wave against register-resident weights (256 registers), operands already in registers (no global
// loads, no flags) -- with the accompanying work switched on piece by piece:
//   MODE 0  MFMAs only (operands and masked operands are loop-invariant registers)
//   MODE 1  + per-chunk bf8 decode of the third limb and the three dropout-mask ANDs (the VALU work of the real loop)
//   MODE 2  + the four third-limb weight fragments read from LDS per chunk (where the compiler puts them)
//   MODE 3  = 2 with the fragments requested one chunk ahead
// CH = accumulator chains per column tile: 4 as in the kernel (S0, S1, S2a / S2b pattern), 6 = one accumulator per product.
// Reports shader cycles per task (192 MFMAs = 3,072 matrix-pipe cycles) of block 0.
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

struct cvae_m4 { unsigned q[4]; };
__device__ __forceinline__ cvae_m4 expand_bits(unsigned x) {
    const unsigned y = x | (x << 12);
    cvae_m4 m;
#pragma unroll
    for (int q = 0; q < 4; ++q) m.q[q] = ((y >> q) & 0x00010001u) * 0xFFFFu;
    return m;
}
__device__ __forceinline__ f32x4 mask_h8(f32x4 v, const cvae_m4& m) {
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, (float)v[q]) & m.q[q]);
    return o;
}

template <int MODE, int CH>
__global__ __launch_bounds__(256, 1) void k_mm(const float* wsrc, const float* osrc, float* dst, long long* cyc, int iters) {
    constexpr int C32W = 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* w2l = (float*)CVAE_SMEM;        // [4 waves][2 n][2 paths][8][64 lanes][4]
    f32x4 w0[2][2][C32W], w1[2][2][C32W];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
            for (int ci = 0; ci < C32W; ++ci) {
                const float* w = wsrc + ((((n * 2 + pa) * C32W + ci) * 3) * 256 + lane * 4);
                w0[n][pa][ci] = *(const f32x4*)w;
                w1[n][pa][ci] = *(const f32x4*)(w + 256);
                *(f32x4*)(w2l + ((((wave * 2 + n) * 2 + pa) * C32W + ci) * 64 + lane) * 4) = *(const f32x4*)(w + 512);
            }
    __syncthreads();
    const float* w2w = w2l + (long)wave * 4 * C32W * 256 + lane * 4;
    f32x4 a0[C32W], a1[C32W];
    f32x2 a2[C32W];
#pragma unroll
    for (int ci = 0; ci < C32W; ++ci) {
        a0[ci] = *(const f32x4*)(osrc + (ci * 3) * 256 + lane * 4);
        a1[ci] = *(const f32x4*)(osrc + (ci * 3 + 1) * 256 + lane * 4);
        a2[ci] = *(const f32x2*)(osrc + (ci * 3 + 2) * 256 + lane * 2);
    }
    f32x2 mraw = *(const f32x2*)(osrc + 8192 + lane * 2);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = cvae_clock();
    for (int it = 0; it < iters; ++it) {
        f32x4 s[2][6];
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 6; ++q) s[n][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 w2h[2], w2o[2];
        if (MODE == 3) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                w2h[n] = *(const f32x4*)(w2w + ((n * 2 + 0) * C32W) * 256);
                w2o[n] = *(const f32x4*)(w2w + ((n * 2 + 1) * C32W) * 256);
            }
        }
#pragma unroll
        for (int ci = 0; ci < C32W; ++ci) {
            f32x4 l0 = a0[ci], l1 = a1[ci], l2, m0, m1, m2;
            if (MODE >= 1) {
                l2 = cvae_bf8x8_to_h8(a2[ci]);
                const float mw = mraw[ci >> 2];
                const cvae_m4 mk = expand_bits((__builtin_bit_cast(unsigned, mw) >> (8 * (ci & 3))) & 0xffu);
                m0 = mask_h8(l0, mk); m1 = mask_h8(l1, mk); m2 = mask_h8(l2, mk);
            } else {
                l2 = a1[(ci + 1) % C32W]; m0 = a0[(ci + 2) % C32W]; m1 = a1[(ci + 3) % C32W]; m2 = a0[(ci + 4) % C32W];
            }
            f32x4 n2h[2], n2o[2];
            if (MODE == 3) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    n2h[n] = w2h[n]; n2o[n] = w2o[n];
                    if (ci + 1 < C32W) {
                        n2h[n] = *(const f32x4*)(w2w + ((n * 2 + 0) * C32W + ci + 1) * 256);
                        n2o[n] = *(const f32x4*)(w2w + ((n * 2 + 1) * C32W + ci + 1) * 256);
                    }
                }
                cvae_sched_fence();
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                f32x4 wh2, wo2;
                if (MODE == 2) {
                    wh2 = *(const f32x4*)(w2w + ((n * 2 + 0) * C32W + ci) * 256);
                    wo2 = *(const f32x4*)(w2w + ((n * 2 + 1) * C32W + ci) * 256);
                } else if (MODE == 3) {
                    wh2 = w2h[n]; wo2 = w2o[n];
                } else {
                    wh2 = w1[n][0][(ci + 1) % C32W]; wo2 = w1[n][1][(ci + 1) % C32W];
                }
                if (CH == 4) {
                    s[n][0] = cvae_mfma_16x16x32_f16(l0, w0[n][0][ci], s[n][0]);
                    s[n][1] = cvae_mfma_16x16x32_f16(l0, w1[n][0][ci], s[n][1]);
                    s[n][2] = cvae_mfma_16x16x32_f16(l1, w1[n][0][ci], s[n][2]);
                    s[n][3] = cvae_mfma_16x16x32_f16(l0, wh2, s[n][3]);
                    s[n][1] = cvae_mfma_16x16x32_f16(l1, w0[n][0][ci], s[n][1]);
                    s[n][2] = cvae_mfma_16x16x32_f16(l2, w0[n][0][ci], s[n][2]);
                    s[n][0] = cvae_mfma_16x16x32_f16(m0, w0[n][1][ci], s[n][0]);
                    s[n][1] = cvae_mfma_16x16x32_f16(m0, w1[n][1][ci], s[n][1]);
                    s[n][2] = cvae_mfma_16x16x32_f16(m1, w1[n][1][ci], s[n][2]);
                    s[n][3] = cvae_mfma_16x16x32_f16(m0, wo2, s[n][3]);
                    s[n][1] = cvae_mfma_16x16x32_f16(m1, w0[n][1][ci], s[n][1]);
                    s[n][2] = cvae_mfma_16x16x32_f16(m2, w0[n][1][ci], s[n][2]);
                } else {
                    s[n][0] = cvae_mfma_16x16x32_f16(l0, w0[n][0][ci], s[n][0]);
                    s[n][1] = cvae_mfma_16x16x32_f16(l0, w1[n][0][ci], s[n][1]);
                    s[n][2] = cvae_mfma_16x16x32_f16(l1, w1[n][0][ci], s[n][2]);
                    s[n][3] = cvae_mfma_16x16x32_f16(l0, wh2, s[n][3]);
                    s[n][4] = cvae_mfma_16x16x32_f16(l1, w0[n][0][ci], s[n][4]);
                    s[n][5] = cvae_mfma_16x16x32_f16(l2, w0[n][0][ci], s[n][5]);
                    s[n][0] = cvae_mfma_16x16x32_f16(m0, w0[n][1][ci], s[n][0]);
                    s[n][1] = cvae_mfma_16x16x32_f16(m0, w1[n][1][ci], s[n][1]);
                    s[n][2] = cvae_mfma_16x16x32_f16(m1, w1[n][1][ci], s[n][2]);
                    s[n][3] = cvae_mfma_16x16x32_f16(m0, wo2, s[n][3]);
                    s[n][4] = cvae_mfma_16x16x32_f16(m1, w0[n][1][ci], s[n][4]);
                    s[n][5] = cvae_mfma_16x16x32_f16(m2, w0[n][1][ci], s[n][5]);
                }
            }
            cvae_sched_fence();
            if (MODE == 3) {
#pragma unroll
                for (int n = 0; n < 2; ++n) { w2h[n] = n2h[n]; w2o[n] = n2o[n]; }
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int q = 0; q < 6; ++q) acc += s[n][q];
        mraw[0] += acc[0] * 1e-30f;      // (keeps the iterations dependent on each other)
    }
    const long long t1 = cvae_clock();
    dst[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 400, nblk = 256;
    float *wsrc, *osrc, *dst;
    long long* cyc;
    hipMalloc(&wsrc, 4 << 20); hipMemset(wsrc, 0, 4 << 20);
    hipMalloc(&osrc, 1 << 20); hipMemset(osrc, 0, 1 << 20);
    hipMalloc(&dst, nblk * 256 * 4);
    hipMalloc(&cyc, nblk * 8);
    std::vector<long long> c(nblk);
    auto report = [&](const char* name) {
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(c.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : c) s += (double)v;
        printf("%-64s %8.1f cycles per task (192 MFMAs: %.1f per MFMA)  %s\n", name, s / nblk / iters, s / nblk / iters / 192.0, e == hipSuccess ? "" : hipGetErrorString(e));
    };
    const size_t lds = 4 * 2 * 2 * 8 * 64 * 4 * sizeof(float);
#define RUN(M, C, NAME) hipLaunchKernelGGL((k_mm<M, C>), dim3(nblk), dim3(256), lds, 0, wsrc, osrc, dst, cyc, iters); report(NAME);
    for (int rep = 0; rep < 2; ++rep) {
        RUN(0, 4, "MFMAs only, 4 chains per column tile (kernel's pattern)");
        RUN(0, 6, "MFMAs only, 6 chains per column tile");
        RUN(1, 4, "+ bf8 decode + mask ANDs, 4 chains");
        RUN(1, 6, "+ bf8 decode + mask ANDs, 6 chains");
        RUN(2, 4, "+ third-limb weights from LDS at use, 4 chains");
        RUN(3, 4, "+ third-limb weights from LDS one chunk ahead, 4 chains");
        RUN(3, 6, "+ third-limb weights from LDS one chunk ahead, 6 chains");
    }
    return 0;
}
