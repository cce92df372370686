// Microbenchmark (GPU box only): where does a stage of the LDS-tiled fp32 MFMA GEMM (k_gemm_tn2<3,4>, dW_hh shape) spend its time?
//   exp bit 0: no global loads in the loop; bit 1: no LDS stores; bit 2: no barrier
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>
#include <cvae_kernels.h>
#include <cvae_train_kernels.h>

template <int TM, int TN>
__global__ __launch_bounds__(256) void k_tn2x(const float* __restrict__ A, long lda, const float* __restrict__ Bm, long ldb,
                                              float* __restrict__ C, long ldc, int M, int N1, int N2, int exp) {
    using G = GemmTileCfg<TM, TN>;
    float* sm = (float*)CVAE_SMEM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int a0 = blockIdx.y * G::BM, b0 = blockIdx.x * G::BN;
    long aoff[G::NA], boff[G::NB];
    int asm_[G::NA], bsm_[G::NB];
    bool aok[G::NA], bok[G::NB];
#pragma unroll
    for (int u = 0; u < G::NA; ++u) {
        const int e = tid + 256 * u, r = e / (G::BM / 4), c = 4 * (e % (G::BM / 4));
        aok[u] = e < 4 * G::BM && a0 + c < N1;
        aoff[u] = (long)r * lda + a0 + c;
        asm_[u] = r * G::LDA + c;
    }
#pragma unroll
    for (int u = 0; u < G::NB; ++u) {
        const int e = tid + 256 * u, r = e / (G::BN / 4), c = 4 * (e % (G::BN / 4)), n2 = b0 + c;
        bok[u] = e < 4 * G::BN && n2 < N2;
        boff[u] = (long)r * ldb + n2;
        bsm_[u] = 16 * G::LDA + r * G::LDB + c;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ga[G::NA], gb[G::NB];
    auto gload = [&](int m0) {
#pragma unroll
        for (int u = 0; u < G::NA; ++u) ga[u] = aok[u] ? *(const f32x4*)(A + (long)m0 * lda + aoff[u]) : zero4;
#pragma unroll
        for (int u = 0; u < G::NB; ++u) gb[u] = bok[u] ? *(const f32x4*)(Bm + (long)m0 * ldb + boff[u]) : zero4;
    };
    auto sstore = [&](int stage) {
        float* st = sm + stage * G::STAGE;
#pragma unroll
        for (int u = 0; u < G::NA; ++u)
            if (tid + 256 * u < 4 * G::BM) *(f32x4*)(st + asm_[u]) = ga[u];
#pragma unroll
        for (int u = 0; u < G::NB; ++u)
            if (tid + 256 * u < 4 * G::BN) *(f32x4*)(st + bsm_[u]) = gb[u];
    };
    gload(0);
    sstore(0);
    __syncthreads();
    const bool xl = exp & 1, xs = exp & 2, xb = exp & 4;
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int stage = (m0 >> 4) & 1;
        const bool more = m0 + 16 < M;
        if (more && !xl) gload(m0 + 16);
        cvae_gemm_tile_stage<TM, TN>(sm + stage * G::STAGE, sm + stage * G::STAGE + 16 * G::LDA, wm, wn, lr, kq, acc);
        if (more && !xs) sstore(stage ^ 1);
        if (!xb) __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = b0 + wn * 16 * TN + 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = a0 + wm * 16 * TM + 16 * i + 4 * kq + r;
                if (rowi < N1 && col < N2) C[(long)rowi * ldc + col] = acc[i][j][r];
            }
        }
}

// variant: both operand tiles K-CONTIGUOUS in LDS ([n][16 k + 4 pad]); a lane's four sub-step operands are ONE ds_read_b128,
// all fragment reads of a stage are issued before its MFMAs; global tiles transposed by four coalesced dword loads per lane
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_tn2y(const float* __restrict__ A, long lda, const float* __restrict__ Bm, long ldb,
                                              float* __restrict__ C, long ldc, int M, int N1, int N2, int exp) {
    constexpr int BM = 32 * TM, BN = 32 * TN, LDK = 20, STAGE = (BM + BN) * LDK;
    constexpr int NA = (4 * BM + 255) / 256, NB = (4 * BN + 255) / 256;
    float* sm = (float*)CVAE_SMEM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int a0 = blockIdx.y * BM, b0 = blockIdx.x * BN;
    long aoff[NA], boff[NB];
    int asm_[NA], bsm_[NB];
    bool aok[NA], bok[NB];
#pragma unroll
    for (int u = 0; u < NA; ++u) {
        const int e = tid + 256 * u, n = e % BM, kg = e / BM;
        aok[u] = e < 4 * BM && a0 + n < N1;
        aoff[u] = (long)(4 * kg) * lda + a0 + n;
        asm_[u] = n * LDK + 4 * kg;
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int e = tid + 256 * u, n = e % BN, kg = e / BN;
        bok[u] = e < 4 * BN && b0 + n < N2;
        boff[u] = (long)(4 * kg) * ldb + b0 + n;
        bsm_[u] = BM * LDK + n * LDK + 4 * kg;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ga[NA], gb[NB];
    auto gload = [&](int m0) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            ga[u] = zero4;
            if (aok[u]) {
                const float* s = A + (long)m0 * lda + aoff[u];
                ga[u] = (f32x4){s[0], s[lda], s[2 * lda], s[3 * lda]};
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            gb[u] = zero4;
            if (bok[u]) {
                const float* s = Bm + (long)m0 * ldb + boff[u];
                gb[u] = (f32x4){s[0], s[ldb], s[2 * ldb], s[3 * ldb]};
            }
        }
    };
    auto sstore = [&](int stage) {
        float* st = sm + stage * STAGE;
#pragma unroll
        for (int u = 0; u < NA; ++u)
            if (tid + 256 * u < 4 * BM) *(f32x4*)(st + asm_[u]) = ga[u];
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (tid + 256 * u < 4 * BN) *(f32x4*)(st + bsm_[u]) = gb[u];
    };
    gload(0);
    sstore(0);
    __syncthreads();
    const bool xl = exp & 1, xs = exp & 2, xb = exp & 4;
    for (int m0 = 0; m0 < M; m0 += 16) {
        const int stage = (m0 >> 4) & 1;
        const bool more = m0 + 16 < M;
        const float* As = sm + stage * STAGE;
        const float* Bs = As + BM * LDK;
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const f32x4*)(As + (wm * 16 * TM + 16 * i + lr) * LDK + 4 * kq);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const f32x4*)(Bs + (wn * 16 * TN + 16 * j + lr) * LDK + 4 * kq);
        if (more && !xl) gload(m0 + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = cvae_mfma_16x16x4(a[i][s], b[j][s], acc[i][j]);
        if (more && !xs) sstore(stage ^ 1);
        if (!xb) __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = b0 + wn * 16 * TN + 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = a0 + wm * 16 * TM + 16 * i + 4 * kq + r;
                if (rowi < N1 && col < N2) C[(long)rowi * ldc + col] = acc[i][j][r];
            }
        }
}

int main() {
    const int M = 5120, N1 = 3072, N2 = 1024;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * N1 * 4);
    hipMalloc(&B, (size_t)M * N2 * 4);
    hipMalloc(&C, (size_t)N1 * N2 * 4);
    hipMemset(A, 0, (size_t)M * N1 * 4);
    hipMemset(B, 0, (size_t)M * N2 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        hipEventRecord(e0, 0);
        const int n = 10;
        for (int i = 0; i < n; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-60s %8.1f us  %6.1f TFLOP/s\n", name, 1e3 * ms / n, 2.0 * M * N1 * N2 / (1e9 * ms / n));
    };
    const char* names[8] = {"baseline", "no global loads", "no LDS stores", "no loads, no stores", "no barrier", "no loads, no barrier",
                            "no stores, no barrier", "LDS reads + MFMA only"};
    for (int exp = 0; exp < 8; ++exp) {
        char nm[128];
        snprintf(nm, sizeof nm, "tn2<3,4> 96x128 tiles, 256 blocks: %s", names[exp]);
        run(nm, [&] {
            hipLaunchKernelGGL((k_tn2x<3, 4>), dim3(N2 / 128, N1 / 96), dim3(256), (GemmTileCfg<3, 4>::lds_bytes), 0, A, (long)N1, B, (long)N2, C,
                               (long)N2, M, N1, N2, exp);
        });
    }
    for (int exp = 0; exp < 8; exp += 7) {
        char nm[128];
        snprintf(nm, sizeof nm, "tn2<2,4> 64x128 tiles, 384 blocks: %s", names[exp]);
        run(nm, [&] {
            hipLaunchKernelGGL((k_tn2x<2, 4>), dim3(N2 / 128, N1 / 64), dim3(256), (GemmTileCfg<2, 4>::lds_bytes), 0, A, (long)N1, B, (long)N2, C,
                               (long)N2, M, N1, N2, exp);
        });
        snprintf(nm, sizeof nm, "tn2<2,2> 64x64 tiles, 768 blocks: %s", names[exp]);
        run(nm, [&] {
            hipLaunchKernelGGL((k_tn2x<2, 2>), dim3(N2 / 64, N1 / 64), dim3(256), (GemmTileCfg<2, 2>::lds_bytes), 0, A, (long)N1, B, (long)N2, C,
                               (long)N2, M, N1, N2, exp);
        });
        snprintf(nm, sizeof nm, "tn2<4,4> 128x128 tiles, 192 blocks: %s", names[exp]);
        run(nm, [&] {
            hipLaunchKernelGGL((k_tn2x<4, 4>), dim3(N2 / 128, N1 / 128), dim3(256), (GemmTileCfg<4, 4>::lds_bytes), 0, A, (long)N1, B, (long)N2, C,
                               (long)N2, M, N1, N2, exp);
        });
    }
    for (int exp = 0; exp < 8; exp += 7) {
        char nm[128];
        snprintf(nm, sizeof nm, "b128 <3,4> 96x128 tiles, 256 blocks: %s", names[exp]);
        run(nm, [&] { hipLaunchKernelGGL((k_tn2y<3, 4>), dim3(N2 / 128, N1 / 96), dim3(256), 2 * (96 + 128) * 20 * 4, 0, A, (long)N1, B, (long)N2, C, (long)N2, M, N1, N2, exp); });
        snprintf(nm, sizeof nm, "b128 <2,4> 64x128 tiles, 384 blocks: %s", names[exp]);
        run(nm, [&] { hipLaunchKernelGGL((k_tn2y<2, 4>), dim3(N2 / 128, N1 / 64), dim3(256), 2 * (64 + 128) * 20 * 4, 0, A, (long)N1, B, (long)N2, C, (long)N2, M, N1, N2, exp); });
        snprintf(nm, sizeof nm, "b128 <2,2> 64x64 tiles, 768 blocks: %s", names[exp]);
        run(nm, [&] { hipLaunchKernelGGL((k_tn2y<2, 2>), dim3(N2 / 64, N1 / 64), dim3(256), 2 * (64 + 64) * 20 * 4, 0, A, (long)N1, B, (long)N2, C, (long)N2, M, N1, N2, exp); });
        snprintf(nm, sizeof nm, "b128 <4,4> 128x128 tiles, 192 blocks: %s", names[exp]);
        run(nm, [&] { hipLaunchKernelGGL((k_tn2y<4, 4>), dim3(N2 / 128, N1 / 128), dim3(256), 2 * (128 + 128) * 20 * 4, 0, A, (long)N1, B, (long)N2, C, (long)N2, M, N1, N2, exp); });
        snprintf(nm, sizeof nm, "b128 <3,2> 96x64 tiles, 512 blocks: %s", names[exp]);
        run(nm, [&] { hipLaunchKernelGGL((k_tn2y<3, 2>), dim3(N2 / 64, N1 / 96), dim3(256), 2 * (96 + 64) * 20 * 4, 0, A, (long)N1, B, (long)N2, C, (long)N2, M, N1, N2, exp); });
    }
    return 0;
}
