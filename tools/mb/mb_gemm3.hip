// Microbenchmark (GPU box only): the limb GEMM (cvae_gemm3.h: fp32-exact products as three fp16 limbs, six f16 MFMAs per product)
// on the big shapes of the training step, with its split passes, against fp64 on sampled entries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cyclevae-vc_amd/csrc -I include -I tools/mb tools/mb/mb_gemm3.hip -o tools/mb/mb_gemm3
#include <cvae_intrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "cvae_gemm3.h"

static long up(long x, long m) { return (x + m - 1) / m * m; }

// C[M][N] = sum_k A(m,k) B(n,k); ta / tb: operand stored transposed ([k][row])
static void run(const char* what, int M, int N, int K, int ta, int tb, int kz, int eight = 0) {
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    srand(1);
    for (auto& v : A) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *dC, *dP;
    const int Mp = (int)up(M, 128), Np = (int)up(N, 128), Kp = (int)up(K, 32);
    unsigned short *A3, *B3;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4); hipMalloc(&dP, (size_t)kz * M * N * 4);
    hipMalloc(&A3, (size_t)Mp * Kp * 6); hipMalloc(&B3, (size_t)Np * Kp * 6);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1, e2;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    Gemm3Epi ep = {nullptr, 0, nullptr, 0, 0, 0};
    float ms_split = 0, ms_gemm = 0;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_split3, dim3(Kp / 32, Mp / 128), dim3(256), 0, 0, dA, ta ? (long)M : (long)K, M, K, ta, 0, 0L, A3, Mp, Kp);
        hipLaunchKernelGGL(k_split3, dim3(Kp / 32, Np / 128), dim3(256), 0, 0, dB, tb ? (long)N : (long)K, N, K, tb, 0, 0L, B3, Np, Kp);
        hipEventRecord(e1);
        if (eight)
            hipLaunchKernelGGL(k_gemm3_nt8, dim3(Np / 128, Mp / 128, kz), dim3(512), 4 * CVAE_G3_TILE_HALVES * 2, 0, A3, B3, dC, (long)N, M, N, Kp, ep,
                               kz > 1 ? dP : nullptr);
        else
        hipLaunchKernelGGL(k_gemm3_nt, dim3(Np / 128, Mp / 128, kz), dim3(256), 4 * CVAE_G3_TILE_HALVES * 2, 0, A3, B3, dC, (long)N, M, N, Kp, ep,
                           kz > 1 ? dP : nullptr);
        hipEventRecord(e2);
        hipEventSynchronize(e2);
        if (it >= 2) {
            float a, b;
            hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
            ms_split += a / 4; ms_gemm += b / 4;
        }
    }
    hipError_t err = hipGetLastError();
    std::vector<float> C((size_t)M * N), P;
    if (kz > 1) {
        P.resize((size_t)kz * M * N);
        hipMemcpy(P.data(), dP, P.size() * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < C.size(); ++i) { float s = 0; for (int z = 0; z < kz; ++z) s += P[(size_t)z * M * N + i]; C[i] = s; }
    } else {
        hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    }
    double worst = 0, scale = 0;
    for (int s = 0; s < 400; ++s) {
        const int m = rand() % M, n = rand() % N;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(ta ? A[(size_t)k * M + m] : A[(size_t)m * K + k]) * (double)(tb ? B[(size_t)k * N + n] : B[(size_t)n * K + k]);
        worst = fmax(worst, fabs(ref - C[(size_t)m * N + n]));
        scale = fmax(scale, fabs(ref));
    }
    const double gf = 2.0 * M * N * K / 1e9;
    printf("%-28s M=%d N=%d K=%d kz=%d blocks=%d: split %.1f us, gemm %.1f us = %.1f TFLOP/s (with split %.1f)  max|d| %.2e of %.2e  %s\n", what, M, N, K, kz,
           (Mp / 128) * (Np / 128) * kz, 1e3 * ms_split, 1e3 * ms_gemm, gf / ms_gemm, gf / (ms_gemm + ms_split), worst, scale, hipGetErrorString(err));
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dP); hipFree(A3); hipFree(B3);
}

int main() {
    run("gi (nt)", 5120, 3072, 496, 0, 0, 1);
    run("dX = dgi . W_ix^T (nt)", 5120, 486, 3072, 0, 0, 1);
    run("dX, 4 K slices", 5120, 486, 3072, 0, 0, 4);
    run("dW_hh (tn)", 3072, 1024, 5120, 1, 1, 1);
    run("dW_hh, 2 K slices", 3072, 1024, 5120, 1, 1, 2);
    run("dW_ih (tn)", 3072, 486, 5120, 1, 1, 2);
    run("dW_hh stacked rows (tn)", 3072, 1024, 10240, 1, 1, 2);
    run("8 waves: gi", 5120, 3072, 496, 0, 0, 1, 1);
    run("8 waves: dX 4 slices", 5120, 486, 3072, 0, 0, 4, 1);
    run("8 waves: dW_hh", 3072, 1024, 5120, 1, 1, 1, 1);
    run("8 waves: dW_hh 2 slices", 3072, 1024, 5120, 1, 1, 2, 1);
    run("8 waves: dW_hh 4 slices", 3072, 1024, 5120, 1, 1, 4, 1);
    run("8 waves: dW_ih 2 slices", 3072, 486, 5120, 1, 1, 2, 1);
    return 0;
}
