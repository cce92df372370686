// Microbenchmark (GPU box only), round 6, measured and NOT adopted: k_gemm3_nt (tools/mb/cvae_gemm3p.h: operands as pre-split fp16
// limb planes, six f16 MFMAs per product) on the big GEMM shapes of a 64-row training pass, checked against fp64 on sampled entries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cyclevae-vc_amd/csrc -I include -I tools/mb tools/mb/mb_gemm3p.hip -o tools/mb/mb_gemm3p
//   mb_gemm3p: the shapes; mb_gemm3p x: the same with the loads / the MFMAs left out in turn (results wrong by construction)
#include <cvae_intrin.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "cvae_gemm3p.h"

static long up(long x, long m) { return (x + m - 1) / m * m; }

static int g_xcd_map = 1, g_exp = 0;
static void run(const char* what, int M, int N, int K, int kz, bool transposed_inputs) {
    const int Mp = (int)up(M, 128), Np = (int)up(N, 128), Kp = (int)up(K, 32);
    std::vector<float> A((size_t)M * K), B((size_t)N * K);
    srand(1);
    for (auto& v : A) v = ((float)rand() / RAND_MAX - 0.5f) * 0.01f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f);
    float *dA, *dB, *dC, *dP;
    unsigned short *pA, *pB;
    unsigned* dCnt;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMalloc(&pA, (size_t)3 * Mp * Kp * 2); hipMalloc(&pB, (size_t)3 * Np * Kp * 2);
    const int Mt = Mp / 128, Nt = Np / 128;
    hipMalloc(&dP, (size_t)kz * Mt * Nt * 65536); hipMalloc(&dCnt, 4096 * 4);
    hipMemset(dCnt, 0, 4096 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms_split = 0.f;
    if (!transposed_inputs) {
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_split3_rows, dim3((unsigned)(((long)Mp * (Kp / 8) + 255) / 256)), dim3(256), 0, 0, (const float*)dA, (long)K, M, K, pA, (long)Mp * Kp, Mp, Kp, 256.0f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_split, e0, e1);
        }
        hipLaunchKernelGGL(k_split3_rows, dim3((unsigned)(((long)Np * (Kp / 8) + 255) / 256)), dim3(256), 0, 0, (const float*)dB, (long)K, N, K, pB, (long)Np * Kp, Np, Kp, 1.0f);
    } else {   // inputs stored [K][M] / [K][N] (time-major rows): the weight-gradient form, split + transposed
        std::vector<float> At((size_t)K * M), Bt((size_t)K * N);
        for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) At[(size_t)k * M + m] = A[(size_t)m * K + k];
        for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) Bt[(size_t)k * N + n] = B[(size_t)n * K + k];
        hipMemcpy(dA, At.data(), At.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_split3_t, dim3(Kp / 64 + (Kp % 64 ? 1 : 0), Mp / 64), dim3(256), 3 * 64 * 72 * 2, 0, (const float*)dA, (long)M, K, M, pA, (long)Mp * Kp, 0, Kp, Mp, 256.0f);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms_split, e0, e1);
        }
        hipLaunchKernelGGL(k_split3_t, dim3(Kp / 64 + (Kp % 64 ? 1 : 0), Np / 64), dim3(256), 3 * 64 * 72 * 2, 0, (const float*)dB, (long)N, K, N, pB, (long)Np * Kp, 0, Kp, Np, 1.0f);
    }
    Gemm3Params p{};
    p.A = pA; p.a_plane = (long)Mp * Kp;
    p.B = pB; p.b_plane = (long)Np * Kp;
    p.C = dC; p.ldc = N; p.bias = nullptr; p.M = M; p.N = N; p.K = Kp; p.a_brk = 1 << 30; p.a_skip = 0; p.accumulate = 0; p.scale = 1.0f / 256.0f;
    p.kchunk = (int)up((Kp + kz - 1) / kz, 32);
    const int nz = (Kp + p.kchunk - 1) / p.kchunk;
    p.part = nz > 1 ? dP : nullptr; p.cnt = dCnt; p.mask = nullptr;
    p.gx = Nt; p.gy = Mt; p.gz = nz; p.xcd_map = g_xcd_map; p.exp = g_exp;
    hipFuncSetAttribute((const void*)k_gemm3_nt, hipFuncAttributeMaxDynamicSharedMemorySize, CVAE_G3_LDS);
    float best = 1e9f;
    for (int it = 0; it < 6; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_gemm3_nt, dim3(8 * ((Nt * Mt * nz + 7) / 8)), dim3(256), CVAE_G3_LDS, 0, p);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it > 0 && ms < best) best = ms;
    }
    hipError_t err = hipGetLastError();
    std::vector<float> C((size_t)M * N);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0.0, cmax = 0.0;
    for (int s = 0; s < 400; ++s) {
        const int m = (int)((long)rand() % M), n = (int)((long)rand() % N);
        double ref = 0.0;
        for (int k = 0; k < K; ++k) ref += (double)A[(size_t)m * K + k] * (double)B[(size_t)n * K + k];
        worst = fmax(worst, fabs(ref - (double)C[(size_t)m * N + n]));
        cmax = fmax(cmax, fabs(ref));
    }
    printf("xcd%d %-28s M=%5d N=%5d K=%5d kz=%d  %7.1f us  %6.1f TFLOP/s (fp32-equivalent)  split(A) %5.1f us  max|d|/max|ref| %.2e  %s\n", g_xcd_map, what, M, N, K, nz,
           1e3 * best, 2.0 * M * N * K / (1e9 * best), 1e3 * ms_split, worst / cmax, err == hipSuccess ? "" : hipGetErrorString(err));
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(pA); hipFree(pB); hipFree(dP); hipFree(dCnt);
}

int main(int argc, char** argv) {
    g_xcd_map = 0;
    for (g_exp = 0; g_exp < 4; ++g_exp) {
        printf("exp=%d (bit 0: no global fetch / stash, bit 1: no MFMAs)\n", g_exp);
        run("dW_hh = dgh^T . h", 3072, 1024, 5120, 1, true);
        run("dW_hh = dgh^T . h", 3072, 1024, 5120, 4, true);
        run("gi = x . W_ix^T", 5120, 3072, 496, 1, false);
        if (argc < 2) break;
    }
    return 0;
}
