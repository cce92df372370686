// Microbenchmark (GPU box only): k_gemm_x3 (tools/mb/cvae_gemm_x3.h: operands split into fp16 limb triples on their way into LDS, six
// f16 MFMAs per product) on the big shapes of the training step, against fp64 on sampled entries.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I cyclevae-vc_amd/csrc -I include -I tools/mb tools/mb/mb_gemm_x3.hip -o tools/mb/mb_gemm_x3   (-DCVAE_GX3_NSET=4: deeper prefetch)
//   mb_gemm_x3: the shapes; mb_gemm_x3 x: where a stage spends its time (work left out piece by piece, results wrong by construction)
#include <cvae_intrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include <cvae_kernels.h>
#include <cvae_train_kernels.h>
#include "cvae_gemm_x3.h"

static long up(long x, long m) { return (x + m - 1) / m * m; }

// C[I][J] = sum_k A(i,k) B(j,k); t: operands stored k-strided ([k][row]) -- both or neither
template <int T, int FLAGS>
static void run(const char* what, int I, int J, int K, int kz, float ascale = 1.0f) {
    std::vector<float> A((size_t)I * K), B((size_t)J * K);
    srand(1);
    for (auto& v : A) v = ((float)rand() / RAND_MAX - 0.5f) / ascale;
    for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *dC, *dP;
    unsigned* dCnt;
    int* dStatus;
    const int It = (int)up(I, 128) / 128, Jt = (int)up(J, 128) / 128;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)I * J * 4);
    hipMalloc(&dP, (size_t)kz * It * Jt * 16384 * 4); hipMalloc(&dCnt, 4096 * 4); hipMalloc(&dStatus, 4);
    hipMemset(dCnt, 0, 4096 * 4); hipMemset(dStatus, 0, 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    GemmX3Params p{};
    p.A = dA; p.lda = T ? I : K; p.abytes = (unsigned)(A.size() * 4);
    p.B = dB; p.ldb = T ? J : K; p.bbytes = (unsigned)(B.size() * 4);
    p.C = dC; p.ldc = J; p.I = I; p.J = J; p.K = K;
    p.kchunk = (int)up((K + kz - 1) / kz, 32);
    p.ascale = ascale; p.oscale = 1.0f / ascale; p.bias = nullptr; p.accumulate = 0;
    p.part = kz > 1 ? dP : nullptr; p.cnt = dCnt; p.status = dStatus;
    hipFuncSetAttribute((const void*)k_gemm_x3<T, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, CVAE_GX3_LDS);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    const int reps = 10;
    for (int it = 0; it < 2 + reps; ++it) {
        if (it == 2) hipEventRecord(e0);
        hipLaunchKernelGGL((k_gemm_x3<T, FLAGS>), dim3(Jt, It, kz), dim3(256), CVAE_GX3_LDS, 0, p);
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    hipError_t err = hipGetLastError();
    std::vector<float> C((size_t)I * J);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int status = 0;
    hipMemcpy(&status, dStatus, 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int s = 0; s < 600; ++s) {
        const int i = s < 8 ? (s & 1 ? I - 1 : 0) : rand() % I, j = s < 8 ? (s & 2 ? J - 1 : 0) : rand() % J;
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)(T ? A[(size_t)k * I + i] : A[(size_t)i * K + k]) * (double)(T ? B[(size_t)k * J + j] : B[(size_t)j * K + k]);
        worst = fmax(worst, fabs(ref - C[(size_t)i * J + j]));
        scale = fmax(scale, fabs(ref));
    }
    const double gf = 2.0 * I * J * K / 1e9;
    printf("%-30s I=%d J=%d K=%d kz=%d blocks=%d: %.1f us = %.1f TFLOP/s  max|d| %.2e of %.2e  status %d  %s\n", what, I, J, K, kz, It * Jt * kz,
           1e3 * ms, gf / ms, worst, scale, status, hipGetErrorString(err));
    fflush(stdout);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dP); hipFree(dCnt); hipFree(dStatus);
}

int main(int argc, char** argv) {
    if (argc > 1) {      // where does a stage spend its time?  (results are wrong by construction)
        run<1, 1>("dW_hh 4 slices: as is", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x4001>("  plain tile order", 3072, 1024, 5120, 4, 256.0f);
        run<1, 1>("dW_hh 1 slice: as is", 3072, 1024, 5120, 1, 256.0f);
        run<1, 0x4001>("  plain tile order", 3072, 1024, 5120, 1, 256.0f);
        run<0, 2>("gi: as is", 5120, 3072, 496, 1);
        run<0, 0x4002>("  plain tile order", 5120, 3072, 496, 1);
        run<0, 0x102>("  no global loads", 5120, 3072, 496, 1);
        run<0, 0x202>("  no LDS stores", 5120, 3072, 496, 1);
        run<0, 0x1f02>("  MFMAs only", 5120, 3072, 496, 1);
        run<1, 0x101>("  no global loads", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x201>("  no LDS stores", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x401>("  no split", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x801>("  no barrier", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x1001>("  no fragment reads", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x2001>("  no MFMAs", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x701>("  no loads, stores, split", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x1701>("  MFMAs + barrier only", 3072, 1024, 5120, 4, 256.0f);
        run<1, 0x1f01>("  MFMAs only", 3072, 1024, 5120, 4, 256.0f);
        // per-stage time from the difference of two depths (one block per CU and launch: 192 blocks)
        run<1, 1>("as is, K=5120 kz=1", 3072, 1024, 5120, 1, 256.0f);
        run<1, 1>("as is, K=10240 kz=1", 3072, 1024, 10240, 1, 256.0f);
        run<1, 0x1f01>("MFMAs only, K=5120 kz=1", 3072, 1024, 5120, 1, 256.0f);
        run<1, 0x1f01>("MFMAs only, K=10240 kz=1", 3072, 1024, 10240, 1, 256.0f);
        run<1, 0x1701>("MFMAs + barrier, K=5120 kz=1", 3072, 1024, 5120, 1, 256.0f);
        run<1, 0x1701>("MFMAs + barrier, K=10240 kz=1", 3072, 1024, 10240, 1, 256.0f);
        run<1, 0x0701>("MFMAs + barrier + frag reads, K=5120", 3072, 1024, 5120, 1, 256.0f);
        run<1, 0x0701>("MFMAs + barrier + frag reads, K=10240", 3072, 1024, 10240, 1, 256.0f);
        run<1, 0x0301>("all but loads + stores, K=5120", 3072, 1024, 5120, 1, 256.0f);
        run<1, 0x0301>("all but loads + stores, K=10240", 3072, 1024, 10240, 1, 256.0f);
        run<1, 0x0101>("all but loads, K=5120", 3072, 1024, 5120, 1, 256.0f);
        run<1, 0x0101>("all but loads, K=10240", 3072, 1024, 10240, 1, 256.0f);
        return 0;
    }
    run<0, 2>("gi (k-contiguous)", 5120, 3072, 496, 1);
    run<0, 1>("dX = dgi . W_ix^T", 5120, 486, 3072, 1, 256.0f);
    run<0, 1>("dX, 4 slices", 5120, 486, 3072, 4, 256.0f);
    run<0, 1>("dX, 8 slices", 5120, 486, 3072, 8, 256.0f);
    run<1, 1>("dW_hh (k-strided)", 3072, 1024, 5120, 1, 256.0f);
    run<1, 1>("dW_hh, 2 slices", 3072, 1024, 5120, 2, 256.0f);
    run<1, 1>("dW_hh, 4 slices", 3072, 1024, 5120, 4, 256.0f);
    run<1, 5>("dW_ih, 4 slices", 3072, 486, 5120, 4, 256.0f);
    run<1, 5>("dW_ih, 8 slices", 3072, 486, 5120, 8, 256.0f);
    run<1, 1>("dW_hh stacked rows, 4 slices", 3072, 1024, 10240, 4, 256.0f);
    run<0, 0>("small odd: 200 x 70 x 104", 200, 70, 104, 1);
    run<1, 0>("small odd strided: 200 x 70 x 104", 200, 70, 104, 1);
    return 0;
}
