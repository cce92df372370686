// Microbenchmark (GPU box only): what a kernel boundary costs on one stream as a function of the bytes the kernel leaves DIRTY in
// the L2s (round 5: the training step's timeline shows 10-17 us between a GEMM and the next launch, 0-6 us behind small kernels).
//   MODE 0  plain global stores (write-back L2: the end-of-kernel release writes the dirty lines back)
//   MODE 1  sc1 write-through stores (nothing dirty at the end)
//   MODE 2  nontemporal stores
// Chain of N dependent launches (ping-pong between two buffers), wall time per launch by HIP events; for each size also the time
// of the same bytes written by ONE launch of N times the grid, i.e. the pure streaming cost.
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>
#include <cvae_kernels.h>
#include <cvae_train_kernels.h>

template <int MODE>
__global__ __launch_bounds__(256) void k_w(const float* src, float* dst, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f32x4 v = ((const f32x4*)src)[i];
    f32x4 o = v;
    o[0] += 1.0f;
    if (MODE == 0) ((f32x4*)dst)[i] = o;
    else if (MODE == 1) {
        cvae_buf b = cvae_make_buf(dst, 0x7fffffff);
        cvae_buf_store_f4_sc1(b, (unsigned)(i * 16), 0, o);
    } else __builtin_nontemporal_store(o, (f32x4*)dst + i);
}

template <int MODE>
float chain(float* a, float* b, long n4, int N, hipStream_t st) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int g = (int)((n4 + 255) / 256);
    for (int w = 0; w < 2; ++w) {
        hipEventRecord(e0, st);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL((k_w<MODE>), dim3(g), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n4);
        hipEventRecord(e1, st);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return 1e3f * ms / N;
}

static const int N_CHAIN = 400;
int main() {
    hipStream_t st;
    hipStreamCreate(&st);
    const long maxb = 256L << 20;
    float *a, *b;
    hipMalloc(&a, maxb); hipMalloc(&b, maxb);
    hipMemset(a, 0, maxb); hipMemset(b, 0, maxb);
    const int N = 200;
    printf("bytes per launch | us per launch in a chain of %d: plain  sc1  nontemporal\n", N);
    for (long kb : {4L, 64L, 512L, 2048L, 8192L, 32768L, 131072L}) {
        const long n4 = kb * 1024 / 16;
        const float t0 = chain<0>(a, b, n4, N, st), t1 = chain<1>(a, b, n4, N, st), t2 = chain<2>(a, b, n4, N, st);
        printf("%8ld KiB  %8.2f %8.2f %8.2f\n", kb, t0, t1, t2);
    }
    // the library's own small GEMM (k_gemm_nt2<1,1>, M=80 N=162 K=162: conv0 of a one-utterance pass) in a dependent chain, alone
    // and alternating with a one-block elementwise kernel; with and without dynamic LDS on the small kernel
    {
        const int M = 80, N = 162, K = 164;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const EpiMask em0{nullptr, 0, 0, 0};
        const float* wB = b + 1048576;
        const size_t lds11 = GemmTileCfg<1, 1>::lds_bytes;
        auto gemm = [&](const float* A, float* C) {
            hipLaunchKernelGGL((k_gemm_nt2<1, 1>), dim3((N + 31) / 32, (M + 31) / 32, 1), dim3(256), lds11, st, A, (long)K,
                               K, 0L, wB, (long)K, (const float*)nullptr, C, (long)K, M, N, K, 0, K, (float*)nullptr,
                               (unsigned*)nullptr, em0);
        };
        auto tiny = [&](const float* A, float* C) { hipLaunchKernelGGL((k_w<0>), dim3(13), dim3(256), 0, st, A, C, (long)(M * K / 4)); };
        auto tiny_lds = [&](const float* A, float* C) { hipLaunchKernelGGL((k_w<0>), dim3(13), dim3(256), 8192, st, A, C, (long)(M * K / 4)); };
        auto run = [&](const char* nm, int mode) {
            float ms = 0;
            for (int w = 0; w < 2; ++w) {
                hipEventRecord(e0, st);
                for (int i = 0; i < N_CHAIN; ++i) {
                    const float* A = (i & 1) ? b : a;
                    float* C = (i & 1) ? a : b;
                    if (mode == 0) gemm(A, C);
                    else if (mode == 1) tiny(A, C);
                    else if (mode == 2) { if (i & 1) gemm(A, C); else tiny(A, C); }
                    else if (mode == 3) tiny_lds(A, C);
                    else { if (i & 1) tiny_lds(A, C); else tiny(A, C); }
                }
                hipEventRecord(e1, st);
                hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            printf("%-50s %7.2f us per launch\n", nm, 1e3f * ms / N_CHAIN);
        };
        run("gemm_nt2<1,1> chain", 0);
        run("tiny chain", 1);
        run("gemm / tiny alternating", 2);
        run("tiny with 8 KB dynamic LDS chain", 3);
        run("tiny with / without LDS alternating", 4);
    }
    return 0;
}
