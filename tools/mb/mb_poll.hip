// Microbenchmark (GPU box only): cost of ONE poll iteration of k_gru_steps_ll -- every thread of a 256-thread block loads NQ
// 16-byte words (block total NQ * 4 KB) with a given cache policy, then the block votes (__syncthreads_and).
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

template <int NQ, int AUX, bool VOTE>
__global__ __launch_bounds__(256, 1) void k_poll(const float* src, float* dst, long long* cyc, int iters, int nactive) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nactive) return;
    const cvae_buf b = cvae_make_buf(src, 1u << 20);
    f32x4 acc = {0, 0, 0, 0};
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        f32x4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            v[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(b, (tid * NQ + q) * 16, 0, AUX | (int)0x80000000));
        bool ok = true;
#pragma unroll
        for (int q = 0; q < NQ; ++q) { acc += v[q]; ok = ok && v[q][3] == 0.0f; }
        if (VOTE) { if (!__syncthreads_and(ok ? 1 : 0)) break; }
        else asm volatile("" ::: "memory");
    }
    const long long t1 = wall_clock64();
    dst[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *src, *dst;
    long long* cyc;
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    hipMalloc(&dst, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    std::vector<long long> c(256);
    auto run = [&](const char* name, auto kern, int nactive) {
        hipMemset(cyc, 0, 256 * 8);
        hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, (const float*)src, dst, cyc, iters, nactive);
        hipDeviceSynchronize();
        hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < nactive; ++i) s += (double)c[i];
        printf("%-44s %3d blocks polling: %7.0f ns per iteration\n", name, nactive, 10.0 * s / nactive / iters);
        fflush(stdout);
    };
    for (int n : {1, 8, 32, 256}) {
        run("1 word/thread (4 KB), sc1, vote", k_poll<1, 16, true>, n);
        run("4 words/thread (16 KB), sc1, vote", k_poll<4, 16, true>, n);
        run("4 words/thread (16 KB), sc1, no vote", k_poll<4, 16, false>, n);
        run("4 words/thread (16 KB), sc0, vote", k_poll<4, 1, true>, n);
        run("1 word/thread (4 KB), sc0, vote", k_poll<1, 1, true>, n);
        run("4 words/thread (16 KB), plain, vote", k_poll<4, 0, true>, n);
    }
    return 0;
}
