// Microbenchmark (GPU box only): how fast do all 256 CUs pull the per-step operand set of k_gru_steps_v6 / v5 from L2?
//   every block: 4 waves x NL loads of 1 KiB (16 B per lane), all CUs of a row tile read the SAME bytes (like the recurrence)
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

template <int NL, bool SC1>
__global__ __launch_bounds__(256, 1) void k_ld(const float* src, float* dst, long long* cyc, int iters, int shared) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const cvae_buf b = cvae_make_buf(src, 64u << 20);
    // shared = 1: blocks of a tile (blockIdx / 128) read the same region; 0: every block its own region
    const unsigned region = shared ? (unsigned)(blockIdx.x / 128) : (unsigned)blockIdx.x;
    const unsigned base = region * (unsigned)(4 * NL * 1024) + (unsigned)wave * (NL * 1024);
    f32x4 acc = {0, 0, 0, 0};
    const long long t0 = cvae_clock();
    for (int it = 0; it < iters; ++it) {
        f32x4 v[NL];
#pragma unroll
        for (int s = 0; s < NL; ++s)
            v[s] = SC1 ? cvae_buf_load_f4_sc1(b, lane * 16u, base + s * 1024u) : cvae_buf_load_f4(b, lane * 16u, base + s * 1024u);
#pragma unroll
        for (int s = 0; s < NL; ++s) acc += v[s];
        __syncthreads();
    }
    const long long t1 = cvae_clock();
    dst[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 200, nblk = 256;
    float *src, *dst;
    long long* cyc;
    hipMalloc(&src, 64u << 20);
    hipMemset(src, 0, 64u << 20);
    hipMalloc(&dst, nblk * 256 * 4);
    hipMalloc(&cyc, nblk * 8);
    std::vector<long long> c(nblk);
    auto report = [&](const char* name, int kb) {
        hipDeviceSynchronize();
        hipMemcpy(c.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : c) s += (double)v;
        const double per = s / nblk / iters;
        printf("%-46s %9.1f cycles per round, %5.1f B/clk/CU\n", name, per, kb * 1024.0 / per);
    };
    for (int sh = 1; sh >= 0; --sh) {
        printf("-- %s\n", sh ? "all blocks of a tile read the same region" : "every block reads its own region");
        hipLaunchKernelGGL((k_ld<16, true>), dim3(nblk), dim3(256), 0, 0, src, dst, cyc, iters, sh); report("64 KB per CU, sc1 loads", 64);
        hipLaunchKernelGGL((k_ld<16, false>), dim3(nblk), dim3(256), 0, 0, src, dst, cyc, iters, sh); report("64 KB per CU, plain loads", 64);
        hipLaunchKernelGGL((k_ld<32, true>), dim3(nblk), dim3(256), 0, 0, src, dst, cyc, iters, sh); report("128 KB per CU, sc1 loads", 128);
        hipLaunchKernelGGL((k_ld<32, false>), dim3(nblk), dim3(256), 0, 0, src, dst, cyc, iters, sh); report("128 KB per CU, plain loads", 128);
    }
    return 0;
}
