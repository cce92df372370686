// Microbenchmark (GPU box only): the same per-step operand pull as mb_load.hip (every CU reads 64 / 128 KB that all CUs of a row
// tile share, L2-resident), but through the LDS-DMA path (global_load_lds_dwordx4: 1 KiB per wave instruction straight into LDS)
// instead of VGPR loads -- is the CU's load path faster that way?  (The reverse training recurrence is bound by it: 640 KB per CU
// and step at 43-46 B/clk with buffer_load_dwordx4.)
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

template <int NL, int MODE>   // MODE 0: buffer_load_dwordx4 into VGPRs (reference), 1: global_load_lds 16 B, 2: = 1 followed by ds_read_b128 of everything
__global__ __launch_bounds__(256, 1) void k_ld(const float* src, float* dst, long long* cyc, int iters, int shared) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const cvae_buf b = cvae_make_buf(src, 64u << 20);
    const unsigned region = shared ? (unsigned)(blockIdx.x / 128) : (unsigned)blockIdx.x;
    const unsigned base = region * (unsigned)(4 * NL * 1024) + (unsigned)wave * (NL * 1024);
    f32x4 acc = {0, 0, 0, 0};
    __attribute__((address_space(3))) unsigned char* lds = (__attribute__((address_space(3))) unsigned char*)smem_ + wave * (NL * 1024);
    const long long t0 = cvae_clock();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            f32x4 v[NL];
#pragma unroll
            for (int s = 0; s < NL; ++s) v[s] = cvae_buf_load_f4(b, lane * 16u, base + s * 1024u);
#pragma unroll
            for (int s = 0; s < NL; ++s) acc += v[s];
        } else {
#pragma unroll
            for (int s = 0; s < NL; ++s)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const unsigned char*)src + base + s * 1024u + lane * 16u),
                                                 (__attribute__((address_space(3))) void*)(lds + s * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (MODE == 2) {
#pragma unroll
                for (int s = 0; s < NL; ++s) acc += *(const __attribute__((address_space(3))) f32x4*)(lds + s * 1024 + lane * 16);
            }
        }
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
        hipError_t e = hipDeviceSynchronize();
        hipMemcpy(c.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : c) s += (double)v;
        const double per = s / nblk / iters;
        printf("%-58s %9.1f cycles per round, %5.1f B/clk/CU %s\n", name, per, kb * 1024.0 / per, e == hipSuccess ? "" : hipGetErrorString(e));
    };
    for (int sh = 1; sh >= 0; --sh) {
        printf("-- %s\n", sh ? "all blocks of a tile read the same region" : "every block reads its own region");
        hipLaunchKernelGGL((k_ld<16, 0>), dim3(nblk), dim3(256), 64 << 10, 0, src, dst, cyc, iters, sh); report("64 KB per CU, buffer_load_dwordx4 -> VGPR", 64);
        hipLaunchKernelGGL((k_ld<16, 1>), dim3(nblk), dim3(256), 64 << 10, 0, src, dst, cyc, iters, sh); report("64 KB per CU, global_load_lds 16 B", 64);
        hipLaunchKernelGGL((k_ld<16, 2>), dim3(nblk), dim3(256), 64 << 10, 0, src, dst, cyc, iters, sh); report("64 KB per CU, global_load_lds 16 B + ds_read_b128", 64);
        hipLaunchKernelGGL((k_ld<32, 0>), dim3(nblk), dim3(256), 128 << 10, 0, src, dst, cyc, iters, sh); report("128 KB per CU, buffer_load_dwordx4 -> VGPR", 128);
        hipLaunchKernelGGL((k_ld<32, 1>), dim3(nblk), dim3(256), 128 << 10, 0, src, dst, cyc, iters, sh); report("128 KB per CU, global_load_lds 16 B", 128);
        hipLaunchKernelGGL((k_ld<32, 2>), dim3(nblk), dim3(256), 128 << 10, 0, src, dst, cyc, iters, sh); report("128 KB per CU, global_load_lds 16 B + ds_read_b128", 128);
    }
    return 0;
}
