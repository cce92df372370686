// Microbenchmark (GPU box only): cost of the in-register three-limb split next to the six-MFMA product of k_gru_steps_v6.
//   hipcc --offload-arch=gfx950 -O3 -I cyclevae-vc_amd/csrc tools/mb/mb_split.hip -o /tmp/mb_split && /tmp/mb_split
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

template <int MODE>   // 0: MFMA only (limbs precomputed), 1: split only, 2: split + MFMA (as in the kernel), 3: limbs loaded (no split), 4: split of one value chain per step
__global__ __launch_bounds__(256, 1) void k_mb(const float* src, float* dst, long long* cyc, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    f32x4 w0[16], w1[16], w2[16], hc[32];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        w0[s] = *(const f32x4*)(src + (s * 3 + 0) * 256 + lane * 4);
        w1[s] = *(const f32x4*)(src + (s * 3 + 1) * 256 + lane * 4);
        w2[s] = *(const f32x4*)(src + (s * 3 + 2) * 256 + lane * 4);
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) hc[s] = *(const f32x4*)(src + 16384 + s * 256 + lane * 4);
    f32x16 a0, a1, a2, a3;
    for (int q = 0; q < 16; ++q) { a0[q] = 0; a1[q] = 0; a2[q] = 0; a3[q] = 0; }
    f32x4 keep = {0, 0, 0, 0};
    const long long t0 = cvae_clock();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            f32x4 l0, l1, l2;
            if (MODE == 0 || MODE == 3) { l0 = hc[2 * s]; l1 = hc[2 * s + 1]; l2 = hc[(2 * s + 2) & 31]; }
            else cvae_split3_pack8(hc[2 * s], hc[2 * s + 1], l0, l1, l2);
            if (MODE != 1) {
                a0 = cvae_mfma_32x32x16_f16(l0, w0[s], a0);
                a1 = cvae_mfma_32x32x16_f16(l0, w1[s], a1);
                a2 = cvae_mfma_32x32x16_f16(l1, w1[s], a2);
                a3 = cvae_mfma_32x32x16_f16(l0, w2[s], a3);
                a1 = cvae_mfma_32x32x16_f16(l1, w0[s], a1);
                a2 = cvae_mfma_32x32x16_f16(l2, w0[s], a2);
            } else {
                keep[0] += l0[0] + l1[1] + l2[2];
                keep[1] += l0[3] + l1[2] + l2[0];
            }
        }
        // make the next iteration's inputs depend on this one (no hoisting), cheaply
        hc[0][0] += keep[0] * 1e-30f + a0[0] * 1e-30f;
    }
    const long long t1 = cvae_clock();
    float sum = keep[0] + keep[1];
    for (int q = 0; q < 16; ++q) sum += a0[q] + a1[q] + a2[q] + a3[q];
    dst[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 200, nblk = 256;
    std::vector<float> h(16384 + 32 * 256);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.0f - 1.0f;
    float *src, *dst;
    long long* cyc;
    hipMalloc(&src, h.size() * 4);
    hipMalloc(&dst, nblk * 256 * 4);
    hipMalloc(&cyc, nblk * 8);
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<long long> c(nblk);
    auto report = [&](const char* name) {
        hipDeviceSynchronize();
        hipMemcpy(c.data(), cyc, nblk * 8, hipMemcpyDeviceToHost);
        double s = 0;
        for (auto v : c) s += (double)v;
        printf("%-40s %10.1f cycles per 16-step phase (96 MFMA / 128 values per lane)\n", name, s / nblk / iters);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_mb<0>, dim3(nblk), dim3(256), 0, 0, src, dst, cyc, iters); report("MFMA only");
        hipLaunchKernelGGL(k_mb<1>, dim3(nblk), dim3(256), 0, 0, src, dst, cyc, iters); report("split only");
        hipLaunchKernelGGL(k_mb<2>, dim3(nblk), dim3(256), 0, 0, src, dst, cyc, iters); report("split + MFMA");
    }
    return 0;
}
