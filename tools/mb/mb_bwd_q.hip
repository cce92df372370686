// Microbenchmark (GPU box only): the matrix phase of ONE task of the exact reverse training recurrence, per wave, operands in
// registers (no global loads, no flags), to price taking dq = dnp * r off the exchange (round 5):
//   MODE 0  today's k_train_bwd_steps_x3<32>: 32 k-steps x 6 v_mfma_f32_16x16x32_f16 on [drp, dzp, dnp, dq] x [W_hh^T | F^T] with
//           structural zero rows; per step one bf8 decode of the operand's third limb and one of the weights' (from LDS)
//   MODE 1  the zero rows dropped: 16 steps on (drp, dzp) x all 16 columns + 8 steps on dnp with ONE weight fragment
//           [W_hn^T | F_n^T] used twice -- once with dnp (columns 8..15 count), once with dq (columns 0..7 count) -- dq given
//   MODE 2  = 1 with dq REBUILT by the consumer: limb triple of dnp -> fp32 (exact), times the taped reset gate (fp32, in
//           registers), split into a fresh triple (cvae_split3_pack8): the VALU work that replaces a quarter of the operand loads
// Reports shader cycles per task of block 0..255 (192 MFMAs = 3,072 matrix-pipe cycles in every mode).
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void h8_to_f32(f32x4 l0, f32x4 l1, f32x4 l2, float* x) {
    // x = l0 + l1 / 2^11 + l2 / 2^22, exact in fp32 for a triple that came from one fp32 value
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const auto a = __builtin_bit_cast(__attribute__((ext_vector_type(2))) _Float16, (float)l0[e]);
        const auto b = __builtin_bit_cast(__attribute__((ext_vector_type(2))) _Float16, (float)l1[e]);
        const auto c = __builtin_bit_cast(__attribute__((ext_vector_type(2))) _Float16, (float)l2[e]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float t = __builtin_fmaf((float)c[h], 1.0f / 2048.0f, (float)b[h]);
            x[2 * e + h] = __builtin_fmaf(t, 1.0f / 2048.0f, (float)a[h]);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_bq(const float* wsrc, const float* osrc, float* dst, long long* cyc, int iters) {
    constexpr int NS = MODE == 0 ? 32 : 24;     // weight fragments per wave
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    float* w2l = (float*)CVAE_SMEM;             // third limbs of the weights as bf8: [4 waves][NS][64 lanes][8 bytes]
    f32x4 w0[NS], w1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float* w = wsrc + (long)s * 640 + lane * 4;
        w0[s] = *(const f32x4*)w;
        w1[s] = *(const f32x4*)(w + 256);
        *(f32x2*)(w2l + (wave * NS + s) * 128 + lane * 2) = *(const f32x2*)(wsrc + (long)s * 640 + 512 + lane * 2);
    }
    __syncthreads();
    const float* w2w = w2l + wave * NS * 128 + lane * 2;
    // operands: 8 fragments in registers, reused round robin (the kernel streams them through a ring of 8)
    f32x4 g0[8], g1[8];
    f32x2 g2[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        g0[s] = *(const f32x4*)(osrc + (s * 3) * 256 + lane * 4);
        g1[s] = *(const f32x4*)(osrc + (s * 3 + 1) * 256 + lane * 4);
        g2[s] = *(const f32x2*)(osrc + (s * 3 + 2) * 256 + lane * 2);
    }
    f32x4 rb[16];                               // taped reset gates of the wave's dnp values: 8 steps x 8 floats per lane
#pragma unroll
    for (int s = 0; s < 16; ++s) rb[s] = *(const f32x4*)(osrc + 8192 + s * 256 + lane * 4);
    float carry = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long t0 = cvae_clock();
    for (int it = 0; it < iters; ++it) {
        f32x4 a[4], n[4], q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = n[k] = q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) {
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                const f32x4 l0 = g0[s & 7], l1 = g1[s & 7], l2 = cvae_bf8x8_to_h8(g2[s & 7]);
                const f32x4 w2 = cvae_bf8x8_to_h8(*(const f32x2*)(w2w + s * 128));
                a[0] = cvae_mfma_16x16x32_f16(l0, w0[s], a[0]);
                a[1] = cvae_mfma_16x16x32_f16(l0, w1[s], a[1]);
                a[2] = cvae_mfma_16x16x32_f16(l1, w1[s], a[2]);
                a[3] = cvae_mfma_16x16x32_f16(l0, w2, a[3]);
                a[1] = cvae_mfma_16x16x32_f16(l1, w0[s], a[1]);
                a[2] = cvae_mfma_16x16x32_f16(l2, w0[s], a[2]);
                cvae_sched_fence();
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {           // two (drp, dzp) steps
                    const int s = 3 * i + j;
                    const f32x4 l0 = g0[(2 * i + j) & 7], l1 = g1[(2 * i + j) & 7], l2 = cvae_bf8x8_to_h8(g2[(2 * i + j) & 7]);
                    const f32x4 w2 = cvae_bf8x8_to_h8(*(const f32x2*)(w2w + s * 128));
                    a[0] = cvae_mfma_16x16x32_f16(l0, w0[s], a[0]);
                    a[1] = cvae_mfma_16x16x32_f16(l0, w1[s], a[1]);
                    a[2] = cvae_mfma_16x16x32_f16(l1, w1[s], a[2]);
                    a[3] = cvae_mfma_16x16x32_f16(l0, w2, a[3]);
                    a[1] = cvae_mfma_16x16x32_f16(l1, w0[s], a[1]);
                    a[2] = cvae_mfma_16x16x32_f16(l2, w0[s], a[2]);
                    if (MODE == 1) cvae_sched_fence();
                }
                {                                       // one dnp step: the fragment [W_hn^T | F_n^T] with dnp and with dq
                    const int s = 3 * i + 2;
                    const f32x4 l0 = g0[(i + 3) & 7], l1 = g1[(i + 3) & 7], l2 = cvae_bf8x8_to_h8(g2[(i + 3) & 7]);
                    const f32x4 w2 = cvae_bf8x8_to_h8(*(const f32x2*)(w2w + s * 128));
                    f32x4 q0, q1, q2;
                    if (MODE == 2) {
                        float x[8];
                        h8_to_f32(l0, l1, l2, x);
                        const f32x4 ra = rb[2 * i], rc = rb[2 * i + 1];
                        f32x4 va, vb;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { va[e] = x[e] * (ra[e] + carry); vb[e] = x[4 + e] * (rc[e] + carry); }
                        cvae_split3_pack8(va, vb, q0, q1, q2);
                    } else {
                        q0 = g0[(i + 5) & 7]; q1 = g1[(i + 5) & 7]; q2 = g1[(i + 6) & 7];
                    }
                    n[0] = cvae_mfma_16x16x32_f16(l0, w0[s], n[0]);
                    n[1] = cvae_mfma_16x16x32_f16(l0, w1[s], n[1]);
                    n[2] = cvae_mfma_16x16x32_f16(l1, w1[s], n[2]);
                    n[3] = cvae_mfma_16x16x32_f16(l0, w2, n[3]);
                    n[1] = cvae_mfma_16x16x32_f16(l1, w0[s], n[1]);
                    n[2] = cvae_mfma_16x16x32_f16(l2, w0[s], n[2]);
                    q[0] = cvae_mfma_16x16x32_f16(q0, w0[s], q[0]);
                    q[1] = cvae_mfma_16x16x32_f16(q0, w1[s], q[1]);
                    q[2] = cvae_mfma_16x16x32_f16(q1, w1[s], q[2]);
                    q[3] = cvae_mfma_16x16x32_f16(q0, w2, q[3]);
                    q[1] = cvae_mfma_16x16x32_f16(q1, w0[s], q[1]);
                    q[2] = cvae_mfma_16x16x32_f16(q2, w0[s], q[2]);
                    if (MODE == 1) cvae_sched_fence();
                }
                if (MODE == 2) cvae_sched_fence();      // (per group of three steps: the rebuild may move under the group's 24 MFMAs, not further)
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) acc += a[k] + ((lane & 15) < 8 ? q[k] : n[k]);
        carry = acc[0] * 1e-30f;                        // (keeps the iterations dependent on each other)
        g0[0][0] += carry;
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
        printf("%-72s %8.1f cycles per task (192 MFMAs: %.1f per MFMA)  %s\n", name, s / nblk / iters, s / nblk / iters / 192.0, e == hipSuccess ? "" : hipGetErrorString(e));
    };
    const size_t lds = 4 * 32 * 128 * sizeof(float);
#define RUN(M, NAME) hipLaunchKernelGGL((k_bq<M>), dim3(nblk), dim3(256), lds, 0, wsrc, osrc, dst, cyc, iters); report(NAME);
    for (int rep = 0; rep < 2; ++rep) {
        RUN(0, "today: 32 steps on [drp dzp dnp dq], zero rows in the weights");
        RUN(1, "zero rows dropped: 16 (drp dzp) steps + 8 dnp steps x 2 products, dq given");
        RUN(2, "the same, dq = dnp * r rebuilt by the consumer (split3_pack8)");
    }
    return 0;
}
