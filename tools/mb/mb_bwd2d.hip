// Microbenchmark (GPU box only): would a 2-D decomposition of the exact reverse training recurrence be faster?  (round 5)
// Today (k_train_bwd_steps_x3): block = 8 units x all K = 4H gate gradients -> every CU pulls 16 rows x 4096 x 5 B = 327 KB per task
// from L2, 654 KB per step at 64 rows: L2 -> CU bandwidth bounds the step (23.6K cycles).  Here: a 16 x 16 grid of CUs, CU (i, j)
// owns the K slice i (64 units x 4 gate gradients = 256 k) x the N slice j (64 units x 2 paths = 128 columns) of [W_hh^T | F^T]
// (160 KB as limb triples, as today), and a step is TWO dependent phases per 16-row tile:
//   phase 1  load the K slice of the tile's gate gradients (20 KB, from the 16 CUs that finalised those units), 96 MFMAs per wave,
//            add the four waves' sums in LDS, write the [16 x 128] fp32 partial sums as 16 pieces of 512 B, one per reducer
//   phase 2  CU (i, j) reduces units 64j + 4i .. +4 (8 columns): load the 16 pieces (8 KB), add them in fixed order, cell backward of
//            16 rows x 4 units, split into limb triples, publish 1280 B for the 16 CUs of K slice j
// 150 KB per CU and step instead of 654 KB, at the price of a second hand-off per step, hidden by the NT tiles a CU cycles through.
// Synthetic data (the arithmetic is the real kernel's: six MFMAs per product, split3, bf8 third limbs); reports cycles per step.
#include <cvae_intrin.h>
#include <stdio.h>
#include <vector>

struct P2 {
    char* G; unsigned* fG; char* P; unsigned* fP; const float* w; const float* tape; float* dg; long long* cyc; int* status;
    int T, NT, pf;
};

__global__ __launch_bounds__(256, 1) void k_bwd2d(P2 p) {
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int i = blockIdx.x & 15, j = blockIdx.x >> 4, NT = p.NT;
    float* red = (float*)CVAE_SMEM;                         // [4 waves][16 rows][132]: the waves' partial sums; reused by phase 2
    float* w2l = red + 4 * 16 * 132;                        // third limbs of the weights (bf8): [4 waves][2 s][8 n][64 lanes][8 B]
    unsigned short* pub = (unsigned short*)(w2l + 4 * 16 * 128);   // 1280 B
    f32x4 w0[2][8], w1[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const float* src = p.w + ((long)((blockIdx.x * 4 + wave) * 2 + s) * 8 + n) * 640;
            w0[s][n] = *(const f32x4*)(src + lane * 4);
            w1[s][n] = *(const f32x4*)(src + 256 + lane * 4);
            *(f32x2*)(w2l + ((wave * 2 + s) * 8 + n) * 128 + lane * 2) = *(const f32x2*)(src + 512 + lane * 2);
        }
    __syncthreads();
    const float* w2w = w2l + wave * 16 * 128 + lane * 2;
    const cvae_buf gb = cvae_make_buf(p.G, (unsigned)((long)p.T * NT * 256 * 1280));
    const cvae_buf pb = cvae_make_buf(p.P, (unsigned)((long)p.T * NT * 256 * 8192));
    long long pc[6] = {0, 0, 0, 0, 0, 0};
    float keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < p.T; ++t) {
        // ---------------- phase 1, every tile
        for (int tile = 0; tile < NT; ++tile) {
            long long c0 = cvae_clock();
            f32x4 l0[2], l1[2];
            f32x2 l2[2];
            if (t > 0) {
                unsigned spins = 0;
                for (;;) {       // the four producers of this wave's K share
                    unsigned f = (unsigned)t;
                    if (lane < 4) f = cvae_atomic_load_agent(p.fG + (tile * 16 + i) * 16 + 4 * wave + lane);
                    if (cvae_wave_all(f >= (unsigned)t)) break;
                    cvae_sleep();
                    if (++spins > (1u << 20)) { p.status[0] = 1; break; }
                }
                cvae_compiler_fence();
                if (blockIdx.x == 0) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned so = (unsigned)((((t - 1) * NT + tile) * 16 + i) * 16 + 4 * wave + 2 * s + (kq >> 1)) * 1280u;
                    l0[s] = cvae_buf_load_f4(gb, (unsigned)((kq & 1) * 256 + lr * 16), so);
                    l1[s] = cvae_buf_load_f4(gb, (unsigned)(512 + (kq & 1) * 256 + lr * 16), so);
                    l2[s] = cvae_buf_load_f2(gb, (unsigned)(1024 + (kq & 1) * 128 + lr * 8), so);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 2; ++s) { l0[s] = l1[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; l2[s] = (f32x2){0.f, 0.f}; }
            }
            f32x4 a0[8], a1[8], a2[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) a0[n] = a1[n] = a2[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4 g2 = cvae_bf8x8_to_h8(l2[s]);
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    const f32x4 w2 = cvae_bf8x8_to_h8(*(const f32x2*)(w2w + (s * 8 + n) * 128));
                    a0[n] = cvae_mfma_16x16x32_f16(l0[s], w0[s][n], a0[n]);
                    a1[n] = cvae_mfma_16x16x32_f16(l0[s], w1[s][n], a1[n]);
                    a2[n] = cvae_mfma_16x16x32_f16(l1[s], w1[s][n], a2[n]);
                    a2[n] = cvae_mfma_16x16x32_f16(l0[s], w2, a2[n]);
                    a1[n] = cvae_mfma_16x16x32_f16(l1[s], w0[s][n], a1[n]);
                    a2[n] = cvae_mfma_16x16x32_f16(g2, w0[s][n], a2[n]);
                }
            }
            constexpr float S1 = 1.0f / 2048.0f;
#pragma unroll
            for (int n = 0; n < 8; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[(wave * 16 + kq * 4 + q) * 132 + n * 16 + lr] = a0[n][q] + (a1[n][q] + a2[n][q] * S1) * S1;
            if (blockIdx.x == 0) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
            __syncthreads();
            {   // thread (reducer r, row q): the eight columns reducer r finalises, summed over the waves; 32 B to its piece
                const int r = tid >> 4, q = tid & 15;
                f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    o0 += *(const f32x4*)(red + (w * 16 + q) * 132 + r * 8);
                    o1 += *(const f32x4*)(red + (w * 16 + q) * 132 + r * 8 + 4);
                }
                const unsigned so = (unsigned)((((t * NT + tile) * 16 + j) * 16 + r) * 16 + i) * 512u;
                cvae_buf_store_f4_sc1(pb, (unsigned)(q * 32), so, o0);
                cvae_buf_store_f4_sc1(pb, (unsigned)(q * 32 + 16), so, o1);
            }
            cvae_drain_vmem();
            __syncthreads();
            if (tid == 0) cvae_atomic_store_agent(p.fP + (tile * 16 + j) * 16 + i, (unsigned)(t + 1));
            if (blockIdx.x == 0) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        }
        // ---------------- phase 2, every tile
        for (int tile = 0; tile < NT; ++tile) {
            long long c0 = cvae_clock();
            // what the cell backward needs from the tape does not depend on the recurrence: requested in front of the poll
            float tp[7];
            if (tid < 64) {
                const float* tq = p.tape + ((long)(t * NT + tile) * 256 + blockIdx.x) * 512 + tid * 8;
#pragma unroll
                for (int q = 0; q < 7; ++q) tp[q] = tq[q];
            }
            unsigned spins = 0;
            for (;;) {
                unsigned f = (unsigned)(t + 1);
                if (lane < 16) f = cvae_atomic_load_agent(p.fP + (tile * 16 + j) * 16 + lane);
                if (cvae_wave_all(f >= (unsigned)(t + 1))) break;
                cvae_sleep();
                if (++spins > (1u << 20)) { p.status[0] = 2; break; }
            }
            cvae_compiler_fence();
            if (blockIdx.x == 0) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
            {
                const unsigned so = (unsigned)(((t * NT + tile) * 16 + j) * 16 + i) * 8192u;
                const f32x4 v0 = cvae_buf_load_f4(pb, (unsigned)(tid * 32), so), v1 = cvae_buf_load_f4(pb, (unsigned)(tid * 32 + 16), so);
                *(f32x4*)(red + tid * 8) = v0;          // [src][row][8]
                *(f32x4*)(red + tid * 8 + 4) = v1;
            }
            __syncthreads();
            if (tid < 128) {
                float s = 0.f;
#pragma unroll
                for (int src = 0; src < 16; ++src) s += red[src * 128 + tid];
                red[2048 + tid] = s;
            }
            __syncthreads();
            if (tid < 64) {
                const int row = tid >> 2, u = tid & 3;
                const float sa = red[2048 + row * 8 + u], sb = red[2048 + row * 8 + 4 + u];
                const float dht = keep[tile & 7] + sa * (1.0f / 256.0f) + tp[5] * (tp[6] + sb * (1.0f / 256.0f));
                const float r = tp[0], z = tp[1], n = tp[2], qq = tp[3], hp = tp[4];
                const float dn = dht * (1.0f - z), dz = dht * (hp - n);
                float v[4];
                v[2] = dn * (1.0f - n * n); v[3] = v[2] * r; v[0] = v[2] * qq * r * (1.0f - r); v[1] = dz * z * (1.0f - z);
                keep[tile & 7] = dht * z;
                float* dgo = p.dg + ((long)(t * NT + tile) * 256 + blockIdx.x) * 512 + tid * 8;
#pragma unroll
                for (int cm = 0; cm < 4; ++cm) {
                    dgo[cm] = v[cm];
                    unsigned short h0, h1;
                    unsigned char h2;
                    cvae_split3_f16b8(v[cm] * 256.0f, h0, h1, h2);
                    const int kl = 4 * u + cm;
                    pub[((kl >> 3) * 16 + row) * 8 + (kl & 7)] = h0;
                    pub[256 + ((kl >> 3) * 16 + row) * 8 + (kl & 7)] = h1;
                    ((unsigned char*)(pub + 512))[((kl >> 3) * 16 + row) * 8 + (kl & 7)] = h2;
                }
            }
            __syncthreads();
            if (blockIdx.x == 0) { const long long c1 = cvae_clock(); pc[4] += c1 - c0; c0 = c1; }
            if (tid < 64) {
                const unsigned so = (unsigned)(((t * NT + tile) * 16 + j) * 16 + i) * 1280u;
                cvae_buf_store_f4_sc1(gb, (unsigned)(tid * 16), so, *(const f32x4*)(pub + tid * 8));
                if (tid < 16) cvae_buf_store_f4_sc1(gb, (unsigned)(1024 + tid * 16), so, *(const f32x4*)(pub + 512 + tid * 8));
                cvae_drain_vmem();
                cvae_wave_barrier();
                if (tid == 0) cvae_atomic_store_agent(p.fG + (tile * 16 + j) * 16 + i, (unsigned)(t + 1));
            }
            if (blockIdx.x == 0) { const long long c1 = cvae_clock(); pc[5] += c1 - c0; c0 = c1; }
        }
    }
    if (blockIdx.x == 0 && tid == 0)
        for (int q = 0; q < 6; ++q) p.cyc[q] = pc[q];
}

int main(int argc, char** argv) {
    const int T = 80;
    for (int NT : {4, 8}) {
        P2 p;
        p.T = T; p.NT = NT; p.pf = 0;
        const size_t gbytes = (size_t)T * NT * 256 * 1280, pbytes = (size_t)T * NT * 256 * 8192, tbytes = (size_t)T * NT * 256 * 512 * 4;
        hipMalloc(&p.G, gbytes); hipMemset(p.G, 0, gbytes);
        hipMalloc(&p.P, pbytes); hipMemset(p.P, 0, pbytes);
        hipMalloc(&p.fG, NT * 256 * 4); hipMalloc(&p.fP, NT * 256 * 4);
        float* w; hipMalloc(&w, (size_t)256 * 4 * 16 * 640 * 4); hipMemset(w, 0, (size_t)256 * 4 * 16 * 640 * 4); p.w = w;
        float* tape; hipMalloc(&tape, tbytes); hipMemset(tape, 0, tbytes); p.tape = tape;
        hipMalloc(&p.dg, tbytes);
        hipMalloc(&p.cyc, 64); hipMalloc(&p.status, 16); hipMemset(p.status, 0, 16);
        const size_t lds = (size_t)(4 * 16 * 132 + 4 * 16 * 128) * 4 + 1280 + 64;
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(p.fG, 0, NT * 256 * 4); hipMemset(p.fP, 0, NT * 256 * 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipError_t le = cvae_launch_coop(k_bwd2d, dim3(256), dim3(256), lds, (hipStream_t)0, p);
            hipEventRecord(e1, 0);
            hipError_t e = hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long c[6]; int st[4];
            hipMemcpy(c, p.cyc, 48, hipMemcpyDeviceToHost); hipMemcpy(st, p.status, 16, hipMemcpyDeviceToHost);
            double tot = 0; for (int q = 0; q < 6; ++q) tot += (double)c[q];
            printf("tiles %d (%3d rows) T %d: %.1f us per launch = %.2f us per step | block 0 cycles per step: poll1 %.0f  load+mfma %.0f  combine+store %.0f  poll2 %.0f  reduce+cell %.0f  publish %.0f  total %.0f | status %d %s %s\n",
                   NT, NT * 16, T, 1e3 * ms, 1e3 * ms / T, c[0] / (double)T, c[1] / (double)T, c[2] / (double)T, c[3] / (double)T, c[4] / (double)T, c[5] / (double)T,
                   tot / T, st[0], le == hipSuccess ? "" : hipGetErrorString(le), e == hipSuccess ? "" : hipGetErrorString(e));
        }
        hipFree(p.G); hipFree(p.P); hipFree(w); hipFree(tape); hipFree(p.dg);
    }
    return 0;
}
