// k_gemm3_nt: C[M x N] (+)= scale * A[M x K] . B[N x K]^T (+ bias) with EXACT fp32 operands carried as three fp16 limb planes
// (x = l0 + l1/2^11 + l2/2^22, cvae_split3_f16: the arithmetic of the exact-operand recurrences, cvae_exact3.h) and fp32
// accumulation: S0 = a0 b0, S1 = a0 b1 + a1 b0, S2 = a1 b1 + a0 b2 + a2 b0, C = S0 + (S1 + S2/2^11)/2^11 -- six
// v_mfma_f32_32x32x16_f16 per 16 k; the dropped terms are below 2^-33 of a product.  The f16 matrix pipe runs 16x the rate of the
// fp32-input MFMA (2,500 vs 157 TFLOP/s dense), so six products cost 3/8 of one fp32-input product.
//
// The fp32-input GEMMs of the training step (k_gemm_nt2 / k_gemm_tn2) convert nothing but are bound by that pipe at 65-100 TFLOP/s;
// a version that split its operands on the way into LDS (k_gemm_tn3, round 5, not kept) paid ~7 VALU operations per element per
// block that loads it and came out even.  Here the split happens ONCE per element, in a bandwidth-bound pass of its own
// (k_split3_rows / k_split3_t: also the transposition that turns a weight-gradient contraction over rows into this NT form), and
// the weights are split when the train image is built; the GEMM's inner loop is loads, LDS traffic and MFMAs only.
//
// Operand format ("limb planes"): for X [R x K]: halves X_l[r * ld + k], l = 0..2, plane l at X + l * plane (in halves); R padded
// to a multiple of 128 and K to a multiple of 32 with zeros, so tile loads need no bounds checks; ld in halves, a multiple of 8.
// Block = 128 x 128 outputs, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles x 3 sums = 192 accumulator registers; K stages
// of 32 through a double-buffered LDS image (rows of 32 halves + 16 bytes of padding: conflict-free ds_read_b128 fragments).
#pragma once
#include <cvae_intrin.h>

struct Gemm3Params {
    const unsigned short* A;     // limb planes [3][Mp][lda]
    long a_plane, lda;
    const unsigned short* B;     // limb planes [3][Np][ldb]
    long b_plane, ldb;
    float* C;
    long ldc;
    const float* bias;           // [N] or null
    int M, N, K;                 // K: multiple of 32 (the planes' zero padding included)
    int a_brk, a_skip;           // rows r >= a_brk of A are read from row r + a_skip (two row ranges of one plane set; multiples of 128)
    int accumulate;
    float scale;                 // the product is multiplied by this (operands that travel scaled, e.g. gate gradients x 2^8)
    int kchunk;                  // K per blockIdx.z slice (multiple of 32)
    float* part;                 // split contraction: [slices][tiles][128 x 128] partial sums + arrival counters, or null
    unsigned* cnt;
    const float* mask;           // optional epilogue (cvae_epi_mask): batch-major dropout mask [B][T][N]
    int mB, mBp, mT;
};

#define CVAE_G3_RSB 80           // bytes per LDS row: 32 halves + 16 bytes
#define CVAE_G3_PLANE (128 * CVAE_G3_RSB)
#define CVAE_G3_STAGE (6 * CVAE_G3_PLANE)          // A planes 0..2, B planes 0..2
#define CVAE_G3_LDS (2 * CVAE_G3_STAGE)

__global__ __launch_bounds__(256, 1) void k_gemm3_nt(Gemm3Params p) {
    unsigned char* sm = (unsigned char*)CVAE_SMEM;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lm = lane & 31, k8 = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int kbeg = blockIdx.z * p.kchunk, kend = kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K;
    // global -> LDS pieces of 8 halves: 512 per plane and operand (128 rows x 4), two per thread
    const unsigned short* ga[2];
    const unsigned short* gb[2];
    int so[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int id = tid + 256 * h, r = id >> 2, pc = id & 3;
        int ar = m0 + r;
        if (ar >= p.a_brk) ar += p.a_skip;
        ga[h] = p.A + (long)ar * p.lda + pc * 8;
        gb[h] = p.B + (long)(n0 + r) * p.ldb + pc * 8;
        so[h] = r * CVAE_G3_RSB + pc * 16;
    }
    // Software pipeline (one wave per SIMD: nothing else hides a latency).  Global loads run TWO stages ahead of the MFMAs that
    // use them (two register sets), LDS fragment reads one 16-k step ahead (two fragment sets):
    //   stage s (LDS buffer s & 1):  read F1(s) | fetch(s + 2) | MFMAs on F0(s) | stash(s + 1) -> other buffer | barrier |
    //                                read F0(s + 1) | MFMAs on F1(s)
    // The barrier (behind a wait for the wave's own LDS reads) both publishes stage s + 1 and retires buffer s & 1: the next
    // stage's stash may overwrite it.
    f32x4 gr[2][2][3][2];        // [set][operand][plane][piece]
    auto fetch = [&](int set, int k) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                gr[set][0][pl][h] = *(const f32x4*)(ga[h] + pl * p.a_plane + k);
                gr[set][1][pl][h] = *(const f32x4*)(gb[h] + pl * p.b_plane + k);
            }
    };
    auto stash = [&](int set, int buf) {
        unsigned char* st = sm + buf * CVAE_G3_STAGE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                *(f32x4*)(st + pl * CVAE_G3_PLANE + so[h]) = gr[set][0][pl][h];
                *(f32x4*)(st + (3 + pl) * CVAE_G3_PLANE + so[h]) = gr[set][1][pl][h];
            }
    };
    f32x16 acc[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][s][q] = 0.0f;
    const int aoff = (wm * 64 + lm) * CVAE_G3_RSB + k8 * 16, boff = (wn * 64 + lm) * CVAE_G3_RSB + k8 * 16;
    f32x4 fa[2][2][3], fb[2][2][3];      // [fragment set][tile][plane]
    auto frags = [&](int set, int buf, int ks) {
        const unsigned char* st = sm + buf * CVAE_G3_STAGE;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[set][i][pl] = *(const f32x4*)(st + pl * CVAE_G3_PLANE + aoff + i * 32 * CVAE_G3_RSB + ks * 32);
                fb[set][i][pl] = *(const f32x4*)(st + (3 + pl) * CVAE_G3_PLANE + boff + i * 32 * CVAE_G3_RSB + ks * 32);
            }
    };
    // term by term over the four tiles: an accumulator is touched again four MFMAs later at the earliest
    auto mfmas = [&](int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][0] = cvae_mfma_32x32x16_f16(fa[set][i][0], fb[set][j][0], acc[i][j][0]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][1] = cvae_mfma_32x32x16_f16(fa[set][i][0], fb[set][j][1], acc[i][j][1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][2] = cvae_mfma_32x32x16_f16(fa[set][i][1], fb[set][j][1], acc[i][j][2]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][1] = cvae_mfma_32x32x16_f16(fa[set][i][1], fb[set][j][0], acc[i][j][1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][2] = cvae_mfma_32x32x16_f16(fa[set][i][0], fb[set][j][2], acc[i][j][2]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][2] = cvae_mfma_32x32x16_f16(fa[set][i][2], fb[set][j][0], acc[i][j][2]);
    };
    const int nst = kend > kbeg ? (kend - kbeg) >> 5 : 0;
    if (nst > 0) {
        fetch(0, kbeg);
        stash(0, 0);
        if (nst > 1) fetch(1, kbeg + 32);
    }
    __syncthreads();
    if (nst > 0) frags(0, 0, 0);
    auto stage = [&](int s, int par) {         // par = s & 1, a compile-time constant at both call sites
        frags(1, par, 1);
        if (s + 2 < nst) fetch(par, kbeg + 32 * (s + 2));       // (set `par` was stashed one stage ago)
        cvae_sched_fence();
        mfmas(0);
        cvae_sched_fence();
        if (s + 1 < nst) stash(par ^ 1, par ^ 1);
        cvae_drain_lgkm();
        __syncthreads();
        if (s + 1 < nst) frags(0, par ^ 1, 0);
        cvae_sched_fence();
        mfmas(1);
        cvae_sched_fence();
    };
    for (int s = 0; s < nst; s += 2) {
        stage(s, 0);
        if (s + 1 < nst) stage(s + 1, 1);
    }
    // C = scale * (S0 + (S1 + S2 / 2^11) / 2^11)
    constexpr float S1 = 1.0f / 2048.0f;
    f32x16 c[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) c[i][j][q] = (acc[i][j][0][q] + (acc[i][j][1][q] + acc[i][j][2][q] * S1) * S1) * p.scale;
    if (p.part) {
        // split contraction: slabs in accumulator order, ticket, the last arriver adds them in slice order (cvae_split_combine)
        const int nz = gridDim.z;
        const unsigned tile = blockIdx.y * gridDim.x + blockIdx.x, ntile = gridDim.x * gridDim.y;
        const cvae_buf pb = cvae_make_buf(p.part, (unsigned)((size_t)nz * ntile * 65536));
        const unsigned mine = ((unsigned)blockIdx.z * ntile + tile) * 65536u + (unsigned)tid * 16u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
                    cvae_buf_store_f4_sc1(pb, mine + (unsigned)(((i * 2 + j) * 4 + q4) * 4096), 0,
                                          (f32x4){c[i][j][4 * q4], c[i][j][4 * q4 + 1], c[i][j][4 * q4 + 2], c[i][j][4 * q4 + 3]});
        cvae_drain_vmem();
        __syncthreads();
        unsigned* tk = (unsigned*)sm;
        if (tid == 0) tk[0] = cvae_atomic_add_agent(p.cnt + tile, 1u);
        __syncthreads();
        if (tk[0] != (unsigned)(nz - 1)) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                    for (int z = 0; z < nz; ++z)
                        v += cvae_buf_load_f4_sc1(pb, ((unsigned)z * ntile + tile) * 65536u + (unsigned)(((i * 2 + j) * 4 + q4) * 4096) + (unsigned)tid * 16u, 0);
#pragma unroll
                    for (int e = 0; e < 4; ++e) c[i][j][4 * q4 + e] = v[e];
                }
        if (tid == 0) cvae_atomic_store_agent(p.cnt + tile, 0u);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn * 64 + j * 32 + lm;
            if (col >= p.N) continue;
            const float bv = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int rowi = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * k8;
                if (rowi < p.M) {
                    float* cp = p.C + (long)rowi * p.ldc + col;
                    float v = c[i][j][q] + bv + (p.accumulate ? *cp : 0.0f);
                    if (p.mask) {
                        const int b = rowi % p.mBp, f = rowi / p.mBp;
                        v = b < p.mB ? v * p.mask[((long)b * p.mT + f) * p.N + col] : 0.0f;
                    }
                    *cp = v;
                }
            }
        }
}

// fp32 X [R x C] (row stride ldx) -> limb planes out[l][r * ldo + c] for r < Rp, c < Cp (zeros outside R x C); values are
// multiplied by `scale` first.  One thread per 8 consecutive columns (16-byte stores).
__global__ void k_split3_rows(const float* __restrict__ X, long ldx, int R, int C, unsigned short* __restrict__ out, long plane, long ldo,
                              int Rp, int Cp, float scale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c8 = Cp >> 3;
    if (idx >= (long)Rp * c8) return;
    const int r = (int)(idx / c8), c0 = (int)(idx % c8) * 8;
    unsigned short l[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = (r < R && c0 + e < C) ? X[(long)r * ldx + c0 + e] * scale : 0.0f;
        cvae_split3_f16(v, l[0][e], l[1][e], l[2][e]);
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
        f32x4 w;
        __builtin_memcpy(&w, l[pl], 16);
        *(f32x4*)(out + pl * plane + (long)r * ldo + c0) = w;
    }
}

// fp32 X [R x C] (row stride ldx) -> TRANSPOSED limb planes out[l][(crow0 + c) * ldo + r] for c < Cp, r < Rp (zeros outside R x C):
// the contraction index of a weight-gradient product (the time-major rows) becomes the contiguous one.  Block = 64 x 64 tile
// through LDS.
__global__ __launch_bounds__(256) void k_split3_t(const float* __restrict__ X, long ldx, int R, int C, unsigned short* __restrict__ out, long plane,
                                                   long ldo, int crow0, int Rp, int Cp, float scale) {
    unsigned short* t = (unsigned short*)CVAE_SMEM;        // [3][64 c][72 r]
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, rr = e >> 6, cc = e & 63;
        const float v = (r0 + rr < R && c0 + cc < C) ? X[(long)(r0 + rr) * ldx + c0 + cc] * scale : 0.0f;
        unsigned short a, b, c;
        cvae_split3_f16(v, a, b, c);
        t[(0 * 64 + cc) * 72 + rr] = a;
        t[(1 * 64 + cc) * 72 + rr] = b;
        t[(2 * 64 + cc) * 72 + rr] = c;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int e = tid + 256 * it;           // 3 planes x 64 c x 8 pieces of 8 halves
        const int pl = e >> 9, cc = (e >> 3) & 63, pc = e & 7;
        if (c0 + cc < Cp && r0 + pc * 8 < Rp)
            *(f32x4*)(out + pl * plane + (long)(crow0 + c0 + cc) * ldo + r0 + pc * 8) = *(const f32x4*)(t + (pl * 64 + cc) * 72 + pc * 8);
    }
}
