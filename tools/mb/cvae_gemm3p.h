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
// Operand format ("limb planes", in MFMA FRAGMENT ORDER): for X [R x K], R padded to a multiple of 128 and K to a multiple of 32 with
// zeros: plane l (l = 0..2) at X + l * plane (in halves) holds one KiB per (32 rows x 16 k) fragment of v_mfma_f32_32x32x16_f16 --
// [r / 32][k / 16][lane = r % 32 + 32 * (k % 16 / 8)][k % 8] -- so a wave's operand fragment is ONE contiguous KiB that goes from
// L2 straight into the registers the MFMA reads: no LDS staging, no barrier, every wave streams on its own (measured,
// tools/mb/mb_gemm3p: the LDS-staged forms -- register-staged or LDS-DMA, padded or swizzled, one or two stages ahead -- all sat at
// 2.2K cycles of load + ds_write + ds_read per 32-k stage next to 2.0K cycles of MFMAs, barely overlapped).  cvae_g3_at() is the index.
// Block = 128 x 128 outputs, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles x 3 sums = 192 accumulator registers; K stages
// of 32 through a double-buffered LDS image (rows of 32 halves + 16 bytes of padding: conflict-free ds_read_b128 fragments).
#pragma once
#include <cvae_intrin.h>

__host__ __device__ __forceinline__ long cvae_g3_at(long r, long k, long kblocks) {      // kblocks = K / 16
    return (((r >> 5) * kblocks + (k >> 4)) << 9) + (((r & 31) + 32 * ((k >> 3) & 1)) << 3) + (k & 7);
}

struct Gemm3Params {
    const unsigned short* A;     // limb planes in fragment order [3][Mp / 32][K / 16][64 lanes][8]
    long a_plane;
    const unsigned short* B;     // limb planes in fragment order [3][Np / 32][K / 16][64 lanes][8]
    long b_plane;
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
    int gx, gy, gz;              // tiles along N, along M, contraction slices; the launch is ONE-dimensional: 8 * ceil(gx gy gz / 8) blocks
    int exp;                     // measurement only (tools/mb/mb_gemm3p): bit 0 no global fetch / stash after the first stage, bit 1 no MFMAs
    int xcd_map;                 // 1: workgroup b (XCD b % 8, MI355X dispatch order) takes tile (b % 8) * per + b / 8: an XCD's blocks work on
                                 // neighbouring tiles (same A rows, consecutive B columns), so its L2 fetches a slice of the operands instead of all of them
};

#define CVAE_G3_LDS 1024                            // (only the ticket word of a split contraction)
#ifndef CVAE_G3_RING
#define CVAE_G3_RING 3                              // 16-k steps of operand fragments in flight per wave (48 registers each)
#endif

__global__ __launch_bounds__(256, 1) void k_gemm3_nt(Gemm3Params p) {
    unsigned char* sm = (unsigned char*)CVAE_SMEM;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lm = lane & 31, k8 = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned nb = (unsigned)(p.gx * p.gy * p.gz), per = (nb + 7) >> 3;
    const unsigned tix = p.xcd_map ? (blockIdx.x & 7) * per + (blockIdx.x >> 3) : blockIdx.x;
    if (tix >= nb) return;                     // (padding blocks of the one-dimensional launch)
    const int bx = (int)(tix % p.gx), by = (int)((tix / p.gx) % p.gy), bz = (int)(tix / (p.gx * p.gy));
    const int m0 = by * 128, n0 = bx * 128;
    const int kbeg = bz * p.kchunk, kend = kbeg + p.kchunk < p.K ? kbeg + p.kchunk : p.K;
    const long kblocks = p.K >> 4;
    int ar = m0 + wm * 64;
    if (m0 >= p.a_brk) ar += p.a_skip;
    // fragment (32-row block rb, 16-k step kk) of plane l: X + l * plane + ((rb * kblocks + kk) << 9) + lane * 8 halves
    const unsigned short* ga = p.A + (((long)(ar >> 5) * kblocks) << 9) + lane * 8;
    const unsigned short* gb = p.B + (((long)((n0 + wn * 64) >> 5) * kblocks) << 9) + lane * 8;
    const long rstep = kblocks << 9;           // to the next 32-row block
    f32x16 acc[2][2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][s][q] = 0.0f;
    constexpr int RD = CVAE_G3_RING;
    f32x4 fa[RD][2][3], fb[RD][2][3];          // [ring slot][tile][plane]
    auto load = [&](int slot, int kk) {        // 12 KiB per wave and 16-k step, every fragment one contiguous KiB
        const long ko = (long)kk << 9;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[slot][i][pl] = *(const f32x4*)(ga + pl * p.a_plane + i * rstep + ko);
                fb[slot][i][pl] = *(const f32x4*)(gb + pl * p.b_plane + i * rstep + ko);
            }
    };
    // term by term over the four tiles: an accumulator is touched again four MFMAs later at the earliest
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][0] = cvae_mfma_32x32x16_f16(fa[slot][i][0], fb[slot][j][0], acc[i][j][0]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][1] = cvae_mfma_32x32x16_f16(fa[slot][i][0], fb[slot][j][1], acc[i][j][1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][2] = cvae_mfma_32x32x16_f16(fa[slot][i][1], fb[slot][j][1], acc[i][j][2]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][1] = cvae_mfma_32x32x16_f16(fa[slot][i][1], fb[slot][j][0], acc[i][j][1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][2] = cvae_mfma_32x32x16_f16(fa[slot][i][0], fb[slot][j][2], acc[i][j][2]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j][2] = cvae_mfma_32x32x16_f16(fa[slot][i][2], fb[slot][j][0], acc[i][j][2]);
    };
    const int k0 = kbeg >> 4, nk = kend > kbeg ? (kend - kbeg) >> 4 : 0;
#pragma unroll
    for (int q = 0; q < RD; ++q)
        if (q < nk) load(q, k0 + q);
    for (int kk = 0; kk < nk; kk += RD) {
#pragma unroll
        for (int q = 0; q < RD; ++q) {
            if (kk + q < nk) {
                if (!(p.exp & 2)) mfmas(q);
                cvae_sched_fence();
                if (kk + q + RD < nk && !(p.exp & 1)) load(q, k0 + kk + q + RD);
            }
        }
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
        const int nz = p.gz;
        const unsigned tile = by * p.gx + bx, ntile = p.gx * p.gy;
        const cvae_buf pb = cvae_make_buf(p.part, (unsigned)((size_t)nz * ntile * 65536));
        const unsigned mine = ((unsigned)bz * ntile + tile) * 65536u + (unsigned)tid * 16u;
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

// fp32 X [R x C] (row stride ldx) -> blocked limb planes out[l][cvae_g3_at(r, c, Cp / 16)] for r < Rp, c < Cp (zeros outside R x C);
// values are multiplied by `scale` first.  One thread per 8 consecutive columns (16-byte stores).
__global__ void k_split3_rows(const float* __restrict__ X, long ldx, int R, int C, unsigned short* __restrict__ out, long plane,
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
        *(f32x4*)(out + pl * plane + cvae_g3_at(r, c0, Cp >> 4)) = w;
    }
}

// fp32 X [R x C] (row stride ldx) -> TRANSPOSED blocked limb planes out[l][cvae_g3_at(crow0 + c, r, Rp / 16)] for c < Cp, r < Rp (zeros
// outside R x C): the contraction index of a weight-gradient product (the time-major rows) becomes the contiguous one.  Block =
// 64 x 64 tile through LDS.
__global__ __launch_bounds__(256) void k_split3_t(const float* __restrict__ X, long ldx, int R, int C, unsigned short* __restrict__ out, long plane,
                                                   int crow0, int Rp, int Cp, float scale) {
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
            *(f32x4*)(out + pl * plane + cvae_g3_at(crow0 + c0 + cc, r0 + pc * 8, Rp >> 4)) = *(const f32x4*)(t + (pl * 64 + cc) * 72 + pc * 8);
    }
}
