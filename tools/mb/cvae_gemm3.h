// MEASUREMENT ONLY (tools/mb/mb_gemm3.hip): not part of the library.  Result on MI355X (profiles/r03_notes_training.md): 110-125 TFLOP/s
// fp32-equivalent, no better than the fp32-input MFMA GEMMs it was meant to replace once the split passes are counted.
// GEMMs of the training step with fp32-exact products on the f16 matrix pipe: C[m][n] = sum_k A[m][k] * B[n][k], every operand
// the exact sum of three fp16 limbs (x = l0 + l1/2^11 + l2/2^22, cvae_split3_f16), a product accumulated as
//     S0 = a0.b0    S1 = a0.b1 + a1.b0    S2 = a1.b1 + a0.b2 + a2.b0     C = S0 + (S1 + S2/2^11)/2^11
// six v_mfma_f32_32x32x16_f16 per 16 k (192 matrix-pipe cycles per 32 x 32 x 16 block product) where the fp32-input MFMA needs
// eight v_mfma_f32_32x32x2_f32 (512 cycles).  The reference does these contractions in fp32 (conv / GRU input GEMMs and their
// autograd, gru_vae.py:353-357, :392; train...:1419); dropped terms are below 2^-33 of a product.
//
// Operands are split ONCE per use by k_split3 into a "tile image": [rows/128][K/32] blocks of 24 KiB =
//     [2 steps of 16 k][3 limbs][2 kh][128 rows][8 halves]
// i.e. exactly what a workgroup stages in LDS for one 32-k stage of a 128-row operand tile, lane-linear for the MFMA operand
// fragment (lane (r, kh) of a 32-row sub-tile reads 16 B at ((step*3 + limb)*2 + kh)*2048 + (32*sub + r)*16: conflict-free
// ds_read_b128).  The staging loads are therefore six perfectly coalesced 16-byte loads per thread and stage.  The split kernel
// reads the fp32 matrix either as it lies ([row][k], k contiguous) or TRANSPOSED ([k][row]: the weight-gradient contractions run
// over the rows of both operands), pads rows and K with zeros.
#pragma once
#include <cvae_intrin.h>

#define CVAE_G3_TILE_HALVES 12288   // 24 KiB

// src element (row r, contraction index k) = transposed ? src[k * ld + r] : src[r * ld + k];  r < rows, k < K, else 0.
// seglen > 0 (non-transposed only): "segmented rows" as in k_gemm_nt_seg: k = j*seglen + c reads src[r * ld + j * segstride + c].
// dst: tile image, rows_p = up(rows, 128), Kp = up(K, 32).  One thread per (tile, 16-byte piece).
__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ src, long ld, int rows, int K, int transposed, int seglen,
                                                long segstride, unsigned short* __restrict__ dst, int rows_p, int Kp) {
    __shared__ float tile[32][129];                       // transposed source: a 32 (k) x 128 (rows) patch
    const int tk = blockIdx.x, tr = blockIdx.y, tid = threadIdx.x;   // one block per (row tile, 32-k stage)
    unsigned short* out = dst + ((long)tr * (Kp >> 5) + tk) * CVAE_G3_TILE_HALVES;
    const int r0 = tr * 128, k0 = tk * 32;
    if (transposed) {
        // coalesced along rows: thread reads src[(k0 + kk) * ld + r0 + c]
        for (int e = tid; e < 32 * 128; e += 256) {
            const int kk = e >> 7, c = e & 127;
            tile[kk][c] = (k0 + kk < K && r0 + c < rows) ? src[(long)(k0 + kk) * ld + r0 + c] : 0.0f;
        }
        __syncthreads();
    }
    // 128 rows x 4 pieces of 8 k: thread (row, piece)
    for (int e = tid; e < 512; e += 256) {
        const int row = e & 127, piece = e >> 7, step = piece >> 1, kh = piece & 1;
        unsigned short l0[8], l1[8], l2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kk = 8 * piece + q, k = k0 + kk;
            float v;
            if (transposed) {
                v = tile[kk][row];
            } else if (k < K && r0 + row < rows) {
                const long off = seglen > 0 ? (long)(k / seglen) * segstride + k % seglen : k;
                v = src[(long)(r0 + row) * ld + off];
            } else {
                v = 0.0f;
            }
            cvae_split3_f16(v, l0[q], l1[q], l2[q]);
        }
        auto pack = [](const unsigned short* h) {
            return (f32x4){__builtin_bit_cast(float, (unsigned)h[0] | ((unsigned)h[1] << 16)), __builtin_bit_cast(float, (unsigned)h[2] | ((unsigned)h[3] << 16)),
                           __builtin_bit_cast(float, (unsigned)h[4] | ((unsigned)h[5] << 16)), __builtin_bit_cast(float, (unsigned)h[6] | ((unsigned)h[7] << 16))};
        };
        unsigned short* o = out + ((step * 3) * 2 + kh) * 1024 + row * 8;
        *(f32x4*)(o) = pack(l0);
        *(f32x4*)(o + 2 * 1024) = pack(l1);
        *(f32x4*)(o + 4 * 1024) = pack(l2);
    }
}

struct Gemm3Epi {
    const float* bias;      // [N] or null
    int accumulate;         // C += result
    const float* mask;      // EpiMask of the fp32 GEMMs: C[(f*Bp + b)][c] *= mask[(b*T + f)*N + c], rows b >= B -> 0; null: none
    int mB, mBp, mT;
};

// C[M][N] (ldc) = A3 . B3^T over the padded K; A3 / B3: tile images of rows_p(M) x Kp and rows_p(N) x Kp.  Block = 128 x 128 of C,
// 4 waves in 2 x 2, each 64 x 64 = 2 x 2 MFMA tiles with S0 / S1 / S2 accumulators (192 registers); one 32-k stage per barrier,
// global -> register -> LDS double buffering.  kz > 1: blockIdx.z takes every kz-th stage and writes a partial tile to
// part[z][M][N] (summed in fixed order by k_sum_parts).
__global__ __launch_bounds__(256, 1) void k_gemm3_nt(const unsigned short* __restrict__ A3, const unsigned short* __restrict__ B3, float* __restrict__ C,
                                                     long ldc, int M, int N, int Kp, Gemm3Epi ep, float* __restrict__ part) {
    constexpr float S1 = 1.0f / 2048.0f;
    unsigned short* sm = (unsigned short*)CVAE_SMEM;      // [2 buffers][A tile | B tile] = 2 x 48 KiB
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lc = lane & 31, kh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int nst = Kp >> 5, kz = gridDim.z, z = blockIdx.z;
    const unsigned short* Ag = A3 + (long)blockIdx.y * nst * CVAE_G3_TILE_HALVES;
    const unsigned short* Bg = B3 + (long)blockIdx.x * nst * CVAE_G3_TILE_HALVES;
    f32x16 s0[2][2], s1[2][2], s2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) { s0[i][j][q] = 0.f; s1[i][j][q] = 0.f; s2[i][j][q] = 0.f; }
    f32x4 ga[6], gb[6];
    auto gload = [&](int st) {
        const unsigned short* a = Ag + (long)st * CVAE_G3_TILE_HALVES + tid * 8;
        const unsigned short* b = Bg + (long)st * CVAE_G3_TILE_HALVES + tid * 8;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            ga[u] = *(const f32x4*)(a + u * 2048);
            gb[u] = *(const f32x4*)(b + u * 2048);
        }
    };
    auto sstore = [&](int buf) {
        unsigned short* a = sm + buf * 2 * CVAE_G3_TILE_HALVES + tid * 8;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            *(f32x4*)(a + u * 2048) = ga[u];
            *(f32x4*)(a + CVAE_G3_TILE_HALVES + u * 2048) = gb[u];
        }
    };
    int st = z;
    if (st < nst) {
        gload(st);
        sstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (; st < nst; st += kz) {
        const bool more = st + kz < nst;
        if (more) gload(st + kz);
        const unsigned short* a = sm + buf * 2 * CVAE_G3_TILE_HALVES;
        const unsigned short* b = a + CVAE_G3_TILE_HALVES;
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            f32x4 af[2][3], bf[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    af[i][m] = *(const f32x4*)(a + ((step * 3 + m) * 2 + kh) * 1024 + (64 * wm + 32 * i + lc) * 8);
                    bf[i][m] = *(const f32x4*)(b + ((step * 3 + m) * 2 + kh) * 1024 + (64 * wn + 32 * i + lc) * 8);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    s0[i][j] = cvae_mfma_32x32x16_f16(af[i][0], bf[j][0], s0[i][j]);
                    s1[i][j] = cvae_mfma_32x32x16_f16(af[i][0], bf[j][1], s1[i][j]);
                    s2[i][j] = cvae_mfma_32x32x16_f16(af[i][1], bf[j][1], s2[i][j]);
                    s1[i][j] = cvae_mfma_32x32x16_f16(af[i][1], bf[j][0], s1[i][j]);
                    s2[i][j] = cvae_mfma_32x32x16_f16(af[i][0], bf[j][2], s2[i][j]);
                    s2[i][j] = cvae_mfma_32x32x16_f16(af[i][2], bf[j][0], s2[i][j]);
                }
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // epilogue: D[row = (q&3) + 8*(q>>2) + 4*kh][col = lc] of each 32 x 32 tile
    const int m0 = blockIdx.y * 128 + 64 * wm, n0 = blockIdx.x * 128 + 64 * wn;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + 32 * j + lc;
            if (col >= N) continue;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = m0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (row >= M) continue;
                float v = s0[i][j][q] + (s1[i][j][q] + s2[i][j][q] * S1) * S1;
                if (part) {
                    part[((long)z * M + row) * N + col] = v;
                    continue;
                }
                if (ep.bias) v += ep.bias[col];
                if (ep.mask) {
                    const int f = row / ep.mBp, bb = row - f * ep.mBp;
                    v = bb < ep.mB ? v * ep.mask[((long)bb * ep.mT + f) * N + col] : 0.0f;
                }
                float* c = C + (long)row * ldc + col;
                *c = v + (ep.accumulate ? *c : 0.0f);
            }
        }
}


// Second form: the same block tile with EIGHT waves (two per SIMD), wave (wm, wn) = 64 rows x 32 columns = 2 x 1 MFMA tiles (96
// accumulator registers), so that one wave's LDS reads, staging stores and barrier waits run under its SIMD partner's MFMAs.
__global__ __launch_bounds__(512, 2) void k_gemm3_nt8(const unsigned short* __restrict__ A3, const unsigned short* __restrict__ B3, float* __restrict__ C,
                                                      long ldc, int M, int N, int Kp, Gemm3Epi ep, float* __restrict__ part) {
    constexpr float S1 = 1.0f / 2048.0f;
    unsigned short* sm = (unsigned short*)CVAE_SMEM;      // [2 buffers][A tile | B tile] = 2 x 48 KiB
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lc = lane & 31, kh = lane >> 5;
    const int wm = wave >> 2, wn = wave & 3;
    const int nst = Kp >> 5, kz = gridDim.z, z = blockIdx.z;
    const unsigned short* Ag = A3 + (long)blockIdx.y * nst * CVAE_G3_TILE_HALVES;
    const unsigned short* Bg = B3 + (long)blockIdx.x * nst * CVAE_G3_TILE_HALVES;
    f32x16 s0[2], s1[2], s2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) { s0[i][q] = 0.f; s1[i][q] = 0.f; s2[i][q] = 0.f; }
    f32x4 ga[3], gb[3];
    auto gload = [&](int st) {
        const unsigned short* a = Ag + (long)st * CVAE_G3_TILE_HALVES + tid * 8;
        const unsigned short* b = Bg + (long)st * CVAE_G3_TILE_HALVES + tid * 8;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            ga[u] = *(const f32x4*)(a + u * 4096);
            gb[u] = *(const f32x4*)(b + u * 4096);
        }
    };
    auto sstore = [&](int buf) {
        unsigned short* a = sm + buf * 2 * CVAE_G3_TILE_HALVES + tid * 8;
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            *(f32x4*)(a + u * 4096) = ga[u];
            *(f32x4*)(a + CVAE_G3_TILE_HALVES + u * 4096) = gb[u];
        }
    };
    int st = z;
    if (st < nst) {
        gload(st);
        sstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (; st < nst; st += kz) {
        const bool more = st + kz < nst;
        if (more) gload(st + kz);
        const unsigned short* a = sm + buf * 2 * CVAE_G3_TILE_HALVES;
        const unsigned short* b = a + CVAE_G3_TILE_HALVES;
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            f32x4 af[2][3], bf[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
#pragma unroll
                for (int i = 0; i < 2; ++i) af[i][m] = *(const f32x4*)(a + ((step * 3 + m) * 2 + kh) * 1024 + (64 * wm + 32 * i + lc) * 8);
                bf[m] = *(const f32x4*)(b + ((step * 3 + m) * 2 + kh) * 1024 + (32 * wn + lc) * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) s0[i] = cvae_mfma_32x32x16_f16(af[i][0], bf[0], s0[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) s1[i] = cvae_mfma_32x32x16_f16(af[i][0], bf[1], s1[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) s2[i] = cvae_mfma_32x32x16_f16(af[i][1], bf[1], s2[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) s1[i] = cvae_mfma_32x32x16_f16(af[i][1], bf[0], s1[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) s2[i] = cvae_mfma_32x32x16_f16(af[i][0], bf[2], s2[i]);
#pragma unroll
            for (int i = 0; i < 2; ++i) s2[i] = cvae_mfma_32x32x16_f16(af[i][2], bf[0], s2[i]);
        }
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    const int m0 = blockIdx.y * 128 + 64 * wm, n0 = blockIdx.x * 128 + 32 * wn;
    const int col = n0 + lc;
    if (col >= N) return;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = m0 + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (row >= M) continue;
            float v = s0[i][q] + (s1[i][q] + s2[i][q] * S1) * S1;
            if (part) {
                part[((long)z * M + row) * N + col] = v;
                continue;
            }
            if (ep.bias) v += ep.bias[col];
            if (ep.mask) {
                const int f = row / ep.mBp, bb = row - f * ep.mBp;
                v = bb < ep.mB ? v * ep.mask[((long)bb * ep.mT + f) * N + col] : 0.0f;
            }
            float* c = C + (long)row * ldc + col;
            *c = v + (ep.accumulate ? *c : 0.0f);
        }
}
