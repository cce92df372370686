// Training-side kernels: train-mode forward with dropout, BPTT backward, Adam.
//
// Layout rule for every [frames x batch] buffer here: TIME-MAJOR, row = frame*Bp + b (Bp = batch rows padded to 16).
// A frame shift (conv taps, y_{t-1}, h_{t-1}) is then a constant row offset for the whole batch, so convolutions and
// their gradients are GEMMs whose A (or B) rows consist of `nseg` equal segments `segstride` floats apart.
#pragma once
#include <cvae_intrin.h>
#include <stdint.h>

// C[m][n] (+)= sum_k A[m*lda + seg(k)] * Bm[n*ldb + k] + bias[n],  seg(k) = (k / seglen)*segstride + k % seglen.
// K and seglen are multiples of 16 (so a 16-k chunk never straddles segments); segstride may be negative.
template <int TM, int TN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void k_gemm_nt_seg(const float* __restrict__ A, long lda, int seglen,
                                                                long segstride, const float* __restrict__ Bm, long ldb,
                                                                const float* __restrict__ bias, float* __restrict__ C,
                                                                long ldc, int M, int N, int K, int accumulate) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m0 = (blockIdx.y * WGM + wm) * TM * 16, n0 = (blockIdx.x * WGN + wn) * TN * 16;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long arow[TM], brow[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        int r = m0 + 16 * i + lr;
        r = r < M ? r : M - 1;
        arow[i] = (long)r * lda + 4 * kq;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int c = n0 + 16 * j + lr;
        c = c < N ? c : N - 1;
        brow[j] = (long)c * ldb + 4 * kq;
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        const long aoff = (long)(k0 / seglen) * segstride + (k0 % seglen);
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const float4*)(A + arow[i] + aoff);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const float4*)(Bm + brow[j] + k0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                acc[i][j] = cvae_mfma_16x16x4(a[i].x, b[j].x, acc[i][j]);
                acc[i][j] = cvae_mfma_16x16x4(a[i].y, b[j].y, acc[i][j]);
                acc[i][j] = cvae_mfma_16x16x4(a[i].z, b[j].z, acc[i][j]);
                acc[i][j] = cvae_mfma_16x16x4(a[i].w, b[j].w, acc[i][j]);
            }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + 16 * j + lr;
            const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = m0 + 16 * i + 4 * kq + r;
                if (rowi < M && col < N) {
                    float* c = C + (long)rowi * ldc + col;
                    *c = acc[i][j][r] + bv + (accumulate ? *c : 0.0f);
                }
            }
        }
}

// C[n1*ldc + n2] (+)= sum_m A[m*lda + n1] * Bm[m*ldb + seg(n2)]   (contraction over ROWS: weight gradients)
// seg(n2) = (n2 / seglen)*segstride + n2 % seglen.  Any M (rows beyond it count as zeros).  Block tile 64 x 64, waves 2 x 2 of 32 x 32.
__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ A, long lda, const float* __restrict__ Bm,
                                                 long ldb, int seglen, long segstride, float* __restrict__ C, long ldc,
                                                 int M, int N1, int N2, int accumulate) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int a0 = blockIdx.y * 64 + wm * 32, b0 = blockIdx.x * 64 + wn * 32;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    long acol[2], bcol[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int c = a0 + 16 * i + lr;
        acol[i] = c < N1 ? c : N1 - 1;
        int d = b0 + 16 * i + lr;
        d = d < N2 ? d : N2 - 1;
        bcol[i] = (long)(d / seglen) * segstride + (d % seglen);
    }
    for (int m0 = 0; m0 < M; m0 += 16) {
        float a[2][4], b[2][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long r = m0 + 4 * kq + q;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][q] = r < M ? A[r * lda + acol[i]] : 0.0f;
                b[i][q] = r < M ? Bm[r * ldb + bcol[i]] : 0.0f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = cvae_mfma_16x16x4(a[i][q], b[j][q], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = b0 + 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = a0 + 16 * i + 4 * kq + r;
                if (rowi < N1 && col < N2) {
                    float* c = C + (long)rowi * ldc + col;
                    *c = acc[i][j][r] + (accumulate ? *c : 0.0f);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------------
// LDS-tiled fp32 MFMA GEMMs for the big [frames*batch]-row products (weight gradients, input-gate GEMM, dx).
// Block tile (32*TM) x (32*TN), 2 x 2 waves of (16*TM) x (16*TN); 16 contraction rows per stage, two LDS stages, one
// barrier per stage.  Both operand tiles sit in LDS K-MAJOR ([16][tile + 4]); lane (i = lane & 15, kq = lane >> 4) feeds
// MFMA sub-step s with row 4*kq + s of both tiles (any K order is fine as long as A and B agree), so a wave's 64
// ds_read_b32 hit 64 different banks (leading dimension = 4 mod 16).
// ------------------------------------------------------------------------------------------------------
template <int TM, int TN>
struct GemmTileCfg {
    static constexpr int BM = 32 * TM, BN = 32 * TN, LDA = BM + 4, LDB = BN + 4, STAGE = 16 * (LDA + LDB);
    static constexpr int NA = (4 * BM + 255) / 256, NB = (4 * BN + 255) / 256;   // float4 per thread and operand tile
    static constexpr size_t lds_bytes = (size_t)2 * STAGE * sizeof(float);
};

template <int TM, int TN>
__device__ __forceinline__ void cvae_gemm_tile_stage(const float* As, const float* Bs, int wm, int wn, int lr, int kq,
                                                     f32x4 (&acc)[TM][TN]) {
    using G = GemmTileCfg<TM, TN>;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = As[(4 * kq + s) * G::LDA + wm * 16 * TM + 16 * i + lr];
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = Bs[(4 * kq + s) * G::LDB + wn * 16 * TN + 16 * j + lr];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = cvae_mfma_16x16x4(a[i], b[j], acc[i][j]);
    }
}

// In-launch combine of a split contraction (gridDim.z slices per output tile): every slice block writes its accumulators as a
// slab in ACCUMULATOR ORDER ([slice][tile][i][j][thread] x 16 bytes: one coalesced write-through store per thread and MFMA tile),
// drains them, takes a ticket on the tile's counter; the block that draws the last ticket adds the slabs of all slices IN SLICE
// ORDER (the result does not depend on which block arrives last: deterministic), resets the counter for the next launch and
// runs the normal epilogue.  sc1 stores + sc1 loads on both sides: no L2 write-back / L1 invalidate fence (cdna guide 6 G16,
// "split-K arrival counters").  Returns true in the block that holds the sums, false in the others (they are done).
// Replaces the separate k_sum_parts launch of rounds 2-3 (123 launches and 2.5 ms of kernel time per training step at B = 64).
template <int TM, int TN>
__device__ __forceinline__ bool cvae_split_combine(f32x4 (&acc)[TM][TN], float* part, unsigned* cnt, float* lds) {
    const int tid = threadIdx.x, nz = gridDim.z;
    const unsigned tile = blockIdx.y * gridDim.x + blockIdx.x, ntile = gridDim.x * gridDim.y;
    const cvae_buf pb = cvae_make_buf(part, (unsigned)((size_t)nz * ntile * TM * TN * 4096));
    const unsigned slab = TM * TN * 4096;                                   // bytes of one (slice, tile) slab
    const unsigned mine = ((unsigned)blockIdx.z * ntile + tile) * slab + tid * 16;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) cvae_buf_store_f4_sc1(pb, mine + (i * TN + j) * 4096, 0, acc[i][j]);
    cvae_drain_vmem();
    __syncthreads();                          // (also: every wave is done with the LDS stages)
    unsigned* tk = (unsigned*)lds;
    if (tid == 0) tk[0] = cvae_atomic_add_agent(cnt + tile, 1u);
    __syncthreads();
    if (tk[0] != (unsigned)(nz - 1)) return false;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int z = 0; z < nz; ++z) v += cvae_buf_load_f4_sc1(pb, ((unsigned)z * ntile + tile) * slab + (i * TN + j) * 4096 + tid * 16, 0);
            acc[i][j] = v;
        }
    if (tid == 0) cvae_atomic_store_agent(cnt + tile, 0u);
    return true;
}

// C[n1*ldc + n2] (+)= sum_m A[m*lda + n1] * Bm[m*ldb + seg(n2)]   (same contract as k_gemm_tn).
// Needs lda, ldb, seglen, segstride multiples of 4 and 16-byte aligned A, Bm (float4 tile loads); any M (rows beyond the slice's
// end are loaded as zeros).
template <int TM, int TN>
__global__ __launch_bounds__(256) void k_gemm_tn2(const float* __restrict__ A, long lda, const float* __restrict__ Bm,
                                                  long ldb, int seglen, long segstride, float* __restrict__ C, long ldc,
                                                  int M, int N1, int N2, int accumulate, int mchunk, float* part,
                                                  unsigned* cnt) {
    using G = GemmTileCfg<TM, TN>;
    float* sm = (float*)CVAE_SMEM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int a0 = blockIdx.y * G::BM, b0 = blockIdx.x * G::BN;
    // row slice of this block (split over the contraction: gridDim.z slices of mchunk rows, partial tiles go to `part`)
    const int mbeg = blockIdx.z * mchunk, mend = mbeg + mchunk < M ? mbeg + mchunk : M;
    long aoff[G::NA], boff[G::NB];
    int asm_[G::NA], bsm_[G::NB], arow[G::NA], brow[G::NB];
    bool aok[G::NA], bok[G::NB];
#pragma unroll
    for (int u = 0; u < G::NA; ++u) {
        const int e = tid + 256 * u, r = e / (G::BM / 4), c = 4 * (e % (G::BM / 4));
        aok[u] = e < 4 * G::BM && a0 + c < N1;
        aoff[u] = (long)r * lda + a0 + c;
        asm_[u] = r * G::LDA + c;
        arow[u] = r;
    }
#pragma unroll
    for (int u = 0; u < G::NB; ++u) {
        const int e = tid + 256 * u, r = e / (G::BN / 4), c = 4 * (e % (G::BN / 4)), n2 = b0 + c;
        bok[u] = e < 4 * G::BN && n2 < N2;
        boff[u] = (long)r * ldb + (long)(n2 / seglen) * segstride + (n2 % seglen);
        bsm_[u] = 16 * G::LDA + r * G::LDB + c;
        brow[u] = r;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ga[G::NA], gb[G::NB];
    auto gload = [&](int m0) {
#pragma unroll
        for (int u = 0; u < G::NA; ++u) ga[u] = aok[u] && m0 + arow[u] < mend ? *(const f32x4*)(A + (long)m0 * lda + aoff[u]) : zero4;
#pragma unroll
        for (int u = 0; u < G::NB; ++u) gb[u] = bok[u] && m0 + brow[u] < mend ? *(const f32x4*)(Bm + (long)m0 * ldb + boff[u]) : zero4;
    };
    auto sstore = [&](int stage) {
        float* st = sm + stage * G::STAGE;
#pragma unroll
        for (int u = 0; u < G::NA; ++u)
            if (tid + 256 * u < 4 * G::BM) *(f32x4*)(st + asm_[u]) = ga[u];
#pragma unroll
        for (int u = 0; u < G::NB; ++u)
            if (tid + 256 * u < 4 * G::BN) *(f32x4*)(st + bsm_[u]) = gb[u];
    };
    gload(mbeg);
    sstore(0);
    __syncthreads();
    for (int m0 = mbeg; m0 < mend; m0 += 16) {
        const int stage = ((m0 - mbeg) >> 4) & 1;
        const bool more = m0 + 16 < mend;
        if (more) gload(m0 + 16);
        cvae_gemm_tile_stage<TM, TN>(sm + stage * G::STAGE, sm + stage * G::STAGE + 16 * G::LDA, wm, wn, lr, kq, acc);
        if (more) sstore(stage ^ 1);
        __syncthreads();
    }
    if (part && !cvae_split_combine<TM, TN>(acc, part, cnt, sm)) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = b0 + wn * 16 * TN + 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = a0 + wm * 16 * TM + 16 * i + 4 * kq + r;
                if (rowi < N1 && col < N2) {
                    float* c = C + (long)rowi * ldc + col;
                    *c = acc[i][j][r] + (accumulate ? *c : 0.0f);
                }
            }
        }
}

// Optional epilogue of the time-major GEMMs: the inverted-dropout mask of conv_drop (gru_vae.py:355) or of its gradient, stored
// batch-major [B][T][N]: row r = f*Bp + b of the output is multiplied by mask[(b*T + f)*N + col]; batch padding rows become 0.
struct EpiMask {
    const float* mask;
    int B, Bp, T;
};
__device__ __forceinline__ float cvae_epi_mask(const EpiMask& em, int rowi, int col, int N, float v) {
    if (!em.mask) return v;
    const int b = rowi % em.Bp, f = rowi / em.Bp;
    return b < em.B ? v * em.mask[((long)b * em.T + f) * N + col] : 0.0f;
}

// part[rs][n] = sum over the rows m = rs*mchunk + rl, rl + 16, ... of A[m*lda + n]: first half of the column sums (bias
// gradients).  Block = 64 columns (16 float4 lanes) x 16 row lanes, LDS tree in fixed order.
// With `cnt`: the block that takes the last ticket of its 64-column group adds the row slices in slice order and writes
// out[n] (+)= sum (in-launch second stage, same protocol as cvae_split_combine; 4-byte sc1 stores / loads).
__global__ __launch_bounds__(256) void k_colsum_part(const float* __restrict__ A, long lda, float* __restrict__ part, int M,
                                                     int N, int mchunk, unsigned* cnt, float* out, int accumulate) {
    f32x4* red = (f32x4*)CVAE_SMEM;  // [16][17]
    const int cg = threadIdx.x & 15, rl = threadIdx.x >> 4, n = blockIdx.x * 64 + 4 * cg;
    const int mbeg = blockIdx.y * mchunk, mend = mbeg + mchunk < M ? mbeg + mchunk : M;
    f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (n < N) {
        int m = mbeg + rl;
        for (; m + 16 < mend; m += 32) {
            s0 += *(const f32x4*)(A + (long)m * lda + n);
            s1 += *(const f32x4*)(A + (long)(m + 16) * lda + n);
        }
        if (m < mend) s0 += *(const f32x4*)(A + (long)m * lda + n);
    }
    red[rl * 17 + cg] = s0 + s1;
    __syncthreads();
    f32x4 t = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (rl == 0 && n < N) {
        t = red[cg];
        for (int r = 1; r < 16; ++r) t += red[r * 17 + cg];
    }
    if (!cnt) {
        if (rl == 0 && n < N)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (n + q < N) part[(long)blockIdx.y * N + n + q] = t[q];
        return;
    }
    // slab [slice][column group][64]: written through, ticket, last arriver sums in slice order
    const int nz = gridDim.y, NP = gridDim.x * 64;
    const cvae_buf pb = cvae_make_buf(part, (unsigned)((size_t)nz * NP * 4));
    if (rl == 0) cvae_buf_store_f4_sc1(pb, ((unsigned)blockIdx.y * NP + blockIdx.x * 64 + 4 * cg) * 4, 0, t);
    cvae_drain_vmem();
    __syncthreads();
    unsigned* tk = (unsigned*)CVAE_SMEM;
    if (threadIdx.x == 0) tk[0] = cvae_atomic_add_agent(cnt + blockIdx.x, 1u);
    __syncthreads();
    if (tk[0] != (unsigned)(nz - 1)) return;
    if (rl == 0 && n < N) {
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int z = 0; z < nz; ++z) v += cvae_buf_load_f4_sc1(pb, ((unsigned)z * NP + blockIdx.x * 64 + 4 * cg) * 4, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (n + q < N) out[n + q] = v[q] + (accumulate ? out[n + q] : 0.0f);
    }
    if (threadIdx.x == 0) cvae_atomic_store_agent(cnt + blockIdx.x, 0u);
}

// C[m][n] (+)= sum_k A[m*lda + seg(k)] * Bm[n*ldb + k] + bias[n]   (same contract as k_gemm_nt_seg; lda, ldb, segstride
// multiples of 4, 16-byte aligned operands).  The [rows][16 k] global tiles are transposed on their way into LDS.
// (128 x 128 tiles: four waves per SIMD asked for -- 109 registers with the accumulators in VGPRs instead of 90 + 64 and three waves;
// the input-gate GEMM's 960 workgroups then fit the chip's 1,024 slots in one round: 173 -> 167 us stand-alone)
template <int TM, int TN>
__global__ __launch_bounds__(256, (TM * TN >= 16 ? 4 : 1)) void k_gemm_nt2(const float* __restrict__ A, long lda, int seglen, long segstride,
                                                  const float* __restrict__ Bm, long ldb, const float* __restrict__ bias,
                                                  float* __restrict__ C, long ldc, int M, int N, int K, int accumulate,
                                                  int kchunk, float* part, unsigned* cnt, EpiMask em) {
    using G = GemmTileCfg<TM, TN>;
    float* sm = (float*)CVAE_SMEM;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * G::BM, n0 = blockIdx.x * G::BN;
    // slice of the contraction of this block (gridDim.z slices of kchunk, a multiple of 16; partial tiles go to `part`, the last
    // block of the tile adds them in slice order and applies bias / accumulate / mask: cvae_split_combine)
    const int kbeg = blockIdx.z * kchunk, kend = kbeg + kchunk < K ? kbeg + kchunk : K;
    long aoff[G::NA], boff[G::NB];
    int asm_[G::NA], bsm_[G::NB];
    bool aok[G::NA], bok[G::NB];
#pragma unroll
    for (int u = 0; u < G::NA; ++u) {
        const int e = tid + 256 * u, r = e >> 2, kg = e & 3;
        aok[u] = e < 4 * G::BM && m0 + r < M;
        aoff[u] = (long)(m0 + r) * lda + 4 * kg;
        asm_[u] = 4 * kg * G::LDA + r;
    }
#pragma unroll
    for (int u = 0; u < G::NB; ++u) {
        const int e = tid + 256 * u, r = e >> 2, kg = e & 3;
        bok[u] = e < 4 * G::BN && n0 + r < N;
        boff[u] = (long)(n0 + r) * ldb + 4 * kg;
        bsm_[u] = 16 * G::LDA + 4 * kg * G::LDB + r;
    }
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ga[G::NA], gb[G::NB];
    auto gload = [&](int k0) {
        const long ao = (long)(k0 / seglen) * segstride + (k0 % seglen);
#pragma unroll
        for (int u = 0; u < G::NA; ++u) ga[u] = aok[u] ? *(const f32x4*)(A + aoff[u] + ao) : zero4;
#pragma unroll
        for (int u = 0; u < G::NB; ++u) gb[u] = bok[u] ? *(const f32x4*)(Bm + boff[u] + k0) : zero4;
    };
    auto sstore = [&](int stage) {
        float* st = sm + stage * G::STAGE;
#pragma unroll
        for (int u = 0; u < G::NA; ++u)
            if (tid + 256 * u < 4 * G::BM) {
#pragma unroll
                for (int q = 0; q < 4; ++q) st[asm_[u] + q * G::LDA] = ga[u][q];
            }
#pragma unroll
        for (int u = 0; u < G::NB; ++u)
            if (tid + 256 * u < 4 * G::BN) {
#pragma unroll
                for (int q = 0; q < 4; ++q) st[bsm_[u] + q * G::LDB] = gb[u][q];
            }
    };
    gload(kbeg);
    sstore(0);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        const int stage = ((k0 - kbeg) >> 4) & 1;
        const bool more = k0 + 16 < kend;
        if (more) gload(k0 + 16);
        cvae_gemm_tile_stage<TM, TN>(sm + stage * G::STAGE, sm + stage * G::STAGE + 16 * G::LDA, wm, wn, lr, kq, acc);
        if (more) sstore(stage ^ 1);
        __syncthreads();
    }
    if (part && !cvae_split_combine<TM, TN>(acc, part, cnt, sm)) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * 16 * TN + 16 * j + lr;
            const float bv = (bias && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rowi = m0 + wm * 16 * TM + 16 * i + 4 * kq + r;
                if (rowi < M && col < N) {
                    float* c = C + (long)rowi * ldc + col;
                    *c = cvae_epi_mask(em, rowi, col, N, acc[i][j][r] + bv + (accumulate ? *c : 0.0f));
                }
            }
        }
}

// Small-M GEMM for the per-step products of the backward recurrence: C[m][n] (+)= sum_k A[m*lda+k] * Bm[n*ldb+k].
// Block = 16 rows x 16*NTN columns, the 4 waves split K (multiple of 16) and reduce through LDS.
template <int NTN>
__global__ __launch_bounds__(256) void k_gemm_ks(const float* __restrict__ A, long lda, const float* __restrict__ Bm,
                                                 long ldb, float* __restrict__ C, long ldc, int M, int N, int K,
                                                 int accumulate) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int nchk = K >> 4, c_lo = (nchk * wave) >> 2, c_hi = (nchk * (wave + 1)) >> 2;
    const int m0 = blockIdx.y * 16, n0 = blockIdx.x * 16 * NTN;
    float* red = (float*)CVAE_SMEM;  // [4][16][16*NTN + 4]
    const int S = 16 * NTN + 4;
    f32x4 acc[NTN];
#pragma unroll
    for (int n = 0; n < NTN; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int ar = m0 + lr;
    ar = ar < M ? ar : M - 1;
    const float* ap = A + (long)ar * lda + 4 * kq;
    long bro[NTN];
#pragma unroll
    for (int n = 0; n < NTN; ++n) {
        int c = n0 + 16 * n + lr;
        c = c < N ? c : N - 1;
        bro[n] = (long)c * ldb + 4 * kq;
    }
    // chunks in groups of 4 with all loads of a group issued before its MFMAs (the chain of load -> MFMA round trips
    // otherwise dominates these small launches)
    int c = c_lo;
    for (; c + 4 <= c_hi; c += 4) {
        float4 a4[4], b4[4][NTN];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a4[u] = *(const float4*)(ap + 16 * (c + u));
#pragma unroll
            for (int n = 0; n < NTN; ++n) b4[u][n] = *(const float4*)(Bm + bro[n] + 16 * (c + u));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int n = 0; n < NTN; ++n) {
                acc[n] = cvae_mfma_16x16x4(a4[u].x, b4[u][n].x, acc[n]);
                acc[n] = cvae_mfma_16x16x4(a4[u].y, b4[u][n].y, acc[n]);
                acc[n] = cvae_mfma_16x16x4(a4[u].z, b4[u][n].z, acc[n]);
                acc[n] = cvae_mfma_16x16x4(a4[u].w, b4[u][n].w, acc[n]);
            }
    }
    for (; c < c_hi; ++c) {
        const float4 a4 = *(const float4*)(ap + 16 * c);
        float4 b4[NTN];
#pragma unroll
        for (int n = 0; n < NTN; ++n) b4[n] = *(const float4*)(Bm + bro[n] + 16 * c);
#pragma unroll
        for (int n = 0; n < NTN; ++n) {
            acc[n] = cvae_mfma_16x16x4(a4.x, b4[n].x, acc[n]);
            acc[n] = cvae_mfma_16x16x4(a4.y, b4[n].y, acc[n]);
            acc[n] = cvae_mfma_16x16x4(a4.z, b4[n].z, acc[n]);
            acc[n] = cvae_mfma_16x16x4(a4.w, b4[n].w, acc[n]);
        }
    }
#pragma unroll
    for (int n = 0; n < NTN; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(wave * 16 + kq * 4 + q) * S + n * 16 + lr] = acc[n][q];
    __syncthreads();
    for (int e = tid; e < 16 * 16 * NTN; e += 256) {
        const int col = e % (16 * NTN), r = e / (16 * NTN);
        if (m0 + r < M && n0 + col < N) {
            float* c = C + (long)(m0 + r) * ldc + n0 + col;
            const float v = red[(0 * 16 + r) * S + col] + red[(1 * 16 + r) * S + col] + red[(2 * 16 + r) * S + col] +
                            red[(3 * 16 + r) * S + col];
            *c = v + (accumulate ? *c : 0.0f);
        }
    }
}

// out[n] (+)= sum_m A[m*lda + n]   (bias gradients).  Block = 16 columns x 16 row lanes, fixed-order LDS tree: deterministic.
__global__ __launch_bounds__(256) void k_colsum(const float* A, long lda, float* out, int M, int N, int accumulate) {
    float* part = (float*)CVAE_SMEM;  // [16][17]
    const int c = threadIdx.x & 15, rl = threadIdx.x >> 4, n = blockIdx.x * 16 + c;
    float s = 0.0f;
    if (n < N)
        for (int m = rl; m < M; m += 16) s += A[(long)m * lda + n];
    part[rl * 17 + c] = s;
    __syncthreads();
    if (rl == 0 && n < N) {
        float t = 0.0f;
        for (int r = 0; r < 16; ++r) t += part[r * 17 + c];
        out[n] = t + (accumulate ? out[n] : 0.0f);
    }
}

// ------------------------------------------------------------------------------------------------------
// prepare-time kernels for the training image
// ------------------------------------------------------------------------------------------------------
// conv weights as GEMM operands over time-major segmented rows, forward and transposed (input-gradient) forms:
//   w0r[i][k*Cq + c]  = conv0.w[i][c][k]      w0t[c][k*C3p + i] = conv0.w[i][c][k]
//   w1r[o][j*C3p + i] = conv1.w[o][i][j]      w1t[i][j*C9p + o] = conv1.w[o][i][j]
// mode 0..3 selects which; `n` = total elements of the destination (padding entries are written as zero).
__global__ void k_prep_conv_train(const float* w, float* dst, int mode, int C, int ks, int Cq, int C3p, int C9p) {
    const int C3 = ks * C, C9 = ks * ks * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long n;
    if (mode == 0) n = (long)C3 * ks * Cq;
    else if (mode == 1) n = (long)Cq * ks * C3p;
    else if (mode == 2) n = (long)C9 * ks * C3p;
    else n = (long)C3 * ks * C9p;
    if (idx >= n) return;
    float v = 0.0f;
    if (mode == 0) {          // [C3][ks*Cq]
        const int c = (int)(idx % Cq), k = (int)((idx / Cq) % ks), i = (int)(idx / ((long)Cq * ks));
        if (c < C) v = w[((long)i * C + c) * ks + k];
    } else if (mode == 1) {   // [Cq][ks*C3p]
        const int i = (int)(idx % C3p), k = (int)((idx / C3p) % ks), c = (int)(idx / ((long)C3p * ks));
        if (c < C && i < C3) v = w[((long)i * C + c) * ks + k];
    } else if (mode == 2) {   // [C9][ks*C3p]
        const int i = (int)(idx % C3p), j = (int)((idx / C3p) % ks), o = (int)(idx / ((long)C3p * ks));
        if (i < C3) v = w[((long)o * C3 + i) * ks + j];
    } else {                  // [C3][ks*C9p]
        const int o = (int)(idx % C9p), j = (int)((idx / C9p) % ks), i = (int)(idx / ((long)C9p * ks));
        if (o < C9) v = w[((long)o * C3 + i) * ks + j];
    }
    dst[idx] = v;
}

// w0t of an encoder: conv0^T with the dense scale_in^T folded in (fp64 accumulation), so that the data-gradient GEMM of conv0 yields
// d loss / d x directly: dst[c][k*C3p + i] = sum_q scale_in.w[q][c] * conv0.w[i][q][k]   (k_scale_in_bwd then only gathers)
__global__ void k_prep_w0t_sin(const float* w, const float* sin_w, float* dst, int C, int ks, int Cq, int C3p) {
    const int C3 = ks * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)Cq * ks * C3p) return;
    const int i = (int)(idx % C3p), k = (int)((idx / C3p) % ks), c = (int)(idx / ((long)C3p * ks));
    double v = 0.0;
    if (c < C && i < C3)
        for (int q = 0; q < C; ++q) v += (double)sin_w[(long)q * C + c] * (double)w[((long)i * C + q) * ks + k];
    dst[idx] = (float)v;
}

// wbp[ct][c][lr][k] = (n = 16*ct + lr) < H ? whhT[n][16c + k] : (n - H < Co ? wyT[n - H][16c + k] : 0)
__global__ void k_prep_wbp(const float* whhT, const float* wyT, float* wbp, int H, int Co, int Cop) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x, H3 = 3L * H, nchk = H3 >> 4;
    if (idx < (long)(H + Cop) * H3) {
        const int k = (int)(idx & 15), lr = (int)((idx >> 4) & 15);
        const long c = (idx >> 8) % nchk, ct = (idx >> 8) / nchk;
        const long n = 16 * ct + lr, ka = 16 * c + k;
        wbp[idx] = n < H ? whhT[n * H3 + ka] : (n - H < Co ? wyT[(n - H) * H3 + ka] : 0.0f);
    }
}

// wix[n][c] = W_ih[n][c] (c < C9, zero padded to C9p);  cfold_t[n] = b_ih[n] + (n < 2H ? b_hh[n] : 0) + W_ih[n, C9:] . b_o
__global__ void k_prep_wix(const float* wih, const float* bih, const float* bhh, const float* bo, float* wix, float* cfold_t,
                           int C9, int C9p, int Co, int tot, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)3 * H * C9p) {
        const int c = (int)(idx % C9p), n = (int)(idx / C9p);
        wix[idx] = c < C9 ? wih[(long)n * tot + c] : 0.0f;
    } else if (idx < (long)3 * H * C9p + 3 * H) {
        const int n = (int)(idx - (long)3 * H * C9p);
        double s = (double)bih[n] + (n < 2 * H ? (double)bhh[n] : 0.0);
        for (int q = 0; q < Co; ++q) s += (double)wih[(long)n * tot + C9 + q] * (double)bo[q];
        cfold_t[n] = (float)s;
    }
}

// wrec_t[g][c2][col][kk], c2 < 2*nch: operand [h ; o] with o = mask*h (train mode keeps the feedback fold, on the masked
// state):  c2 < nch  -> a=0: W_hr, 1: W_hz, 2: 0, 3: W_hn       c2 >= nch -> a=0: F_r, 1: F_z, 2: F_n, 3: 0,  F = W_ih[:,C9:]*out_1.w
// F[n][k] = sum_q W_ih[n][C9 + q] * out_1.w[q][k]  (n over the 3H gate rows): the feedback fold, accumulated in fp64 and rounded
// once; the recurrent images of the forward and the reverse kernel are laid out from it
__global__ void k_prep_ffold(const float* wih, const float* wo, float* F, int C9, int Co, int tot, int H) {
    // thread = four consecutive k of one gate row n (H % 4 == 0): a feedback channel's weight is fetched and widened once for four
    // sums (scalar loads: the caller's out_1.w may sit at any 4-byte offset of a flat parameter buffer)
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)3 * H * (H >> 2)) {
        const int k = 4 * (int)(idx % (H >> 2)), n = (int)(idx / (H >> 2));
        const float* wrow = wih + (long)n * tot + C9;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll 4
        for (int q = 0; q < Co; ++q) {
            const float* w = wo + (long)q * H + k;
            const double a = (double)wrow[q];
            s0 += a * (double)w[0];
            s1 += a * (double)w[1];
            s2 += a * (double)w[2];
            s3 += a * (double)w[3];
        }
        *(f32x4*)(F + (long)n * H + k) = (f32x4){(float)s0, (float)s1, (float)s2, (float)s3};
    }
}

__global__ void k_prep_wrec_train(const float* F, const float* whh, float* wrec_t, int H) {
    const int nch = H >> 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)(H >> 2) * 2 * nch * 256) {
        const int kk = (int)(idx & 15), col = (int)((idx >> 4) & 15);
        const int c2 = (int)((idx >> 8) % (2 * nch)), g = (int)((idx >> 8) / (2 * nch));
        const int a = col >> 2, u = col & 3, j = 4 * g + u;
        float s = 0.0f;
        if (c2 < nch) {
            const int k = 16 * c2 + kk;
            if (a != 2) s = whh[(long)((a == 3 ? 2 : a) * H + j) * H + k];
        } else if (a < 3) {
            s = F[(long)(a * H + j) * H + 16 * (c2 - nch) + kk];
        }
        wrec_t[idx] = s;
    }
}

// ------------------------------------------------------------------------------------------------------
// forward (train mode)
// ------------------------------------------------------------------------------------------------------
// inverted-dropout mask: out[i] = (u >= p) / (1 - p), u from Philox keyed (seed, stream, GLOBAL element index): element i of a
// local [outer][B][inner] array (outer = 1 for the row-major conv mask, T for the time-major GRU mask) is element
// (o * Bg + row0 + b) * inner + k of the data-parallel job's array (cvae_set_draw_origin), so a row gets the same mask on
// whichever rank it lands.  Bg = B, row0 = 0 is the single-rank numbering.
// parts > 1 (cvae_set_draw_parts): the local batch is `parts` stacked copies of the job's rows (rec || cv of one decoder launch);
// copy c of row b is row c*Bg + row0 + b of a job-wide array of parts*Bg rows.
__global__ void k_mask_gen(float* out, long n, uint64_t seed, uint64_t stream, float p, long B, long inner, long Bg, long row0, long parts) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) {
        const long k = idx % inner, ob = idx / inner, bl = ob % B, o_ = ob / B, Bl = B / parts;
        const long b = (bl / Bl) * Bg + row0 + bl % Bl;
        const unsigned long long gi = (unsigned long long)((o_ * parts * Bg + b) * inner + k);
        uint32_t o[4];
        cvae_philox((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed,
                    (uint32_t)(seed >> 32), o);
        const float u = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        out[idx] = u >= p ? 1.0f / (1.0f - p) : 0.0f;
    }
}

// both masks of a pass (conv_drop [B][T*C9] and gru_drop [T][B][H]) in ONE launch: element idx < n1 belongs to the first
__global__ void k_mask_gen2(float* out1, long n1, long inner1, float* out2, long n2, long inner2, uint64_t seed, float p, long B,
                            long Bg, long row0, long parts) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n1 + n2) return;
    const bool second = idx >= n1;
    if (second) idx -= n1;
    const long inner = second ? inner2 : inner1;
    const uint64_t stream = second ? 2 : 1;
    const long k = idx % inner, ob = idx / inner, bl = ob % B, o_ = ob / B, Bl = B / parts;
    const long b = (bl / Bl) * Bg + row0 + bl % Bl;
    const unsigned long long gi = (unsigned long long)((o_ * parts * Bg + b) * inner + k);
    uint32_t o[4];
    cvae_philox((uint32_t)gi, (uint32_t)(gi >> 32), (uint32_t)stream, (uint32_t)(stream >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    const float u = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    (second ? out2 : out1)[idx] = u >= p ? 1.0f / (1.0f - p) : 0.0f;
}

struct TrainProParams {
    const float* x;      // [B][T][C]
    const float* y_in;   // [B][Co]
    const float* h_in;   // [B][H] or null
    const float* sin_w;  // [C][C] or null
    const float* sin_b;
    const float* bo;
    int B, Bp, T, C, Cq, pad, Co, Cop, H;
    long mtot;           // (T+1)*Bp
    float* xnp;          // time-major [(T+2*pad)*Bp][Cq], pre-zeroed
    float* hbuf;         // chunk-major slots; slot 0 written here
    float* obuf;
    float* hrow;         // [(T+1)*Bp][H]
    float* orow;
    float* ybuf;         // [(T+1)*Bp][Cop]; slot 0 = y_in
    float* dy;           // [B][Co] = y_in - out_1.b
    int nA, nH;          // block ranges: [0,nA) input rows, [nA,nA+nH) slot-0 state, rest y_in / dy
    int nY;              // blocks of the y_in role
    float* zero_ptr;     // words to clear (flags, counters), zero_n of them, by the blocks behind the other roles
    long zero_n;
};

__global__ void k_train_prologue(TrainProParams p) {
    const int tid = threadIdx.x, blk = blockIdx.x;
    if (blk < p.nA) {
        float* row = (float*)CVAE_SMEM;
        const int t = blk % p.T, b = blk / p.T;
        const float* xr = p.x + ((long)b * p.T + t) * p.C;
        for (int q = tid; q < p.C; q += 64) row[q] = xr[q];
        __syncthreads();
        float* dst = p.xnp + ((long)(t + p.pad) * p.Bp + b) * p.Cq;
        for (int q = tid; q < p.C; q += 64) {
            float v;
            if (p.sin_w) {
                v = p.sin_b[q];
                for (int r = 0; r < p.C; ++r) v += p.sin_w[(long)q * p.C + r] * row[r];
            } else {
                v = row[q];
            }
            dst[q] = v;
        }
    } else if (blk < p.nA + p.nH) {
        const long base = (long)(blk - p.nA) * 1024;
        for (int e = 0; e < 16; ++e) {
            const long idx = base + e * 64 + tid;
            if (idx < (long)p.Bp * p.H) {
                const int k = (int)(idx % p.H), r = (int)(idx / p.H);
                const float v = (p.h_in && r < p.B) ? p.h_in[(long)r * p.H + k] : 0.0f;
                p.hbuf[((long)(k >> 4) * p.mtot + r) * 16 + (k & 15)] = v;
                p.obuf[((long)(k >> 4) * p.mtot + r) * 16 + (k & 15)] = 0.0f;   // folded feedback sees o_{-1} = 0 (dy carries y_in)
                p.hrow[(long)r * p.H + k] = v;
                p.orow[(long)r * p.H + k] = 0.0f;
            }
        }
    } else if (blk < p.nA + p.nH + p.nY) {
        const int idx = (blk - p.nA - p.nH) * 64 + tid;
        if (idx < p.Bp * p.Cop) {
            const int q = idx % p.Cop, b = idx / p.Cop;
            const float v = (b < p.B && q < p.Co) ? p.y_in[(long)b * p.Co + q] : 0.0f;
            p.ybuf[idx] = v;
            if (b < p.B && q < p.Co) p.dy[(long)b * p.Co + q] = v - p.bo[q];
        }
    } else {      // the flags / status words of the recurrence and the arrival counters of the pass's split GEMMs (one launch less)
        const long idx = (long)(blk - p.nA - p.nH - p.nY) * 64 + tid;
        if (idx < p.zero_n) p.zero_ptr[idx] = 0.0f;
    }
}

// dst[(f*Bp + b)*ld + c] = src[(f*Bp + b)*ld + c] * mask[(b*T + f)*Cn + c]   (c < Cn, b < B; everything else 0)
// conv_drop in the forward (gru_vae.py:355) and its gradient in the backward.
__global__ void k_mul_mask_tm(float* dst, const float* src, const float* mask, int B, int Bp, int T, int Cn, int ld) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)T * Bp * ld) {
        const int c = (int)(idx % ld), b = (int)((idx / ld) % Bp), f = (int)(idx / ((long)ld * Bp));
        dst[idx] = (b < B && c < Cn) ? src[idx] * mask[((long)b * T + f) * Cn + c] : 0.0f;
    }
}

struct TrainStepParams {
    float* hbuf;
    float* obuf;
    long mtot;
    float* hrow;
    float* orow;
    const float* wrec_t;
    const float* gi;      // [T*Bp][3H] time-major
    const float* bhn;
    const float* gmask;   // [T][B][H]
    float* tape;          // [T*Bp][4H]: r, z, n, q = W_hn.h + b_hn
    const float* wyT;
    const float* dy;
    int Co, B, Bp, H, t;
};

// One train-mode GRU step (gru_vae.py:379-381): gates from [h_{t-1} ; o_{t-1}] (o = gru_drop(h), feedback folded on o),
// h_t carried un-dropped, o_t = mask_t * h_t published for the next step and the projection; gate values taped.
// NCT = 16-column tiles (4 hidden units x 4 gate columns each) per block.  Every block streams the WHOLE operand [Bp x 2H] through
// its waves (they split K); with one column tile per block and two blocks per CU that is 80 bytes per clock and CU of operand and
// weight loads against ~45-64 the load path delivers -- the step ran at half its matrix-pipe time (66.7 us at hu2048, B = 64,
// rocprofv3).  Two column tiles per block reuse every operand register for twice the MFMAs: 23 bytes per clock.
// NRB = 16-row tiles per trip over the weights (4, or 8 when the pass has that many: the stacked rec || cv pass then streams the
// weights once instead of twice).
// NW = waves per block (they split K).  8 (two waves per SIMD, twice the loads in flight) measured SLOWER at hu2048: 179.1 against
// 171.4 ms per step; 4 is what runs.
template <int NCT, int NRB = 4, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_gru_step_train(TrainStepParams p) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int g0 = NCT * (int)blockIdx.x, H = p.H, nch = H >> 4, nch2 = 2 * nch, t = p.t;
    const int c_lo = (nch2 * wave) / NW, c_hi = (nch2 * (wave + 1)) / NW;
    float* red = (float*)CVAE_SMEM;  // [NCT][NW][16 NRB][20]
    const float* wg = p.wrec_t + (long)g0 * nch2 * 256 + lr * 16 + kq * 4;
    const int nrt = p.Bp >> 4;
    const long slot = (long)t * p.Bp * 16;
    for (int rt0 = 0; rt0 < nrt; rt0 += NRB) {
        f32x4 acc[NCT][NRB];
#pragma unroll
        for (int n = 0; n < NCT; ++n)
#pragma unroll
            for (int i = 0; i < NRB; ++i) acc[n][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // several chunks per trip; the operand loads of trip k+1 are issued BEFORE the MFMAs of trip k (two register sets): at
        // H = 2048 the weights (134 MB per step, more than L2 holds) come from HBM / Infinity Cache, and with load -> wait -> MFMA
        // in sequence every trip paid that latency in full
        constexpr int NCH = NCT == 1 ? 4 : 2;              // chunks per trip (4 with two column tiles: no gain, 186.6 vs 184.5 ms per hu2048 step)
        float4 b4[2][NCH][NCT], a4[2][NCH][NRB];
        auto load_trip = [&](int c, int s) {
#pragma unroll
            for (int e = 0; e < NCH; ++e) {
                const int cc = c + e < c_hi ? c + e : c_lo;   // tail / past the end: a valid chunk, its MFMAs are skipped
#pragma unroll
                for (int n = 0; n < NCT; ++n) b4[s][e][n] = *(const float4*)(wg + ((long)n * nch2 + cc) * 256);
                const float* src = cc < nch ? p.hbuf + (long)cc * p.mtot * 16 : p.obuf + (long)(cc - nch) * p.mtot * 16;
                const float* hc = src + slot + lr * 16 + kq * 4;
#pragma unroll
                for (int i = 0; i < NRB; ++i) {
                    const int rt = rt0 + i < nrt ? rt0 + i : rt0;
                    a4[s][e][i] = *(const float4*)(hc + (long)rt * 256);
                }
            }
        };
        auto mfma_trip = [&](int c, int s) {
#pragma unroll
            for (int e = 0; e < NCH; ++e) {
                if (c + e < c_hi) {
#pragma unroll
                    for (int i = 0; i < NRB; ++i) {
                        if (rt0 + i < nrt) {
#pragma unroll
                            for (int n = 0; n < NCT; ++n) {
                                acc[n][i] = cvae_mfma_16x16x4(a4[s][e][i].x, b4[s][e][n].x, acc[n][i]);
                                acc[n][i] = cvae_mfma_16x16x4(a4[s][e][i].y, b4[s][e][n].y, acc[n][i]);
                                acc[n][i] = cvae_mfma_16x16x4(a4[s][e][i].z, b4[s][e][n].z, acc[n][i]);
                                acc[n][i] = cvae_mfma_16x16x4(a4[s][e][i].w, b4[s][e][n].w, acc[n][i]);
                            }
                        }
                    }
                }
            }
        };
        load_trip(c_lo, 0);
        for (int c = c_lo; c < c_hi; c += 2 * NCH) {
            load_trip(c + NCH, 1);
            mfma_trip(c, 0);
            load_trip(c + 2 * NCH, 0);
            mfma_trip(c + NCH, 1);
        }
#pragma unroll
        for (int n = 0; n < NCT; ++n)
#pragma unroll
            for (int i = 0; i < NRB; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[((n * NW + wave) * (16 * NRB) + i * 16 + kq * 4 + r) * 20 + lr] = acc[n][i][r];
        __syncthreads();
#pragma unroll
        for (int nn = 0; nn < NCT * (NRB / 4); ++nn) {
            if (tid >= 256) break;      // (NW = 8: the second half of the block only multiplies)
            const int n = nn % NCT, row = (nn / NCT) * 64 + (tid >> 2), u = tid & 3, g = g0 + n, j = 4 * g + u;
            const long hcol = (long)(g >> 2) * p.mtot * 16 + (g & 3) * 4 + u;
            const int grow = rt0 * 16 + row;
            if (grow < p.Bp) {
                float rg = 0.f, zg = 0.f, ng = 0.f, q = 0.f, hn = 0.f, on = 0.f;
                if (grow < p.B) {
                    float s[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                    {
                        s[a] = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) s[a] += red[((n * NW + w) * (16 * NRB) + row) * 20 + a * 4 + u];
                    }
                    const float* gip = p.gi + ((long)t * p.Bp + grow) * 3 * H;
                    float g0_ = gip[j], g1 = gip[H + j], g2 = gip[2 * H + j];
                    if (t == 0) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, g0_, g1, g2);
                    rg = cvae_sigmoid(g0_ + s[0]);
                    zg = cvae_sigmoid(g1 + s[1]);
                    q = s[3] + p.bhn[j];
                    ng = tanhf(g2 + s[2] + rg * q);
                    const float hold = p.hbuf[hcol + slot + (long)grow * 16];
                    hn = ng + zg * (hold - ng);
                    on = hn * p.gmask[((long)t * p.B + grow) * H + j];
                }
                const long nxt = slot + (long)p.Bp * 16 + (long)grow * 16;
                p.hbuf[hcol + nxt] = hn;
                p.obuf[hcol + nxt] = on;
                p.hrow[((long)(t + 1) * p.Bp + grow) * H + j] = hn;
                p.orow[((long)(t + 1) * p.Bp + grow) * H + j] = on;
                float* tp = p.tape + ((long)t * p.Bp + grow) * 4 * H + j;
                tp[0] = rg; tp[H] = zg; tp[2 * H] = ng; tp[3 * H] = q;
            }
        }
        __syncthreads();
    }
}

struct TrainEpiParams {
    const float* ybuf;   // [(T+1)*Bp][Cop], slot t+1 = raw y_t
    const float* hrow;
    const float* sout_w; // [Co][Co] or null
    const float* sout_b;
    int clamp_from, B, Bp, T, Co, Cop, H;
    float clamp_min;
    float* trj_out;      // [B][T][Co]
    float* y_last;       // [B][Co] or null
    float* h_last;       // [B][H] or null
};

__global__ void k_train_epilogue(TrainEpiParams p) {
    float* row = (float*)CVAE_SMEM;
    const int t = blockIdx.x % p.T, b = blockIdx.x / p.T;
    const float* yr = p.ybuf + ((long)(t + 1) * p.Bp + b) * p.Cop;
    for (int c = threadIdx.x; c < p.Co; c += blockDim.x) row[c] = yr[c];
    __syncthreads();
    for (int c = threadIdx.x; c < p.Co; c += blockDim.x) {
        float v;
        if (p.sout_w) {
            v = p.sout_b[c];
            for (int q = 0; q < p.Co; ++q) v += p.sout_w[(long)c * p.Co + q] * row[q];
        } else {
            v = row[c];
            if (p.clamp_from >= 0 && c >= p.clamp_from) v = fmaxf(v, p.clamp_min);
        }
        p.trj_out[((long)b * p.T + t) * p.Co + c] = v;
        if (p.y_last && t == p.T - 1) p.y_last[(long)b * p.Co + c] = row[c];
    }
    if (p.h_last && t == p.T - 1)
        for (int k = threadIdx.x; k < p.H; k += blockDim.x) p.h_last[(long)b * p.H + k] = p.hrow[((long)p.T * p.Bp + b) * p.H + k];
}

// ------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------
// dYl[(t*Bp+b)][c] = d loss / d raw y_t through scale_out^T (dense) or the clamp's pass-through mask; zero in padding
__global__ void k_bwd_dy(const float* dout, const float* ybuf, const float* sout_w, int clamp_from, float clamp_min, float* dyl, int B, int Bp,
                         int T, int Co, int Cop) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)T * Bp * Cop) {
        const int c = (int)(idx % Cop), b = (int)((idx / Cop) % Bp), t = (int)(idx / ((long)Cop * Bp));
        float v = 0.0f;
        if (b < B && c < Co) {
            const float* d = dout + ((long)b * T + t) * Co;
            if (sout_w) {
                for (int q = 0; q < Co; ++q) v += sout_w[(long)q * Co + c] * d[q];
            } else {
                v = d[c];
                if (clamp_from >= 0 && c >= clamp_from && ybuf[((long)(t + 1) * Bp + b) * Cop + c] < clamp_min) v = 0.0f;
            }
        }
        dyl[idx] = v;
    }
}

// The scale_out^T form of the same (decoder passes) with the Co x Co matrix and 16 rows of dout staged in LDS: the plain kernel
// re-reads the matrix from memory for every output (70-78 us per 64-row pass on MI355X, on the backward's critical path in front of
// every decoder recurrence; this form: a few us).  Same summation order (q ascending), same bits.
#define CVAE_BWD_DY_ROWS 16
__global__ __launch_bounds__(256) void k_bwd_dy_sout(const float* dout, const float* sout_w, float* dyl, int B, int Bp, int T, int Co,
                                                     int Cop) {
    float* S = (float*)CVAE_SMEM;                 // [Co][Co]
    float* D = S + Co * Co;                       // [CVAE_BWD_DY_ROWS][Co]
    const long row0 = (long)blockIdx.x * CVAE_BWD_DY_ROWS, M = (long)T * Bp;
    for (int i = threadIdx.x; i < Co * Co; i += blockDim.x) S[i] = sout_w[i];
    for (int i = threadIdx.x; i < CVAE_BWD_DY_ROWS * Co; i += blockDim.x) {
        const long r = row0 + i / Co;
        const int q = i % Co, b = (int)(r % Bp), t = (int)(r / Bp);
        D[i] = (r < M && b < B) ? dout[((long)b * T + t) * Co + q] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < CVAE_BWD_DY_ROWS * Cop; i += blockDim.x) {
        const int rr = i / Cop, c = i % Cop;
        const long r = row0 + rr;
        if (r >= M) continue;
        float v = 0.0f;
        if ((int)(r % Bp) < B && c < Co) {
            const float* d = D + rr * Co;
            for (int q = 0; q < Co; ++q) v += S[q * Co + c] * d[q];
        }
        dyl[r * Cop + c] = v;
    }
}

struct BwdStepParams {
    const float* dyl;    // [T*Bp][Cop]
    float* dytot;        // [T*Bp][Cop]
    const float* dyfb;   // [Bp][Cop]: d loss / d y_t arriving through step t+1's input (zeros at t = T-1)
    const float* part;   // null, or [nparts][Bp][H + Cop]: K-split partial sums of step t+1's products (k_bwd_step_gemm):
    int nparts;          //   columns [0, H) add to dh, columns [H, H + Co) replace dyfb
    const float* wo;     // out_1.w [Cop][H]
    float* dh;           // [Bp][H] in: d loss / d h_t from step t+1; out: the z-path part of d loss / d h_{t-1}
    const float* tape;
    const float* hrow;
    const float* gmask;
    float* dgi;          // [T*Bp][3H] grads of the input-side pre-activations (r, z, n)
    float* dgh;          // [T*Bp][3H] grads of the hidden-side pre-activations (r, z, r*... n uses dn_pre*r)
    float* dgic;         // this step's rows once more, chunk-major [3H/16][Bp][16]: what k_bwd_step_gemm reads (see there)
    float* dghc;
    int B, Bp, H, Co, Cop, t;
};

// One step of the reverse recurrence: projection + dropout + GRU cell backward for every (b, j); the two matrix products
// that carry gradients to step t-1 (W_hh^T dgh, W_ih[:,C9:]^T dgi) follow as k_gemm_ks launches.
__global__ __launch_bounds__(256) void k_gru_step_bwd(BwdStepParams p) {
    float* dyt = (float*)CVAE_SMEM;  // [Cop]
    const int tid = threadIdx.x, b = blockIdx.y, j = blockIdx.x * 256 + tid, H = p.H, t = p.t;
    const long rowi = (long)t * p.Bp + b;
    if (tid < p.Cop) {
        float v = 0.0f;
        if (b < p.B && tid < p.Co) {
            v = p.dyl[rowi * p.Cop + tid];
            if (p.part) {
                const long ldp = H + p.Cop;
#pragma unroll 8
                for (int s = 0; s < p.nparts; ++s) v += p.part[((long)s * p.Bp + b) * ldp + H + tid];
            } else {
                v += p.dyfb[(long)b * p.Cop + tid];
            }
        }
        dyt[tid] = v;
        if (blockIdx.x == 0) p.dytot[rowi * p.Cop + tid] = v;
    }
    // what the cell needs besides d loss / d y_t does not depend on it: requested BEFORE the barrier, so that these loads and the
    // partial-sum loads above are in flight together (they used to follow the barrier: two memory latencies in a row)
    const bool livej = j < H && b < p.B;
    float dhin = 0.f, gm = 0.f, r = 0.f, z = 0.f, n = 0.f, q = 0.f, hp = 0.f;
    if (livej) {
        dhin = p.dh[(long)b * H + j];
        if (p.part) {
            const long ldp = H + p.Cop;
#pragma unroll 8
            for (int s = 0; s < p.nparts; ++s) dhin += p.part[((long)s * p.Bp + b) * ldp + j];
        }
        gm = p.gmask[((long)t * p.B + b) * H + j];
        const float* tp = p.tape + rowi * 4 * H + j;
        r = tp[0]; z = tp[H]; n = tp[2 * H]; q = tp[3 * H];
        hp = p.hrow[rowi * H + j];   // slot t = h_{t-1}
    }
    __syncthreads();
    if (j < H) {
        float drp = 0.f, dzp = 0.f, dnp = 0.f, dq = 0.f, dhz = 0.f;
        if (b < p.B) {
            float dov = 0.0f;
#pragma unroll 10
            for (int c = 0; c < p.Co; ++c) dov += dyt[c] * p.wo[(long)c * H + j];
            const float dht = dhin + gm * dov;
            const float dn = dht * (1.0f - z), dz = dht * (hp - n);
            dnp = dn * (1.0f - n * n);
            dq = dnp * r;
            drp = dnp * q * r * (1.0f - r);
            dzp = dz * z * (1.0f - z);
            dhz = dht * z;
        }
        p.dh[(long)b * H + j] = dhz;
        float* gi = p.dgi + rowi * 3 * H + j;
        float* gh = p.dgh + rowi * 3 * H + j;
        gi[0] = drp; gi[H] = dzp; gi[2 * H] = dnp;
        gh[0] = drp; gh[H] = dzp; gh[2 * H] = dq;
        if (p.dgic) {
            const long nc = H >> 4, at = ((long)(j >> 4) * p.Bp + b) * 16 + (j & 15), gs = nc * p.Bp * 16;
            p.dgic[at] = drp; p.dgic[gs + at] = dzp; p.dgic[2 * gs + at] = dnp;
            p.dghc[at] = drp; p.dghc[gs + at] = dzp; p.dghc[2 * gs + at] = dq;
        }
    }
}

// The two products that carry gradients from step t to step t-1, as ONE launch that fills the chip:
//   part[ks][b][j]     = sum_{k in slice ks} dgh_t[b][k] * whhT[j][k]     (j < H:  W_hh^T dgh_t)
//   part[ks][b][H + c] = sum_{k in slice ks} dgi_t[b][k] * wyT[c][k]      (c < Co: W_ih[:, C9:]^T dgi_t)
// grid = ((H + Cop)/16 column blocks, KS slices of K = 3H, row-tile groups); the 4 waves split a slice once more and meet in
// LDS; a block keeps its weight fragments for up to NRT row tiles (weights are read once per 16*NRT batch rows).
// k_gru_step_bwd of step t-1 adds the KS partial sums (fixed order: deterministic).
struct BwdGemmParams {
    const float* dgh;    // rows of step t, CHUNK-MAJOR [3H/16][Bp][16] (k_gru_step_bwd's second copy): a 16-row operand tile of one
    const float* dgi;    //   chunk is one contiguous KiB.  Row-major rows (stride 3H floats = 24 KB at hu2048) put the 16 rows of every
                         //   load instruction on ONE L2 channel
    const float* wbp;    // [whhT ; wyT ; 0] packed as MFMA fragments: [(H + Cop)/16 col tiles][3H/16 chunks][16 cols][16 k]
    float* part;         // [KS][Bp][H + Cop]
    int Bp, H, Co, Cop;
};

// NTN column tiles (16 columns each) per block: every block reads its K slice of the gate-gradient rows once for ALL of them.  With
// one tile per block (hu1024 shapes: 68 tiles, plenty of blocks) the rows are re-read by every column block -- at hu2048 (132 tiles)
// that was 207 MB of L2 -> CU traffic per step next to 52 MB of weights and 43 us per launch; NTN = 4 quarters it.
template <int NRT, int NTN>
__global__ __launch_bounds__(256) void k_bwd_step_gemm(BwdGemmParams p) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, H3 = 3 * H, nchk = H3 >> 4, KS = gridDim.y, ks = blockIdx.y;
    const int sl_lo = (nchk * ks) / KS, sl_hi = (nchk * (ks + 1)) / KS, nsl = sl_hi - sl_lo;
    const int c_lo = sl_lo + (nsl * wave) / 4, c_hi = sl_lo + (nsl * (wave + 1)) / 4;
    const int ntile = (H + p.Cop) >> 4, nrt = p.Bp >> 4, rt0 = blockIdx.z * NRT;
    // a block's tiles are all on the same side of the H boundary (H / 16 is a multiple of NTN for the sizes it is built for)
    const int tile0 = blockIdx.x * NTN;
    const bool hid = tile0 * 16 < H;
    const float* A = hid ? p.dgh : p.dgi;
    const float* bp[NTN];
#pragma unroll
    for (int n = 0; n < NTN; ++n) {
        const int tl = tile0 + n < ntile ? tile0 + n : ntile - 1;
        bp[n] = p.wbp + (long)tl * nchk * 256 + lr * 16 + 4 * kq;   // chunk c: + 256*c (one coalesced 1 KiB read)
    }
    const float* ap[NRT];
#pragma unroll
    for (int r = 0; r < NRT; ++r) {
        const int rt = rt0 + r < nrt ? rt0 + r : nrt - 1;
        ap[r] = A + (long)(rt * 16 + lr) * 16 + 4 * kq;     // chunk c: + c * Bp * 16
    }
    f32x4 acc[NTN][NRT];
#pragma unroll
    for (int n = 0; n < NTN; ++n)
#pragma unroll
        for (int r = 0; r < NRT; ++r) acc[n][r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int c = c_lo;
    constexpr int RND = NTN > 1 ? 3 : (NRT == 4 ? 6 : 8);   // chunks per round: every load of a round is in flight before its first MFMA
    for (; c + RND <= c_hi; c += RND) {     // (hu1024, 8 slices: the wave's whole share is one round of 6)
        f32x4 b4[RND][NTN], a4[RND][NRT];
#pragma unroll
        for (int u = 0; u < RND; ++u) {
#pragma unroll
            for (int n = 0; n < NTN; ++n) b4[u][n] = *(const f32x4*)(bp[n] + 256 * (c + u));
#pragma unroll
            for (int r = 0; r < NRT; ++r) a4[u][r] = *(const f32x4*)(ap[r] + (long)(c + u) * p.Bp * 16);
        }
#pragma unroll
        for (int u = 0; u < RND; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int n = 0; n < NTN; ++n)
#pragma unroll
                    for (int r = 0; r < NRT; ++r) acc[n][r] = cvae_mfma_16x16x4(a4[u][r][q], b4[u][n][q], acc[n][r]);
    }
    for (; c < c_hi; ++c) {
        f32x4 b4[NTN];
#pragma unroll
        for (int n = 0; n < NTN; ++n) b4[n] = *(const f32x4*)(bp[n] + 256 * c);
#pragma unroll
        for (int r = 0; r < NRT; ++r) {
            const f32x4 a4 = *(const f32x4*)(ap[r] + (long)c * p.Bp * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int n = 0; n < NTN; ++n) acc[n][r] = cvae_mfma_16x16x4(a4[q], b4[n][q], acc[n][r]);
        }
    }
    float* red = (float*)CVAE_SMEM;   // [4 waves][NTN][NRT][16 rows][20]
#pragma unroll
    for (int n = 0; n < NTN; ++n)
#pragma unroll
        for (int r = 0; r < NRT; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[(((wave * NTN + n) * NRT + r) * 16 + kq * 4 + q) * 20 + lr] = acc[n][r][q];
    __syncthreads();
    const long ldp = H + p.Cop;
    for (int e = tid; e < NTN * NRT * 256; e += 256) {
        const int n = e / (NRT * 256), r = (e >> 8) % NRT, row = (e >> 4) & 15, col = e & 15;
        if (rt0 + r < nrt && tile0 + n < ntile) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += red[(((w * NTN + n) * NRT + r) * 16 + row) * 20 + col];
            p.part[((long)ks * p.Bp + (rt0 + r) * 16 + row) * ldp + (tile0 + n) * 16 + col] = v;
        }
    }
}

// dx[b][t][c'] = sum_c scale_in.w[c][c'] * dxn[(t+pad)*Bp + b][c]  (or a plain copy without scale_in)
__global__ void k_scale_in_bwd(const float* dxn, const float* sin_w, float* dx, int B, int Bp, int T, int C, int Cq, int pad) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)B * T * C) {
        const int c = (int)(idx % C), t = (int)((idx / C) % T), b = (int)(idx / ((long)C * T));
        const float* d = dxn + ((long)(t + pad) * Bp + b) * Cq;
        float v = 0.0f;
        if (sin_w)
            for (int q = 0; q < C; ++q) v += sin_w[(long)q * C + c] * d[q];
        else
            v = d[c];
        dx[idx] = v;
    }
}

// conv weight gradient from its GEMM form: dw[(o*I + i)*ks + j] (+)= tmp[o*ldt + j*seglen + i]
__global__ void k_unpack_dw(const float* tmp, long ldt, int seglen, float* dw, int O, int I, int ks, int accumulate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)O * I * ks) {
        const int j = (int)(idx % ks), i = (int)((idx / ks) % I), o = (int)(idx / ((long)ks * I));
        const float v = tmp[(long)o * ldt + (long)j * seglen + i];
        dw[idx] = v + (accumulate ? dw[idx] : 0.0f);
    }
}

// torch.optim.Adam (no weight decay, no amsgrad), step counted from 1 (train_gru_cyclevae_gauss_batch.py:377,1420)
// gate: null, or a word the device can read (the pinned status sink): non-zero = a kernel of this step reported a failed hand-off
// or a range overflow, the gradients are invalid and the update is SKIPPED (parameters and moments stay as they were)
__global__ void k_adam(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                       float bc1, float bc2_sqrt, const int* gate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gate && *(const volatile int*)gate != 0) return;
    if (idx < n) {
        const float gi = g[idx];
        const float mi = b1 * m[idx] + (1.0f - b1) * gi;
        const float vi = b2 * v[idx] + (1.0f - b2) * gi * gi;
        m[idx] = mi;
        v[idx] = vi;
        p[idx] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// cvae_adam_step_counted: the step counter and the bias corrections live on the device.  state[0] = updates applied so far,
// state[1], state[2] = bit patterns of bc1 = 1 - beta1^step and sqrt(1 - beta2^step) of the update being applied.
__global__ void k_adam_tick(int* state, float b1, float b2, const int* gate) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (gate && *(const volatile int*)gate != 0) return;
        const int step = state[0] + 1;
        state[0] = step;
        ((float*)state)[1] = 1.0f - powf(b1, (float)step);
        ((float*)state)[2] = sqrtf(1.0f - powf(b2, (float)step));
    }
}

__global__ void k_adam_counted(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                               const int* state, const int* gate) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gate && *(const volatile int*)gate != 0) return;
    if (idx < n) {
        const float bc1 = ((const float*)state)[1], bc2_sqrt = ((const float*)state)[2];
        const float gi = g[idx];
        const float mi = b1 * m[idx] + (1.0f - b1) * gi;
        const float vi = b2 * v[idx] + (1.0f - b2) * gi * gi;
        m[idx] = mi;
        v[idx] = vi;
        p[idx] -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}

// cvae_status_latch: latch[0] = max(latch[0], sink[0]); sink[0] = 0 -- the step's status word moves from the pinned host sink the
// kernels report into to a DEVICE word in stream order, so that the host never has to clear the sink while steps are in flight
__global__ void k_status_latch(int* latch, volatile int* sink) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const int s = sink[0], l = latch[0];
        if (s != 0) {
            latch[0] = s > l ? s : l;
            sink[0] = 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// persistent recurrences for training (H = 64*CPWx/..., see the per-kernel notes)
// ------------------------------------------------------------------------------------------------------
// wrec_t8[jg][tile][c2][col][kk]: k_prep_wrec_train's matrix regrouped for blocks of 8 hidden units: block jg owns units
// 8jg..8jg+7 = two 16-column MFMA tiles (tile n: units 8jg+4n..+3, col = a*4 + u), c2 < 2*nch over the operand [h ; o].
__global__ void k_prep_wrec_t8(const float* wrec_t, float* wrec_t8, int H) {
    const int nch2 = 2 * (H >> 4);
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)(H >> 3) * 2 * nch2 * 256) {
        const int within = (int)(idx & 255), c2 = (int)((idx >> 8) % nch2), tile = (int)(((idx >> 8) / nch2) & 1);
        const int jg = (int)((idx >> 8) / nch2 / 2);
        wrec_t8[idx] = wrec_t[((long)(2 * jg + tile) * nch2 + c2) * 256 + within];   // wrec_t group g = 4 units = 2*jg + tile
    }
}

struct TrainFwdParams {
    float* hbuf;          // k_train_fwd_steps: [H/8][(T+1)*Bp][8] (slots >= 1 only; slot 0 is read from hrow / orow)
    float* obuf;
    long mtot;
    float* hrow;          // [(T+1)*Bp][H]
    float* orow;
    const float* wrec_t8;
    const float* gi;      // [T*Bp][3H] time-major
    const float* bhn;
    const float* gmask;   // [T][B][H]
    float* tape;          // [T*Bp][4H]
    const float* wyT;
    const float* dy;
    int Co, B, Bp, H, T, rts;
    unsigned* flags;      // [Bp/16][H/8], zeroed before launch: flags[i][jg] = t  <=>  this block's 8 units of h_t, o_t are published
    int* status;
    long long* prof;      // null, or 8 cycle sums of block 0: poll, loads+MFMA, reduce+gates, publish
    int backoff;          // k_train_fwd_steps_h, one row tile per block: s_sleep units (64 cycles) before the first poll of a step
};

// All T train-mode GRU steps in one cooperative launch.  Block (jg, i): hidden units 8jg..8jg+7 (32 MFMA columns, weights
// for the operand [h ; o] = 2 tiles x CPW2 chunks x float4 per lane, register-resident) x row tiles i, i+rts, ...
// Waves 0,1 take the h half of K, waves 2,3 the o half.  Same hand-off as the eval kernels: write-through stores, sc1 loads,
// one flag per (row tile, block), per-wave polling; no grid barrier.  The exchanged state lives in 8-unit planes
// ([H/8][rows][8]) so that a block's 16 rows x 8 units are ONE contiguous 512-byte piece: with 16-unit planes two blocks
// wrote the two 32-byte halves of every 64-byte row, and those partial-line write-through stores made the publish 5x slower.
template <int CPW2>
__global__ __launch_bounds__(256, 1) void k_train_fwd_steps(TrainFwdParams p) {
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, nch = H >> 4, ng = H >> 3, nrt = p.Bp >> 4;
    const int jg = blockIdx.x % ng, ti = blockIdx.x / ng, rts = p.rts;
    const int c_lo = wave * CPW2;                       // over [0, 2*nch): h chunks then o chunks
    const bool from_o = c_lo >= nch;
    const int cs = from_o ? c_lo - nch : c_lo;          // first source chunk inside hbuf / obuf
    float* red = (float*)CVAE_SMEM;                     // [4 waves][16 rows][36]
    float* hsh = red + 4 * 16 * 36;                     // [2][16 rows][8]: h, o of this block's units
    const unsigned mtot = (unsigned)p.mtot;
    const unsigned bytes = (unsigned)((long)nch * p.mtot * 64);
    const cvae_buf hb = cvae_make_buf(p.hbuf, bytes), ob = cvae_make_buf(p.obuf, bytes);
    const cvae_buf src = from_o ? ob : hb;
    // fragment of 16-wide chunk c, lane (lr, kq): units 16c + 4kq .. +3 = plane 2c + (kq >> 1), floats (kq & 1)*4 .. +3 of row lr
    const unsigned voff = ((unsigned)(kq >> 1) * mtot + (unsigned)lr) * 32u + (unsigned)(kq & 1) * 16u;
    const float* src0 = from_o ? p.orow : p.hrow;       // slot 0 (written row-major by the prologue)
    f32x4 w[2][CPW2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ci = 0; ci < CPW2; ++ci)
            w[n][ci] = *(const f32x4*)(p.wrec_t8 + (((long)jg * 2 + n) * (2 * nch) + c_lo + ci) * 256 + lr * 16 + kq * 4);
    // gate threads are the UPPER half of the block (waves 2,3): their bulk stores (tape, row-major copies) then never sit in
    // front of wave 0's vmcnt(0) drain, which precedes the flag store
    const int gt = tid - 128, row = (gt >> 3) & 15, u8 = gt & 7, j = 8 * jg + u8;
    const bool gate = tid >= 128;
    const float bhn = p.bhn[j];
    long long pc[4] = {0, 0, 0, 0};
    const bool prof = p.prof && blockIdx.x == 0;
    for (int t = 0; t < p.T; ++t) {
        for (int i = ti; i < nrt; i += rts) {
            long long c0 = prof ? cvae_clock() : 0;
            if (t > 0) {   // chunks [cs, cs+CPW2) of slot t: chunk c is published by blocks 2c and 2c+1
                unsigned spins = 0;
                for (;;) {
                    unsigned f = (unsigned)t;
                    if (lane < 2 * CPW2) f = cvae_atomic_load_agent(p.flags + (long)i * ng + 2 * cs + lane);
                    if (cvae_wave_all(f >= (unsigned)t)) break;
                    cvae_sleep();
                    if (++spins > (1u << 22)) {
                        p.status[0] = 3;
                        break;
                    }
                }
            }
            cvae_compiler_fence();
            if (prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
            const unsigned row0 = (unsigned)(t * p.Bp + i * 16);
            f32x4 a4[CPW2];
#pragma unroll
            for (int ci = 0; ci < CPW2; ++ci)
                a4[ci] = t == 0 ? *(const f32x4*)(src0 + (long)(row0 + lr) * H + 16 * (cs + ci) + 4 * kq)
                                : cvae_buf_load_f4_sc1(src, voff, ((unsigned)(2 * (cs + ci)) * mtot + row0) * 32u);
            const int grow = i * 16 + row;
            const bool live = gate && grow < p.B;
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, hold = 0.f, msk = 0.f;
            if (live) {
                const float* gip = p.gi + ((long)t * p.Bp + grow) * 3 * H;
                g0 = gip[j]; g1 = gip[H + j]; g2 = gip[2 * H + j];
                if (t == 0) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, g0, g1, g2);
                hold = t == 0 ? p.hrow[(long)(row0 + row) * H + j]
                              : cvae_buf_load_f1_sc1(hb, (unsigned)(u8 * 4), ((unsigned)jg * mtot + row0 + (unsigned)row) * 32u);
                msk = p.gmask[((long)t * p.B + grow) * H + j];
            }
            f32x4 acc[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ci = 0; ci < CPW2; ++ci)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0] = cvae_mfma_16x16x4(a4[ci][q], w[0][ci][q], acc[0]);
                    acc[1] = cvae_mfma_16x16x4(a4[ci][q], w[1][ci][q], acc[1]);
                }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[(wave * 16 + kq * 4 + q) * 36 + n * 16 + lr] = acc[n][q];
            if (prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
            __syncthreads();
            if (gate) {
                float rg = 0.f, zg = 0.f, ng_ = 0.f, qq = 0.f, hn = 0.f, on = 0.f;
                if (live) {
                    const int col = (u8 >> 2) * 16 + (u8 & 3);   // tile n = u8>>2, unit inside the tile u8&3; + a*4
                    float s[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        s[a] = red[(0 * 16 + row) * 36 + col + a * 4] + red[(1 * 16 + row) * 36 + col + a * 4] +
                               red[(2 * 16 + row) * 36 + col + a * 4] + red[(3 * 16 + row) * 36 + col + a * 4];
                    rg = cvae_sigmoid(g0 + s[0]);
                    zg = cvae_sigmoid(g1 + s[1]);
                    qq = s[3] + bhn;
                    ng_ = tanhf(g2 + s[2] + rg * qq);
                    hn = ng_ + zg * (hold - ng_);
                    on = hn * msk;
                }
                hsh[row * 8 + u8] = hn;
                hsh[128 + row * 8 + u8] = on;
                if (grow < p.Bp) {
                    p.hrow[((long)(t + 1) * p.Bp + grow) * H + j] = hn;
                    p.orow[((long)(t + 1) * p.Bp + grow) * H + j] = on;
                    float* tp = p.tape + ((long)t * p.Bp + grow) * 4 * H + j;
                    tp[0] = rg; tp[H] = zg; tp[2 * H] = ng_; tp[3 * H] = qq;
                }
            }
            __syncthreads();
            if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
            if (tid < 64) {   // wave 0: lanes 0..31 publish h (16 rows x 32 B = one 512-byte piece), lanes 32..63 publish o
                const int which = tid >> 5, l = tid & 31;
                const f32x4 v = *(const f32x4*)(hsh + which * 128 + l * 4);
                const unsigned so = ((unsigned)jg * mtot + row0 + (unsigned)p.Bp) * 32u;
                cvae_buf_store_f4_sc1(which ? ob : hb, (unsigned)l * 16u, so, v);
                cvae_drain_vmem();
                cvae_wave_barrier();
                if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * ng + jg, (unsigned)(t + 1));
            }
            if (prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
        }
    }
    if (prof && tid == 0)
        for (int q = 0; q < 4; ++q) p.prof[q] = pc[q];
}

// wrec_t8h[jg][n][c32][hl][lane][e]: wrec_t8 as fp16 pairs in the operand order of v_mfma_f32_16x16x32_f16 (lane: col =
// lane & 15, kq = lane >> 4 -> k = 32*c32 + 8*kq + e over the operand [h ; o]; hl = 0 hi halves, 1 lo halves, x = hi + lo/2048)
__global__ void k_prep_wrec_t8h(const float* wrec_t8, float* wrec_t8h, int H) {
    const int nch = H >> 4;                       // K = 2H = nch chunks of 32
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)(H >> 3) * 2 * nch * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int c32 = (int)((idx >> 9) % nch), n = (int)(((idx >> 9) / nch) & 1), jg = (int)((idx >> 9) / nch / 2);
        const int lr = lane & 15, kq = lane >> 4, k = 32 * c32 + 8 * kq + e;
        const float w = wrec_t8[(((long)jg * 2 + n) * (2 * nch) + (k >> 4)) * 256 + lr * 16 + (k & 15)];
        unsigned short hi, lo;
        cvae_split_f16(w, hi, lo);
        unsigned short* dst = (unsigned short*)wrec_t8h + ((((long)jg * 2 + n) * nch + c32) * 2) * 512 + lane * 8 + e;
        dst[0] = hi;
        dst[512] = lo;
    }
}

// k_train_fwd_steps with the matrix product in split fp16 (the form of k_gru_steps_v5): weights and the exchanged h / o as
// (hi, lo) pairs of halves, x = hi + lo/2048, three v_mfma_f32_16x16x32_f16 per 32 k with fp32 accumulation, three independent
// accumulator sets.  The exchanged state is tile-planar like the buffers of k_gru_steps_v6 / k_train_bwd_steps: 2 KiB per
// (slot, 32-k chunk, 16-row tile) = { hi [4 kq][16 rows][8 halves] | lo likewise }, a block's 8 units being one kq quarter, so
// every operand load instruction reads one contiguous KiB and every publish store a run of whole lines nobody else writes;
// operands stream through a ring of 8 chunks with plain first-touch loads.  Slot 0 is split on the fly from the row-major fp32 copy; gate math,
// the tape and the row-major hrow / orow copies (backward, projection) stay fp32.  C32W = 32-k chunks per wave (H/64).
template <int C32W>
__global__ __launch_bounds__(256, 1) void k_train_fwd_steps_h(TrainFwdParams p) {
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, nch = H >> 4, ng = H >> 3, nrt = p.Bp >> 4, n32h = H >> 5;   // n32h: 32-k chunks per half of K
    const int jg = blockIdx.x % ng, ti = blockIdx.x / ng, rts = p.rts;
    const int c_lo = wave * C32W;                       // over [0, 2*n32h): h chunks then o chunks
    const bool from_o = c_lo >= n32h;
    const int cs = from_o ? c_lo - n32h : c_lo;         // first source 32-k chunk inside the h / o planes
    float* red = (float*)CVAE_SMEM;                     // [4 waves][16 rows][36]
    float* hsh = red + 4 * 16 * 36;                     // [2][16 rows][8]: h, o of this block's units
    const unsigned mtot = (unsigned)p.mtot;
    const unsigned bytes = (unsigned)((long)nch * p.mtot * 64);
    const cvae_buf hb = cvae_make_buf(p.hbuf, bytes), ob = cvae_make_buf(p.obuf, bytes);
    const cvae_buf src = from_o ? ob : hb;
    // 32-k chunk c of a half, slot s, tile i: 2 KiB at ((s*n32h + c)*nrt + i)*2048; lane (lr, kq): 16 B of hi at lane*16, of lo at 1024 + lane*16
    const unsigned voff = (unsigned)lane * 16u;
    constexpr int RD = C32W < 8 ? C32W : 8;             // operand ring: chunks in flight per wave
    const float* src0 = from_o ? p.orow : p.hrow;       // slot 0 (row-major fp32, written by the prologue)
    f32x4 wh[2][C32W], wl[2][C32W];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int ci = 0; ci < C32W; ++ci) {
            const float* w = p.wrec_t8 + ((((long)jg * 2 + n) * nch + c_lo + ci) * 2) * 256 + lane * 4;
            wh[n][ci] = *(const f32x4*)w;
            wl[n][ci] = *(const f32x4*)(w + 256);
        }
    const int gt = tid - 128, row = (gt >> 3) & 15, u8 = gt & 7, j = 8 * jg + u8;
    const bool gate = tid >= 128;
    const float bhn = p.bhn[j];
    long long pc[4] = {0, 0, 0, 0};
    const bool prof = p.prof && blockIdx.x == 0;
    const int ntile = ti < nrt ? (nrt - ti + rts - 1) / rts : 0;
    float hkeep0 = 0.f, hkeep1 = 0.f;                   // this gate thread's previous h per tile (up to two tiles per block)
    const int backoff = p.backoff;                      // x 64 cycles
    // The flags of the NEXT task are requested before this task's reduce / cell / publish: with several row tiles per block
    // they were raised a whole task ago, and a poll that starts after the publish costs a memory round trip before the first
    // operand load can go out even then.  Flags only grow, so an early value is a valid lower bound.
    unsigned fnext = 0u;
    for (int t = 0; t < p.T; ++t) {
        int tcount = 0;
        for (int i = ti; i < nrt; i += rts, ++tcount) {
            long long c0 = prof ? cvae_clock() : 0;
            if (t > 0 && !cvae_wave_all(fnext >= (unsigned)t)) {   // 8-unit planes [4cs, 4(cs + C32W)) of slot t: plane q is published by block q
                unsigned spins = 0;
                if (ntile == 1)   // nothing can be up right after this block's own publish: do not poll through that window
                    for (int q = 0; q < backoff; ++q) cvae_sleep_64();
                for (;;) {
                    unsigned f = (unsigned)t;
                    if (lane < 4 * C32W) f = cvae_atomic_load_agent(p.flags + (long)i * ng + 4 * cs + lane);
                    if (cvae_wave_all(f >= (unsigned)t)) break;
                    cvae_sleep();
                    if (++spins > (1u << 22)) {
                        p.status[0] = 3;
                        break;
                    }
                }
            }
            cvae_compiler_fence();
            if (prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
            const unsigned row0 = (unsigned)(t * p.Bp + i * 16);
            f32x4 a_hi[C32W], a_lo[C32W];
            auto load_op = [&](int ci) {
                const unsigned so = (((unsigned)t * (unsigned)n32h + (unsigned)(cs + ci)) * (unsigned)nrt + (unsigned)i) * 2048u;
                a_hi[ci] = cvae_buf_load_f4(src, voff, so);
                a_lo[ci] = cvae_buf_load_f4(src, voff, so + 1024u);
            };
            if (t == 0) {
#pragma unroll
                for (int ci = 0; ci < C32W; ++ci) {
                    const float* s0 = src0 + (long)(row0 + lr) * H + 32 * (cs + ci) + 8 * kq;
                    const f32x4 v0 = *(const f32x4*)s0, v1 = *(const f32x4*)(s0 + 4);
                    unsigned ph[4], pl[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float x0 = e < 2 ? v0[2 * e] : v1[2 * e - 4], x1 = e < 2 ? v0[2 * e + 1] : v1[2 * e - 3];
                        unsigned short h0, l0, h1, l1;
                        cvae_split_f16(x0, h0, l0);
                        cvae_split_f16(x1, h1, l1);
                        ph[e] = (unsigned)h0 | ((unsigned)h1 << 16);
                        pl[e] = (unsigned)l0 | ((unsigned)l1 << 16);
                    }
                    a_hi[ci] = (f32x4){__builtin_bit_cast(float, ph[0]), __builtin_bit_cast(float, ph[1]),
                                       __builtin_bit_cast(float, ph[2]), __builtin_bit_cast(float, ph[3])};
                    a_lo[ci] = (f32x4){__builtin_bit_cast(float, pl[0]), __builtin_bit_cast(float, pl[1]),
                                       __builtin_bit_cast(float, pl[2]), __builtin_bit_cast(float, pl[3])};
                }
            } else {
#pragma unroll
                for (int ci = 0; ci < RD; ++ci) load_op(ci);
            }
            const int grow = i * 16 + row;
            const bool live = gate && grow < p.B;
            const bool keep1 = ntile == 2 && (tcount & 1);
            float g0 = 0.f, g1 = 0.f, g2 = 0.f, hold = keep1 ? hkeep1 : hkeep0, msk = 0.f;
            if (live) {
                const float* gip = p.gi + ((long)t * p.Bp + grow) * 3 * H;
                g0 = gip[j]; g1 = gip[H + j]; g2 = gip[2 * H + j];
                if (t == 0) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, g0, g1, g2);
                if (t == 0 || ntile > 2) hold = p.hrow[(long)(row0 + row) * H + j];   // row-major fp32 copy (t > 0: written by this thread)
                msk = p.gmask[((long)t * p.B + grow) * H + j];
            }
            f32x4 acc[2], accx[2], accy[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                accx[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                accy[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int ci = 0; ci < C32W; ++ci) {
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[n] = cvae_mfma_16x16x32_f16(a_hi[ci], wh[n][ci], acc[n]);
#pragma unroll
                for (int n = 0; n < 2; ++n) accx[n] = cvae_mfma_16x16x32_f16(a_hi[ci], wl[n][ci], accx[n]);
#pragma unroll
                for (int n = 0; n < 2; ++n) accy[n] = cvae_mfma_16x16x32_f16(a_lo[ci], wh[n][ci], accy[n]);
                cvae_sched_fence();
                if (t > 0 && ci + RD < C32W) load_op(ci + RD);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[(wave * 16 + kq * 4 + q) * 36 + n * 16 + lr] = acc[n][q] + (accx[n][q] + accy[n][q]) * (1.0f / 2048.0f);
            {   // next task of this block: the following tile of step t, or this block's first tile of step t + 1
                const bool wrap = i + rts >= nrt;
                const int i_n = wrap ? ti : i + rts, t_n = wrap ? t + 1 : t;
                fnext = (unsigned)t_n;
                if (t_n > 0 && t_n < p.T && lane < 4 * C32W) fnext = cvae_atomic_load_agent(p.flags + (long)i_n * ng + 4 * cs + lane);
                if (t_n == 0) fnext = 0u;
            }
            if (prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
            __syncthreads();
            if (gate) {
                float rg = 0.f, zg = 0.f, ng_ = 0.f, qq = 0.f, hn = 0.f, on = 0.f;
                if (live) {
                    const int col = (u8 >> 2) * 16 + (u8 & 3);
                    float s[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        s[a] = red[(0 * 16 + row) * 36 + col + a * 4] + red[(1 * 16 + row) * 36 + col + a * 4] +
                               red[(2 * 16 + row) * 36 + col + a * 4] + red[(3 * 16 + row) * 36 + col + a * 4];
                    rg = cvae_sigmoid(g0 + s[0]);
                    zg = cvae_sigmoid(g1 + s[1]);
                    qq = s[3] + bhn;
                    ng_ = tanhf(g2 + s[2] + rg * qq);
                    hn = ng_ + zg * (hold - ng_);
                    on = hn * msk;
                }
                if (keep1) hkeep1 = hn; else hkeep0 = hn;
                hsh[row * 8 + u8] = hn;
                hsh[128 + row * 8 + u8] = on;
                if (grow < p.Bp) {
                    p.hrow[((long)(t + 1) * p.Bp + grow) * H + j] = hn;
                    p.orow[((long)(t + 1) * p.Bp + grow) * H + j] = on;
                    float* tp = p.tape + ((long)t * p.Bp + grow) * 4 * H + j;
                    tp[0] = rg; tp[H] = zg; tp[2 * H] = ng_; tp[3 * H] = qq;
                }
            }
            __syncthreads();
            if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
            if (tid < 64) {   // wave 0: lanes 0..31 publish h (16 rows x 16 B of hi, then of lo: two 256-byte runs), lanes 32..63 publish o
                const int which = tid >> 5, l = tid & 31, r = l & 15, part = l >> 4;
                const float* hv = hsh + which * 128 + r * 8;
                unsigned pk[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned short h0, l0, h1, l1;
                    cvae_split_f16(hv[2 * e], h0, l0);
                    cvae_split_f16(hv[2 * e + 1], h1, l1);
                    pk[e] = part == 0 ? ((unsigned)h0 | ((unsigned)h1 << 16)) : ((unsigned)l0 | ((unsigned)l1 << 16));
                }
                const f32x4 v = (f32x4){__builtin_bit_cast(float, pk[0]), __builtin_bit_cast(float, pk[1]),
                                        __builtin_bit_cast(float, pk[2]), __builtin_bit_cast(float, pk[3])};
                // slot t+1, chunk jg/4, tile i; this block's units are quarter jg%4 of the chunk's 32 k
                const unsigned so = (((unsigned)(t + 1) * (unsigned)n32h + (unsigned)(jg >> 2)) * (unsigned)nrt + (unsigned)i) * 2048u;
                cvae_buf_store_f4_sc1(which ? ob : hb, (unsigned)part * 1024u + (unsigned)(jg & 3) * 256u + (unsigned)r * 16u, so, v);
                cvae_drain_vmem();
                cvae_wave_barrier();
                if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * ng + jg, (unsigned)(t + 1));
            }
            if (prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
        }
    }
    if (prof && tid == 0)
        for (int q = 0; q < 4; ++q) p.prof[q] = pc[q];
}

#include "cvae_train_bwd.h"
#include "cvae_train_x3.h"
#include "cvae_train_w3.h"
#include "cvae_train_ll.h"
