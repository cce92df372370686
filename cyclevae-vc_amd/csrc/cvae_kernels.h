// Device kernels of the CycleVAE hot path for gfx950 (wave64, MFMA f32 16x16x4).
//
// Data layouts (all float32):
//   xnp   [B][Tp][Cp]        normalised, zero-padded input; Tp = T + R - 1 (R = ks^2 taps), Cp = ceil4(Cin).
//                            Frame t's conv receptive field is the R*Cp contiguous floats starting at row t.
//   gx    [B*Tp][3H]         input-side gate pre-activations (gate order r,z,n like torch.nn.GRU)
//   hbuf  [H/16][Mtot][16]   hidden state, "chunk-major": Mtot = (T+1)*Bp rows (slot s = rows s*Bp..), so a
//                            16-row x 16-k MFMA operand tile is one contiguous, fully coalesced 1 KiB block
//   wrec  [H/4][H/16][16][16] recurrent weights per 4-unit group g: column col = a*4+u (a: r,z,n_in,n_h; unit
//                            j = 4g+u), k-contiguous; the autoregressive feedback W_ih[:,R*C:]*out_1 is folded in
//   y     [T*Bp][Cop]        raw projections, row = t*Bp + b
#pragma once
#include <cvae_intrin.h>
#include <stdint.h>

struct CvaeSeg {
    const float* ptr;
    int width;
    int row_stride;
};

// ------------------------------------------------------------------------------------------------------
// Philox4x32-10 -> N(0,1)  (on-device stand-in for the reference's torch.randn, gru_vae.py:91-94)
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cvae_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                            uint32_t k1, uint32_t out[4]) {
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Keyed by (seed; GLOBAL frame index, latent dim QUAD, draw id): the frame index counts batch rows from the data-parallel job's
// row 0 (cvae_set_draw_origin), so a row draws the same eps on whichever rank it lands.  (Draw ids are small: the word that held
// their upper half carries the frame index's upper half, which leaves every stream with frame < 2^32 as it was.)
// One Philox block gives the FOUR normals of dims 4*quad .. 4*quad+3: words (0, 1) -> Box-Muller radius and angle -> the cosine
// and the sine branch, words (2, 3) likewise.  (Rounds 1-4 spent a whole block and one branch on every value: the 300-draw
// latent means of a ten-pair stage-6 call -- 190 M values -- were 0.87 ms of its 7.4.)
__device__ __forceinline__ void cvae_randn_pair(uint32_t w0, uint32_t w1, float& zc, float& zs) {
    const float u1 = ((float)(w0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(w1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float r = sqrtf(-2.0f * logf(u1)), a = 6.283185307179586f * u2;
    zc = r * cosf(a);
    zs = r * sinf(a);
}
__device__ __forceinline__ void cvae_randn4(uint64_t seed, uint64_t draw, uint64_t frame, uint32_t quad, float z[4]) {
    uint32_t o[4];
    cvae_philox((uint32_t)frame, quad, (uint32_t)draw, (uint32_t)(frame >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    cvae_randn_pair(o[0], o[1], z[0], z[1]);
    cvae_randn_pair(o[2], o[3], z[2], z[3]);
}
// one of them (same bits as cvae_randn4 gives for that dim)
__device__ __forceinline__ float cvae_randn(uint64_t seed, uint64_t draw, uint64_t frame, uint32_t dim) {
    uint32_t o[4];
    cvae_philox((uint32_t)frame, dim >> 2, (uint32_t)draw, (uint32_t)(frame >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
    float zc, zs;
    cvae_randn_pair((dim & 2) ? o[2] : o[0], (dim & 2) ? o[3] : o[1], zc, zs);
    return (dim & 1) ? zs : zc;
}

// ------------------------------------------------------------------------------------------------------
// prepare-time kernels (run when the weights change)
// ------------------------------------------------------------------------------------------------------
// mfull[o][d][c] = sum_i conv1.w[o][i][j] * conv0.w[i][c][k], tap d = ks*j + k   (gru_vae.py:49-51,62-64)
__global__ void k_prep_mfull(const float* w0, const float* w1, double* mfull, int C, int ks) {
    const int c1 = ks * C, c2 = ks * ks * C, R = ks * ks;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)c2 * R * C) {
        const int c = (int)(idx % C), d = (int)((idx / C) % R), o = (int)(idx / ((long)C * R));
        const int j = d / ks, k = d % ks;
        double s = 0.0;
        for (int i = 0; i < c1; ++i) s += (double)w1[((long)o * c1 + i) * ks + j] * (double)w0[((long)i * C + c) * ks + k];
        mfull[idx] = s;
    }
}

// bprime[o] = conv1.b[o] + sum_j sum_i conv1.w[o][i][j] * conv0.b[i]   (conv0's bias reaches every padded tap)
__global__ void k_prep_bprime(const float* b0, const float* w1, const float* b1, double* bprime, int C, int ks) {
    const int c1 = ks * C, c2 = ks * ks * C;
    const int o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o < c2) {
        double s = (double)b1[o];
        for (int i = 0; i < c1; ++i)
            for (int j = 0; j < ks; ++j) s += (double)w1[((long)o * c1 + i) * ks + j] * (double)b0[i];
        bprime[o] = s;
    }
}

// afold[n][d*Cp + c] = sum_o W_ih[n][o] * mfull[o][d][c]  (zero in the Cp / Kfe padding)
__global__ void k_prep_afold(const float* wih, const double* mfull, float* afold, int C, int Cp, int ks, int tot,
                             int Kfe, int H3) {
    const int R = ks * ks, c2 = R * C;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)H3 * Kfe) {
        const int kc = (int)(idx % Kfe), n = (int)(idx / Kfe);
        const int d = kc / Cp, c = kc % Cp;
        double s = 0.0;
        if (d < R && c < C)
            for (int o = 0; o < c2; ++o) s += (double)wih[(long)n * tot + o] * mfull[((long)o * R + d) * C + c];
        afold[idx] = (float)s;
    }
}

// cfold[n] = b_ih[n] + W_ih[n,:c2].bprime + W_ih[n,c2:].b_o + (n < 2H ? b_hh[n] : 0)
__global__ void k_prep_cfold(const float* wih, const float* bih, const float* bhh, const float* bo,
                             const double* bprime, float* cfold, int c2, int Co, int tot, int H) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < 3 * H) {
        double s = (double)bih[n];
        for (int o = 0; o < c2; ++o) s += (double)wih[(long)n * tot + o] * bprime[o];
        for (int c = 0; c < Co; ++c) s += (double)wih[(long)n * tot + c2 + c] * (double)bo[c];
        if (n < 2 * H) s += (double)bhh[n];
        cfold[n] = (float)s;
    }
}

// wrec[g][c][col][kk]: col = a*4+u, unit j = 4g+u, k = 16c+kk;  F = W_ih[:,c2:] * out_1.w  (feedback fold)
//   a=0: W_hr + F_r   a=1: W_hz + F_z   a=2: F_n   a=3: W_hn
__global__ void k_prep_wrec(const float* wih, const float* whh, const float* wo, float* wrec, int c2, int Co,
                            int tot, int H) {
    const int nch = H >> 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)(H >> 2) * nch * 256) {
        const int kk = (int)(idx & 15), col = (int)((idx >> 4) & 15);
        const int c = (int)((idx >> 8) % nch), g = (int)((idx >> 8) / nch);
        const int a = col >> 2, u = col & 3, j = 4 * g + u, k = 16 * c + kk;
        const int gate = a < 3 ? a : 2;
        double s = 0.0;
        if (a < 3) {
            const float* wrow = wih + (long)(gate * H + j) * tot + c2;
            for (int q = 0; q < Co; ++q) s += (double)wrow[q] * (double)wo[(long)q * H + k];
        }
        if (a != 2) s += (double)whh[(long)(gate * H + j) * H + k];
        wrec[idx] = (float)s;
    }
}

// wrec2[j][a][c][u][kk]: the same folded recurrent weights, grouped for the 2-D decomposition: block j owns the 16
// hidden units 16j..16j+15 (= h chunk j); a = accumulator kind (r, z, n_in, n_h) is one 16-column MFMA tile.
__global__ void k_prep_wrec2(const float* wih, const float* whh, const float* wo, float* wrec2, int c2, int Co,
                             int tot, int H) {
    const int nch = H >> 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)nch * 4 * nch * 256) {
        const int kk = (int)(idx & 15), u = (int)((idx >> 4) & 15);
        const int c = (int)((idx >> 8) % nch), a = (int)(((idx >> 8) / nch) & 3), jg = (int)((idx >> 8) / nch / 4);
        const int j = 16 * jg + u, k = 16 * c + kk;
        const int gate = a < 3 ? a : 2;
        double s = 0.0;
        if (a < 3) {
            const float* wrow = wih + (long)(gate * H + j) * tot + c2;
            for (int q = 0; q < Co; ++q) s += (double)wrow[q] * (double)wo[(long)q * H + k];
        }
        if (a != 2) s += (double)whh[(long)(gate * H + j) * H + k];
        wrec2[idx] = (float)s;
    }
}

// wo2[c][k] = sum_q scale_out.w[c][q] * out_1.w[q][k] (or out_1.w when there is no scale_out); rows c >= Co are zero.
// bo2[c] = scale_out.w[c,:] . out_1.b + scale_out.b[c]   (idx in [Cop*H, Cop*H + Cop) computes the bias)
__global__ void k_prep_wo2(const float* wo, const float* bo, const float* sw, const float* sb, float* wo2, float* bo2,
                           int Co, int Cop, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)Cop * H) {
        const int k = (int)(idx % H), c = (int)(idx / H);
        double s = 0.0;
        if (c < Co) {
            if (sw)
                for (int q = 0; q < Co; ++q) s += (double)sw[(long)c * Co + q] * (double)wo[(long)q * H + k];
            else
                s = (double)wo[(long)c * H + k];
        }
        wo2[idx] = (float)s;
    } else if (idx < (long)Cop * H + Cop) {
        const int c = (int)(idx - (long)Cop * H);
        double s = 0.0;
        if (c < Co) {
            if (sw) {
                s = (double)sb[c];
                for (int q = 0; q < Co; ++q) s += (double)sw[(long)c * Co + q] * (double)bo[q];
            } else {
                s = (double)bo[c];
            }
        }
        bo2[c] = (float)s;
    }
}

// generic strided 2-D copy: dst[r*dld + c] = src[r*sld + c]
__global__ void k_copy2d(float* dst, long dld, const float* src, long sld, int rows, int cols) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)rows * cols) {
        const int c = (int)(idx % cols), r = (int)(idx / cols);
        dst[(long)r * dld + c] = src[(long)r * sld + c];
    }
}

// ------------------------------------------------------------------------------------------------------
// per-pass kernels
// ------------------------------------------------------------------------------------------------------
struct ProCell {
    CvaeSeg seg0, seg1;
    const float* lat;
    const float* eps;
    uint64_t seed, draw;
    const float* y_in;  // [B][Co]
    const float* h_in;  // [B][H] or null
    int frames;         // valid frames of this cell (<= T; 0 = T): later frames are zero after normalisation, like the conv padding
    int n_draws;        // > 1: z uses the MEAN of n_draws eps (draw ids draw .. draw+n-1; eps [n_draws][B][T][L])
    int ctx_before, ctx_after;   // window of a longer utterance (cvae_pass_input, ABI 5): real frames in front of frame 0 / behind the last
    long draw_frame0;            // Philox frame index of the window's frame 0
    long eps_stride;             // floats between two draws in eps (0: B*T*L)
};
// frame t of the padded input of cell c holds data (else zeros: conv padding, or beyond a shorter cell's end)
#define CVAE_FRAME_VALID(p, c, t) ((t) >= -(c).ctx_before && (t) < ((c).frames > 0 ? (c).frames : (p).T) + (c).ctx_after)

#define CVAE_MAX_CELLS 32     // one 32-row tile of the dataflow kernels (ProParams stays under the 4 KiB kernel-argument limit)
struct ProParams {
    ProCell cell[CVAE_MAX_CELLS];
    int ncell, L;
    uint64_t frame0;     // global frame index of this pass's (row 0, frame 0): draw-origin row * T
    const float* sin_w;  // [C][C] or null
    const float* sin_b;
    const float* wo;     // out_1.w [Cop][H]
    const float* bo;
    int B, T, C, Cp, pad, Co, H, Bp, nslack, nzero;
    long mtot;
    float* xnp;          // [ncell*B][Tp][Cp] + nslack floats kept zero
    float* hbuf;
    float* hx;           // null, or slot 0 as fp16 triples for k_gru_steps_v6 ([H/16][mtot/32][limb][kh][32 rows][8 halves])
    float* xt;           // null, or xnp as fp16 triples in the layout k_gru_steps_v6 reads: [Bp/32][Tp][Cp/8][limb][32 rows][8 halves]
                         //   (rows >= ncell*B are zero); frame t's window is then 9*Cp/8 consecutive pieces of 3 x 512 bytes
    int nxt_slack;       // 16-bit words kept zero behind xt (read by the K padding of the last frames)
    float* hs;           // null, or the fp16-pair copy of slot 0 for k_gru_steps_v5 ([H/16][mtot][16 hi | 16 lo] halves)
    float* xs;           // null, or xnp as fp16 pairs for k_gru_steps_v5: hi plane then lo plane, xs_plane halves each,
    long xs_plane;       //   same [row][Tp][Cp] indexing as xnp (Cp % 8 == 0: 8 consecutive halves are one 16-byte operand)
    float* dy;           // [ncell*B][Co]: y_in - out_1(h_in)
    unsigned* zero_words;
    int* zero_status;    // null, or the pass's 8 status words: cleared here instead of by a launch of their own in front of the prologue
    unsigned* ll_counter;   // null, or the workspace's launch counter of k_gru_steps_ll: incremented here, read there as the tag nonce
    int nA, nH, nD;      // block ranges: [0,nA) assemble rows, [nA,nA+nH) slot-0 init, then dy, then gx0, last block zeroing
    int nG;              // 0, or blocks of the frame-0 feedback correction gx0 (only when no cell carries a state in: dy = y_in - out_1.b)
    float* gx0;          // [ncell*B][3H]: W_ih[:, R*C:] . dy -- what the recurrent kernel adds to the gates of frame 0 (else it forms
    const float* wyT;    //   the sums itself, cvae_t0_fix: 4 loads per channel and thread in front of its first step); wyT [Co][3H]
};
static_assert(sizeof(ProParams) <= 4096, "ProParams is a kernel argument: 4 KiB at most");

// Everything a pass needs before its GEMM, in one launch of 64-thread blocks (role by block range):
//   assemble : input row [seg0 ; seg1 | z] -> scale_in (dense CxC, gru_vae.py:336) -> zero-padded xnp
//   slot-0   : hbuf slot 0 <- h_in or zeros (batch padding rows zero)
//   dy       : y_in - (out_1.b + out_1.w . h_in): what frame 0 must add through W_ih[:,R*C:] because the folded
//              recurrent matrix assumes y_{-1} = out_1(h_{-1})
//   zeroing  : xnp slack read by the K padding, barrier / flag words
// one value of a pass's input row: [seg0 ; seg1 | z] (z: reparameterised draw, single or the mean of n_draws)
// emean: null, or the block's mean draw per latent dim (cvae_mean_draws)
__device__ __forceinline__ float cvae_input_value(const ProParams& p, const ProCell& c, int b, int t, int q, const float* emean = nullptr) {
    const long fr = (long)b * p.T + t;
    if (q < c.seg0.width) return c.seg0.ptr[fr * c.seg0.row_stride + q];
    if (!c.lat) return c.seg1.ptr[fr * c.seg1.row_stride + (q - c.seg0.width)];
    const int l = q - c.seg0.width;
    float e;
    if (c.n_draws > 1 && emean) {
        e = emean[l];
    } else if (c.n_draws > 1) {    // mean of the draws (decode_gru-cyclevae_gauss.py:304-305: mean of n_smpl_dec samples)
        e = 0.0f;
        const long es = c.eps_stride ? c.eps_stride : (long)p.B * p.T * p.L;
        for (int k = 0; k < c.n_draws; ++k)
            e += c.eps ? c.eps[(long)k * es + fr * p.L + l]
                       : cvae_randn(c.seed, c.draw + (uint64_t)k, (uint64_t)(fr + c.draw_frame0) + p.frame0, (uint32_t)l);
        e *= 1.0f / (float)c.n_draws;
    } else {
        e = c.eps ? c.eps[fr * p.L + l] : cvae_randn(c.seed, c.draw, (uint64_t)(fr + c.draw_frame0) + p.frame0, (uint32_t)l);
    }
    return c.lat[fr * 2 * p.L + l] + expf(c.lat[fr * 2 * p.L + p.L + l] * 0.5f) * e;
}

// The mean of a frame's n_draws draws with all 256 threads of a block: slice s of the draws (k = s, s + nsl, ...) per latent dim
// in parallel, the slices added in fixed order (300 Philox + Box-Muller evaluations in one lane made the stage-6 prologue 540 us).
// part: [256] floats, emean: [L] floats of LDS.  Every thread of the block must call it.
__device__ __forceinline__ void cvae_mean_draws(const ProParams& p, const ProCell& c, int b, int t, float* part, float* emean) {
    // thread = (dim quad, slice of the draws); part: [256 / (L/4) slices][L] floats (<= 1024), emean: [L]; L % 4 == 0
    const int tid = threadIdx.x, L = p.L, nq = L >> 2, nsl = 256 / nq, qd = tid % nq, sl = tid / nq;
    const long fr = (long)b * p.T + t;
    const long es = c.eps_stride ? c.eps_stride : (long)p.B * p.T * p.L;
    if (sl < nsl) {
        float e[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = sl; k < c.n_draws; k += nsl) {
            float z[4];
            if (c.eps) {
                const float* ep = c.eps + (long)k * es + fr * p.L + 4 * qd;
                z[0] = ep[0]; z[1] = ep[1]; z[2] = ep[2]; z[3] = ep[3];
            } else {
                cvae_randn4(c.seed, c.draw + (uint64_t)k, (uint64_t)(fr + c.draw_frame0) + p.frame0, (uint32_t)qd, z);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] += z[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) part[sl * L + 4 * qd + j] = e[j];
    }
    __syncthreads();
    if (tid < L) {
        float m = 0.0f;
        for (int q = 0; q < nsl; ++q) m += part[q * L + tid];
        emean[tid] = m * (1.0f / (float)c.n_draws);
    }
    __syncthreads();
}

// Blocks of 256 threads when the limb-triple input of k_gru_steps_v6 is built (p.xt): an assemble block then owns one
// (32-row tile, padded frame) and writes its Cp/8 pieces of 1280 B as whole coalesced runs from an LDS image; the other roles use
// the first 64 threads.  Blocks of 64 threads otherwise (one assemble block per (row, padded frame)).
__global__ void k_prologue(ProParams p) {
    const int tid = threadIdx.x, blk = blockIdx.x;
    const int Tp = p.T + 2 * p.pad;
    if (blk < p.nA && p.xt) {
        float* raw = (float*)CVAE_SMEM;                            // [32][C + 1]
        unsigned char* img = (unsigned char*)(raw + 32 * (p.C + 1));   // [Cp/8][1280]
        const int tile = blk / Tp, tp = blk % Tp, t = tp - p.pad, np = p.Cp >> 3;
        // the means of many draws first, one (row, dim quad) per thread: [32][L] behind the image (cells with n_draws > 1, L % 4 == 0)
        float* tmean = (float*)(img + np * 1280);
        const bool quads = p.L > 0 && (p.L & 3) == 0;
        if (quads) {
            const int nq = p.L >> 2;
            for (int item = tid; item < 32 * nq; item += 256) {
                const int r = item / nq, qd = item - r * nq, bb = tile * 32 + r;
                if (bb >= p.ncell * p.B) continue;
                const ProCell& c = p.cell[bb / p.B];
                if (!c.lat || c.n_draws <= 1 || !CVAE_FRAME_VALID(p, c, t)) continue;
                const long fr = (long)(bb % p.B) * p.T + t, es = c.eps_stride ? c.eps_stride : (long)p.B * p.T * p.L;
                float e[4] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < c.n_draws; ++k) {
                    float z[4];
                    if (c.eps) {
                        const float* ep = c.eps + (long)k * es + fr * p.L + 4 * qd;
                        z[0] = ep[0]; z[1] = ep[1]; z[2] = ep[2]; z[3] = ep[3];
                    } else {
                        cvae_randn4(c.seed, c.draw + (uint64_t)k, (uint64_t)(fr + c.draw_frame0) + p.frame0, (uint32_t)qd, z);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) e[j] += z[j];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) tmean[r * p.L + 4 * qd + j] = e[j] * (1.0f / (float)c.n_draws);
            }
            __syncthreads();
        }
        for (int idx = tid; idx < 32 * p.C; idx += 256) {
            const int r = idx / p.C, q = idx - r * p.C, bb = tile * 32 + r;
            float v = 0.0f;
            if (bb < p.ncell * p.B) {
                const ProCell& c = p.cell[bb / p.B];
                if (CVAE_FRAME_VALID(p, c, t)) v = cvae_input_value(p, c, bb % p.B, t, q, quads ? tmean + r * p.L : nullptr);
            }
            raw[r * (p.C + 1) + q] = v;
        }
        __syncthreads();
        // thread = (channel q = tid % 64 (+ 64, ...), rows tid / 64 + 4 i): one scale_in weight load feeds eight rows' FMAs, the
        // rows' inputs are LDS broadcasts (a wave shares its row)
        const ProCell* const cells = p.cell;
        for (int q = tid & 63; q < p.Cp; q += 64) {
            float v[8];
            bool ok[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = (tid >> 6) + 4 * i, bb = tile * 32 + r;
                ok[i] = false;
                if (bb < p.ncell * p.B) {
                    const ProCell& c = cells[bb / p.B];
                    ok[i] = CVAE_FRAME_VALID(p, c, t) && q < p.C;
                }
                v[i] = !ok[i] ? 0.0f : (p.sin_w ? p.sin_b[q] : raw[r * (p.C + 1) + q]);
            }
            if (p.sin_w && q < p.C)
                for (int k = 0; k < p.C; ++k) {
                    const float w = p.sin_w[(long)q * p.C + k];     // (staging the matrix in LDS first was measured slower: 22 vs 18 us)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] += w * raw[((tid >> 6) + 4 * i) * (p.C + 1) + k];
                }
            unsigned char* pc = img + (q >> 3) * 1280;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = (tid >> 6) + 4 * i;
                unsigned short l0, l1;
                unsigned char l2;
                cvae_split3_f16b8(ok[i] ? v[i] : 0.0f, l0, l1, l2);
                ((unsigned short*)pc)[r * 8 + (q & 7)] = l0;
                ((unsigned short*)(pc + 512))[r * 8 + (q & 7)] = l1;
                pc[1024 + r * 8 + (q & 7)] = l2;
            }
        }
        __syncthreads();
        f32x4* dst = (f32x4*)((unsigned char*)p.xt + ((long)tile * Tp + tp) * np * 1280);
        for (int e = tid; e < np * 80; e += 256) dst[e] = ((const f32x4*)img)[e];
        return;
    }
    if (p.nG > 0 && blk >= p.nA + p.nH + p.nD && blk < p.nA + p.nH + p.nD + p.nG) {
        // gx0[bb][col .. col+3] = sum_c wyT[c][col ..] * (y_in[bb][c] - out_1.b[c]): every thread of the block, four columns each
        const long q4 = ((long)(blk - p.nA - p.nH - p.nD) * blockDim.x + tid) * 4, H3 = 3L * p.H;
        if (q4 < (long)p.ncell * p.B * H3) {
            const int bb = (int)(q4 / H3), col = (int)(q4 - (long)bb * H3);
            const float* yi = p.cell[bb / p.B].y_in + (long)(bb % p.B) * p.Co;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int c = 0;
            for (; c + 8 <= p.Co; c += 8) {          // eight channels' loads in flight before the first use (see cvae_t0_fix)
                float d[8], w[8][4];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    d[q] = yi[c + q] - p.bo[c + q];
                    const float* wp = p.wyT + (long)(c + q) * H3 + col;
                    w[q][0] = wp[0]; w[q][1] = wp[1]; w[q][2] = wp[2]; w[q][3] = wp[3];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) { a0 += w[q][0] * d[q]; a1 += w[q][1] * d[q]; a2 += w[q][2] * d[q]; a3 += w[q][3] * d[q]; }
            }
            for (; c < p.Co; ++c) {
                const float d = yi[c] - p.bo[c];
                const float* w = p.wyT + (long)c * H3 + col;
                a0 += w[0] * d; a1 += w[1] * d; a2 += w[2] * d; a3 += w[3] * d;
            }
            float* o = p.gx0 + (long)bb * H3 + col;
            o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
        }
        return;
    }
    const float* emean = nullptr;
    if (blk < p.nA && blockDim.x == 256) {      // (256-thread launch without xt: a cell takes the mean of many draws)
        const int tp = blk % Tp, bb = blk / Tp, t = tp - p.pad;
        if (bb < p.ncell * p.B) {
            const ProCell& c = p.cell[bb / p.B];
            if (c.lat && c.n_draws > 1 && p.L <= 256 && (p.L & 3) == 0 && CVAE_FRAME_VALID(p, c, t)) {
                float* part = (float*)CVAE_SMEM + p.C;
                cvae_mean_draws(p, c, bb % p.B, t, part, part + 1024);
                emean = part + 1024;
            }
        }
    }
    if (tid >= 64) return;          // (256-thread launch: every other role is written for 64 threads)
    if (blk < p.nA) {
        float* row = (float*)CVAE_SMEM;
        const int tp = blk % Tp, bb = blk / Tp, t = tp - p.pad;
        const bool real_row = bb < p.ncell * p.B;           // (with xt the range covers the batch padding rows too: zeros)
        const int ci = real_row ? bb / p.B : 0, b = real_row ? bb % p.B : 0;
        const ProCell& c = p.cell[ci];
        const bool valid = real_row && CVAE_FRAME_VALID(p, c, t);
        const long fr = (long)b * p.T + t;
        if (valid)
            for (int q = tid; q < p.C; q += 64) row[q] = cvae_input_value(p, c, b, t, q, emean);
        __syncthreads();
        for (int q = tid; q < p.Cp; q += 64) {
            float v = 0.0f;
            if (valid && q < p.C) {
                if (p.sin_w) {
                    v = p.sin_b[q];
                    for (int r = 0; r < p.C; ++r) v += p.sin_w[(long)q * p.C + r] * row[r];
                } else {
                    v = row[q];
                }
            }
            if (real_row) p.xnp[((long)bb * Tp + tp) * p.Cp + q] = v;
            if (p.xs) {
                unsigned short hi, lo;
                cvae_split_f16(v, hi, lo);
                unsigned short* xh = (unsigned short*)p.xs + ((long)bb * Tp + tp) * p.Cp + q;
                xh[0] = hi;
                xh[p.xs_plane] = lo;
            }
        }
    } else if (blk < p.nA + p.nH) {
        const long base = (long)(blk - p.nA) * 1024;
        bool any_h = false;
        for (int c = 0; c < p.ncell; ++c) any_h = any_h || p.cell[c].h_in != nullptr;
        if (!any_h) {   // fresh pass (the usual case): slot 0 is all zeros in every representation -- wide stores
            const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
            const long nel = (long)p.Bp * p.H, hb_ = blk - p.nA;
            for (int e = 0; e < 4; ++e) {                   // hbuf: 256 float4 of this block's 1024 elements
                const long idx = base + 4 * (e * 64 + tid);
                if (idx < nel) *(f32x4*)(p.hbuf + ((idx >> 4) / p.Bp * p.mtot + (idx >> 4) % p.Bp) * 16 + (idx & 15)) = z4;
            }
            if (p.hx) {                                     // 2560 B per (chunk, tile): 5 bytes per element
                const long per = (long)(p.Bp >> 5) * 160, npc = (long)(p.H >> 4) * per;
                for (int e = 0; e < 5; ++e) {
                    const long pi = hb_ * 320 + e * 64 + tid;
                    if (pi < npc) *(f32x4*)((unsigned char*)p.hx + (pi / per) * (p.mtot >> 5) * 2560 + (pi % per) * 16) = z4;
                }
            }
            if (p.hs) {                                     // 64 B per row and chunk
                const long per = (long)p.Bp * 4, npc = (long)(p.H >> 4) * per;
                for (int e = 0; e < 4; ++e) {
                    const long pi = hb_ * 256 + e * 64 + tid;
                    if (pi < npc) *(f32x4*)((unsigned char*)p.hs + (pi / per) * p.mtot * 64 + (pi % per) * 16) = z4;
                }
            }
            return;
        }
        for (int e = 0; e < 16; ++e) {
            const long idx = base + e * 64 + tid;
            if (idx < (long)p.Bp * p.H) {
                const int kk = (int)(idx & 15), r = (int)((idx >> 4) % p.Bp), ch = (int)((idx >> 4) / p.Bp);
                float v = 0.0f;
                if (r < p.ncell * p.B) {
                    const float* h_in = p.cell[r / p.B].h_in;
                    if (h_in) v = h_in[(long)(r % p.B) * p.H + 16 * ch + kk];
                }
                p.hbuf[((long)ch * p.mtot + r) * 16 + kk] = v;
                if (p.hx) {   // 2560 bytes per (chunk, tile): l0 [kh][32 rows][8 halves] | l1 likewise | l2 [kh][32 rows][8 bytes]
                    unsigned short l0, l1;
                    unsigned char l2;
                    cvae_split3_f16b8(v, l0, l1, l2);
                    unsigned char* pc = (unsigned char*)p.hx + ((long)ch * (p.mtot >> 5) + (r >> 5)) * 2560;
                    const int at = (kk >> 3) * 256 + (r & 31) * 8 + (kk & 7);
                    ((unsigned short*)pc)[at] = l0;
                    ((unsigned short*)(pc + 1024))[at] = l1;
                    pc[2048 + at] = l2;
                }
                if (p.hs) {   // the fp16-pair copy the split-precision recurrence reads
                    unsigned short hi, lo;
                    cvae_split_f16(v, hi, lo);
                    unsigned short* hrow = (unsigned short*)p.hs + ((long)ch * p.mtot + r) * 32;
                    hrow[kk] = hi;
                    hrow[16 + kk] = lo;
                }
            }
        }
    } else if (blk < p.nA + p.nH + p.nD) {
        const int idx = (blk - p.nA - p.nH) * 64 + tid;
        if (idx < p.ncell * p.B * p.Co) {
            const int q = idx % p.Co, bb = idx / p.Co;
            const ProCell& c = p.cell[bb / p.B];
            const int b = bb % p.B;
            // (no y_in: the window continues the recurrence that left h_in -- its first frame is fed out_1(h_in) by the fold itself;
            //  the H-long dot product below is one thread's serial chain, ~90 us at H = 1024: not spent on a zero)
            float yh = p.bo[q];
            if (c.h_in && c.y_in)
                for (int k = 0; k < p.H; ++k) yh += p.wo[(long)q * p.H + k] * c.h_in[(long)b * p.H + k];
            p.dy[idx] = c.y_in ? c.y_in[(long)b * p.Co + q] - yh : 0.0f;
        }
    } else {
        for (int q = tid; q < p.nslack; q += 64) p.xnp[(long)p.ncell * p.B * Tp * p.Cp + q] = 0.0f;
        if (p.xs)
            for (int q = tid; q < p.nslack; q += 64) {
                unsigned short* xh = (unsigned short*)p.xs + (long)p.ncell * p.B * Tp * p.Cp + q;
                xh[0] = 0;
                xh[p.xs_plane] = 0;
            }
        if (p.xt)
            for (int q = tid; q < p.nxt_slack; q += 64) ((unsigned short*)p.xt)[(long)(p.Bp >> 5) * Tp * (p.Cp >> 3) * 640 + q] = 0;
        for (int q = tid; q < p.nzero; q += 64) p.zero_words[q] = 0u;
        if (p.zero_status && tid < 8) p.zero_status[tid] = 0;
        if (p.ll_counter && tid == 0) *p.ll_counter += 1u;
    }
}

// z = mu + exp(log_var/2)*eps  (sampling_vae_batch, gru_vae.py:85-98)
__global__ void k_sample(const float* lat, int rows, int L, const float* eps, uint64_t seed, uint64_t draw,
                         float* z, float* eps_out, uint64_t frame0) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)rows * L) {
        const int l = (int)(idx % L);
        const long n = idx / L;
        const float e = eps ? eps[idx] : cvae_randn(seed, draw, (uint64_t)n + frame0, (uint32_t)l);
        z[idx] = lat[n * 2 * L + l] + expf(lat[n * 2 * L + L + l] * 0.5f) * e;
        if (eps_out) eps_out[idx] = e;
    }
}

// z = mu - exp(log_scale) * sign(eps) * log1p(-2|eps|), eps ~ U(-0.4999, 0.5)   (sampling_vae_laplace, gru_vae.py:101-112: the inverse
// CDF of the Laplace distribution).  eps supplied, or one Philox word per element mapped to the reference's
// interval: u = -0.4999 + 0.9999 * (w + 0.5) / 2^32.
__global__ void k_sample_laplace(const float* lat, int rows, int L, const float* eps, uint64_t seed, uint64_t draw, float* z,
                                 float* eps_out, uint64_t frame0) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)rows * L) {
        const int l = (int)(idx % L);
        const long n = idx / L;
        float e;
        if (eps) {
            e = eps[idx];
        } else {
            const uint64_t frame = (uint64_t)n + frame0;
            uint32_t o[4];
            cvae_philox((uint32_t)frame, (uint32_t)l >> 2, (uint32_t)draw, (uint32_t)(frame >> 32), (uint32_t)seed, (uint32_t)(seed >> 32), o);
            e = -0.4999f + 0.9999f * (((float)o[l & 3] + 0.5f) * 2.3283064365386963e-10f);
            e = fminf(fmaxf(e, -0.4999f), 0.49999997f);      // (|e| < 0.5 keeps log1p(-2|e|) finite, as the reference's interval does)
        }
        const float sgn = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
        z[idx] = lat[n * 2 * L + l] - expf(lat[n * 2 * L + L + l]) * sgn * log1pf(-2.0f * fabsf(e));
        if (eps_out) eps_out[idx] = e;
    }
}
// its gradient: d mu = dz, d log_scale = dz * (z - mu)
__global__ void k_sample_laplace_bwd(const float* dz, const float* lat, const float* z, int rows, int L, float* dlat) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)rows * L) {
        const int l = (int)(idx % L);
        const long n = idx / L;
        const float g = dz[idx];
        dlat[n * 2 * L + l] = g;
        dlat[n * 2 * L + L + l] = g * (z[idx] - lat[n * 2 * L + l]);
    }
}

// C[m][n] = sum_k A[m][k] * Bm[n][k] + bias[n]   (both operands k-contiguous, K a multiple of 16).
// f32 MFMA 16x16x4; each wave owns a (16*TM) x (16*TN) tile and loads its A/B fragments straight from
// global memory as 16-byte pieces (lane: row lane&15, k-quad lane>>4), 4 MFMAs per loaded quad.
// A_CHUNKED: A is chunk-major [K/16][a_plane rows][16] instead of row-major with leading dimension lda.
template <int TM, int TN, int WGM, int WGN, bool A_CHUNKED>
__global__ __launch_bounds__(64 * WGM * WGN) void k_gemm_nt(const float* __restrict__ A, long lda, long a_plane,
                                                            const float* __restrict__ Bm, long ldb,
                                                            const float* __restrict__ bias, float* __restrict__ C,
                                                            long ldc, int M, int N, int K) {
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
        arow[i] = A_CHUNKED ? (long)r * 16 + 4 * kq : (long)r * lda + 4 * kq;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int c = n0 + 16 * j + lr;
        c = c < N ? c : N - 1;
        brow[j] = (long)c * ldb + 4 * kq;
    }
    for (int k0 = 0; k0 < K; k0 += 16) {
        float4 a[TM], b[TN];
        const long aoff = A_CHUNKED ? (long)(k0 >> 4) * a_plane * 16 : (long)k0;
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
                if (rowi < M && col < N) C[(long)rowi * ldc + col] = acc[i][j][r] + bv;
            }
        }
}

// transposing copy: dst[c*rows + r] = src[r*sld + c]
__global__ void k_copy2d_t(float* dst, const float* src, long sld, int rows, int cols) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)rows * cols) {
        const int r = (int)(idx % rows), c = (int)(idx / rows);
        dst[idx] = src[(long)r * sld + c];
    }
}

struct StepParams {
    float* hbuf;
    long mtot;          // rows per chunk plane = (T+1)*Bp
    const float* wrec;
    const float* gx;    // [B][Tp][3H]
    long gx_bstride;    // Tp*3H
    const float* bhn;   // b_hh[2H:]
    int B, Bp, H, T, t0;
    unsigned* bar;      // grid-barrier counter (zeroed before every launch)
    int* status;        // status[0] = 1 on barrier timeout
    unsigned nwg;
    long long* prof;    // null, or [gridDim.x][4] cycle sums: loads+MFMA, reduce+gates+store, drain, barrier wait
    const float* wyT;   // W_ih[:, R*C:] transposed [Co][3H]
    const float* dy;    // [rows][Co] frame-0 feedback correction, or null when gx already carries it
    int Co;
};

// gx[b,0,:] += W_ih[:, R*C:] . dy[b]  for hidden unit j (the three gates), see k_prologue.
// Eight feedback channels per trip with all 32 loads issued before the first use: written as a plain loop the compiler waited
// for every channel's loads in turn (unknown trip count, reference arguments), one memory round trip per channel -- 34 us in
// front of the first step of EVERY launch of the recurrent kernels (tools/launch_timing.py: first task 85.7K ticks against 11.2K
// for a steady one).  The sums run in the same order as before (channel ascending per gate): same bits.
// BW: channels per trip (4 BW registers of loads in flight; the register-bound fp32 kernel takes 2).
template <int BW = 8>
__device__ __forceinline__ void cvae_t0_fix(const float* wyT, const float* dy, int Co, int H, int j, int grow, float& gr_,
                                            float& gz_, float& gn_) {
    const float* d = dy + (long)grow * Co;
    float gr = gr_, gz = gz_, gn = gn_;
    int c = 0;
    for (; c + BW <= Co; c += BW) {
        float wr[BW], wz[BW], wn[BW], dd[BW];
#pragma unroll
        for (int q = 0; q < BW; ++q) {
            const float* w = wyT + (long)(c + q) * 3 * H + j;
            wr[q] = w[0];
            wz[q] = w[H];
            wn[q] = w[2 * H];
            dd[q] = d[c + q];
        }
#pragma unroll
        for (int q = 0; q < BW; ++q) {
            gr += wr[q] * dd[q];
            gz += wz[q] * dd[q];
            gn += wn[q] * dd[q];
        }
    }
    for (; c < Co; ++c) {
        const float* w = wyT + (long)c * 3 * H + j;
        gr += w[0] * d[c];
        gz += w[H] * d[c];
        gn += w[2 * H] * d[c];
    }
    gr_ = gr; gz_ = gz; gn_ = gn;
}

// Whole-grid barrier on one monotonic counter: every wave drains its stores, lane 0 releases at agent scope,
// arrives, polls relaxed, then one agent acquire (MI355X_MICROARCH price list "barrier-counter").  Spins are
// bounded: on timeout status[0] is raised and the kernel runs to completion with garbage instead of hanging.
__device__ __forceinline__ void cvae_grid_barrier(unsigned* bar, unsigned target, int* status) {
    cvae_drain_vmem();
    __syncthreads();
    if (threadIdx.x == 0) {
        cvae_release_agent();
        cvae_atomic_add_agent(bar, 1u);
        unsigned spins = 0;
        while (cvae_atomic_load_agent(bar) < target) {
            cvae_sleep();
            if (++spins > (1u << 22)) {
                status[0] = 1;
                break;
            }
        }
        cvae_acquire_agent();
    }
    __syncthreads();
}

__device__ __forceinline__ float cvae_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// The autoregressive GRU recurrence (gru_vae.py:364-394, eval).  Block g owns hidden units 4g..4g+3, i.e.
// one 16-column MFMA tile (r, z, n_in, n_h of four units); its 4 waves split K = H and reduce through LDS.
// PERSIST: all T steps in one cooperative launch (grid barrier between steps); else only step p.t0.
template <bool PERSIST>
__global__ __launch_bounds__(256) void k_gru_steps(StepParams p) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int g = blockIdx.x, H = p.H, nch = H >> 4;
    const int c_lo = (nch * wave) >> 2, c_hi = (nch * (wave + 1)) >> 2;
    float* red = (float*)CVAE_SMEM;  // [4 waves][64 rows][20]
    const float* wg = p.wrec + (long)g * nch * 256 + lr * 16 + kq * 4;
    const int nrt = p.Bp >> 4;
    const int t_begin = PERSIST ? 0 : p.t0, t_end = PERSIST ? p.T : p.t0 + 1;
    const int row = tid >> 2, u = tid & 3, j = 4 * g + u;
    const long hcol = (long)(g >> 2) * p.mtot * 16 + (g & 3) * 4 + u;  // this thread's unit inside hbuf
    for (int t = t_begin; t < t_end; ++t) {
        const float* hprev = p.hbuf + (long)t * p.Bp * 16;
        float* hnext = p.hbuf + (long)(t + 1) * p.Bp * 16;
        for (int rt0 = 0; rt0 < nrt; rt0 += 4) {
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int c = c_lo; c < c_hi; ++c) {
                const float4 b4 = *(const float4*)(wg + (long)c * 256);
                const float* hc = hprev + (long)c * p.mtot * 16 + lr * 16 + kq * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (rt0 + i < nrt) {
                        const float4 a4 = *(const float4*)(hc + (long)(rt0 + i) * 256);
                        acc[i] = cvae_mfma_16x16x4(a4.x, b4.x, acc[i]);
                        acc[i] = cvae_mfma_16x16x4(a4.y, b4.y, acc[i]);
                        acc[i] = cvae_mfma_16x16x4(a4.z, b4.z, acc[i]);
                        acc[i] = cvae_mfma_16x16x4(a4.w, b4.w, acc[i]);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * 64 + i * 16 + kq * 4 + r) * 20 + lr] = acc[i][r];
            __syncthreads();
            const int grow = rt0 * 16 + row;
            if (grow < p.Bp) {
                float s[4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    s[a] = red[(0 * 64 + row) * 20 + a * 4 + u] + red[(1 * 64 + row) * 20 + a * 4 + u] +
                           red[(2 * 64 + row) * 20 + a * 4 + u] + red[(3 * 64 + row) * 20 + a * 4 + u];
                float hn = 0.0f;
                if (grow < p.B) {
                    const float* gxp = p.gx + (long)grow * p.gx_bstride + (long)t * 3 * H;
                    float g0 = gxp[j], g1 = gxp[H + j], g2 = gxp[2 * H + j];
                    if (t == 0 && p.dy) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, g0, g1, g2);
                    const float r = cvae_sigmoid(g0 + s[0]);
                    const float z = cvae_sigmoid(g1 + s[1]);
                    const float n = tanhf(g2 + s[2] + r * (s[3] + p.bhn[j]));
                    const float hold = hprev[hcol + (long)grow * 16];
                    hn = n + z * (hold - n);
                }
                hnext[hcol + (long)grow * 16] = hn;
            }
            __syncthreads();
        }
        if (PERSIST && t + 1 < t_end) cvae_grid_barrier(p.bar, (unsigned)(t + 1) * p.nwg, p.status);
    }
}

// fast gate nonlinearities on the hardware exp (v_exp_f32): ~1e-6 relative, three orders inside the MCD budget
__device__ __forceinline__ float cvae_sigmoid_fast(float x) { return cvae_fast_rcp(1.0f + cvae_fast_exp(-x)); }
__device__ __forceinline__ float cvae_tanh_fast(float x) {
    const float e = cvae_fast_exp(-2.0f * fabsf(x));
    const float t = (1.0f - e) * cvae_fast_rcp(1.0f + e);
    return x < 0.0f ? -t : t;
}

struct Step2Params {
    float* hbuf;        // chunk-major [H/16][mtot][16]
    long mtot;
    const float* wrec2; // [H/16][4][H/16][16][16]
    const float* gx;    // [B][Tp][3H]
    long gx_bstride;
    const float* bhn;
    int B, Bp, H, T;
    unsigned* flags;    // [Bp/16 row tiles][H/16 chunks], zeroed before launch: flags[i][c] = t  <=>  h_t chunk c of tile i published
    int* status;
    long long* prof;    // null or [blocks][4] cycle sums: wait, loads+MFMA, reduce+gates+publish, (unused)
    const float* wyT;   // see StepParams
    const float* dy;
    int Co;
};

// Persistent recurrence, 2-D decomposition for H = 64*CPW.  Block (j, i0): hidden units 16j..16j+15 (h chunk j, 64 MFMA
// columns: r, z, n_in, n_h tiles) x row tiles i0, i0+gridDim.y, ...  Its weights (4*CPW float4 per lane per wave) stay in
// registers for the whole launch; per step and row tile a block reads only that tile's 16 rows of h (64 KiB), not the
// whole batch.  Row tiles are independent recurrences: there is no grid barrier, a wave waits only for the flags of the
// 16*... chunks it is about to load (written by the 64 blocks of the same row tile), so the publish->visible latency of
// one row tile hides behind the MFMAs of the block's other row tiles (two stacked decoder passes, or B > 64).
template <int CPW>
__global__ __launch_bounds__(256, 1) void k_gru_steps_v2(Step2Params p) {
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int jg = blockIdx.x, H = p.H, nch = 4 * CPW, nrt = p.Bp >> 4;
    const int c_lo = wave * CPW;
    float* red = (float*)CVAE_SMEM;       // [4 waves][16 rows][84]
    float* hsh = red + 4 * 16 * 84;       // [16 rows][16 units]
    const int row = tid >> 4, u = tid & 15, j = 16 * jg + u;
    const unsigned mtot = (unsigned)p.mtot;
    const cvae_buf hb = cvae_make_buf(p.hbuf, (unsigned)((long)nch * p.mtot * 64));
    const unsigned voff = (unsigned)(lr * 16 + kq * 4) * 4u;
    f32x4 w[4][CPW];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci)
            w[a][ci] = *(const f32x4*)(p.wrec2 + (((long)jg * 4 + a) * nch + c_lo + ci) * 256 + lr * 16 + kq * 4);
    const float bhn = p.bhn[j];
    long long pc[3] = {0, 0, 0};
    for (int t = 0; t < p.T; ++t) {
        for (int i = blockIdx.y; i < nrt; i += gridDim.y) {
            long long c0 = p.prof ? cvae_clock() : 0;
            // wait until h_t chunks [c_lo, c_lo+CPW) of row tile i are published (slot 0 comes from k_hinit)
            if (t > 0) {
                unsigned spins = 0;
                for (;;) {
                    unsigned f = (unsigned)t;
                    if (lane < CPW) f = cvae_atomic_load_agent(p.flags + (long)i * nch + c_lo + lane);
                    if (cvae_wave_all(f >= (unsigned)t)) break;
                    cvae_sleep();
                    if (++spins > (1u << 22)) {
                        p.status[0] = 2;
                        break;
                    }
                }
            }
            cvae_compiler_fence();   // operand loads must stay below the flag poll
            if (p.prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
            const unsigned row0 = (unsigned)(t * p.Bp + i * 16);
            f32x4 a4[CPW];
#pragma unroll
            for (int ci = 0; ci < CPW; ++ci)
                a4[ci] = cvae_buf_load_f4_sc1(hb, voff, ((unsigned)(c_lo + ci) * mtot + row0) * 64u);
            const int grow = i * 16 + row;
            const bool live = grow < p.B;
            float gxr = 0.f, gxz = 0.f, gxn = 0.f, hold = 0.f;
            if (live) {
                const float* gxp = p.gx + (long)grow * p.gx_bstride + (long)t * 3 * H;
                gxr = gxp[j];
                gxz = gxp[H + j];
                gxn = gxp[2 * H + j];
                if (t == 0 && p.dy) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, gxr, gxz, gxn);
                hold = cvae_buf_load_f1_sc1(hb, (unsigned)(u * 4), ((unsigned)jg * mtot + row0 + (unsigned)row) * 64u);
            }
            f32x4 acc[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ci = 0; ci < CPW; ++ci)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int a = 0; a < 4; ++a) acc[a] = cvae_mfma_16x16x4(a4[ci][q], w[a][ci][q], acc[a]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[(wave * 16 + kq * 4 + q) * 84 + a * 16 + lr] = acc[a][q];
            if (p.prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
            __syncthreads();
            {
                float hn = 0.0f;
                if (live) {
                    float s[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a)
                        s[a] = red[(0 * 16 + row) * 84 + a * 16 + u] + red[(1 * 16 + row) * 84 + a * 16 + u] +
                               red[(2 * 16 + row) * 84 + a * 16 + u] + red[(3 * 16 + row) * 84 + a * 16 + u];
                    const float rg = cvae_sigmoid_fast(gxr + s[0]);
                    const float zg = cvae_sigmoid_fast(gxz + s[1]);
                    const float ng = cvae_tanh_fast(gxn + s[2] + rg * (s[3] + bhn));
                    hn = ng + zg * (hold - ng);
                }
                hsh[row * 16 + u] = hn;
            }
            __syncthreads();
            if (tid < 64) {   // wave 0: 16 rows x 64 B = one contiguous 1 KiB block of chunk jg, slot t+1
                const f32x4 v = *(const f32x4*)(hsh + tid * 4);
                cvae_buf_store_f4_sc1(hb, (unsigned)tid * 16u, ((unsigned)jg * mtot + row0 + (unsigned)p.Bp) * 64u, v);
                cvae_drain_vmem();      // every lane's write-through store has left ...
                cvae_wave_barrier();    // ... (all 64 lanes are this one wave) before lane 0 raises the flag
                if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * nch + jg, (unsigned)(t + 1));
            }
            if (p.prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        }
    }
    if (p.prof && tid == 0)
        for (int q = 0; q < 3; ++q) p.prof[((long)blockIdx.y * gridDim.x + jg) * 4 + q] = pc[q];
}

struct Step3Params {
    float* hbuf;         // chunk-major [H/16][mtot][16]
    long mtot;
    const float* wrec2;  // [H/16][4][H/16][16][16]
    const float* afold2; // the kernel's front-end weight image: afold3 (v4) or afold_h (v5)
    const float* cfold;  // [3H]
    const float* xnp;    // [rows][Tp][Cp] normalised, padded input
    int Tp, Cp;
    const float* bhn;
    int B, Bp, H, T;
    unsigned* flags;     // [Bp/16][H/16], zeroed before launch
    int* status;
    long long* prof;     // null or [blocks][4] cycle sums: front-end, flag wait, loads+MFMA, reduce+gates+publish
    const float* wyT;
    const float* dy;
    int Co;
    int rts;             // row tiles handled concurrently by the grid (grid = H/16 * rts blocks)
    float* hs;           // v5 only: exchanged state as fp16 pairs, [H/16][mtot][16 hi halves | 16 lo halves] (64 B per row)
    const float* xs;     // v5 only: xnp as fp16 pairs (hi plane, lo plane of xs_plane halves each)
    long xs_plane;
    const float* wrec_h; // v5 only: recurrent weights as packed fp16 pairs, [H/16][4][H/32][hi, lo][64 lanes][8 halves]
    int exp;             // measurement-only switches: bits 2-3 pick the wave that reports the phase counters, bits 8.. override the poll back-off
};

// afold3[jg][wave][ci][a][lane][4] = the folded front-end weights (afold [3H][Kfe]) pre-arranged as the exact LDS image of
// k_gru_steps_v4: lane L (lr = L & 15, kq = L >> 4) of `wave` finds its B-fragment for (chunk wave*KFW+ci, gate a) at L*16 B.
__global__ void k_prep_afold3(const float* afold, float* afold3, int H, int Kfe, int KFW) {
    const int nch = H >> 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)nch * 4 * KFW * 3 * 256) {
        const int q = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
        const int a = (int)((idx >> 8) % 3), ci = (int)(((idx >> 8) / 3) % KFW);
        const int wave = (int)(((idx >> 8) / 3 / KFW) & 3), jg = (int)((idx >> 8) / 3 / KFW / 4);
        const int lr = lane & 15, kq = lane >> 4;
        const int k = 16 * (wave * KFW + ci) + 4 * kq + q;
        afold3[idx] = k < Kfe ? afold[(long)(a * H + 16 * jg + lr) * Kfe + k] : 0.0f;
    }
}

// k_gru_steps_v2 with the front-end inside the step: the folded conv0*conv1*W_ih product for frame t,
// A_fold[48 cols of this block] . xnp[b, t:t+R, :], has no dependence on h, so its MFMAs (3 tiles x KFW chunks per wave)
// are issued BEFORE the wave starts polling for h_t and accumulate into the same r / z / n_in accumulators: the hoisted
// [B*T, R*C] x [R*C, 3H] GEMM and its gx buffer disappear, the work lands in what used to be hand-off wait.  (a) The
// front-end weights sit in LDS (lane-linear image: conflict-free ds_read_b128), which leaves registers for (b) TWO
// h-operand sets: while task k's MFMAs run, task k+1's operand tiles
// are already in flight whenever its flags are up (always the case with >= 2 independent row tiles per block: stacked
// decoder passes, B > 64); with one tile per block the next task is the next time step and the kernel falls back to
// "publish, then poll".
template <int CPW, int KFW>
__global__ __launch_bounds__(256, 1) void k_gru_steps_v4(Step3Params p) {
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, nch = 4 * CPW, nrt = p.Bp >> 4;
    const int rts = p.rts, jg = blockIdx.x % nch, ti = blockIdx.x / nch;
    const int c_lo = wave * CPW;
    float* red = (float*)CVAE_SMEM;                    // [4 waves][16 rows][84]
    float* hsh = red + 4 * 16 * 84;                    // [16 rows][16 units]
    float* wfl = hsh + 16 * 16;                        // [4 waves][KFW][3][64 lanes][4]
    const int row = tid >> 4, u = tid & 15, j = 16 * jg + u;
    const unsigned mtot = (unsigned)p.mtot;
    const cvae_buf hb = cvae_make_buf(p.hbuf, (unsigned)((long)nch * p.mtot * 64));
    const unsigned voff = (unsigned)(lr * 16 + kq * 4) * 4u;
    f32x4 w[4][CPW];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci)
            w[a][ci] = *(const f32x4*)(p.wrec2 + (((long)jg * 4 + a) * nch + c_lo + ci) * 256 + lr * 16 + kq * 4);
    {   // this wave's slice of the front-end weights -> LDS (straight copy of the prepared image)
        const float* src = p.afold2 + ((long)jg * 4 + wave) * (KFW * 3 * 256);
        float* dst = wfl + wave * (KFW * 3 * 256);
#pragma unroll
        for (int e = 0; e < KFW * 3; ++e) *(f32x4*)(dst + e * 256 + lane * 4) = *(const f32x4*)(src + e * 256 + lane * 4);
    }
    __syncthreads();
    const float* wfw = wfl + wave * (KFW * 3 * 256) + lane * 4;
    const float bhn = p.bhn[j];
    const float cf0 = p.cfold[j], cf1 = p.cfold[H + j], cf2 = p.cfold[2 * H + j];
    const int ntile = ti < nrt ? (nrt - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    long long pc[4] = {0, 0, 0, 0};

    f32x4 x4[KFW], hA[CPW], hB[CPW];
    auto load_x = [&](int k) {
        const int tt = k / ntile, ii = ti + (k % ntile) * rts;
        int xb = ii * 16 + lr;
        xb = xb < p.B ? xb : p.B - 1;
        const float* xrow = p.xnp + ((long)xb * p.Tp + tt) * p.Cp + (wave * KFW) * 16 + kq * 4;
#pragma unroll
        for (int ci = 0; ci < KFW; ++ci) x4[ci] = *(const f32x4*)(xrow + ci * 16);
    };
    // flags of the chunks this wave needs for task k are all up?  (one relaxed load per lane < CPW, wave vote)
    auto flags_up = [&](int k) -> bool {
        const int tt = k / ntile, ii = ti + (k % ntile) * rts;
        if (tt == 0) return true;                      // slot 0 comes from the prologue kernel
        unsigned f = (unsigned)tt;
        if (lane < CPW) f = cvae_atomic_load_agent(p.flags + (long)ii * nch + c_lo + lane);
        return cvae_wave_all(f >= (unsigned)tt);
    };
    auto wait_flags = [&](int k) {
        unsigned spins = 0;
        while (!flags_up(k)) {
            cvae_sleep();
            if (++spins > (1u << 22)) {
                p.status[0] = 2;
                break;
            }
        }
        cvae_compiler_fence();                         // operand loads stay below the poll
    };
    auto load_h = [&](int k, f32x4 (&h)[CPW]) {
        const int tt = k / ntile, ii = ti + (k % ntile) * rts;
        const unsigned row0 = (unsigned)(tt * p.Bp + ii * 16);
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) h[ci] = cvae_buf_load_f4_sc1(hb, voff, ((unsigned)(c_lo + ci) * mtot + row0) * 64u);
    };
    // one task: front-end MFMAs, (operands of this task if not requested yet), early request of the next task's operands
    // when its flags are already up, recurrent MFMAs, gates, publish.  Returns whether the next task's operands are in flight.
    // h_{t-1} of this thread's (row, unit): produced by this very thread one step earlier, so it is carried in a register
    // (slot k % ntile for ntile <= 2) instead of being re-loaded -- a load issued here would sit in front of the MFMAs,
    // whose first s_waitcnt after the loop back-edge is a conservative vmcnt(0).
    float hkeep0 = 0.f, hkeep1 = 0.f;
    auto task = [&](int k, f32x4 (&hc)[CPW], f32x4 (&hn)[CPW], bool have) -> bool {
        long long c0 = p.prof ? cvae_clock() : 0;
        const int t = k / ntile, i = ti + (k % ntile) * rts;
        const unsigned row0 = (unsigned)(t * p.Bp + i * 16);
        f32x4 acc[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ci = 0; ci < KFW; ++ci) {
            f32x4 wf[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) wf[a] = *(const f32x4*)(wfw + (ci * 3 + a) * 256);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int a = 0; a < 3; ++a) acc[a] = cvae_mfma_16x16x4(x4[ci][q], wf[a][q], acc[a]);
        }
        if (p.prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
        if (!have) {            // single tile per block (next task = next time step), or the early request missed
            wait_flags(k);
            load_h(k, hc);
        }
        if (p.prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        const int grow = i * 16 + row;
        const bool live = grow < p.B;
        const bool keep1 = ntile == 2 && (k & 1);
        float gxr = cf0, gxz = cf1, gxn = cf2, hold = keep1 ? hkeep1 : hkeep0;
        if (live) {
            if (t == 0 && p.dy) cvae_t0_fix<2>(p.wyT, p.dy, p.Co, H, j, grow, gxr, gxz, gxn);
            if (t == 0 || ntile > 2)
                hold = cvae_buf_load_f1_sc1(hb, (unsigned)(row * 64 + u * 4), ((unsigned)jg * mtot + row0) * 64u);   // uniform soff
        }
        // Probe for the next task's operands while this task's MFMAs run: with two tiles per block the next task's
        // producers published only one task ago, so the flag load goes out at the half-way point and is looked at when the
        // MFMAs are done; the gates and the next front-end cover the operand round trip.
        const bool probe = k + 1 < ntask && ntile > 1;
        const int kn = k + 1, tn = kn / ntile, in_ = ti + (kn % ntile) * rts;
        unsigned fprobe = 0u;
        bool next_issued = false;
#pragma unroll
        for (int ci = 0; ci < CPW; ++ci) {
            if (ci == CPW / 2) {   // half-way: next task's front-end operands and (several tiles per block) its flags
                cvae_sched_fence();
                if (probe) load_x(k + 1);   // must have drained before wave 0's publish (see below)
                if (probe) {
                    fprobe = (unsigned)tn;
                    if (tn > 0 && lane < CPW) fprobe = cvae_atomic_load_agent(p.flags + (long)in_ * nch + c_lo + lane);
                }
                cvae_sched_fence();
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = cvae_mfma_16x16x4(hc[ci][q], w[a][ci][q], acc[a]);
        }
        // The flag load has had half of the MFMA phase to return.  Waves 1-3 request the next operands right away; wave 0
        // must publish first: its s_waitcnt vmcnt(0) in front of the flag store would otherwise also wait for these loads
        // (vmcnt cannot tell them from the write-through stores) and delay every consumer of this block's h.
        if (!probe && k + 1 < ntask) load_x(k + 1);   // one tile per block: lands under reduce + gates + publish
        const bool probe_hit = probe && cvae_wave_all(fprobe >= (unsigned)tn);
        if (probe_hit && wave != 0) {
            cvae_compiler_fence();
            load_h(kn, hn);
            next_issued = true;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[(wave * 16 + kq * 4 + q) * 84 + a * 16 + lr] = acc[a][q];
        if (p.prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        __syncthreads();
        {
            float hn_ = 0.0f;
            if (live) {
                float s[4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    s[a] = red[(0 * 16 + row) * 84 + a * 16 + u] + red[(1 * 16 + row) * 84 + a * 16 + u] +
                           red[(2 * 16 + row) * 84 + a * 16 + u] + red[(3 * 16 + row) * 84 + a * 16 + u];
                const float rg = cvae_sigmoid_fast(gxr + s[0]);
                const float zg = cvae_sigmoid_fast(gxz + s[1]);
                const float ng = cvae_tanh_fast(gxn + s[2] + rg * (s[3] + bhn));
                hn_ = ng + zg * (hold - ng);
            }
            if (keep1) hkeep1 = hn_; else hkeep0 = hn_;
            hsh[row * 16 + u] = hn_;
        }
        __syncthreads();
        if (tid < 64) {   // wave 0: 16 rows x 64 B = one contiguous 1 KiB block of chunk jg, slot t+1
            const f32x4 v = *(const f32x4*)(hsh + tid * 4);
            cvae_buf_store_f4_sc1(hb, (unsigned)tid * 16u, ((unsigned)jg * mtot + row0 + (unsigned)p.Bp) * 64u, v);
            cvae_drain_vmem();      // every lane's write-through store has left ...
            cvae_wave_barrier();    // ... (all 64 lanes are this one wave) before lane 0 raises the flag
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * nch + jg, (unsigned)(t + 1));
        }
        if (probe_hit && wave == 0) {
            cvae_compiler_fence();
            load_h(kn, hn);
            next_issued = true;
            if (p.prof && tid == 0) cvae_atomic_add_agent((unsigned*)p.status + 1, 1u);   // diagnostics: early requests
        }
        if (p.prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
        return next_issued;
    };

    if (ntask > 0) {
        load_x(0);
        bool have = false;
        int k = 0;
        for (; k + 2 <= ntask; k += 2) {
            have = task(k, hA, hB, have);
            have = task(k + 1, hB, hA, have);
        }
        if (k < ntask) task(k, hA, hB, have);
    }
    if (p.prof && tid == 64 * ((p.exp >> 2) & 3))    // measurement: exp bits 2-3 pick the reporting wave
        for (int q = 0; q < 4; ++q) p.prof[(long)blockIdx.x * 4 + q] = pc[q];
}

// wrec_h[jg][a][c32][hl][lane][e]: the recurrent weights of wrec2 (B[k][col], col = tile a / unit lane&15) as fp16 pairs in
// the operand order of v_mfma_f32_16x16x32_f16: lane (j = lane & 15, kq = lane >> 4) holds k = 32*c32 + 8*kq + e, e = 0..7;
// hl = 0: hi halves, hl = 1: lo halves (x = hi + lo/2048, cvae_split_f16).
__global__ void k_prep_wrec_h(const float* wrec2, float* wrec_h, int H) {
    const int nch = H >> 4, n32 = H >> 5;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (jg, a, c32, lane, e)
    if (idx < (long)nch * 4 * n32 * 64 * 8) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int c32 = (int)((idx >> 9) % n32), a = (int)(((idx >> 9) / n32) & 3), jg = (int)((idx >> 9) / n32 / 4);
        const int lr = lane & 15, kq = lane >> 4, k = 32 * c32 + 8 * kq + e;
        const float w = wrec2[(((long)jg * 4 + a) * nch + (k >> 4)) * 256 + lr * 16 + (k & 15)];
        unsigned short hi, lo;
        cvae_split_f16(w, hi, lo);
        unsigned short* dst = (unsigned short*)wrec_h + ((((long)jg * 4 + a) * n32 + c32) * 2) * 512 + lane * 8 + e;
        dst[0] = hi;
        dst[512] = lo;
    }
}

// afold_h[jg][wave][cf][a][hl][lane][e]: the folded front-end weights (afold [3H][Kfe]) as fp16 pairs in the operand order of
// v_mfma_f32_16x16x32_f16 and as the exact LDS image of k_gru_steps_v5: wave w owns the 32-k chunks NF32*w .. NF32*w + NF32-1,
// lane (col = lane & 15 -> unit 16*jg + col of gate a, kq = lane >> 4) holds k = 32*chunk + 8*kq + e; zero beyond Kfe.
__global__ void k_prep_afold_h(const float* afold, float* afold_h, int H, int Kfe, int NF32) {
    const int nch = H >> 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (jg, wave, cf, a, lane, e)
    if (idx < (long)nch * 4 * NF32 * 3 * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int a = (int)((idx >> 9) % 3), cf = (int)(((idx >> 9) / 3) % NF32);
        const int wave = (int)(((idx >> 9) / 3 / NF32) & 3), jg = (int)((idx >> 9) / 3 / NF32 / 4);
        const int lr = lane & 15, kq = lane >> 4, k = 32 * (wave * NF32 + cf) + 8 * kq + e;
        const float w = k < Kfe ? afold[(long)(a * H + 16 * jg + lr) * Kfe + k] : 0.0f;
        unsigned short hi, lo;
        cvae_split_f16(w, hi, lo);
        unsigned short* dst = (unsigned short*)afold_h + (((((long)jg * 4 + wave) * NF32 + cf) * 3 + a) * 2) * 512 + lane * 8 + e;
        dst[0] = hi;
        dst[512] = lo;
    }
}

// k_gru_steps_v4 with the recurrent product in SPLIT fp16: both the weights and the exchanged state are kept as pairs
// (hi, lo) of halves with x = hi + lo/2048 (22 significant bits), and W.h = hi.hi + (hi.lo + lo.hi)/2048 runs as three
// v_mfma_f32_16x16x32_f16 per 32 k (fp32 accumulation; the dropped lo.lo term is 2^-22 of the product).  Per wave and step that
// is 96 instructions of 16 matrix-pipe cycles instead of 256 of 32.  Register budget is unchanged (a packed pair is 32
// bits); the exchanged row keeps its 64 bytes ([16 hi | 16 lo] halves per 16-unit chunk), so publish and operand loads move
// the same bytes as in v4.  The front-end runs in the same three-product form on fp16 pairs of the normalised input (xs) and of
// the folded weights (afold_h, in LDS); gate math and the carried h are fp32; the fp32 h is still written (chunk-major hbuf)
// for the projection kernel.
// Compared with v4 there is ONE operand set and no probing for the next task: the registers go to three independent
// accumulator sets (hi.hi, hi.lo, lo.hi: twelve accumulation chains, no MFMA waits for the one issued before it), which is
// worth more now that the MFMA phase is short.  Row tiles of a block are processed one after the other with the same
// arithmetic, so a row's result does not depend on how many tiles share its block (bitwise row independence).
// NC32 = 32-k chunks per wave (H/128; H = 64: one chunk on waves 0 and 1).
template <int CPW, int KFW>
__global__ __launch_bounds__(256, 1) void k_gru_steps_v5(Step3Params p) {
    constexpr int NC32 = CPW >= 2 ? CPW / 2 : 1;
    constexpr int NF32 = (KFW + 1) / 2;                // 32-k chunks of the front-end per wave
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, nch = 4 * CPW, n32 = H >> 5, nrt = p.Bp >> 4;
    const int rts = p.rts, jg = blockIdx.x % nch, ti = blockIdx.x / nch;
    const bool has_k = wave * NC32 < n32;                  // (H = 64: waves 2, 3 have no K share)
    const int c32_lo = has_k ? wave * NC32 : 0;
    float* red = (float*)CVAE_SMEM;                    // [4 waves][16 rows][84]
    float* hsh = red + 4 * 16 * 84;                    // [16 rows][16 units]
    float* wfl = hsh + 16 * 16;                        // [4 waves][NF32][3][hi, lo][64 lanes][8 halves]
    const int row = tid >> 4, u = tid & 15, j = 16 * jg + u;
    const unsigned mtot = (unsigned)p.mtot;
    const cvae_buf hb = cvae_make_buf(p.hbuf, (unsigned)((long)nch * p.mtot * 64));
    const cvae_buf sb = cvae_make_buf(p.hs, (unsigned)((long)nch * p.mtot * 64));
    // operand of 32-k chunk c32, lane (lr, kq): units 32*c32 + 8*kq .. +7 = 16-unit chunk 2*c32 + (kq >> 1), halves (kq & 1)*8 .. +7
    const unsigned voff = ((unsigned)(kq >> 1) * mtot + (unsigned)lr) * 64u + (unsigned)(kq & 1) * 16u;
    f32x4 wh[4][NC32], wl[4][NC32];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int ci = 0; ci < NC32; ++ci) {
            const float* src = p.wrec_h + ((((long)jg * 4 + a) * n32 + c32_lo + ci) * 2) * 256 + lane * 4;
            wh[a][ci] = *(const f32x4*)src;
            wl[a][ci] = *(const f32x4*)(src + 256);
        }
    {   // this wave's slice of the front-end weight pairs -> LDS (straight copy of the prepared image)
        const float* src = p.afold2 + ((long)jg * 4 + wave) * (NF32 * 6 * 256);
        float* dst = wfl + wave * (NF32 * 6 * 256);
#pragma unroll
        for (int e = 0; e < NF32 * 6; ++e) *(f32x4*)(dst + e * 256 + lane * 4) = *(const f32x4*)(src + e * 256 + lane * 4);
    }
    __syncthreads();
    const float* wfw = wfl + wave * (NF32 * 6 * 256) + lane * 4;
    const float bhn = p.bhn[j];
    const float cf0 = p.cfold[j], cf1 = p.cfold[H + j], cf2 = p.cfold[2 * H + j];
    const int ntile = ti < nrt ? (nrt - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    long long pc[4] = {0, 0, 0, 0};
    f32x4 x4[2 * NF32];                 // front-end operands: [2*cf] hi halves, [2*cf + 1] lo halves of 32-k chunk cf
    auto load_x = [&](int k) {          // ... of task k (rows of its tile, window t..t+R-1: contiguous in the pair planes)
        const int tt = k / ntile, ii = ti + (k % ntile) * rts;
        int xb = ii * 16 + lr;
        xb = xb < p.B ? xb : p.B - 1;
        const unsigned short* xrow = (const unsigned short*)p.xs + ((long)xb * p.Tp + tt) * p.Cp + (wave * NF32) * 32 + kq * 8;
#pragma unroll
        for (int cf = 0; cf < NF32; ++cf) {
            x4[2 * cf] = *(const f32x4*)(xrow + cf * 32);
            x4[2 * cf + 1] = *(const f32x4*)(xrow + p.xs_plane + cf * 32);
        }
    };
    // h_{t-1} of this thread's (row, unit): produced by this very thread one step earlier, carried in a register per tile
    // (up to two tiles per block); with more tiles it is re-read from the pair buffer
    float hkeep0 = 0.f, hkeep1 = 0.f;
    const int backoff = (p.exp >> 8) ? (p.exp >> 8) - 1 : 20;     // x 64 cycles; swept 0..64 on MI355X (measurement override: exp bits 8..)
    if (ntask > 0) load_x(0);
    unsigned fpre = 0u;                                // flags of the NEXT task, read at the end of the current one (several tiles per block)
    for (int k = 0; k < ntask; ++k) {
        long long c0 = p.prof ? cvae_clock() : 0;
        const int t = k / ntile, i = ti + (k % ntile) * rts;
        const unsigned row0 = (unsigned)(t * p.Bp + i * 16);
        f32x4 acc[4], accx[4], accy[4];                // front-end + hi.hi | hi.lo | lo.hi (the last two on the x 2048 scale)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            acc[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accy[a] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int cf = 0; cf < NF32; ++cf) {     // front-end, same three-product form as the recurrent part below
            f32x4 wfh[3], wfl_[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                wfh[a] = *(const f32x4*)(wfw + ((cf * 3 + a) * 2) * 256);
                wfl_[a] = *(const f32x4*)(wfw + ((cf * 3 + a) * 2 + 1) * 256);
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) acc[a] = cvae_mfma_16x16x32_f16(x4[2 * cf], wfh[a], acc[a]);
#pragma unroll
            for (int a = 0; a < 3; ++a) accx[a] = cvae_mfma_16x16x32_f16(x4[2 * cf], wfl_[a], accx[a]);
#pragma unroll
            for (int a = 0; a < 3; ++a) accy[a] = cvae_mfma_16x16x32_f16(x4[2 * cf + 1], wfh[a], accy[a]);
        }
        if (p.prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
        // Several tiles per block: this task's producers published a whole task ago, and the flag read issued at the end of
        // the previous task has returned under the front-end: no poll round trip when it shows them up.
        const bool pre_ok = ntile > 1 && k > 0 && cvae_wave_all(fpre >= (unsigned)t);
        if (t > 0 && has_k && !pre_ok) {   // the 16-unit chunks of this wave's K share (two per 32-k chunk) are published?
            unsigned spins = 0;
            // One tile per block: nothing can be up before ~2K cycles after this block's own publish (the front-end used
            // to fill that time); polling through it only loads the memory system the publishers and loaders need.
            if (ntile == 1)
                for (int q = 0; q < backoff; ++q) cvae_sleep_64();
            for (;;) {
                unsigned f = (unsigned)t;
                if (lane < 2 * NC32 && 2 * c32_lo + lane < nch) f = cvae_atomic_load_agent(p.flags + (long)i * nch + 2 * c32_lo + lane);
                if (cvae_wave_all(f >= (unsigned)t)) break;
                cvae_sleep();
                if (++spins > (1u << 22)) {
                    p.status[0] = 2;
                    break;
                }
            }
        }
        cvae_compiler_fence();                         // operand loads stay below the poll
        if (p.prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        f32x4 hc[2 * NC32];                            // [2*ci] hi halves, [2*ci + 1] lo halves
        // rows of the tile that are batch padding (B = 1..3 at stage 6: 15..13 of 16) are not loaded: their operand rows are
        // zero and nobody reads their results, and the hand-off moves 1/16 .. 3/16 of the bytes
        const bool row_live = i * 16 + lr < p.B;
#pragma unroll
        for (int ci = 0; ci < 2 * NC32; ++ci) hc[ci] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has_k && row_live) {
#pragma unroll
            for (int ci = 0; ci < NC32; ++ci) {
                const unsigned so = ((unsigned)(2 * (c32_lo + ci)) * mtot + row0) * 64u;
                hc[2 * ci] = cvae_buf_load_f4_sc1(sb, voff, so);
                hc[2 * ci + 1] = cvae_buf_load_f4_sc1(sb, voff + 32u, so);
            }
        }
        const int grow = i * 16 + row;
        const bool live = grow < p.B;
        const bool keep1 = ntile == 2 && (k & 1);
        float gxr = cf0, gxz = cf1, gxn = cf2, hold = keep1 ? hkeep1 : hkeep0;
        if (live) {
            const unsigned so = ((unsigned)jg * mtot + row0) * 64u;      // wave-uniform part; the thread's row goes into voff
            if (t == 0) {
                if (p.dy) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, gxr, gxz, gxn);
                hold = cvae_buf_load_f1_sc1(hb, (unsigned)(row * 64 + u * 4), so);
            } else if (ntile > 2) {   // more than two tiles per block: re-read this thread's own h from the pair buffer
                const unsigned w0 = __builtin_bit_cast(unsigned, cvae_buf_load_f1_sc1(sb, (unsigned)(row * 64 + (u >> 1) * 4), so));
                const unsigned w1 = __builtin_bit_cast(unsigned, cvae_buf_load_f1_sc1(sb, (unsigned)(row * 64 + 32 + (u >> 1) * 4), so));
                const unsigned short hi = (unsigned short)((u & 1) ? (w0 >> 16) : (w0 & 0xffffu));
                const unsigned short lo = (unsigned short)((u & 1) ? (w1 >> 16) : (w1 & 0xffffu));
                hold = cvae_f16_bits_to_f32(hi) + cvae_f16_bits_to_f32(lo) * (1.0f / 2048.0f);
            }
        }
        if (has_k) {
#pragma unroll
            for (int ci = 0; ci < NC32; ++ci) {
#pragma unroll
                for (int a = 0; a < 4; ++a) acc[a] = cvae_mfma_16x16x32_f16(hc[2 * ci], wh[a][ci], acc[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a) accx[a] = cvae_mfma_16x16x32_f16(hc[2 * ci], wl[a][ci], accx[a]);
#pragma unroll
                for (int a = 0; a < 4; ++a) accy[a] = cvae_mfma_16x16x32_f16(hc[2 * ci + 1], wh[a][ci], accy[a]);
            }
        }
        if (k + 1 < ntask) load_x(k + 1);   // next task's front-end operands: they land under reduce + gates + publish
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                red[(wave * 16 + kq * 4 + q) * 84 + a * 16 + lr] = acc[a][q] + (accx[a][q] + accy[a][q]) * (1.0f / 2048.0f);
        if (p.prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        __syncthreads();
        {
            float hn_ = 0.0f;
            if (live) {
                float s[4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    s[a] = red[(0 * 16 + row) * 84 + a * 16 + u] + red[(1 * 16 + row) * 84 + a * 16 + u] +
                           red[(2 * 16 + row) * 84 + a * 16 + u] + red[(3 * 16 + row) * 84 + a * 16 + u];
                const float rg = cvae_sigmoid_fast(gxr + s[0]);
                const float zg = cvae_sigmoid_fast(gxz + s[1]);
                const float ng = cvae_tanh_fast(gxn + s[2] + rg * (s[3] + bhn));
                hn_ = ng + zg * (hold - ng);
            }
            if (keep1) hkeep1 = hn_; else hkeep0 = hn_;
            hsh[row * 16 + u] = hn_;
        }
        __syncthreads();
        if (tid < 64) {   // wave 0: 16 rows x 64 B of fp16 pairs = one contiguous 1 KiB block of chunk jg, slot t+1
            const int r = tid >> 2, part = tid & 3;           // part 0,1: hi halves of units 0-7 / 8-15; 2,3: lo halves
            const float* hv = hsh + r * 16 + (part & 1) * 8;
            unsigned pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                unsigned short h0, l0, h1, l1;
                cvae_split_f16(hv[2 * e], h0, l0);
                cvae_split_f16(hv[2 * e + 1], h1, l1);
                pk[e] = part < 2 ? ((unsigned)h0 | ((unsigned)h1 << 16)) : ((unsigned)l0 | ((unsigned)l1 << 16));
            }
            const f32x4 v = (f32x4){__builtin_bit_cast(float, pk[0]), __builtin_bit_cast(float, pk[1]),
                                    __builtin_bit_cast(float, pk[2]), __builtin_bit_cast(float, pk[3])};
            if (i * 16 + r < p.B)   // (padding rows are never read, see the operand loads)
                cvae_buf_store_f4_sc1(sb, (unsigned)tid * 16u, ((unsigned)jg * mtot + row0 + (unsigned)p.Bp) * 64u, v);
            cvae_drain_vmem();      // every lane's write-through store has left ...
            cvae_wave_barrier();    // ... (all 64 lanes are this one wave) before lane 0 raises the flag
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * nch + jg, (unsigned)(t + 1));
        } else if (tid < 128) {   // wave 1: the fp32 copy for the projection kernel (read after this launch: plain stores)
            const int l = tid - 64;
            *(f32x4*)(p.hbuf + ((long)jg * p.mtot + row0 + p.Bp) * 16 + l * 4) = *(const f32x4*)(hsh + l * 4);
        }
        if (ntile > 1 && k + 1 < ntask) {   // (behind wave 0's publish, so its drain never waits for this load)
            const int kn = k + 1, tn = kn / ntile, in_ = ti + (kn % ntile) * rts;
            fpre = (unsigned)tn;
            if (tn > 0 && has_k && lane < 2 * NC32 && 2 * c32_lo + lane < nch)
                fpre = cvae_atomic_load_agent(p.flags + (long)in_ * nch + 2 * c32_lo + lane);
        }
        if (p.prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
    }
    if (p.prof && tid == 64 * ((p.exp >> 2) & 3))    // measurement: exp bits 2-3 pick the reporting wave
        for (int q = 0; q < 4; ++q) p.prof[(long)blockIdx.x * 4 + q] = pc[q];
}

#include "cvae_exact3.h"
#include "cvae_ll.h"

struct OutParams {
    const float* hbuf;   // chunk-major; slot s rows start at s*Bp
    long mtot;
    const float* wo2;    // [16*NTN][H]: scale_out.w * out_1.w (or out_1.w), rows >= Co zero
    const float* bo2;    // [16*NTN]
    int H, Bp, T, B, ncell, Co, clamp_from;
    float clamp_min;
    float* out[CVAE_MAX_CELLS];       // per cell [B][T][Co]
};

// trj_out = scale_out(out_1(h_t)) (or the clamped out_1(h_t)) for every frame, written straight into [B][T][Co].
// Block = 16 rows (t, 16 b) x all 16*NTN columns; the 4 waves split K = H and reduce through LDS.  scale_out is folded
// into the projection at prepare time (gru_vae.py:371/393 then :402-406; clamp :408-412).
template <int NTN>
__global__ __launch_bounds__(256) void k_outproj(OutParams p) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int nch = p.H >> 4, c_lo = (nch * wave) >> 2, c_hi = (nch * (wave + 1)) >> 2;
    const long m0 = (long)blockIdx.x * 16;            // first row of this block, counted from slot 1
    float* red = (float*)CVAE_SMEM;                   // [4][16][16*NTN + 4]
    const int S = 16 * NTN + 4;
    f32x4 acc[NTN];
#pragma unroll
    for (int n = 0; n < NTN; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* abase = p.hbuf + ((long)p.Bp + m0 + lr) * 16 + kq * 4;
    for (int c = c_lo; c < c_hi; ++c) {
        const float4 a4 = *(const float4*)(abase + (long)c * p.mtot * 16);
        float4 b4[NTN];
#pragma unroll
        for (int n = 0; n < NTN; ++n) b4[n] = *(const float4*)(p.wo2 + (long)(16 * n + lr) * p.H + 16 * c + kq * 4);
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
    {   // one thread row per output row: the (t, b) split of the row index is one 32-bit division per thread
        const int r = tid >> 4, m = (int)m0 + r, t = m / p.Bp, b = m - t * p.Bp;
        if (b < p.ncell * p.B) {
            const int cell = b / p.B, bb = b - cell * p.B;
            float* orow = p.out[cell] + ((long)bb * p.T + t) * p.Co;
            for (int col = tid & 15; col < p.Co; col += 16) {
                float v = red[(0 * 16 + r) * S + col] + red[(1 * 16 + r) * S + col] + red[(2 * 16 + r) * S + col] +
                          red[(3 * 16 + r) * S + col] + p.bo2[col];
                if (p.clamp_from >= 0 && col >= p.clamp_from) v = fmaxf(v, p.clamp_min);
                orow[col] = v;
            }
        }
    }
}

struct EpiParams {
    const float* y;      // [T*Bp][ldy]
    long ldy;
    const float* sout_w; // [Co][Co] or null
    const float* sout_b;
    int clamp_from;      // >= 0: clamp out[c >= clamp_from] to >= clamp_min
    float clamp_min;     // ln(1e-6) (gru_vae.py:412), or -7.2543... for clamp_vae_laplace (gru_vae.py:417)
    int B, Bp, T, Co;
    int b0;              // first y row (within a slot) of this cell
    float* trj_out;      // [B][T][Co]
    float* y_last;       // [B][Co] (raw, gru_vae.py:452) or null
    int t_last;          // the frame whose raw projection is y_last: the cell's last VALID frame (frames - 1; T - 1 for a full cell)
};

// scale_out (dense, gru_vae.py:402-406) or log-variance clamp (gru_vae.py:408-412), transposing (t,b) -> (b,t)
__global__ void k_epilogue(EpiParams p) {
    float* row = (float*)CVAE_SMEM;
    const int t = blockIdx.x % p.T, b = blockIdx.x / p.T;
    const float* yr = p.y + ((long)t * p.Bp + p.b0 + b) * p.ldy;
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
        if (p.y_last && t == p.t_last) p.y_last[(long)b * p.Co + c] = row[c];
    }
}

// h_last[b][k] = hbuf slot T (the caller passes the cell's own frame count: the state behind its last valid frame)
__global__ void k_hlast(const float* hbuf, long mtot, float* h_last, int B, int Bp, int H, int T, int b0) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)B * H) {
        const int k = (int)(idx % H), b = (int)(idx / H);
        h_last[idx] = hbuf[((long)(k >> 4) * mtot + (long)T * Bp + b0 + b) * 16 + (k & 15)];
    }
}
