// Training recurrences with every matrix product on EXACT fp32 operands (reference arithmetic: fp32 autograd through
// nn.GRU / Conv1d, train_gru_cyclevae_gauss_batch.py:1418-1420 via gru_vae.py:376-393), in the geometry of k_gru_steps_v6:
// block = 8 hidden units x 32 batch rows, one v_mfma_f32_32x32x16_f16 tile, operands as three fp16 limbs
// (x = l0 + l1/2^11 + l2/2^22, six MFMAs per product, dropped terms < 2^-33 of a product), state exchanged tile-planar
// as two halves and a bf8 byte per value, written once by the thread that produced it.
//
// Forward (k_train_fwd_steps_x3).  Train mode feeds the GRU with TWO states: the carried h and the dropped o = mask * h
// (gru_drop, gru_vae.py:380: out_1 sees o, so the folded feedback F = W_ih[:, C9:] . out_1.w multiplies o while W_hh multiplies
// h).  Only h is exchanged (limb triples, 5 bytes per value): the CONSUMER forms the operand of the feedback product by masking
// the limbs of h with the dropout bits of its rows (known before the recurrence starts: one 16-byte load per lane and step,
// requested ahead of the flag wait), and the mask's scale 1/(1-p) is folded into the feedback weights when the image is built.
// Exchanging o as a second set of triples doubled what every CU must load per step (320 KB at B = 64) and made the step
// load-bound: measured 24.1K cycles per step against 2 x 9.5K of the pair kernel.  A wave's K share is its quarter of h, used
// twice: 2*KPW weight steps, l0 / l1 register-resident (256 registers per lane at H = 1024), the third limbs of the weights in
// LDS (128 KB).  Gate math, tape and the row-major fp32 copies (backward, projection) are fp32 as before.
// What a 32-row tile buys over the 16-row pair kernel (k_train_fwd_steps_h): the per-task phases that do not shrink with the
// tile -- flag wait, LDS reduction, cell, publish, drain -- are paid once per 32 rows instead of twice.
#pragma once
#include <cvae_intrin.h>

struct TrainFwd3Params {
    float* hx;            // exchanged h, limb triples, tile-planar: [H/16][mtot/32]{ l0 [kh][32 rows][8 halves] | l1 | l2 [kh][32][8 B] }
    const unsigned char* mbits;   // dropout bits of the operand of step t: [T][Bp/32][4 waves][64 lanes][16 steps] bytes (k_train_x3_maskbits)
    long mtot;            // (T + 1) * Bp; slot s = rows s*Bp.. holds the state going INTO step s (slot 0: k_train_x3_slot0)
    const float* w3;      // [H/8][4 waves][2 paths][KPW][3 limbs][64 lanes][8 halves] (k_prep_wrec_x3)
    const float* gi;      // [T*Bp][3H] time-major input-side pre-activations
    const float* bhn;     // [H]
    const float* gmask;   // [T][B][H] dropout mask of the state fed to out_1, scaled by 1/(1-p)
    float* tape;          // [T*Bp][4H]: r, z, n, q = W_hn h + b_hn
    float* hrow;          // [(T+1)*Bp][H] row-major fp32: slot t+1 = h_t
    float* orow;          //                               slot t+1 = o_t
    const float* wyT;     // [Co][3H]
    const float* dy;      // [B][Co]: y_in - b_o (frame-0 feedback correction)
    int Co, B, Bp, H, T, rts;
    int xmap;             // 1: XCD-aware block placement (cvae_block_map)
    unsigned* flags;      // [Bp/32][H/8], zeroed before launch: flags[i][c] = t <=> octet c of row tile i of h_t AND o_t is published
    int* status;
    long long* prof;      // null, or 4 cycle sums of block 0: flag wait, loads + MFMA, reduce + cell, publish + stores
};

// w3[c][wave][path][s][m][lane][e]: lane (col = lane & 31, kh = lane >> 5) holds k = 16*(wave*KPW + s) + 8*kh + e of column
// col = 8*g + u (unit j = 8c + u; g: r, z, n_in, n_h):
//   path 0 (operand h): g = 0: W_hh[j][k], 1: W_hh[H + j][k], 2: 0,            3: W_hh[2H + j][k]
//   path 1 (operand o): g = 0: F[j][k],    1: F[H + j][k],    2: F[2H + j][k], 3: 0          (F: k_prep_ffold)
// oscale = 1/(1-p) of gru_drop: the kernel multiplies F with the MASKED limbs of h (values 0 or h), the scale sits in the weights
// (exact for p = 0.5; otherwise one more fp32 rounding of F, where the reference rounds h/(1-p))
__global__ void k_prep_wrec_x3(const float* F, const float* whh, float* w3, int H, int KPW, float oscale) {
    const int NB = H >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (c, wave, path, s, lane, e)
    if (idx < (long)NB * 4 * 2 * KPW * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long r = idx >> 9;
        const int s = (int)(r % KPW); r /= KPW;
        const int path = (int)(r & 1); r >>= 1;
        const int wave = (int)(r & 3), c = (int)(r >> 2);
        const int col = lane & 31, kh = lane >> 5, g = col >> 3, u = col & 7, j = 8 * c + u;
        const int k = 16 * (wave * KPW + s) + 8 * kh + e;
        float w = 0.0f;
        if (k < H) {
            if (path == 0) {
                if (g != 2) w = whh[(long)((g == 3 ? 2 : g) * H + j) * H + k];
            } else if (g < 3) {
                w = F[(long)(g * H + j) * H + k] * oscale;
            }
        }
        unsigned short l0, l1, l2;
        cvae_split3_f16(w, l0, l1, l2);
        unsigned short* dst = (unsigned short*)w3 + (((((long)c * 4 + wave) * 2 + path) * KPW + s) * 3) * 512 + lane * 8 + e;
        dst[0] = l0;
        dst[512] = l1;
        dst[1024] = l2;
    }
}

// slot 0 of the exchange buffer: limb triples of h_in (hrow slot 0, written by k_train_prologue)
__global__ void k_train_x3_slot0(const float* hrow, float* hx, long mtot, int Bp, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, unit)
    if (idx < (long)Bp * H) {
        const int k = (int)(idx % H), r = (int)(idx / H);
        unsigned short l0, l1;
        unsigned char l2;
        cvae_split3_f16b8(hrow[idx], l0, l1, l2);
        const long base = ((long)(k >> 4) * (mtot >> 5) + (r >> 5)) * 2560;
        const int kh = (k >> 3) & 1, rr = r & 31, e = k & 7;
        unsigned char* h8 = (unsigned char*)hx + base;
        ((unsigned short*)(h8 + kh * 512 + rr * 16))[e] = l0;
        ((unsigned short*)(h8 + 1024 + kh * 512 + rr * 16))[e] = l1;
        h8[2048 + kh * 256 + rr * 8 + e] = l2;
    }
}

// mbits[t][i][wave][lane][s]: which of the 8 values lane (row lc = lane & 31 of tile i, kh = lane >> 5) feeds into its 16-k step s
// of the feedback product of step t survive the dropout of step t-1 (o_{t-1} = mask_{t-1} * h_{t-1}; o_{-1} = 0: step 0 is all
// zeros).  Bit layout of the byte: bits 0..3 = elements 0, 2, 4, 6, bits 4..7 = elements 1, 3, 5, 7 of the lane's 8 consecutive
// units -- the halves of the four operand registers -- so that (x | x << 12) >> q & 0x00010001 selects register q's two halves.
__global__ void k_train_x3_maskbits(const float* gmask, unsigned char* mbits, int T, int B, int Bp, int H, int KPW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (t, i, wave, lane, s)
    const int nrt = Bp >> 5;
    if (idx < (long)T * nrt * 4 * 64 * 16) {
        const int s = (int)(idx & 15), lane = (int)((idx >> 4) & 63), wave = (int)((idx >> 10) & 3);
        const long ti = idx >> 12;
        const int i = (int)(ti % nrt), t = (int)(ti / nrt);
        const int row = 32 * i + (lane & 31), k0 = 16 * (wave * KPW + s) + 8 * (lane >> 5);
        unsigned b = 0;
        if (t > 0 && s < KPW && row < B && k0 < H) {
            const float* m = gmask + ((long)(t - 1) * B + row) * H + k0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m[e] != 0.0f) b |= 1u << ((e >> 1) + 4 * (e & 1));
        }
        mbits[idx] = (unsigned char)b;
    }
}

__device__ __forceinline__ f32x16 cvae_zero16_t() {
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = 0.0f;
    return z;
}

// dropout bits of a lane's 8 values (byte layout of k_train_x3_maskbits) -> one 32-bit mask per operand register (two halves each)
struct cvae_m4 { unsigned q[4]; };
__device__ __forceinline__ cvae_m4 cvae_expand_bits(unsigned x) {
    const unsigned y = x | (x << 12);
    cvae_m4 m;
#pragma unroll
    for (int q = 0; q < 4; ++q) m.q[q] = ((y >> q) & 0x00010001u) * 0xFFFFu;
    return m;
}
__device__ __forceinline__ f32x4 cvae_mask_h8(f32x4 v, const cvae_m4& m) {
    f32x4 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float vq = v[q];
        o[q] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, vq) & m.q[q]);
    }
    return o;
}

template <int KPW>   // 16-k steps per wave = H/64
__global__ __launch_bounds__(256, 1) void k_train_fwd_steps_x3(TrainFwd3Params p) {
    constexpr int RS = 40, NS = 2 * KPW;
    constexpr float S1 = 1.0f / 2048.0f;
    constexpr int RD = KPW < 8 ? KPW : 8;              // operand ring: 16-k steps in flight per wave
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lc = lane & 31, kh = lane >> 5;
    const int H = p.H, NB = H >> 3, nrt = p.Bp >> 5;
    const int rts = p.rts;
    int c, ti;         // option train_xmap: XCD-aware placement (a row-tile group's blocks share XCDs: each L2 pulls only that group's rows)
    cvae_block_map((int)blockIdx.x, NB, rts, p.xmap != 0, c, ti);
    const int s_lo = wave * KPW;
    float* red = (float*)CVAE_SMEM;                    // [4 waves][32 rows][RS]
    float* val = red + 4 * 32 * RS;                    // [6: h, o, r, z, n, q][32 rows][8 units]
    unsigned short* hl = (unsigned short*)(val + 6 * 256);   // publish image of h: l0, l1 [32 rows][8 halves], l2 [32 rows][8 bytes]
    float* w2l = (float*)(hl + 640);                   // third limbs of the weights: [4 waves][NS][64 lanes][8 halves]
    const int row = tid >> 3, u = tid & 7, j = 8 * c + u;
    const unsigned xbytes = (unsigned)((long)(H >> 4) * p.mtot * 80);
    const cvae_buf hxb = cvae_make_buf(p.hx, xbytes);
    const unsigned tstride = (unsigned)(p.mtot >> 5);
    const unsigned voff = (unsigned)kh * 512u + (unsigned)lc * 16u, voff2 = 2048u + (unsigned)kh * 256u + (unsigned)lc * 8u;
    f32x4 w0[NS], w1[NS];                              // [0, KPW): W_hh (operand h), [KPW, 2 KPW): F / (1-p) (operand o = masked h)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float* src = p.w3 + (((((long)c * 4 + wave) * 2 + s / KPW) * KPW + s % KPW) * 3) * 256 + lane * 4;
        w0[s] = *(const f32x4*)src;
        w1[s] = *(const f32x4*)(src + 256);
        *(f32x4*)(w2l + (wave * NS + s) * 256 + lane * 4) = *(const f32x4*)(src + 512);
    }
    __syncthreads();
    const float* w2w = w2l + wave * NS * 256 + lane * 4;
    const float bhn = p.bhn[j];
    const int ntile = ti < nrt ? (nrt - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    float hk0 = 0.f, hk1 = 0.f, hk2 = 0.f, hk3 = 0.f;  // h_{t-1} of this thread's (row, unit) per tile of the block (at most four)
    long long pc[4] = {0, 0, 0, 0};
    const bool prof = p.prof && blockIdx.x == 0;
    unsigned fpre = 0u;
    // What a step needs besides the exchanged state (input-side pre-activations, dropout mask and bits) does not depend on the
    // recurrence.  Loads return in issue order, so requesting it right before the flag poll puts an HBM round trip in front of
    // every poll (measured: flag wait 9.3K cycles per step); it is requested ONE TASK AHEAD instead, behind the last operand
    // refill of the running task, and has landed long before the next poll.
    float ng0 = 0.f, ng1 = 0.f, ng2 = 0.f, nmsk = 0.f;
    f32x4 nmraw = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto prefetch_next = [&](int kn) {
        if (kn >= ntask) return;
        const int tn = kn / ntile, in_ = ti + (kn % ntile) * rts, grn = in_ * 32 + row;
        nmraw = *(const f32x4*)(p.mbits + ((((long)tn * nrt + in_) * 4 + wave) * 64 + lane) * 16);
        if (grn < p.B) {
            const float* gip = p.gi + ((long)tn * p.Bp + grn) * 3 * H;
            ng0 = gip[j]; ng1 = gip[H + j]; ng2 = gip[2 * H + j];
            nmsk = p.gmask[((long)tn * p.B + grn) * H + j];
        }
    };
    prefetch_next(0);
    for (int kk = 0; kk < ntask; ++kk) {
        long long c0 = prof ? cvae_clock() : 0;
        const int t = kk / ntile, tl = kk % ntile, i = ti + tl * rts;
        const unsigned row0 = (unsigned)(t * p.Bp + i * 32), tile0 = row0 >> 5;
        const f32x4 mraw = nmraw;
        const int grow = i * 32 + row;
        const bool live = grow < p.B;
        float g0 = ng0, g1 = ng1, g2 = ng2;
        const float msk = nmsk;
        float hold = tl == 0 ? hk0 : (tl == 1 ? hk1 : (tl == 2 ? hk2 : hk3));
        if (live && t == 0) {
            cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, g0, g1, g2);
            hold = p.hrow[(long)grow * H + j];
        }
        const bool pre_ok = ntile > 1 && kk > 0 && cvae_wave_all(fpre >= (unsigned)t);
        if (t > 0 && !pre_ok) {   // the octets (two per 16-unit chunk) of this wave's K share are published?
            unsigned spins = 0;
            for (;;) {
                unsigned f = (unsigned)t;
                if (lane < 2 * KPW && 2 * s_lo + lane < NB) f = cvae_atomic_load_agent(p.flags + (long)i * NB + 2 * s_lo + lane);
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
        f32x4 hc[2 * RD];
        f32x2 hb2[RD];
        auto load_op = [&](int s) {     // chunk s_lo + s of h, slot t (plain first-touch loads)
            const unsigned so = ((unsigned)(s_lo + s) * tstride + tile0) * 2560u;
            hc[2 * (s % RD)] = cvae_buf_load_f4(hxb, voff, so);
            hc[2 * (s % RD) + 1] = cvae_buf_load_f4(hxb, voff, so + 1024u);
            hb2[s % RD] = cvae_buf_load_f2(hxb, voff2, so);
        };
#pragma unroll
        for (int s = 0; s < RD; ++s) load_op(s);
        f32x16 a0 = cvae_zero16_t(), a1 = cvae_zero16_t(), a2 = cvae_zero16_t(), a3 = cvae_zero16_t();   // S0 | S1 | S2 (two chains)
#pragma unroll
        for (int s = 0; s < KPW; ++s) {
            const f32x4 l0 = hc[2 * (s % RD)], l1 = hc[2 * (s % RD) + 1];
            const f32x4 l2 = cvae_bf8x8_to_h8(hb2[s % RD]);
            {   // W_hh . h
                const f32x4 w2 = *(const f32x4*)(w2w + s * 256);
                a0 = cvae_mfma_32x32x16_f16(l0, w0[s], a0);
                a1 = cvae_mfma_32x32x16_f16(l0, w1[s], a1);
                a2 = cvae_mfma_32x32x16_f16(l1, w1[s], a2);
                a3 = cvae_mfma_32x32x16_f16(l0, w2, a3);
                a1 = cvae_mfma_32x32x16_f16(l1, w0[s], a1);
                a2 = cvae_mfma_32x32x16_f16(l2, w0[s], a2);
            }
            {   // F/(1-p) . (bits * h): the limbs of a dropped value are zeroed, the others are the limbs of h
                const float mw = mraw[s >> 2];
                const cvae_m4 mk = cvae_expand_bits((__builtin_bit_cast(unsigned, mw) >> (8 * (s & 3))) & 0xffu);
                const f32x4 m0 = cvae_mask_h8(l0, mk), m1 = cvae_mask_h8(l1, mk), m2 = cvae_mask_h8(l2, mk);
                const f32x4 w2 = *(const f32x4*)(w2w + (KPW + s) * 256);
                a0 = cvae_mfma_32x32x16_f16(m0, w0[KPW + s], a0);
                a1 = cvae_mfma_32x32x16_f16(m0, w1[KPW + s], a1);
                a2 = cvae_mfma_32x32x16_f16(m1, w1[KPW + s], a2);
                a3 = cvae_mfma_32x32x16_f16(m0, w2, a3);
                a1 = cvae_mfma_32x32x16_f16(m1, w0[KPW + s], a1);
                a2 = cvae_mfma_32x32x16_f16(m2, w0[KPW + s], a2);
            }
            cvae_sched_fence();
            if (s + RD < KPW) load_op(s + RD);
            if (KPW > RD && s + RD == KPW - 1) prefetch_next(kk + 1);     // behind the last operand refill
        }
        if (KPW <= RD) prefetch_next(kk + 1);
        cvae_sched_fence();
#pragma unroll
        for (int q = 0; q < 16; ++q)
            red[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh) * RS + lc] = a0[q] + (a1[q] + (a2[q] + a3[q]) * S1) * S1;
        if (prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        __syncthreads();
        {
            float rg = 0.f, zg = 0.f, ng = 0.f, qq = 0.f, hn = 0.f, on = 0.f;
            if (live) {
                float sg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    sg[g] = red[(0 * 32 + row) * RS + g * 8 + u] + red[(1 * 32 + row) * RS + g * 8 + u] +
                            red[(2 * 32 + row) * RS + g * 8 + u] + red[(3 * 32 + row) * RS + g * 8 + u];
                rg = cvae_sigmoid(g0 + sg[0]);
                zg = cvae_sigmoid(g1 + sg[1]);
                qq = sg[3] + bhn;
                ng = tanhf(g2 + sg[2] + rg * qq);
                hn = ng + zg * (hold - ng);
                on = hn * msk;
            }
            if (tl == 0) hk0 = hn; else if (tl == 1) hk1 = hn; else if (tl == 2) hk2 = hn; else hk3 = hn;
            val[0 * 256 + row * 8 + u] = hn;
            val[1 * 256 + row * 8 + u] = on;
            val[2 * 256 + row * 8 + u] = rg;
            val[3 * 256 + row * 8 + u] = zg;
            val[4 * 256 + row * 8 + u] = ng;
            val[5 * 256 + row * 8 + u] = qq;
            unsigned short l0, l1;          // the split happens HERE, once per value, by the thread that produced it
            unsigned char l2;
            cvae_split3_f16b8(hn, l0, l1, l2);
            hl[row * 8 + u] = l0;
            hl[256 + row * 8 + u] = l1;
            ((unsigned char*)(hl + 512))[row * 8 + u] = l2;
        }
        __syncthreads();
        if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        if (tid < 64) {   // wave 0 publishes the image into slot t+1 (write-through), drains, raises the octet's flag
            const unsigned so = ((unsigned)(c >> 1) * tstride + tile0 + (unsigned)(p.Bp >> 5)) * 2560u;
            cvae_buf_store_f4_sc1(hxb, (unsigned)(c & 1) * 512u + (unsigned)(tid & 31) * 16u, so + (unsigned)(tid >> 5) * 1024u,
                                  *(const f32x4*)(hl + tid * 8));
            if (tid < 32)
                cvae_buf_store_f2_sc1(hxb, 2048u + (unsigned)(c & 1) * 256u + (unsigned)tid * 8u, so, *(const f32x2*)(hl + 512 + tid * 4));
            cvae_drain_vmem();
            cvae_wave_barrier();
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * NB + c, (unsigned)(t + 1));
        } else {          // waves 1..3: the fp32 copies the backward / the projection read after this launch (plain 16-byte stores)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int item = (tid - 64) + it * 192;          // 6 arrays x 32 rows x 2 halves of 16 bytes
                const int arr = item >> 6, r = (item & 63) >> 1, half = item & 1;
                const f32x4 v = *(const f32x4*)(val + arr * 256 + r * 8 + half * 4);
                const long trow = (long)t * p.Bp + i * 32 + r;
                float* dst;
                if (arr == 0) dst = p.hrow + (trow + p.Bp) * H;
                else if (arr == 1) dst = p.orow + (trow + p.Bp) * H;
                else dst = p.tape + trow * 4 * H + (long)(arr - 2) * H;
                *(f32x4*)(dst + 8 * c + half * 4) = v;
            }
        }
        if (ntile > 1 && kk + 1 < ntask) {   // (behind wave 0's publish, so its drain never waits for this load)
            const int kn = kk + 1, tn = kn / ntile, in_ = ti + (kn % ntile) * rts;
            fpre = (unsigned)tn;
            if (tn > 0 && lane < 2 * KPW && 2 * s_lo + lane < NB)
                fpre = cvae_atomic_load_agent(p.flags + (long)in_ * NB + 2 * s_lo + lane);
        }
        if (prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
    }
    if (prof && tid == 0)
        for (int q = 0; q < 4; ++q) p.prof[q] = pc[q];
}

// ------------------------------------------------------------------------------------------------------------------------
// k_train_fwd_steps_x3h: the same exact-operand forward recurrence in the geometry of the pair kernel (k_train_fwd_steps_h):
// 16-row tiles, v_mfma_f32_16x16x32_f16, block = 8 hidden units (two 16-column tiles) x row tiles i, i+rts, ...
// Why both geometries exist.  A dependent step costs one chip-wide hand-off (publish, drain, flag, poll, first operand back:
// ~3 us).  With ONE tile per block that latency is exposed in full every step (32-row kernel at B = 64: flag wait 8.6K of 23.5K
// cycles per step); with two or more tiles per block the other tiles' work runs under it.  At B = 64 on 256 CUs a 32-row tiling
// leaves one tile per block, a 16-row tiling two: this kernel serves passes of up to 64 rows per 128 blocks, the 32-row kernel the
// larger ones (the stacked rec || cv decoder pass), where it needs half the flag waits, reductions and publishes per row.
// Exchange: 2.5 KiB per (slot, 32-k chunk, 16-row tile) = { l0 [4 kq][16 rows][8 halves] | l1 likewise | l2 [4 kq][16 rows][8 B] }.
// ------------------------------------------------------------------------------------------------------------------------
struct TrainFwd3hParams {
    float* hx;            // exchanged h: [slot][H/32 chunks][Bp/16 tiles] x 2560 B
    const unsigned char* mbits;   // [T][Bp/16][4 waves][64 lanes][8 steps] bytes (k_train_x3h_maskbits)
    const float* w3;      // [H/8][2 n][2 paths][H/32][3 limbs][64 lanes][8 halves] (k_prep_wrec_x3h)
    const float* gi;
    const float* bhn;
    const float* gmask;
    float* tape;
    float* hrow;
    float* orow;
    const float* wyT;
    const float* dy;
    int Co, B, Bp, H, T, rts;
    int xmap;             // 1: XCD-aware block placement (cvae_block_map)
    unsigned* flags;      // [Bp/16][H/8]
    int* status;
    long long* prof;
    int backoff;          // x 64 cycles before the first flag poll of a task (blocks with ONE tile: 256 waves polling early slow the publishes they wait for)
};

// w3[jg][n][path][c32][m][lane][e]: lane (col = lane & 15 = a*4 + u: gate a of unit j = 8jg + 4n + u; kq = lane >> 4) holds
// k = 32*c32 + 8*kq + e.  path 0 (operand h): a = 0: W_hh[j][k], 1: W_hh[H+j][k], 2: 0, 3: W_hh[2H+j][k];
// path 1 (operand o = bits * h): a = 0: F[j][k], 1: F[H+j][k], 2: F[2H+j][k], 3: 0, times oscale = 1/(1-p)
__global__ void k_prep_wrec_x3h(const float* F, const float* whh, float* w3, int H, float oscale) {
    const int n32 = H >> 5;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)(H >> 3) * 2 * 2 * n32 * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long r = idx >> 9;
        const int c32 = (int)(r % n32); r /= n32;
        const int path = (int)(r & 1), n = (int)((r >> 1) & 1), jg = (int)(r >> 2);
        const int col = lane & 15, kq = lane >> 4, a = col >> 2, u = col & 3, j = 8 * jg + 4 * n + u, k = 32 * c32 + 8 * kq + e;
        float w = 0.0f;
        if (path == 0) {
            if (a != 2) w = whh[(long)((a == 3 ? 2 : a) * H + j) * H + k];
        } else if (a < 3) {
            w = F[(long)(a * H + j) * H + k] * oscale;
        }
        unsigned short l0, l1, l2;
        cvae_split3_f16(w, l0, l1, l2);
        unsigned short* dst = (unsigned short*)w3 + (((((long)jg * 2 + n) * 2 + path) * n32 + c32) * 3) * 512 + lane * 8 + e;
        dst[0] = l0;
        dst[512] = l1;
        dst[1024] = l2;
    }
}

__global__ void k_train_x3h_slot0(const float* hrow, float* hx, int Bp, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, unit)
    if (idx < (long)Bp * H) {
        const int k = (int)(idx % H), r = (int)(idx / H);
        unsigned short l0, l1;
        unsigned char l2;
        cvae_split3_f16b8(hrow[idx], l0, l1, l2);
        unsigned char* h8 = (unsigned char*)hx + ((long)(k >> 5) * (Bp >> 4) + (r >> 4)) * 2560;
        const int kq = (k >> 3) & 3, rr = r & 15, e = k & 7;
        ((unsigned short*)(h8 + (kq * 16 + rr) * 16))[e] = l0;
        ((unsigned short*)(h8 + 1024 + (kq * 16 + rr) * 16))[e] = l1;
        h8[2048 + (kq * 16 + rr) * 8 + e] = l2;
    }
}

// mbits[t][i][wave][lane][s]: lane (row lr = lane & 15 of 16-row tile i, kq = lane >> 4), 32-k chunk wave*C32W + s: byte layout as
// k_train_x3_maskbits (bits 0..3 = elements 0, 2, 4, 6; bits 4..7 = elements 1, 3, 5, 7); step t uses the dropout of step t-1
__global__ void k_train_x3h_maskbits(const float* gmask, unsigned char* mbits, int T, int B, int Bp, int H, int C32W) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (t, i, wave, lane, s)
    const int nrt = Bp >> 4;
    if (idx < (long)T * nrt * 4 * 64 * 8) {
        const int s = (int)(idx & 7), lane = (int)((idx >> 3) & 63), wave = (int)((idx >> 9) & 3);
        const long ti = idx >> 11;
        const int i = (int)(ti % nrt), t = (int)(ti / nrt);
        const int row = 16 * i + (lane & 15), k0 = 32 * (wave * C32W + s) + 8 * (lane >> 4);
        unsigned b = 0;
        if (t > 0 && s < C32W && row < B && k0 < H) {
            const float* m = gmask + ((long)(t - 1) * B + row) * H + k0;
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (m[e] != 0.0f) b |= 1u << ((e >> 1) + 4 * (e & 1));
        }
        mbits[idx] = (unsigned char)b;
    }
}

template <int C32W, int KW>   // C32W 32-k chunks of h per wave, on the first KW waves (H = 1024: 8 on 4; H = 64: 1 on 2)
__global__ __launch_bounds__(256, 1) void k_train_fwd_steps_x3h(TrainFwd3hParams p) {
    constexpr float S1 = 1.0f / 2048.0f;
    constexpr int RD = C32W < 8 ? C32W : 8;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, ng = H >> 3, nrt = p.Bp >> 4, n32 = H >> 5;
    const int rts = p.rts;
    int jg, ti;
    cvae_block_map((int)blockIdx.x, ng, rts, p.xmap != 0, jg, ti);
    const bool kwave = wave < KW;                        // this wave has a share of K
    const int c_lo = wave * C32W;
    float* red = (float*)CVAE_SMEM;                     // [4 waves][16 rows][36]
    unsigned short* hl = (unsigned short*)(red + 4 * 16 * 36);   // publish image: l0, l1 [16 rows][8 halves], l2 [16 rows][8 bytes] (640 B)
    float* w2l = (float*)(hl + 320);                             // third limbs of the weights: [KW waves][2 n][2 paths][C32W][64 lanes][8 halves]
    const unsigned xbytes = (unsigned)((long)(p.T + 1) * n32 * nrt * 2560);
    const cvae_buf hb = cvae_make_buf(p.hx, xbytes);
    const unsigned voff = (unsigned)lane * 16u, voff2 = 2048u + (unsigned)lane * 8u;
    f32x4 w0[2][2][C32W], w1[2][2][C32W];               // [n][path][chunk]
    if (kwave) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int pa = 0; pa < 2; ++pa)
#pragma unroll
                for (int ci = 0; ci < C32W; ++ci) {
                    const float* w = p.w3 + (((((long)jg * 2 + n) * 2 + pa) * n32 + c_lo + ci) * 3) * 256 + lane * 4;
                    w0[n][pa][ci] = *(const f32x4*)w;
                    w1[n][pa][ci] = *(const f32x4*)(w + 256);
                    *(f32x4*)(w2l + ((((wave * 2 + n) * 2 + pa) * C32W + ci) * 64 + lane) * 4) = *(const f32x4*)(w + 512);
                }
    }
    __syncthreads();
    const float* w2w = w2l + (long)wave * 4 * C32W * 256 + lane * 4;
    const int gt = tid - 128, row = (gt >> 3) & 15, u8 = gt & 7, j = 8 * jg + u8;
    const bool gate = tid >= 128;
    const float bhn = p.bhn[j];
    long long pc[4] = {0, 0, 0, 0};
    const bool prof = p.prof && blockIdx.x == 0;
    const int ntile = ti < nrt ? (nrt - ti + rts - 1) / rts : 0;
    float hkeep0 = 0.f, hkeep1 = 0.f;
    unsigned fnext = 0u;
    for (int t = 0; t < p.T; ++t) {
        int tcount = 0;
        for (int i = ti; i < nrt; i += rts, ++tcount) {
            long long c0 = prof ? cvae_clock() : 0;
            if (kwave && t > 0 && !cvae_wave_all(fnext >= (unsigned)t)) {   // octets [4 c_lo, 4 (c_lo + C32W)) of slot t
                unsigned spins = 0;
                for (int q = 0; q < p.backoff; ++q) cvae_sleep_64();
                for (;;) {
                    unsigned f = (unsigned)t;
                    if (lane < 4 * C32W) f = cvae_atomic_load_agent(p.flags + (long)i * ng + 4 * c_lo + lane);
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
            f32x4 a0[RD], a1[RD];
            f32x2 a2[RD];
            auto load_op = [&](int ci) {
                const unsigned so = (((unsigned)t * (unsigned)n32 + (unsigned)(c_lo + ci)) * (unsigned)nrt + (unsigned)i) * 2560u;
                a0[ci % RD] = cvae_buf_load_f4(hb, voff, so);
                a1[ci % RD] = cvae_buf_load_f4(hb, voff, so + 1024u);
                a2[ci % RD] = cvae_buf_load_f2(hb, voff2, so);
            };
            f32x2 mraw = (f32x2){0.f, 0.f};
            if (kwave) {
#pragma unroll
                for (int ci = 0; ci < RD; ++ci) load_op(ci);
                mraw = *(const f32x2*)(p.mbits + ((((long)t * nrt + i) * 4 + wave) * 64 + lane) * 8);
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
            f32x4 s0[2], s1[2], s2[2], s3[2];            // per column tile: S0 | S1 | S2 (two chains)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                s0[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                s1[n] = s0[n]; s2[n] = s0[n]; s3[n] = s0[n];
            }
            if (kwave) {
#pragma unroll
                for (int ci = 0; ci < C32W; ++ci) {
                    const f32x4 l0 = a0[ci % RD], l1 = a1[ci % RD], l2 = cvae_bf8x8_to_h8(a2[ci % RD]);
                    const float mw = mraw[ci >> 2];
                    const cvae_m4 mk = cvae_expand_bits((__builtin_bit_cast(unsigned, mw) >> (8 * (ci & 3))) & 0xffu);
                    const f32x4 m0 = cvae_mask_h8(l0, mk), m1 = cvae_mask_h8(l1, mk), m2 = cvae_mask_h8(l2, mk);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const f32x4 wh2 = *(const f32x4*)(w2w + ((n * 2 + 0) * C32W + ci) * 256);
                        const f32x4 wo2 = *(const f32x4*)(w2w + ((n * 2 + 1) * C32W + ci) * 256);
                        s0[n] = cvae_mfma_16x16x32_f16(l0, w0[n][0][ci], s0[n]);
                        s1[n] = cvae_mfma_16x16x32_f16(l0, w1[n][0][ci], s1[n]);
                        s2[n] = cvae_mfma_16x16x32_f16(l1, w1[n][0][ci], s2[n]);
                        s3[n] = cvae_mfma_16x16x32_f16(l0, wh2, s3[n]);
                        s1[n] = cvae_mfma_16x16x32_f16(l1, w0[n][0][ci], s1[n]);
                        s2[n] = cvae_mfma_16x16x32_f16(l2, w0[n][0][ci], s2[n]);
                        s0[n] = cvae_mfma_16x16x32_f16(m0, w0[n][1][ci], s0[n]);
                        s1[n] = cvae_mfma_16x16x32_f16(m0, w1[n][1][ci], s1[n]);
                        s2[n] = cvae_mfma_16x16x32_f16(m1, w1[n][1][ci], s2[n]);
                        s3[n] = cvae_mfma_16x16x32_f16(m0, wo2, s3[n]);
                        s1[n] = cvae_mfma_16x16x32_f16(m1, w0[n][1][ci], s1[n]);
                        s2[n] = cvae_mfma_16x16x32_f16(m2, w0[n][1][ci], s2[n]);
                    }
                    cvae_sched_fence();
                    if (ci + RD < C32W) load_op(ci + RD);
                }
            }
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    red[(wave * 16 + kq * 4 + q) * 36 + n * 16 + lr] = s0[n][q] + (s1[n][q] + (s2[n][q] + s3[n][q]) * S1) * S1;
            {   // next task of this block: the following tile of step t, or this block's first tile of step t + 1
                const bool wrap = i + rts >= nrt;
                const int i_n = wrap ? ti : i + rts, t_n = wrap ? t + 1 : t;
                fnext = (unsigned)t_n;
                if (kwave && t_n > 0 && t_n < p.T && lane < 4 * C32W) fnext = cvae_atomic_load_agent(p.flags + (long)i_n * ng + 4 * c_lo + lane);
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
                {   // the split happens here, once per value, by the thread that produced it
                    unsigned short l0, l1;
                    unsigned char l2;
                    cvae_split3_f16b8(hn, l0, l1, l2);
                    hl[row * 8 + u8] = l0;
                    hl[128 + row * 8 + u8] = l1;
                    ((unsigned char*)(hl + 256))[row * 8 + u8] = l2;
                }
                if (grow < p.Bp) {
                    p.hrow[((long)(t + 1) * p.Bp + grow) * H + j] = hn;
                    p.orow[((long)(t + 1) * p.Bp + grow) * H + j] = on;
                    float* tp = p.tape + ((long)t * p.Bp + grow) * 4 * H + j;
                    tp[0] = rg; tp[H] = zg; tp[2 * H] = ng_; tp[3 * H] = qq;
                }
            }
            __syncthreads();
            if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
            if (tid < 64) {   // wave 0: lanes 0..15 publish l0 (16 rows x 16 B), 16..31 l1, 32..47 l2 (16 rows x 8 B) of slot t+1
                const int part = tid >> 4, r = tid & 15;
                const unsigned so = (((unsigned)(t + 1) * (unsigned)n32 + (unsigned)(jg >> 2)) * (unsigned)nrt + (unsigned)i) * 2560u;
                if (part < 2)
                    cvae_buf_store_f4_sc1(hb, (unsigned)part * 1024u + (unsigned)(jg & 3) * 256u + (unsigned)r * 16u, so,
                                          *(const f32x4*)(hl + part * 128 + r * 8));
                else if (part == 2)
                    cvae_buf_store_f2_sc1(hb, 2048u + (unsigned)(jg & 3) * 128u + (unsigned)r * 8u, so, *(const f32x2*)(hl + 256 + r * 4));
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

// ------------------------------------------------------------------------------------------------------------------------
// k_train_bwd_steps_x3: the reverse recurrence of k_train_bwd_steps (cvae_train_bwd.h: same decomposition, same folded feedback
// path, same hand-off) with the exchanged gate gradients and the weights [W_hh^T | F^T] as fp16 TRIPLES: six MFMAs per product,
// fp32-exact operands.  A gate gradient v travels as the triple of v * 2^8 (two halves and a bf8 byte, 5 bytes): bit-exact for
// 2^-24 <= |v| < 2^8 (the scaled value is a normal half down to 2^-16 and its third limb a normal bf8), absolute error
// <= 2^-48 below, status 5 at |v| >= ~234 (the caller then repeats the step on the fp32 per-step path, stage4.Stage4Step).
// The third limbs of the block's 16 x 4096 weights live in LDS (128 KB), l0 / l1 in registers as before.
// Exchange: 2.5 KiB per (step, producer octet, 16-row tile) = { l0 [4 kq][16 rows][8 halves] | l1 likewise | l2 [4 kq][16 rows][8 B] }.
// ------------------------------------------------------------------------------------------------------------------------
// wbk3[c][wave][s]{ l0 [64 lanes][8 halves] | l1 likewise | l2 [64 lanes][8 bytes] }: k_prep_wbk's matrix (cvae_train_bwd.h) as limb
// triples, 2560 B per 32-k step.  The third limb is a bf8 byte (of l2 * 2^6, like the exchanged values: bit-exact for |w| >= 2^-16,
// absolute error <= 2^-40 below), so that a block's third limbs take 64 KB of LDS instead of 128: with 138 KB of LDS per block no
// second workgroup fits on the CU and the weight-gradient GEMMs of the side stream cannot run under the recurrence.
__global__ void k_prep_wbk3(const float* F, const float* whh, float* wbk3, int H, int KPW) {
    const int NB = H >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (c, wave, s, lane, e)
    if (idx < (long)NB * 4 * KPW * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int s = (int)((idx >> 9) % KPW), wave = (int)(((idx >> 9) / KPW) & 3), c = (int)((idx >> 9) / KPW / 4);
        const int col = lane & 15, kq = lane >> 4, k = 32 * (wave * KPW + s) + 8 * kq + e, j = k >> 2, comp = k & 3;
        float v = 0.0f;
        if (j < H) {
            if (col < 8) {
                if (comp != 2) v = whh[(long)((comp == 3 ? 2 : comp) * H + j) * H + 8 * c + col];
            } else if (comp < 3) {
                v = F[(long)(comp * H + j) * H + 8 * c + col - 8];
            }
        }
        unsigned short l0, l1;
        unsigned char l2;
        cvae_split3_f16b8(v, l0, l1, l2);
        unsigned char* base = (unsigned char*)wbk3 + (((long)c * 4 + wave) * KPW + s) * 2560;
        ((unsigned short*)base)[lane * 8 + e] = l0;
        ((unsigned short*)(base + 1024))[lane * 8 + e] = l1;
        base[2048 + lane * 8 + e] = l2;
    }
}

#ifndef CVAE_BWD_RING
// Operand ring depth of the reverse recurrence (k-steps in flight per wave).  5 instead of 8 costs the recurrence nothing (9.74-9.80 ms
// per step either way) and takes the kernel from 256 + 178 to 256 + 140 registers per lane: 116 free, so that a block of the big
// weight-gradient GEMMs (k_gemm_tn2<3,4>, 63 + 48) or two of the small ones fit on a SIMD beside it -- the side stream's GEMMs run
// at 7.2 instead of 9.5 ms of kernel time per step and may be started BEHIND the data-gradient chain (option wgrad_order)
#define CVAE_BWD_RING 5
#endif
template <int KPW>   // 32-k steps per wave = 4H / 128
__global__ __launch_bounds__(256, 1) void k_train_bwd_steps_x3(TrainBwdParams p) {
    constexpr int RD = KPW < CVAE_BWD_RING ? KPW : CVAE_BWD_RING;
    constexpr int RS = 20;
    constexpr float S1 = 1.0f / 2048.0f;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, NB = H >> 3, nt16 = p.Bp >> 4;
    const int rts = p.rts;
    int c, ti;         // option train_xmap: XCD-aware placement (a row-tile group's blocks share XCDs: each L2 pulls only that group's rows)
    cvae_block_map((int)blockIdx.x, NB, rts, p.xmap != 0, c, ti);
    float* red = (float*)CVAE_SMEM;                                   // [4 waves][16 rows][RS]
    unsigned short* pub = (unsigned short*)(red + 4 * 16 * RS);       // l0, l1: [4 kq][16 rows][8 halves] each, l2: [4 kq][16 rows][8 bytes]
    float* w2l = (float*)(pub + 1280);                                // third limbs of the weights: [4 waves][KPW][64 lanes][8 bytes (bf8)]
    const cvae_buf gb = cvae_make_buf(p.gx, (unsigned)((long)p.T * NB * nt16 * 2560));
    f32x4 w0[KPW], w1[KPW];
#pragma unroll
    for (int s = 0; s < KPW; ++s) {
        const float* src = p.wbk + (((long)c * 4 + wave) * KPW + s) * 640;
        w0[s] = *(const f32x4*)(src + lane * 4);
        w1[s] = *(const f32x4*)(src + 256 + lane * 4);
        *(f32x2*)(w2l + (wave * KPW + s) * 128 + lane * 2) = *(const f32x2*)(src + 512 + lane * 2);
    }
    __syncthreads();
    const float* w2w = w2l + wave * KPW * 128 + lane * 2;
    const int row = (tid >> 3) & 15, u = tid & 7, k = 8 * c + u;
    const bool gate_thread = tid < 128;
    // this launch covers the row tiles [tile_lo, tile_lo + tile_n) of the pass (tile_n = 0: all of them): the host cuts a pass with more
    // than two tiles per block into launches of two (rows are independent recurrences; with <= 2 tiles per block the carried z-path
    // gradient stays in registers instead of going through p.dhz)
    const int tile_lo = p.tile_n > 0 ? p.tile_lo : 0, tile_n = p.tile_n > 0 ? p.tile_n : nt16;
    const int ntile = ti < tile_n ? (tile_n - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    float keep0 = 0.f, keep1 = 0.f;
    // What the cell backward of a (row, unit) needs from the tape does not depend on the recurrence.  Loads return in issue order,
    // so requested right before the flag poll they put an HBM round trip in front of it; they are requested ONE TASK AHEAD, behind
    // the last operand refill of the running task (k_train_fwd_steps_x3 has the measurements).
    float ntr = 0.f, ntz = 0.f, ntn = 0.f, ntq = 0.f, nthp = 0.f, ntmask = 0.f, ntdov = 0.f;
    auto prefetch_next = [&](int kn) {
        if (kn >= ntask || !gate_thread) return;
        const int ttn = kn / ntile, tn_ = p.T - 1 - ttn, in_ = tile_lo + ti + (kn % ntile) * rts, grn = in_ * 16 + row;
        if (grn < p.B) {
            const long rn = (long)tn_ * p.Bp + grn;
            const float* tp = p.tape + rn * 4 * H + k;
            ntr = tp[0]; ntz = tp[H]; ntn = tp[2 * H]; ntq = tp[3 * H];
            nthp = p.hrow[rn * H + k];
            ntmask = p.gmask[((long)tn_ * p.B + grn) * H + k];
            ntdov = p.dovl[rn * H + k];
        }
    };
    prefetch_next(0);
    long long pc[4] = {0, 0, 0, 0};
    const bool prof = p.prof && blockIdx.x == 0;
    for (int kk = 0; kk < ntask; ++kk) {
        long long c0 = prof ? cvae_clock() : 0;
        const int tt = kk / ntile, t = p.T - 1 - tt, i = tile_lo + ti + (kk % ntile) * rts;
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        const int grow = i * 16 + row;
        const bool live = gate_thread && grow < p.B;
        const long rowi = (long)t * p.Bp + grow;
        const float tr = ntr, tz = ntz, tn = ntn, tq = ntq, thp = nthp, tmask = ntmask, tdov = ntdov;
        if (tt == 0) prefetch_next(kk + 1);        // (no operand stream in the first step)
        if (tt > 0) {
            unsigned spins = 0;
            for (int q = 0; q < p.backoff; ++q) cvae_sleep_64();
            for (;;) {   // the octets of this wave's K share have published step t+1?
                unsigned f = (unsigned)tt;
                if (lane < KPW) f = cvae_atomic_load_agent(p.flags + (long)i * NB + wave * KPW + lane);
                if (cvae_wave_all(f >= (unsigned)tt)) break;
                cvae_sleep();
                if (++spins > (1u << 22)) {
                    p.status[0] = 4;
                    break;
                }
            }
            cvae_compiler_fence();
            if (prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
            f32x4 gc[2 * RD];
            f32x2 gc2[RD];
            auto load_g = [&](int s) {
                const unsigned so = ((unsigned)((t + 1) * NB + wave * KPW + s) * (unsigned)nt16 + (unsigned)i) * 2560u;
                gc[2 * (s % RD)] = cvae_buf_load_f4(gb, (unsigned)lane * 16u, so);
                gc[2 * (s % RD) + 1] = cvae_buf_load_f4(gb, (unsigned)lane * 16u, so + 1024u);
                gc2[s % RD] = cvae_buf_load_f2(gb, 2048u + (unsigned)lane * 8u, so);
            };
#pragma unroll
            for (int s = 0; s < RD; ++s) load_g(s);
#pragma unroll
            for (int s = 0; s < KPW; ++s) {
                const f32x4 l0 = gc[2 * (s % RD)], l1 = gc[2 * (s % RD) + 1], l2 = cvae_bf8x8_to_h8(gc2[s % RD]);
                const f32x4 w2 = cvae_bf8x8_to_h8(*(const f32x2*)(w2w + s * 128));
                a0 = cvae_mfma_16x16x32_f16(l0, w0[s], a0);
                a1 = cvae_mfma_16x16x32_f16(l0, w1[s], a1);
                a2 = cvae_mfma_16x16x32_f16(l1, w1[s], a2);
                a3 = cvae_mfma_16x16x32_f16(l0, w2, a3);
                a1 = cvae_mfma_16x16x32_f16(l1, w0[s], a1);
                a2 = cvae_mfma_16x16x32_f16(l2, w0[s], a2);
                cvae_sched_fence();
                if (s + RD < KPW) load_g(s + RD);
                if (KPW > RD && s + RD == KPW - 1) prefetch_next(kk + 1);     // behind the last operand refill
            }
            if (KPW <= RD) prefetch_next(kk + 1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(wave * 16 + kq * 4 + q) * RS + lr] = a0[q] + (a1[q] + (a2[q] + a3[q]) * S1) * S1;
        if (prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        __syncthreads();
        if (gate_thread) {
            const bool k1 = ntile == 2 && (kk & 1);
            float v[4] = {0.f, 0.f, 0.f, 0.f}, dhz = 0.f;
            if (live) {
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    sa += red[(w * 16 + row) * RS + u];
                    sb += red[(w * 16 + row) * RS + 8 + u];
                }
                float hold = k1 ? keep1 : keep0;
                if (ntile > 2) hold = tt > 0 ? p.dhz[(long)grow * H + k] : 0.0f;
                const float dht = hold + sa * (1.0f / CVAE_BWD_GSCALE) + tmask * (tdov + sb * (1.0f / CVAE_BWD_GSCALE));
                const float r = tr, z = tz, n = tn, q = tq, hp = thp;
                const float dn = dht * (1.0f - z), dz = dht * (hp - n);
                v[2] = dn * (1.0f - n * n);
                v[3] = v[2] * r;
                v[0] = v[2] * q * r * (1.0f - r);
                v[1] = dz * z * (1.0f - z);
                dhz = dht * z;
            }
            if (ntile > 2) p.dhz[(long)grow * H + k] = dhz;
            else if (k1) keep1 = dhz;
            else keep0 = dhz;
            float* gi = p.dgi + rowi * 3 * H + k;
            float* gh = p.dgh + rowi * 3 * H + k;
            gi[0] = v[0]; gi[H] = v[1]; gi[2 * H] = v[2];
            gh[0] = v[0]; gh[H] = v[1]; gh[2 * H] = v[3];
#pragma unroll
            for (int cm = 0; cm < 4; ++cm) {
                const float sv = v[cm] * CVAE_BWD_GSCALE;
                if (!(fabsf(sv) < p.ovf)) p.status[0] = 5;      // outside the half range (or NaN): the step is invalid
                unsigned short l0, l1;
                unsigned char l2;
                cvae_split3_f16b8(sv, l0, l1, l2);
                const int kl = 4 * u + cm;
                pub[((kl >> 3) * 16 + row) * 8 + (kl & 7)] = l0;
                pub[512 + ((kl >> 3) * 16 + row) * 8 + (kl & 7)] = l1;
                ((unsigned char*)(pub + 1024))[((kl >> 3) * 16 + row) * 8 + (kl & 7)] = l2;
            }
        }
        __syncthreads();
        if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        if (tid < 64) {   // wave 0: 2.5 KiB = the LDS image, lane-linear per limb
            const unsigned so = ((unsigned)(t * NB + c) * (unsigned)nt16 + (unsigned)i) * 2560u;
            cvae_buf_store_f4_sc1(gb, (unsigned)tid * 16u, so, *(const f32x4*)(pub + tid * 8));
            cvae_buf_store_f4_sc1(gb, (unsigned)tid * 16u, so + 1024u, *(const f32x4*)(pub + 512 + tid * 8));
            cvae_buf_store_f2_sc1(gb, 2048u + (unsigned)tid * 8u, so, *(const f32x2*)(pub + 1024 + tid * 4));
            cvae_drain_vmem();
            cvae_wave_barrier();
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * NB + c, (unsigned)(tt + 1));
        }
        if (prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
    }
    if (prof && tid == 0)
        for (int q = 0; q < 4; ++q) p.prof[4 + q] = pc[q];
}
