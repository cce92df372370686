// Training recurrences with every matrix product on EXACT fp32 operands (reference arithmetic: fp32 autograd through
// nn.GRU / Conv1d, train_gru_cyclevae_gauss_batch.py:1418-1420 via gru_vae.py:376-393), in the geometry of k_gru_steps_v6:
// block = 8 hidden units x 32 batch rows, one v_mfma_f32_32x32x16_f16 tile, operands as three fp16 limbs
// (x = l0 + l1/2^11 + l2/2^22, six MFMAs per product, dropped terms < 2^-33 of a product), state exchanged tile-planar
// as two halves and a bf8 byte per value, written once by the thread that produced it.
//
// Forward (k_train_fwd_steps_x3).  Train mode feeds the GRU with TWO states: the carried h and the dropped o = mask * h
// (gru_drop, gru_vae.py:380: out_1 sees o, so the folded feedback F = W_ih[:, C9:] . out_1.w multiplies o while W_hh multiplies
// h).  Both are exchanged as limb triples (hx, ox); a wave's K share is its quarter of h followed by the same quarter of o:
// 2*KPW 16-k steps, weights l0 / l1 register-resident (256 registers per lane at H = 1024), the third limbs of the weights
// in LDS (128 KB).  Gate math, tape and the row-major fp32 copies (backward, projection) are fp32 as before.
// What a 32-row tile buys over the 16-row pair kernel (k_train_fwd_steps_h): the per-task phases that do not shrink with the
// tile -- flag wait, LDS reduction, cell, publish, drain -- are paid once per 32 rows instead of twice.
#pragma once
#include <cvae_intrin.h>

struct TrainFwd3Params {
    float* hx;            // exchanged h, limb triples, tile-planar: [H/16][mtot/32]{ l0 [kh][32 rows][8 halves] | l1 | l2 [kh][32][8 B] }
    float* ox;            // exchanged o = mask * h, likewise
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
    unsigned* flags;      // [Bp/32][H/8], zeroed before launch: flags[i][c] = t <=> octet c of row tile i of h_t AND o_t is published
    int* status;
    long long* prof;      // null, or 4 cycle sums of block 0: flag wait, loads + MFMA, reduce + cell, publish + stores
};

// w3[c][wave][path][s][m][lane][e]: lane (col = lane & 31, kh = lane >> 5) holds k = 16*(wave*KPW + s) + 8*kh + e of column
// col = 8*g + u (unit j = 8c + u; g: r, z, n_in, n_h):
//   path 0 (operand h): g = 0: W_hh[j][k], 1: W_hh[H + j][k], 2: 0,            3: W_hh[2H + j][k]
//   path 1 (operand o): g = 0: F[j][k],    1: F[H + j][k],    2: F[2H + j][k], 3: 0          (F: k_prep_ffold)
__global__ void k_prep_wrec_x3(const float* F, const float* whh, float* w3, int H, int KPW) {
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
                w = F[(long)(g * H + j) * H + k];
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

// slot 0 of the exchange buffers: limb triples of h_in (hrow slot 0, written by k_train_prologue), zeros for o_{-1}
__global__ void k_train_x3_slot0(const float* hrow, float* hx, float* ox, long mtot, int Bp, int H) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (row, unit)
    if (idx < (long)Bp * H) {
        const int k = (int)(idx % H), r = (int)(idx / H);
        unsigned short l0, l1;
        unsigned char l2;
        cvae_split3_f16b8(hrow[idx], l0, l1, l2);
        const long base = ((long)(k >> 4) * (mtot >> 5) + (r >> 5)) * 2560;
        const int kh = (k >> 3) & 1, rr = r & 31, e = k & 7;
        unsigned char* h8 = (unsigned char*)hx + base;
        unsigned char* o8 = (unsigned char*)ox + base;
        ((unsigned short*)(h8 + kh * 512 + rr * 16))[e] = l0;
        ((unsigned short*)(h8 + 1024 + kh * 512 + rr * 16))[e] = l1;
        h8[2048 + kh * 256 + rr * 8 + e] = l2;
        ((unsigned short*)(o8 + kh * 512 + rr * 16))[e] = 0;
        ((unsigned short*)(o8 + 1024 + kh * 512 + rr * 16))[e] = 0;
        o8[2048 + kh * 256 + rr * 8 + e] = 0;
    }
}

__device__ __forceinline__ f32x16 cvae_zero16_t() {
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = 0.0f;
    return z;
}

template <int KPW>   // 16-k steps per wave and path = H/64
__global__ __launch_bounds__(256, 1) void k_train_fwd_steps_x3(TrainFwd3Params p) {
    constexpr int RS = 40, NS = 2 * KPW;
    constexpr float S1 = 1.0f / 2048.0f;
    constexpr int RD = NS < 8 ? NS : 8;                // operand ring: 16-k steps in flight per wave
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lc = lane & 31, kh = lane >> 5;
    const int H = p.H, NB = H >> 3, nrt = p.Bp >> 5;
    const int rts = p.rts, c = blockIdx.x % NB, ti = blockIdx.x / NB;
    const int s_lo = wave * KPW;
    float* red = (float*)CVAE_SMEM;                    // [4 waves][32 rows][RS]
    float* val = red + 4 * 32 * RS;                    // [6: h, o, r, z, n, q][32 rows][8 units]
    unsigned short* hl = (unsigned short*)(val + 6 * 256);   // publish image of h: l0, l1 [32 rows][8 halves], l2 [32 rows][8 bytes]
    unsigned short* ol = hl + 640;                     // the same for o
    float* w2l = (float*)(ol + 640);                   // third limbs of the weights: [4 waves][NS][64 lanes][8 halves]
    const int row = tid >> 3, u = tid & 7, j = 8 * c + u;
    const unsigned xbytes = (unsigned)((long)(H >> 4) * p.mtot * 80);
    const cvae_buf hxb = cvae_make_buf(p.hx, xbytes), oxb = cvae_make_buf(p.ox, xbytes);
    const unsigned tstride = (unsigned)(p.mtot >> 5);
    const unsigned voff = (unsigned)kh * 512u + (unsigned)lc * 16u, voff2 = 2048u + (unsigned)kh * 256u + (unsigned)lc * 8u;
    f32x4 w0[NS], w1[NS];
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
    for (int kk = 0; kk < ntask; ++kk) {
        long long c0 = prof ? cvae_clock() : 0;
        const int t = kk / ntile, tl = kk % ntile, i = ti + tl * rts;
        const unsigned row0 = (unsigned)(t * p.Bp + i * 32), tile0 = row0 >> 5;
        // what the cell needs besides the matrix products does not depend on the recurrence: requested before the flag wait
        const int grow = i * 32 + row;
        const bool live = grow < p.B;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, msk = 0.f;
        float hold = tl == 0 ? hk0 : (tl == 1 ? hk1 : (tl == 2 ? hk2 : hk3));
        if (live) {
            const float* gip = p.gi + ((long)t * p.Bp + grow) * 3 * H;
            g0 = gip[j]; g1 = gip[H + j]; g2 = gip[2 * H + j];
            if (t == 0) {
                cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, g0, g1, g2);
                hold = p.hrow[(long)grow * H + j];
            }
            msk = p.gmask[((long)t * p.B + grow) * H + j];
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
        auto load_op = [&](int s) {     // step s < KPW: chunk s_lo + s of h, else chunk s_lo + s - KPW of o (plain first-touch loads)
            const unsigned so = ((unsigned)(s_lo + s % KPW) * tstride + tile0) * 2560u;
            if (s < KPW) {
                hc[2 * (s % RD)] = cvae_buf_load_f4(hxb, voff, so);
                hc[2 * (s % RD) + 1] = cvae_buf_load_f4(hxb, voff, so + 1024u);
                hb2[s % RD] = cvae_buf_load_f2(hxb, voff2, so);
            } else {
                hc[2 * (s % RD)] = cvae_buf_load_f4(oxb, voff, so);
                hc[2 * (s % RD) + 1] = cvae_buf_load_f4(oxb, voff, so + 1024u);
                hb2[s % RD] = cvae_buf_load_f2(oxb, voff2, so);
            }
        };
#pragma unroll
        for (int s = 0; s < RD; ++s) load_op(s);
        f32x16 a0 = cvae_zero16_t(), a1 = cvae_zero16_t(), a2 = cvae_zero16_t(), a3 = cvae_zero16_t();   // S0 | S1 | S2 (two chains)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const f32x4 l0 = hc[2 * (s % RD)], l1 = hc[2 * (s % RD) + 1];
            const f32x4 l2 = cvae_bf8x8_to_h8(hb2[s % RD]);
            const f32x4 w2 = *(const f32x4*)(w2w + s * 256);
            a0 = cvae_mfma_32x32x16_f16(l0, w0[s], a0);
            a1 = cvae_mfma_32x32x16_f16(l0, w1[s], a1);
            a2 = cvae_mfma_32x32x16_f16(l1, w1[s], a2);
            a3 = cvae_mfma_32x32x16_f16(l0, w2, a3);
            a1 = cvae_mfma_32x32x16_f16(l1, w0[s], a1);
            a2 = cvae_mfma_32x32x16_f16(l2, w0[s], a2);
            cvae_sched_fence();
            if (s + RD < NS) load_op(s + RD);
        }
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
            cvae_split3_f16b8(on, l0, l1, l2);
            ol[row * 8 + u] = l0;
            ol[256 + row * 8 + u] = l1;
            ((unsigned char*)(ol + 512))[row * 8 + u] = l2;
        }
        __syncthreads();
        if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        if (tid < 64) {   // wave 0 publishes both images into slot t+1 (write-through), drains, raises the octet's flag
            const unsigned so = ((unsigned)(c >> 1) * tstride + tile0 + (unsigned)(p.Bp >> 5)) * 2560u;
            const unsigned vo = (unsigned)(c & 1) * 512u + (unsigned)(tid & 31) * 16u, so01 = so + (unsigned)(tid >> 5) * 1024u;
            cvae_buf_store_f4_sc1(hxb, vo, so01, *(const f32x4*)(hl + tid * 8));
            cvae_buf_store_f4_sc1(oxb, vo, so01, *(const f32x4*)(ol + tid * 8));
            if (tid < 32) {
                cvae_buf_store_f2_sc1(hxb, 2048u + (unsigned)(c & 1) * 256u + (unsigned)tid * 8u, so, *(const f32x2*)(hl + 512 + tid * 4));
                cvae_buf_store_f2_sc1(oxb, 2048u + (unsigned)(c & 1) * 256u + (unsigned)tid * 8u, so, *(const f32x2*)(ol + 512 + tid * 4));
            }
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
