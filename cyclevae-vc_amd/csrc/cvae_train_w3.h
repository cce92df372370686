// k_train_bwd_steps_w3: the exact-operand reverse recurrence (k_train_bwd_steps_x3: same arithmetic, same hand-off) in the geometry
// that HALVES what a CU loads per step: block = 16 hidden units x ONE 16-row tile instead of 8 units x two tiles.
//
// Why.  Every block needs ALL 4H gate gradients of its rows per step (5 bytes each); the 8-unit kernel's CU pulls two tiles =
// 640 KB per step through a ~45 B/clk load path (14K of its 23.6K cycles per step, profiles/r05_notes_training.md).  A 16-unit
// block holds twice the columns, so the same 256 CUs cover 64 rows with ONE tile each: 320 KB per CU and step.  Its weights
// (32 columns x 4H x 5 B = 640 KB) only fit a CU with the STRUCTURAL ZEROS of [W_hh^T | F^T] dropped: the state columns read
// (drp, dzp, dq) and never dnp, the feedback columns (drp, dzp, dnp) and never dq (cvae_train_bwd.h) -- 32 x 3H x 5 B = 480 KB:
// l0 / l1 of 6 * GPW fragments per wave in registers (384 at H = 1024, less NL1 fragments whose l1 sits in LDS), the bf8 third
// limbs in LDS (96 KB).  Dropping the zero rows needs two operand packings, so the exchange is laid out per 32-unit producer
// group G as FOUR 32-k chunks of 2560 B (each { l0 [4 kq][16 rows][8 halves] | l1 likewise | l2 [4 kq][16 rows][8 B] }):
//     P0: (drp, dzp) of units 32G .. 32G+15 (k = 2*unit + comp)      -> state AND feedback columns
//     P1: (drp, dzp) of units 32G+16 .. 32G+31                       -> state AND feedback columns
//     Q : dq  of units 32G .. 32G+31 (k = unit)                      -> state columns only     (W_hn^T)
//     N : dnp of units 32G .. 32G+31                                 -> feedback columns only  (F_n^T)
// A producer block (16 units = half h of group G) writes chunk P_h and the kq halves {2h, 2h+1} of Q and N: every piece is a run of
// whole 128-byte lines that no second producer touches.  288 MFMAs per wave and step instead of 384, 320 instead of 640 KB.
// With one tile per block the hand-off latency is exposed (nothing else to run under it); passes of 128 rows give a block two
// tiles, which hide each other's.
#pragma once
#include <cvae_intrin.h>

// wbw[g][wave][Gl][frag]{ l0 [64 lanes][8 halves] | l1 likewise | l2 [64 lanes][8 bytes bf8] } (2560 B per fragment); lane
// (col = lane & 15, kq = lane >> 4) holds k = 8 kq + e of column col (output unit ko = 16 g + col) for producer group
// G = wave * GPW + Gl:
//   frag 0 (state, P0):    unit j = 32G + (k >> 1),      comp k & 1: W_hh[comp H + j][ko]
//   frag 1 (feedback, P0): the same rows of F (k_prep_ffold)
//   frag 2 / 3: the same for P1 (j = 32G + 16 + (k >> 1))
//   frag 4 (state, Q):     j = 32G + k: W_hh[2H + j][ko]        frag 5 (feedback, N): F[2H + j][ko]
// Blocks that share an XCD (workgroup b runs on XCD b % 8: g = x, x + 8, ...) walk their K share in ROTATED order: slot Gl of block g
// holds producer group wave * GPW + (Gl + (g >> 3)) % GPW.  All blocks of a row tile read the same lines at about the same time;
// in the same order every CU of an XCD would wait on the same L2 miss, rotated each line is fetched early by one CU and is an L2
// hit for the others.  (A permutation of the summation order only; baked into the weight image, so register indices stay static.)
__host__ __device__ __forceinline__ int cvae_w3_rot(int g, int GPW) { return (g >> 3) % GPW; }

__global__ void k_prep_wbw3(const float* F, const float* whh, float* wbw, int H, int GPW) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (g, wave, Gl, frag, lane, e)
    if (idx < (long)(H >> 4) * 4 * GPW * 6 * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long r = idx >> 9;
        const int frag = (int)(r % 6); r /= 6;
        const int Gl = (int)(r % GPW); r /= GPW;
        const int wave = (int)(r & 3), g = (int)(r >> 2);
        const int G = wave * GPW + (Gl + cvae_w3_rot(g, GPW)) % GPW, col = lane & 15, kq = lane >> 4, k = 8 * kq + e, ko = 16 * g + col;
        float v = 0.0f;
        if (32 * G < H) {
            const float* src = (frag & 1) ? F : whh;
            if (frag < 4) {
                const int j = 32 * G + 16 * (frag >> 1) + (k >> 1), comp = k & 1;
                v = src[(long)(comp * H + j) * H + ko];
            } else {
                v = src[(long)(2 * H + 32 * G + k) * H + ko];
            }
        }
        unsigned short l0, l1;
        unsigned char l2;
        cvae_split3_f16b8(v, l0, l1, l2);
        unsigned char* base = (unsigned char*)wbw + (idx >> 9) * 2560;
        ((unsigned short*)base)[lane * 8 + e] = l0;
        ((unsigned short*)(base + 1024))[lane * 8 + e] = l1;
        base[2048 + lane * 8 + e] = l2;
    }
}

#ifndef CVAE_BWDW_NL1
#define CVAE_BWDW_NL1 12      // H = 1024: fragments per wave (of 48) whose second limb lives in LDS
#endif
#ifndef CVAE_BWDW_RING
#define CVAE_BWDW_RING 5      // operand ring: 32-k chunks in flight per wave
#endif
// GPW: producer groups (32 units) per wave = H / 128; KW: waves with a share of K (H = 64: two); NL1: fragments per wave whose
// second limb is kept in LDS instead of registers (register budget: 8 * (2 * 6 GPW - NL1) for the weights)
template <int GPW, int KW, int NL1>
__global__ __launch_bounds__(256, 1) void k_train_bwd_steps_w3(TrainBwdParams p) {
    constexpr int NF = 6 * GPW, NS = 4 * GPW;
    constexpr int RD = NS < CVAE_BWDW_RING ? NS : CVAE_BWDW_RING;
    constexpr int RS = 36;
    constexpr float S1 = 1.0f / 2048.0f;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, NG = H >> 4, NG32 = H >> 5, nt16 = p.Bp >> 4;
    const int rts = p.rts;
    int g, ti;
    cvae_block_map((int)blockIdx.x, NG, rts, p.xmap != 0, g, ti);
    const bool kwave = wave < KW;
    float* red = (float*)CVAE_SMEM;                                   // [4 waves][16 rows][RS]: 16 state + 16 feedback sums
    unsigned char* pub = (unsigned char*)(red + 4 * 16 * RS);         // P chunk (2560 B) | Q half (1280 B) | N half (1280 B)
    float* w2l = (float*)(pub + 5120);                                // third limbs: [4 waves][NF][64 lanes][8 bytes (bf8)]
    float* w1l = w2l + 4 * NF * 128;                                  // second limbs of the first NL1 fragments: [4 waves][NL1][64 lanes][8 halves]
    const cvae_buf gb = cvae_make_buf(p.gx, (unsigned)((long)p.T * NG32 * nt16 * 10240));
    f32x4 w0[NF], w1[NF - NL1 > 0 ? NF - NL1 : 1];
    if (kwave) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float* src = p.wbk + (((long)g * 4 + wave) * GPW * 6 + f) * 640;
            w0[f] = *(const f32x4*)(src + lane * 4);
            if (f < NL1) *(f32x4*)(w1l + (wave * NL1 + f) * 256 + lane * 4) = *(const f32x4*)(src + 256 + lane * 4);
            else w1[f - NL1] = *(const f32x4*)(src + 256 + lane * 4);
            *(f32x2*)(w2l + (wave * NF + f) * 128 + lane * 2) = *(const f32x2*)(src + 512 + lane * 2);
        }
    }
    __syncthreads();
    const float* w2w = w2l + wave * NF * 128 + lane * 2;
    const float* w1w = w1l + wave * NL1 * 256 + lane * 4;
    const int rot = cvae_w3_rot(g, GPW);
    const int row = tid >> 4, u = tid & 15, k = 16 * g + u;          // every thread owns one (row, unit) of the tile
    // this launch covers the row tiles [tile_lo, tile_lo + tile_n) of the pass (tile_n = 0: all of them), see k_train_bwd_steps_x3
    const int tile_lo = p.tile_n > 0 ? p.tile_lo : 0, tile_n = p.tile_n > 0 ? p.tile_n : nt16;
    const int ntile = ti < tile_n ? (tile_n - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    float keep0 = 0.f, keep1 = 0.f;
    // tape values of the NEXT task, requested behind the last operand refill of the running one (k_train_fwd_steps_x3)
    float ntr = 0.f, ntz = 0.f, ntn = 0.f, ntq = 0.f, nthp = 0.f, ntmask = 0.f, ntdov = 0.f;
    auto prefetch_next = [&](int kn) {
        if (kn >= ntask) return;
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
        f32x4 s0 = (f32x4){0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0;      // state columns:    S0 | S1 | S2
        f32x4 f0 = s0, f1 = s0, f2 = s0;                               // feedback columns
        const int grow = i * 16 + row;
        const bool live = grow < p.B;
        const long rowi = (long)t * p.Bp + grow;
        const float tr = ntr, tz = ntz, tn = ntn, tq = ntq, thp = nthp, tmask = ntmask, tdov = ntdov;
        if (tt == 0) prefetch_next(kk + 1);        // (no operand stream in the first step)
        if (tt > 0) {
            if (kwave) {
                unsigned spins = 0;
                for (int q = 0; q < p.backoff; ++q) cvae_sleep_64();
                for (;;) {   // both halves of this wave's producer groups have published step t+1?
                    unsigned f = (unsigned)tt;
                    if (lane < 2 * GPW) f = cvae_atomic_load_agent(p.flags + (long)i * NG + 2 * wave * GPW + lane);
                    if (cvae_wave_all(f >= (unsigned)tt)) break;
                    cvae_sleep();
                    if (++spins > (1u << 22)) {
                        p.status[0] = 4;
                        break;
                    }
                }
            }
            cvae_compiler_fence();
            if (prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
            if (kwave) {
                f32x4 gc[2 * RD];
                f32x2 gc2[RD];
                auto load_g = [&](int s) {     // chunk s & 3 (P0, P1, Q, N) of producer group wave * GPW + (s >> 2), step t + 1
                    const unsigned so = (((unsigned)((t + 1) * NG32 + wave * GPW + ((s >> 2) + rot) % GPW) * (unsigned)nt16 + (unsigned)i) * 4u + (unsigned)(s & 3)) * 2560u;
                    gc[2 * (s % RD)] = cvae_buf_load_f4(gb, (unsigned)lane * 16u, so);
                    gc[2 * (s % RD) + 1] = cvae_buf_load_f4(gb, (unsigned)lane * 16u, so + 1024u);
                    gc2[s % RD] = cvae_buf_load_f2(gb, 2048u + (unsigned)lane * 8u, so);
                };
#pragma unroll
                for (int s = 0; s < RD; ++s) load_g(s);
                // one product on limb triples: S0 += a0 b0; S1 += a0 b1 + a1 b0; S2 += a1 b1 + a0 b2 + a2 b0
                auto frag_w1 = [&](int f) { return f < NL1 ? *(const f32x4*)(w1w + f * 256) : w1[f < NL1 ? 0 : f - NL1]; };
                auto frag_w2 = [&](int f) { return cvae_bf8x8_to_h8(*(const f32x2*)(w2w + f * 128)); };
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    const int Gl = s >> 2, c = s & 3;
                    if (c < 2) {            // P0 / P1: the same operand for the state and the feedback columns
                        const f32x4 l0 = gc[2 * (s % RD)], l1 = gc[2 * (s % RD) + 1], l2 = cvae_bf8x8_to_h8(gc2[s % RD]);
                        const int fs = Gl * 6 + 2 * c, ff = fs + 1;
                        const f32x4 sa_ = w0[fs], sb_ = frag_w1(fs), sc_ = frag_w2(fs);
                        const f32x4 fa_ = w0[ff], fb_ = frag_w1(ff), fc_ = frag_w2(ff);
                        s0 = cvae_mfma_16x16x32_f16(l0, sa_, s0);
                        f0 = cvae_mfma_16x16x32_f16(l0, fa_, f0);
                        s1 = cvae_mfma_16x16x32_f16(l0, sb_, s1);
                        f1 = cvae_mfma_16x16x32_f16(l0, fb_, f1);
                        s2 = cvae_mfma_16x16x32_f16(l1, sb_, s2);
                        f2 = cvae_mfma_16x16x32_f16(l1, fb_, f2);
                        s1 = cvae_mfma_16x16x32_f16(l1, sa_, s1);
                        f1 = cvae_mfma_16x16x32_f16(l1, fa_, f1);
                        s2 = cvae_mfma_16x16x32_f16(l0, sc_, s2);
                        f2 = cvae_mfma_16x16x32_f16(l0, fc_, f2);
                        s2 = cvae_mfma_16x16x32_f16(l2, sa_, s2);
                        f2 = cvae_mfma_16x16x32_f16(l2, fa_, f2);
                    } else if (c == 3) {    // Q (state columns) and N (feedback columns) together: two independent chains interleaved
                        const f32x4 q0 = gc[2 * ((s - 1) % RD)], q1 = gc[2 * ((s - 1) % RD) + 1], q2 = cvae_bf8x8_to_h8(gc2[(s - 1) % RD]);
                        const f32x4 n0 = gc[2 * (s % RD)], n1 = gc[2 * (s % RD) + 1], n2 = cvae_bf8x8_to_h8(gc2[s % RD]);
                        const int fs = Gl * 6 + 4, ff = fs + 1;
                        const f32x4 sa_ = w0[fs], sb_ = frag_w1(fs), sc_ = frag_w2(fs);
                        const f32x4 fa_ = w0[ff], fb_ = frag_w1(ff), fc_ = frag_w2(ff);
                        s0 = cvae_mfma_16x16x32_f16(q0, sa_, s0);
                        f0 = cvae_mfma_16x16x32_f16(n0, fa_, f0);
                        s1 = cvae_mfma_16x16x32_f16(q0, sb_, s1);
                        f1 = cvae_mfma_16x16x32_f16(n0, fb_, f1);
                        s2 = cvae_mfma_16x16x32_f16(q1, sb_, s2);
                        f2 = cvae_mfma_16x16x32_f16(n1, fb_, f2);
                        s1 = cvae_mfma_16x16x32_f16(q1, sa_, s1);
                        f1 = cvae_mfma_16x16x32_f16(n1, fa_, f1);
                        s2 = cvae_mfma_16x16x32_f16(q0, sc_, s2);
                        f2 = cvae_mfma_16x16x32_f16(n0, fc_, f2);
                        s2 = cvae_mfma_16x16x32_f16(q2, sa_, s2);
                        f2 = cvae_mfma_16x16x32_f16(n2, fa_, f2);
                    }
                    cvae_sched_fence();
                    // (the slot of chunk Q is refilled together with N's: both are consumed at c == 3)
                    if (c != 2 && c != 3 && s + RD < NS) load_g(s + RD);
                    if (c == 3) {
                        if (s - 1 + RD < NS) load_g(s - 1 + RD);
                        if (s + RD < NS) load_g(s + RD);
                    }
                    if (NS > RD && s == (RD >= 4 ? NS - 4 : NS - 1)) prefetch_next(kk + 1);     // behind the last operand refill
                }
                if (NS <= RD) prefetch_next(kk + 1);
            } else {
                prefetch_next(kk + 1);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            red[(wave * 16 + kq * 4 + q) * RS + lr] = s0[q] + (s1[q] + s2[q] * S1) * S1;
            red[(wave * 16 + kq * 4 + q) * RS + 16 + lr] = f0[q] + (f1[q] + f2[q] * S1) * S1;
        }
        if (prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        __syncthreads();
        {
            const bool k1 = ntile == 2 && (kk & 1);
            float v[4] = {0.f, 0.f, 0.f, 0.f}, dhz = 0.f;
            if (live) {
                float sa = 0.f, sb = 0.f;
#pragma unroll
                for (int w = 0; w < KW; ++w) {
                    sa += red[(w * 16 + row) * RS + u];
                    sb += red[(w * 16 + row) * RS + 16 + u];
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
                // cm 0 / 1: (drp, dzp) -> the P chunk, k = 2u + cm; cm 3: dq -> the Q half, cm 2: dnp -> the N half, k = u
                const int kl = cm < 2 ? 2 * u + cm : u;
                const int at = ((kl >> 3) * 16 + row) * 8 + (kl & 7);
                unsigned char* pb = pub + (cm < 2 ? 0 : (cm == 3 ? 2560 : 3840));
                const int l1o = cm < 2 ? 1024 : 512, l2o = cm < 2 ? 2048 : 1024;
                ((unsigned short*)pb)[at] = l0;
                ((unsigned short*)(pb + l1o))[at] = l1;
                (pb + l2o)[at] = l2;
            }
        }
        __syncthreads();
        if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        if (tid < 64) {   // wave 0 publishes the block's pieces of group g >> 1 (half h = g & 1), drains, raises the flag
            const unsigned base = ((unsigned)(t * NG32 + (g >> 1)) * (unsigned)nt16 + (unsigned)i) * 10240u;
            const unsigned h = (unsigned)(g & 1), soP = base + h * 2560u;
            cvae_buf_store_f4_sc1(gb, (unsigned)tid * 16u, soP, *(const f32x4*)(pub + tid * 16));
            cvae_buf_store_f4_sc1(gb, (unsigned)tid * 16u, soP + 1024u, *(const f32x4*)(pub + 1024 + tid * 16));
            cvae_buf_store_f2_sc1(gb, 2048u + (unsigned)tid * 8u, soP, *(const f32x2*)(pub + 2048 + tid * 8));
            const unsigned ln = (unsigned)(tid & 31), qn = (unsigned)(tid >> 5);          // lanes 0..31: Q, 32..63: N
            const unsigned soQ = base + (2u + qn) * 2560u;
            const unsigned char* src = pub + 2560 + qn * 1280;
            cvae_buf_store_f4_sc1(gb, h * 512u + ln * 16u, soQ, *(const f32x4*)(src + ln * 16));
            cvae_buf_store_f4_sc1(gb, 1024u + h * 512u + ln * 16u, soQ, *(const f32x4*)(src + 512 + ln * 16));
            cvae_buf_store_f2_sc1(gb, 2048u + h * 256u + ln * 8u, soQ, *(const f32x2*)(src + 1024 + ln * 8));
            cvae_drain_vmem();
            cvae_wave_barrier();
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * NG + g, (unsigned)(tt + 1));
        }
        if (prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
    }
    if (prof && tid == 0)
        for (int q = 0; q < 4; ++q) p.prof[4 + q] = pc[q];
}

// ------------------------------------------------------------------------------------------------------------------------
// k_train_fwd_steps_w3: the exact-operand FORWARD training recurrence (k_train_fwd_steps_x3h: same arithmetic, same exchange buffer,
// same dropout bit masks) in the same 16-unit x 16-row-tile geometry, for passes that give a block at least two tiles (the stacked
// rec || cv decoder pass: 128 rows at hu1024).  [W_hh | F/(1-p)] has structural zeros of its own -- the state path never feeds n_in,
// the feedback path never feeds n_h -- and with 16 units per block they fall on whole 16-column MFMA tiles: per 32-k chunk of h
//     state path    h          x  { W_hr, W_hz, W_hn }    -> r, z, n_h        (3 fragments)
//     feedback path bits * h   x  { F_r,  F_z,  F_n  }    -> r, z, n_in       (3 fragments)
// = 36 MFMAs per chunk where the 8-unit kernels issue 48 (two 16-column tiles x two paths x six products, a quarter of the columns
// zero), and the per-task costs that do not shrink with the tile (reduction, cell, publish) are paid per 16 units instead of per 8.
// Measured (MI355X, 128 rows x 80 frames, cycles per step of block 0): 25.8K -> 23.1K; with ONE tile per block (64 rows) the exposed
// hand-off cancels the gain (16.8K vs 16.0K for k_train_fwd_steps_x3h), so those passes stay on the 8-unit kernel.
// Exchange, dropout bits and slot 0: exactly k_train_fwd_steps_x3h's (k_train_x3h_slot0, k_train_x3h_maskbits with C32W = GPW): a
// 16-unit block publishes the kq halves {2h, 2h+1} (h = g & 1) of chunk g >> 1 -- runs of whole 128-byte lines.
// ------------------------------------------------------------------------------------------------------------------------
// w3w[g][wave][c][frag]{ l0 [64 lanes][8 halves] | l1 likewise | l2 [64 lanes][8 bytes bf8] }: lane (col = lane & 15: unit j = 16 g + col,
// kq = lane >> 4) holds k = 32 (wave GPW + c) + 8 kq + e;  frag 0..2: W_hh[frag H + j][k];  frag 3..5: F[(frag - 3) H + j][k] * oscale
__global__ void k_prep_wfw3(const float* F, const float* whh, float* w3w, int H, int GPW, float oscale) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (g, wave, c, frag, lane, e)
    if (idx < (long)(H >> 4) * 4 * GPW * 6 * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        long r = idx >> 9;
        const int frag = (int)(r % 6); r /= 6;
        const int c = (int)(r % GPW); r /= GPW;
        const int wave = (int)(r & 3), g = (int)(r >> 2);
        const int k = 32 * (wave * GPW + c) + 8 * (lane >> 4) + e, j = 16 * g + (lane & 15);
        float v = 0.0f;
        if (k < H) v = frag < 3 ? whh[(long)(frag * H + j) * H + k] : F[(long)((frag - 3) * H + j) * H + k] * oscale;
        unsigned short l0, l1;
        unsigned char l2;
        cvae_split3_f16b8(v, l0, l1, l2);
        unsigned char* base = (unsigned char*)w3w + (idx >> 9) * 2560;
        ((unsigned short*)base)[lane * 8 + e] = l0;
        ((unsigned short*)(base + 1024))[lane * 8 + e] = l1;
        base[2048 + lane * 8 + e] = l2;
    }
}

#ifndef CVAE_FWDW_NL1
#define CVAE_FWDW_NL1 11      // H = 1024: fragments per wave (of 48) whose second limb lives in LDS (what 160 KB leave room for)
#endif
#ifndef CVAE_FWDW_RING
#define CVAE_FWDW_RING 3      // operand ring: 32-k chunks of h in flight per wave (80 KB per CU and task: three cover the latency)
#endif
template <int GPW, int KW, int NL1>     // GPW 32-k chunks of h per wave, on the first KW waves; NL1 second limbs per wave in LDS
__global__ __launch_bounds__(256, 1) void k_train_fwd_steps_w3(TrainFwd3hParams p) {
    constexpr int NF = 6 * GPW;
    constexpr int RD = GPW < CVAE_FWDW_RING ? GPW : CVAE_FWDW_RING;
    constexpr int RS = 68;
    constexpr float S1 = 1.0f / 2048.0f;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, NG = H >> 4, n32 = H >> 5, nrt = p.Bp >> 4;
    const int rts = p.rts;
    int g, ti;
    cvae_block_map((int)blockIdx.x, NG, rts, p.xmap != 0, g, ti);
    const bool kwave = wave < KW;
    const int c_lo = wave * GPW;
    float* red = (float*)CVAE_SMEM;                                   // [4 waves][16 rows][RS]: r | z | n_h | n_in, 16 units each
    unsigned char* hl = (unsigned char*)(red + 4 * 16 * RS);          // publish image: l0 [2 kq][16 rows][8 halves] | l1 likewise | l2 [2 kq][16 rows][8 B]
    float* w2l = (float*)(hl + 1280);                                 // third limbs: [4 waves][NF][64 lanes][8 bytes (bf8)]
    float* w1l = w2l + 4 * NF * 128;                                  // second limbs of the first NL1 fragments: [4 waves][NL1][64 lanes][8 halves]
    const cvae_buf hb = cvae_make_buf(p.hx, (unsigned)((long)(p.T + 1) * n32 * nrt * 2560));
    f32x4 w0[NF], w1[NF - NL1 > 0 ? NF - NL1 : 1];
    if (kwave) {
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const float* src = p.w3 + (((long)g * 4 + wave) * GPW * 6 + f) * 640;
            w0[f] = *(const f32x4*)(src + lane * 4);
            if (f < NL1) *(f32x4*)(w1l + (wave * NL1 + f) * 256 + lane * 4) = *(const f32x4*)(src + 256 + lane * 4);
            else w1[f - NL1] = *(const f32x4*)(src + 256 + lane * 4);
            *(f32x2*)(w2l + (wave * NF + f) * 128 + lane * 2) = *(const f32x2*)(src + 512 + lane * 2);
        }
    }
    __syncthreads();
    const float* w2w = w2l + wave * NF * 128 + lane * 2;
    const float* w1w = w1l + wave * NL1 * 256 + lane * 4;
    const int row = tid >> 4, u = tid & 15, j = 16 * g + u;           // every thread owns one (row, unit) of the tile
    const float bhn = p.bhn[j];
    const int ntile = ti < nrt ? (nrt - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    float hkeep0 = 0.f, hkeep1 = 0.f;
    // what a task needs besides the exchanged state does not depend on the recurrence: requested ONE TASK AHEAD, behind the last
    // operand refill of the running task (k_train_fwd_steps_x3)
    float ng0 = 0.f, ng1 = 0.f, ng2 = 0.f, nmsk = 0.f;
    f32x2 nmraw = (f32x2){0.f, 0.f};
    auto prefetch_next = [&](int kn) {
        if (kn >= ntask) return;
        const int tn = kn / ntile, in_ = ti + (kn % ntile) * rts, grn = in_ * 16 + row;
        if (kwave) nmraw = *(const f32x2*)(p.mbits + ((((long)tn * nrt + in_) * 4 + wave) * 64 + lane) * 8);
        if (grn < p.B) {
            const float* gip = p.gi + ((long)tn * p.Bp + grn) * 3 * H;
            ng0 = gip[j]; ng1 = gip[H + j]; ng2 = gip[2 * H + j];
            nmsk = p.gmask[((long)tn * p.B + grn) * H + j];
        }
    };
    prefetch_next(0);
    long long pc[4] = {0, 0, 0, 0};
    const bool prof = p.prof && blockIdx.x == 0;
    for (int kk = 0; kk < ntask; ++kk) {
        long long c0 = prof ? cvae_clock() : 0;
        const int t = kk / ntile, tl = kk % ntile, i = ti + tl * rts;
        const int grow = i * 16 + row;
        const bool live = grow < p.B;
        float g0 = ng0, g1 = ng1, g2 = ng2;
        const float msk = nmsk;
        const f32x2 mraw = nmraw;
        const bool keep1 = ntile == 2 && tl == 1;
        float hold = keep1 ? hkeep1 : hkeep0;
        if (live && (t == 0 || ntile > 2)) hold = p.hrow[((long)t * p.Bp + grow) * H + j];      // row-major fp32 copy of slot t
        if (live && t == 0) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, g0, g1, g2);
        if (kwave && t > 0) {      // both halves of this wave's chunks of slot t are published?
            unsigned spins = 0;
            for (int q = 0; q < p.backoff; ++q) cvae_sleep_64();
            for (;;) {
                unsigned f = (unsigned)t;
                if (lane < 2 * GPW) f = cvae_atomic_load_agent(p.flags + (long)i * NG + 2 * c_lo + lane);
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
        f32x4 ar0 = (f32x4){0.f, 0.f, 0.f, 0.f}, ar1 = ar0, ar2 = ar0;      // r:    S0 | S1 | S2   (state + feedback path)
        f32x4 az0 = ar0, az1 = ar0, az2 = ar0;                               // z
        f32x4 ah0 = ar0, ah1 = ar0, ah2 = ar0;                               // n_h  (state path only)
        f32x4 ai0 = ar0, ai1 = ar0, ai2 = ar0;                               // n_in (feedback path only)
        if (kwave) {
            f32x4 hc[2 * RD];
            f32x2 hb2[RD];
            auto load_op = [&](int c) {     // chunk c_lo + c of h, slot t (plain first-touch loads)
                const unsigned so = (((unsigned)t * (unsigned)n32 + (unsigned)(c_lo + c)) * (unsigned)nrt + (unsigned)i) * 2560u;
                hc[2 * (c % RD)] = cvae_buf_load_f4(hb, (unsigned)lane * 16u, so);
                hc[2 * (c % RD) + 1] = cvae_buf_load_f4(hb, (unsigned)lane * 16u, so + 1024u);
                hb2[c % RD] = cvae_buf_load_f2(hb, 2048u + (unsigned)lane * 8u, so);
            };
#pragma unroll
            for (int c = 0; c < RD; ++c) load_op(c);
            auto frag_w1 = [&](int f) { return f < NL1 ? *(const f32x4*)(w1w + f * 256) : w1[f < NL1 ? 0 : f - NL1]; };
            auto frag_w2 = [&](int f) { return cvae_bf8x8_to_h8(*(const f32x2*)(w2w + f * 128)); };
#pragma unroll
            for (int c = 0; c < GPW; ++c) {
                const f32x4 l0 = hc[2 * (c % RD)], l1 = hc[2 * (c % RD) + 1], l2 = cvae_bf8x8_to_h8(hb2[c % RD]);
                const float mw = mraw[c >> 2];
                const cvae_m4 mk = cvae_expand_bits((__builtin_bit_cast(unsigned, mw) >> (8 * (c & 3))) & 0xffu);
                const f32x4 m0 = cvae_mask_h8(l0, mk), m1 = cvae_mask_h8(l1, mk), m2 = cvae_mask_h8(l2, mk);
                // one product on limb triples: S0 += a0 b0; S1 += a0 b1 + a1 b0; S2 += a1 b1 + a0 b2 + a2 b0 -- three gates interleaved
#define CVAE_W3_TRIPLE(A0, A1, A2, FR, FZ, FN, R0, R1, R2, Z0, Z1, Z2, N0, N1, N2)                         \
                {                                                                                          \
                    const f32x4 ra = w0[FR], rb = frag_w1(FR), rc = frag_w2(FR);                           \
                    const f32x4 za = w0[FZ], zb = frag_w1(FZ), zc = frag_w2(FZ);                           \
                    const f32x4 na = w0[FN], nb = frag_w1(FN), nc = frag_w2(FN);                           \
                    R0 = cvae_mfma_16x16x32_f16(A0, ra, R0); Z0 = cvae_mfma_16x16x32_f16(A0, za, Z0); N0 = cvae_mfma_16x16x32_f16(A0, na, N0); \
                    R1 = cvae_mfma_16x16x32_f16(A0, rb, R1); Z1 = cvae_mfma_16x16x32_f16(A0, zb, Z1); N1 = cvae_mfma_16x16x32_f16(A0, nb, N1); \
                    R2 = cvae_mfma_16x16x32_f16(A1, rb, R2); Z2 = cvae_mfma_16x16x32_f16(A1, zb, Z2); N2 = cvae_mfma_16x16x32_f16(A1, nb, N2); \
                    R1 = cvae_mfma_16x16x32_f16(A1, ra, R1); Z1 = cvae_mfma_16x16x32_f16(A1, za, Z1); N1 = cvae_mfma_16x16x32_f16(A1, na, N1); \
                    R2 = cvae_mfma_16x16x32_f16(A0, rc, R2); Z2 = cvae_mfma_16x16x32_f16(A0, zc, Z2); N2 = cvae_mfma_16x16x32_f16(A0, nc, N2); \
                    R2 = cvae_mfma_16x16x32_f16(A2, ra, R2); Z2 = cvae_mfma_16x16x32_f16(A2, za, Z2); N2 = cvae_mfma_16x16x32_f16(A2, na, N2); \
                }
                CVAE_W3_TRIPLE(l0, l1, l2, c * 6 + 0, c * 6 + 1, c * 6 + 2, ar0, ar1, ar2, az0, az1, az2, ah0, ah1, ah2)      // W_hh . h
                CVAE_W3_TRIPLE(m0, m1, m2, c * 6 + 3, c * 6 + 4, c * 6 + 5, ar0, ar1, ar2, az0, az1, az2, ai0, ai1, ai2)      // F/(1-p) . (bits * h)
#undef CVAE_W3_TRIPLE
                cvae_sched_fence();
                if (c + RD < GPW) load_op(c + RD);
                if (GPW > RD && c + RD == GPW - 1) prefetch_next(kk + 1);     // behind the last operand refill
            }
            if (GPW <= RD) prefetch_next(kk + 1);
        } else {
            prefetch_next(kk + 1);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float* rr = red + (wave * 16 + kq * 4 + q) * RS + lr;
            rr[0] = ar0[q] + (ar1[q] + ar2[q] * S1) * S1;
            rr[16] = az0[q] + (az1[q] + az2[q] * S1) * S1;
            rr[32] = ah0[q] + (ah1[q] + ah2[q] * S1) * S1;
            rr[48] = ai0[q] + (ai1[q] + ai2[q] * S1) * S1;
        }
        if (prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        __syncthreads();
        {
            float rg = 0.f, zg = 0.f, ng = 0.f, qq = 0.f, hn = 0.f, on = 0.f;
            if (live) {
                float sr = 0.f, sz = 0.f, sh = 0.f, si = 0.f;
#pragma unroll
                for (int w = 0; w < KW; ++w) {
                    const float* rr = red + (w * 16 + row) * RS + u;
                    sr += rr[0]; sz += rr[16]; sh += rr[32]; si += rr[48];
                }
                rg = cvae_sigmoid(g0 + sr);
                zg = cvae_sigmoid(g1 + sz);
                qq = sh + bhn;
                ng = tanhf(g2 + si + rg * qq);
                hn = ng + zg * (hold - ng);
                on = hn * msk;
            }
            if (keep1) hkeep1 = hn; else hkeep0 = hn;
            {   // the split happens here, once per value, by the thread that produced it
                unsigned short l0, l1;
                unsigned char l2;
                cvae_split3_f16b8(hn, l0, l1, l2);
                const int at = ((u >> 3) * 16 + row) * 8 + (u & 7);
                ((unsigned short*)hl)[at] = l0;
                ((unsigned short*)(hl + 512))[at] = l1;
                (hl + 1024)[at] = l2;
            }
            if (grow < p.Bp) {
                p.hrow[((long)(t + 1) * p.Bp + grow) * H + j] = hn;
                p.orow[((long)(t + 1) * p.Bp + grow) * H + j] = on;
                float* tp = p.tape + ((long)t * p.Bp + grow) * 4 * H + j;
                tp[0] = rg; tp[H] = zg; tp[2 * H] = ng; tp[3 * H] = qq;
            }
        }
        __syncthreads();
        if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        if (tid < 64) {   // wave 0: lanes 0..31 publish l0 (512 B), 32..63 l1, then 0..31 l2 (256 B) of slot t + 1; drain; flag
            const unsigned so = (((unsigned)(t + 1) * (unsigned)n32 + (unsigned)(g >> 1)) * (unsigned)nrt + (unsigned)i) * 2560u;
            const unsigned h2 = (unsigned)(g & 1), ln = (unsigned)(tid & 31), part = (unsigned)(tid >> 5);
            cvae_buf_store_f4_sc1(hb, part * 1024u + h2 * 512u + ln * 16u, so, *(const f32x4*)(hl + part * 512 + ln * 16));
            if (tid < 32) cvae_buf_store_f2_sc1(hb, 2048u + h2 * 256u + ln * 8u, so, *(const f32x2*)(hl + 1024 + ln * 8));
            cvae_drain_vmem();
            cvae_wave_barrier();
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * NG + g, (unsigned)(t + 1));
        }
        if (prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
    }
    if (prof && tid == 0)
        for (int q = 0; q < 4; ++q) p.prof[q] = pc[q];
}
