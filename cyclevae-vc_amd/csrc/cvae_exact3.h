// k_gru_steps_v6: the fused front-end + recurrence of one GRU_RNN pass (reference gru_vae.py:353-393) with every matrix
// product carried at FULL fp32 operand width on the f16 matrix pipe.
//
// Why a new decomposition.  k_gru_steps_v5 keeps a block's 64 gate columns x 1024 k of recurrent weights as (hi, lo) fp16
// pairs in registers: 256 of the 512 registers a lane has at one wave per SIMD, 494 in total with accumulators and operands
// (hipcc -Rpass-analysis: VGPRs 256 + AGPRs 238).  A third limb (the missing 2 bits of a 24-bit significand) is another 128
// registers per lane and does not fit; LDS is full of front-end weights.  v6 therefore halves the columns per block and
// doubles the rows: block = 8 hidden units (32 gate columns: r, z, n_in, n_h) x 32 batch rows, one v_mfma_f32_32x32x16_f16
// tile.  256 blocks at B = 64 as before (128 unit octets x 2 row tiles), the weights are replicated 2x instead of 4x over
// the chip, and three limbs per weight take 192 registers.
//
// Arithmetic.  Every fp32 operand x (weights, exchanged state, normalised input) is the exact sum of three halves,
// x = l0 + l1*s + l2*s^2, s = 2^-11 (cvae_split3_f16).  A product of two such sums is accumulated in fp32 as
//     S0 = a0.b0        S1 = a0.b1 + a1.b0        S2 = a1.b1 + a0.b2 + a2.b0        result = S0 + s*(S1 + s*S2)
// six v_mfma_f32_32x32x16_f16 per 16 k.  Products of halves are exact in fp32; the dropped terms (a1.b2, a2.b1, a2.b2) are
// below 2^-33 of the product, i.e. 2^-9 of an fp32 ulp: the result is the fp32-operand product to fp32 accumulation accuracy,
// where the fp32-input MFMA (v_mfma_f32_16x16x4_f32) delivers the same products at 2.7x the matrix-pipe time.
//
// Everything else follows v5: recurrent weights register-resident for the whole launch (the 4 waves split K), folded
// front-end weights in LDS as a lane-linear image, partial sums reduced through LDS, gates with one thread per (row, unit),
// every gate thread splits its new state value into limbs ONCE (two halves and a byte: the third limb has three significant
// bits and travels as bf8), wave 0 publishes the block's 32 rows x 40 B with
// write-through (sc1) stores and raises the octet's flag after draining them; consumers poll flags and stream operands
// through a ring of 8 16-k steps.  The exchange buffer and the input window are "tile-planar": whatever one load or store
// instruction touches is a run of whole cache lines that no second producer writes.  (Measured alternatives -- fp32
// exchanged and split in the consumer's registers, row-major buffers, all loads in flight at once -- are in
// profiles/r02_notes_exact3_kernel.md.)
#pragma once
#include <cvae_intrin.h>

struct Step6Params {
    const float* gx0;     // null, or [B][3H]: the frame-0 feedback correction, ready-made by the prologue (else cvae_t0_fix forms it here)
    const float* w2s;     // null, or the third limbs of the recurrent weights as bf8 bytes for the streamed form (W2S): [H/8][4 waves][KPW][64 lanes][8 B]
    float* hbuf;         // fp32 state, chunk-major [H/16][mtot][16]: slot 0 from the prologue, slots 1..T for k_outproj
    long mtot;
    float* hx;           // EXCHANGED state as limb triples, tile-planar, 2.5 KiB per 16-unit chunk and 32-row tile:
                         //   [H/16][mtot/32]{ l0 [kh][32 rows][8 halves] | l1 likewise | l2 [kh][32 rows][8 bytes, bf8 of l2*2^6] }
                         //   (kh = which 8 of the chunk's 16 units = which octet)
    const float* wrec3;  // [H/8][4 waves][KPW][3 limbs][64 lanes][8 halves]: B operands of the recurrent product
    const float* afold3; // [H/8][4 waves][KFW][3 limbs][64 lanes][8 halves]: B operands of the front-end = its LDS image
    const float* cfold;  // [3H]
    const float* bhn;    // [H]
    const float* xt;     // normalised, padded input as limb triples, tile-planar, 1280 B per 8-channel piece:
                         //   [Bp/32][Tp][Cp/8]{ l0 [32 rows][8 halves] | l1 likewise | l2 [32 rows][8 bytes] }
    int Tp, Cp;
    int B, Bp, H, T;
    unsigned* flags;     // [Bp/32][H/8], zeroed before launch: flags[i][c] = t <=> octet c of row tile i of h_t is published
    int* status;
    long long* prof;     // null or [blocks][4] cycle sums: front-end, flag wait, loads + MFMA, reduce + gates + publish
    const float* wyT;
    const float* dy;
    int Co;
    int rts;             // row tiles handled concurrently by the grid (grid = H/8 * rts blocks)
    int want_f32;        // write the fp32 copy of every new state to hbuf (the raw-projection / h_last paths read it; the limb projection does not)
    int exp;             // measurement-only switches (bit 0: operands from slot 0 every step; bit 1: XCD-aware block mapping; bits 2-3: reporting wave; bits 8..: poll back-off override)
    int backoff;         // x 64 cycles a block with ONE row tile sleeps before the first flag poll of a step (swept per front-end width, cvae_lib.hip)
};

// wrec3[c][wave][s][m][lane][e]: the folded recurrent weights (wrec2, fp32) of unit octet c as fp16 triples in the operand
// order of v_mfma_f32_32x32x16_f16: lane (col = lane & 31, kh = lane >> 5) holds k = 16*(wave*KPW + s) + 8 * kh + e
// of column col = 8*g + u (g: r, z, n_in, n_h; unit j = 8c + u).
__global__ void k_prep_wrec3(const float* wrec2, float* wrec3, int H, int KPW) {
    const int nch = H >> 4, NB = H >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (c, wave, s, lane, e)
    if (idx < (long)NB * 4 * KPW * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int s = (int)((idx >> 9) % KPW), wave = (int)(((idx >> 9) / KPW) & 3), c = (int)((idx >> 9) / KPW / 4);
        const int col = lane & 31, kh = lane >> 5, g = col >> 3, u = col & 7, j = 8 * c + u;
        const int k = 16 * (wave * KPW + s) + 8 * kh + e;
        float w = 0.0f;
        if (k < H) w = wrec2[((((long)(j >> 4) * 4 + g) * nch + (k >> 4)) * 16 + (j & 15)) * 16 + (k & 15)];
        unsigned short l0, l1, l2;
        cvae_split3_f16(w, l0, l1, l2);
        unsigned short* dst = (unsigned short*)wrec3 + ((((long)c * 4 + wave) * KPW + s) * 3) * 512 + lane * 8 + e;
        dst[0] = l0;
        dst[512] = l1;
        dst[1024] = l2;
    }
}

// w2s[c][wave][s][lane][e]: the THIRD limb of the same weights as a bf8 byte (of l2 * 2^6, cvae_split3_f16b8: bit-exact for
// |w| >= 2^-16, absolute error <= 2^-40 below), 512 B per (octet, wave, 16-k step): what the streamed form (W2S) reads every step
__global__ void k_prep_wrec3_l2b(const float* wrec2, unsigned char* w2s, int H, int KPW) {
    const int nch = H >> 4, NB = H >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one thread per (c, wave, s, lane, e)
    if (idx < (long)NB * 4 * KPW * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int s = (int)((idx >> 9) % KPW), wave = (int)(((idx >> 9) / KPW) & 3), c = (int)((idx >> 9) / KPW / 4);
        const int col = lane & 31, kh = lane >> 5, g = col >> 3, u = col & 7, j = 8 * c + u;
        const int k = 16 * (wave * KPW + s) + 8 * kh + e;
        float w = 0.0f;
        if (k < H) w = wrec2[((((long)(j >> 4) * 4 + g) * nch + (k >> 4)) * 16 + (j & 15)) * 16 + (k & 15)];
        unsigned short l0, l1;
        unsigned char l2;
        cvae_split3_f16b8(w, l0, l1, l2);
        w2s[idx] = l2;
    }
}

// afold3[c][wave][s][m][lane][e]: the folded front-end weights (afold [3H][Kfe], fp32) likewise: k = 16*(wave*KFW + s) +
// 8 * kh + e, column 8*g + u of gate g < 3 (the n_h column group takes no input term: zeros); zero beyond Kfe.
__global__ void k_prep_afold3l(const float* afold, float* afold3, int H, int Kfe, int KFW) {
    const int NB = H >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)NB * 4 * KFW * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
        const int s = (int)((idx >> 9) % KFW), wave = (int)(((idx >> 9) / KFW) & 3), c = (int)((idx >> 9) / KFW / 4);
        const int col = lane & 31, kh = lane >> 5, g = col >> 3, u = col & 7;
        const int k = 16 * (wave * KFW + s) + 8 * kh + e;
        const float w = (g < 3 && k < Kfe) ? afold[(long)(g * H + 8 * c + u) * Kfe + k] : 0.0f;
        unsigned short l0, l1, l2;
        cvae_split3_f16(w, l0, l1, l2);
        unsigned short* dst = (unsigned short*)afold3 + ((((long)c * 4 + wave) * KFW + s) * 3) * 512 + lane * 8 + e;
        dst[0] = l0;
        dst[512] = l1;
        dst[1024] = l2;
    }
}

__device__ __forceinline__ f32x16 cvae_zero16() {
    f32x16 z;
#pragma unroll
    for (int q = 0; q < 16; ++q) z[q] = 0.0f;
    return z;
}

// KPW = 16-k steps of the recurrent product per wave (H/64; H = 64: one), KFW = 16-k steps of the front-end per wave.
// LIMBS = 3: exact fp32 operands (six MFMAs per product).  LIMBS = 2: the same kernel on (l0, l1) pairs only -- 22-23 bit
// operands, three MFMAs per product, the arithmetic of k_gru_steps_v5 -- for H = 2048 (the hu2048 stress configuration), whose
// 32 columns x 2048 k per block fill 256 registers per lane with TWO limbs; a third one cannot be resident at that width.
// W2S (H = 2048 with three limbs): the third limbs of the recurrent weights are not resident (l0 and l1 of 32 columns x 2048 k fill
// 256 registers per lane) but STREAMED from L2 every step as bf8 bytes (64 KB per block: the 32 blocks of an XCD share 2 MB, which
// its L2 holds), through a ring of 4 steps; the operand rings shrink (5 recurrent steps, 3 front-end steps in flight) to make room
// for the third-limb registers and the fourth accumulator.
template <int KPW, int KFW, int LIMBS = 3, bool W2S = false>
__global__ __launch_bounds__(256, 1) void k_gru_steps_v6(Step6Params p) {
    constexpr int RS = 40;                             // row stride of the reduction buffer (conflict-free reads and writes)
    constexpr float S1 = 1.0f / 2048.0f;
#ifndef CVAE_V6_RD
#define CVAE_V6_RD 8
#endif
    constexpr int RD0 = W2S ? 5 : CVAE_V6_RD;
    constexpr int RD = KPW < RD0 ? KPW : RD0;          // operand ring: 16-k steps in flight per wave (8: swept 4..16 on MI355X)
    constexpr int RF = W2S && KFW > 3 ? 3 : KFW;       // front-end operands: all requested ahead (they land during the publish)
    constexpr int RW = KPW < 4 ? KPW : 4;              // W2S: third weight limbs in flight
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lc = lane & 31, kh = lane >> 5;
    const int H = p.H, NB = H >> 3, nrt = p.Bp >> 5;
    const int rts = p.rts;
    int c, ti;
    // exp bit 1 selects the XCD-aware mapping (cvae_block_map): measured on MI355X at B = 64 it halves the fabric-side fetch of a
    // launch and is 2 % SLOWER (4.46 vs 4.36 ms per chain, two A/B pairs): 32 CUs of an XCD then wait on the same first-touch misses
    // where the plain mapping spreads a tile's readers over all eight L2s.  The plain mapping stays the default.
    cvae_block_map((int)blockIdx.x, NB, rts, (p.exp & 2) != 0, c, ti);
    const int s_lo = wave * KPW;                       // this wave's first 16-k step = 16-unit chunk of h
    float* red = (float*)CVAE_SMEM;                    // [4 waves][32 rows][RS]
    float* hsh = red + 4 * 32 * RS;                    // [32 rows][8 units]
    unsigned short* hl = (unsigned short*)(hsh + 32 * 8);   // the publish image: l0, l1 [32 rows][8 halves] each, l2 [32 rows][8 bytes]
    float* wfl = hsh + 32 * 8 + 384;                   // [4 waves][KFW][LIMBS][64 lanes][8 halves]
    const int row = tid >> 3, u = tid & 7, j = 8 * c + u;
    const unsigned mtot = (unsigned)p.mtot;
    const cvae_buf hb = cvae_make_buf(p.hbuf, (unsigned)((long)(H >> 4) * p.mtot * 64));
    const cvae_buf xb_ = cvae_make_buf(p.hx, (unsigned)((long)(H >> 4) * p.mtot * 80));
    const unsigned tstride = mtot >> 5;                // 32-row tiles per chunk of the exchange buffer (2.5 KiB each)
    // operand of 16-k step s, lane (lc, kh): limbs 0, 1: 16 B at m*1024 + kh*512 + lc*16 of (chunk s, tile), limb 2: 8 B at
    // 2048 + kh*256 + lc*8 -- every load instruction reads one contiguous run (1 KiB, 1 KiB, 512 B)
    const unsigned voff = (unsigned)kh * 512u + (unsigned)lc * 16u, voff2 = 2048u + (unsigned)kh * 256u + (unsigned)lc * 8u;
    f32x4 w0[KPW], w1[KPW], w2[LIMBS == 3 && !W2S ? KPW : 1];
#pragma unroll
    for (int s = 0; s < KPW; ++s) {
        const float* src = p.wrec3 + ((((long)c * 4 + wave) * KPW + s) * 3) * 256 + lane * 4;
        w0[s] = *(const f32x4*)src;
        w1[s] = *(const f32x4*)(src + 256);
        if constexpr (LIMBS == 3 && !W2S) w2[s] = *(const f32x4*)(src + 512);
    }
    {   // this wave's slice of the front-end weight limbs -> LDS (the prepared image holds three planes per step)
        const float* src = p.afold3 + ((long)c * 4 + wave) * (KFW * 3 * 256);
        float* dst = wfl + wave * (KFW * LIMBS * 256);
#pragma unroll
        for (int s = 0; s < KFW; ++s)
#pragma unroll
            for (int m = 0; m < LIMBS; ++m)
                *(f32x4*)(dst + (s * LIMBS + m) * 256 + lane * 4) = *(const f32x4*)(src + (s * 3 + m) * 256 + lane * 4);
    }
    __syncthreads();
    const float* wfw = wfl + wave * (KFW * LIMBS * 256) + lane * 4;
    const float bhn = p.bhn[j];
    const float cf0 = p.cfold[j], cf1 = p.cfold[H + j], cf2 = p.cfold[2 * H + j];
    const int ntile = ti < nrt ? (nrt - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    long long pc[4] = {0, 0, 0, 0};
    // front-end operands of task k: frame t's window = octet pieces [t*Cp/8, +9*Cp/8) of the tile (k = 8*piece + e), three
    // limb planes of 512 B per piece; 16-k step s of this wave = pieces 2*(wave*KFW + s) + kh
    f32x4 x4[2 * RF];                   // ring slot s % RF: limbs 0, 1 of 16-k step s
    f32x2 x2[RF];                       //                   limb 2 (8 bytes)
    const unsigned char* xw = nullptr;
    auto set_x = [&](int k) {
        const int tt = k / ntile, ii = ti + (k % ntile) * rts;
        xw = (const unsigned char*)p.xt + (((long)ii * p.Tp + tt) * (p.Cp >> 3) + 2 * (wave * KFW) + kh) * 1280;
    };
    auto load_x = [&](int s) {
        x4[2 * (s % RF)] = *(const f32x4*)(xw + s * 2560 + lc * 16);
        x4[2 * (s % RF) + 1] = *(const f32x4*)(xw + s * 2560 + 512 + lc * 16);
        if constexpr (LIMBS == 3) x2[s % RF] = *(const f32x2*)(xw + s * 2560 + 1024 + lc * 8);
    };
    float hkeep0 = 0.f, hkeep1 = 0.f;   // h_{t-1} of this thread's (row, unit), per tile for up to two tiles per block
    const int backoff = (p.exp >> 8) ? (p.exp >> 8) - 1 : p.backoff;     // x 64 cycles before the first poll (exp: measurement override)
    if (ntask > 0) {
        set_x(0);
#pragma unroll
        for (int s = 0; s < RF; ++s) load_x(s);
    }
    unsigned fpre = 0u;                 // flags of the NEXT task, read at the end of the current one (several tiles per block)
    for (int k = 0; k < ntask; ++k) {
        long long c0 = !W2S && p.prof ? cvae_clock() : 0;      // (no phase counters in the W2S form: no registers to spare)
        const int t = k / ntile, i = ti + (k % ntile) * rts;
        const unsigned row0 = (unsigned)(t * p.Bp + i * 32), tile0 = row0 >> 5;   // (Bp is a multiple of 32)
        f32x16 a0 = cvae_zero16(), a1 = cvae_zero16(), a2 = cvae_zero16(), a3 = cvae_zero16();   // S0 | S1 | S2 (two chains)
#pragma unroll
        for (int s = 0; s < KFW; ++s) {     // front-end: independent of h, issued before the poll
            const f32x4 l0 = x4[2 * (s % RF)], l1 = x4[2 * (s % RF) + 1];
            const f32x4 b0 = *(const f32x4*)(wfw + (s * LIMBS + 0) * 256);
            const f32x4 b1 = *(const f32x4*)(wfw + (s * LIMBS + 1) * 256);
            a0 = cvae_mfma_32x32x16_f16(l0, b0, a0);
            a1 = cvae_mfma_32x32x16_f16(l0, b1, a1);
            if constexpr (LIMBS == 3) {
                const f32x4 l2 = cvae_bf8x8_to_h8(x2[s % RF]);
                const f32x4 b2 = *(const f32x4*)(wfw + (s * LIMBS + 2) * 256);
                a2 = cvae_mfma_32x32x16_f16(l1, b1, a2);
                if constexpr (W2S) a2 = cvae_mfma_32x32x16_f16(l0, b2, a2);      // (W2S: one S2 chain, 16 registers fewer)
                else a3 = cvae_mfma_32x32x16_f16(l0, b2, a3);
                a1 = cvae_mfma_32x32x16_f16(l1, b0, a1);
                a2 = cvae_mfma_32x32x16_f16(l2, b0, a2);
            } else {
                a2 = cvae_mfma_32x32x16_f16(l1, b0, a2);      // (second chain of the S1 sum)
            }
            cvae_sched_fence();
            if (s + RF < KFW) load_x(s + RF);          // refill the slot just consumed
        }
        if (!W2S && p.prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
        const bool pre_ok = ntile > 1 && k > 0 && cvae_wave_all(fpre >= (unsigned)t);
        if (t > 0 && !pre_ok) {   // the octets (two per 16-unit chunk) of this wave's K share are published?
            unsigned spins = 0;
            if (ntile == 1)
                for (int q = 0; q < backoff; ++q) cvae_sleep_64();
            for (;;) {
                unsigned f = (unsigned)t;
                if (lane < 2 * KPW && 2 * s_lo + lane < NB) f = cvae_atomic_load_agent(p.flags + (long)i * NB + 2 * s_lo + lane);
                if (cvae_wave_all(f >= (unsigned)t)) break;
                cvae_sleep();
                if (++spins > (1u << 22)) {
                    p.status[0] = 2;
                    break;
                }
            }
        }
        cvae_compiler_fence();                         // operand loads stay below the poll
        if (!W2S && p.prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        // The thread's gate inputs.  At frame 0 the feedback correction keeps 32 registers of loads in flight: the KPW = 32 variants
        // (H = 2048, 256 weight registers) have no room for them next to the operand ring and take them BEFORE it (they spilled
        // otherwise); the others issue them behind the ring's first loads, where their latency overlaps (3 % faster per launch).
        const int grow = i * 32 + row;
        const bool live = grow < p.B;
        const bool keep1 = ntile == 2 && (k & 1);
        float gxr = cf0, gxz = cf1, gxn = cf2, hold = keep1 ? hkeep1 : hkeep0;
        auto gate_inputs = [&]() {
            if (live) {
                if (t == 0 && p.gx0) {
                    const float* g0p = p.gx0 + (long)grow * 3 * H + j;
                    gxr += g0p[0]; gxz += g0p[H]; gxn += g0p[2 * H];
                } else if (t == 0 && p.dy) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, grow, gxr, gxz, gxn);
                if (t == 0) {               // slot 0 comes from the prologue
                    hold = cvae_buf_load_f1_sc1(hb, (unsigned)(row * 64 + (j & 15) * 4), ((unsigned)(j >> 4) * mtot + row0) * 64u);
                } else if (ntile > 2) {     // more than two tiles per block: re-read the own h from the exchange buffer (exact)
                    const unsigned so = ((unsigned)(j >> 4) * tstride + tile0) * 2560u, okh = (unsigned)((j >> 3) & 1);
                    const unsigned vo = okh * 512u + (unsigned)(row * 16 + (u >> 1) * 4);
                    const int sh = (u & 1) * 16;
                    const unsigned q0 = __builtin_bit_cast(unsigned, cvae_buf_load_f1_sc1(xb_, vo, so));
                    const unsigned q1 = __builtin_bit_cast(unsigned, cvae_buf_load_f1_sc1(xb_, vo, so + 1024u));
                    float third = 0.0f;
                    if constexpr (LIMBS == 3) {
                        const unsigned q2 = __builtin_bit_cast(unsigned, cvae_buf_load_f1_sc1(xb_, 2048u + okh * 256u + (unsigned)(row * 8 + (u >> 2) * 4), so));
                        third = cvae_bf8_to_f32((unsigned char)(q2 >> ((u & 3) * 8))) * (S1 / CVAE_L2_SCALE);
                    }
                    hold = cvae_f16_bits_to_f32((unsigned short)(q0 >> sh)) + (cvae_f16_bits_to_f32((unsigned short)(q1 >> sh)) + third) * S1;
                }
            }
        };
        if constexpr (KPW >= 32) gate_inputs();
        // Operand ring: RD 16-k steps in flight per wave.  All four waves see their flags at about the same time; if each
        // issued its whole K share at once the CU's memory pipe would serve them one wave after the other and the last wave
        // would start its MFMAs a full load phase late.  With a ring every wave gets its first operands early and the refills
        // (issued as a slot is consumed) interleave across waves at the rate the MFMAs eat them.  Plain (cached) loads: a
        // slot's lines are read here for the first time since the kernel started, so no cache can hold an older copy.
        f32x4 hc[2 * RD];                              // slot s % RD: limbs 0, 1 of 16-k step s
        f32x2 hb2[RD];                                 //                limb 2 (8 bytes)
        const unsigned tsel = (p.exp & 1) ? (unsigned)i : tile0;   // (measurement switch exp bit 0: read slot 0 every step)
        auto load_h = [&](int s) {
            const unsigned so = ((unsigned)(s_lo + s) * tstride + tsel) * 2560u;
            hc[2 * (s % RD)] = cvae_buf_load_f4(xb_, voff, so);
            hc[2 * (s % RD) + 1] = cvae_buf_load_f4(xb_, voff, so + 1024u);
            if constexpr (LIMBS == 3) hb2[s % RD] = cvae_buf_load_f2(xb_, voff2, so);
        };
        f32x2 w2r[W2S ? RW : 1];
        const float* w2base = W2S ? p.w2s + ((long)c * 4 + wave) * KPW * 128 + lane * 2 : nullptr;
        auto load_w2 = [&](int s) { w2r[s % RW] = *(const f32x2*)(w2base + s * 128); };
        if constexpr (W2S) {
#pragma unroll
            for (int s = 0; s < RW; ++s) load_w2(s);
        }
#pragma unroll
        for (int s = 0; s < RD; ++s) load_h(s);
        if constexpr (KPW < 32) gate_inputs();
#pragma unroll
        for (int s = 0; s < KPW; ++s) {
            const f32x4 l0 = hc[2 * (s % RD)], l1 = hc[2 * (s % RD) + 1];
            a0 = cvae_mfma_32x32x16_f16(l0, w0[s], a0);
            a1 = cvae_mfma_32x32x16_f16(l0, w1[s], a1);
            if constexpr (LIMBS == 3) {
                const f32x4 l2 = cvae_bf8x8_to_h8(hb2[s % RD]);
                a2 = cvae_mfma_32x32x16_f16(l1, w1[s], a2);
                if constexpr (W2S) a2 = cvae_mfma_32x32x16_f16(l0, cvae_bf8x8_to_h8(w2r[s % RW]), a2);
                else a3 = cvae_mfma_32x32x16_f16(l0, w2[s], a3);
                a1 = cvae_mfma_32x32x16_f16(l1, w0[s], a1);
                a2 = cvae_mfma_32x32x16_f16(l2, w0[s], a2);
            } else {
                a2 = cvae_mfma_32x32x16_f16(l1, w0[s], a2);
            }
            cvae_sched_fence();             // keeps the refill where it is written (a hoisted load has no register to land in)
            if (s + RD < KPW) load_h(s + RD);
            if constexpr (W2S) { if (s + RW < KPW) load_w2(s + RW); }
        }
        cvae_sched_fence();
        // next task's front-end operands.  xmode (measurement, exp bits 5-6): 0 = every wave requests them here (they land
        // under reduce + gates + publish), 1 = every wave after the publish, 2 = wave 0 (the publisher) after, the others here
        const int xmode = (p.exp >> 5) & 3;
        if (k + 1 < ntask) set_x(k + 1);
        if (k + 1 < ntask && (xmode == 0 || (xmode == 2 && wave != 0))) {
#pragma unroll
            for (int s = 0; s < RF; ++s) load_x(s);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            red[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh) * RS + lc] =
                LIMBS == 3 ? a0[q] + (a1[q] + (a2[q] + a3[q]) * S1) * S1 : a0[q] + (a1[q] + a2[q]) * S1;
        if (!W2S && p.prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        __syncthreads();
        {
            float hn_ = 0.0f;
            if (live) {
                float sg[4];
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    sg[g] = red[(0 * 32 + row) * RS + g * 8 + u] + red[(1 * 32 + row) * RS + g * 8 + u] +
                            red[(2 * 32 + row) * RS + g * 8 + u] + red[(3 * 32 + row) * RS + g * 8 + u];
                const float rg = cvae_sigmoid_fast(gxr + sg[0]);
                const float zg = cvae_sigmoid_fast(gxz + sg[1]);
                const float ng = cvae_tanh_fast(gxn + sg[2] + rg * (sg[3] + bhn));
                hn_ = ng + zg * (hold - ng);
            }
            if (keep1) hkeep1 = hn_; else hkeep0 = hn_;
            hsh[row * 8 + u] = hn_;
            unsigned short l0, l1;          // the split happens HERE, once per value, by the thread that produced it
            unsigned char l2;
            cvae_split3_f16b8(hn_, l0, l1, l2);
            hl[row * 8 + u] = l0;
            hl[256 + row * 8 + u] = l1;
            ((unsigned char*)(hl + 512))[row * 8 + u] = l2;
        }
        __syncthreads();
        if (tid < 64) {   // wave 0: split once, publish 3 limb pieces of 32 rows x 16 B (512 B runs, whole lines), slot t+1
            const unsigned so = ((unsigned)(c >> 1) * tstride + tile0 + (unsigned)(p.Bp >> 5)) * 2560u;
            // the LDS image, lane-linear: 64 pieces of 16 B (limb tid/32, row tid%32), then 32 pieces of 8 B (third limbs)
            cvae_buf_store_f4_sc1(xb_, (unsigned)(c & 1) * 512u + (unsigned)(tid & 31) * 16u, so + (unsigned)(tid >> 5) * 1024u,
                                  *(const f32x4*)(hl + tid * 8));
            if (LIMBS == 3 && tid < 32)
                cvae_buf_store_f2_sc1(xb_, 2048u + (unsigned)(c & 1) * 256u + (unsigned)tid * 8u, so, *(const f32x2*)(hl + 512 + tid * 4));
            cvae_drain_vmem();      // every lane's write-through stores have left ...
            cvae_wave_barrier();    // ... (all 64 lanes are this one wave) before lane 0 raises the flag
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * NB + c, (unsigned)(t + 1));
        } else if (tid < 128 && p.want_f32) {   // wave 1: the chunk-major fp32 copy for the raw projection / h_last (read after this launch: plain stores)
            const int l = tid - 64, r = l >> 1, half = l & 1;
            *(f32x4*)(p.hbuf + (((long)(c >> 1) * p.mtot + row0 + p.Bp + r) * 16 + (c & 1) * 8 + half * 4)) =
                *(const f32x4*)(hsh + r * 8 + half * 4);
        }
        if (k + 1 < ntask && (xmode == 1 || (xmode == 2 && wave == 0))) {
#pragma unroll
            for (int s = 0; s < RF; ++s) load_x(s);
        }
        if (ntile > 1 && k + 1 < ntask) {   // (behind wave 0's publish, so its drain never waits for this load)
            const int kn = k + 1, tn = kn / ntile, in_ = ti + (kn % ntile) * rts;
            fpre = (unsigned)tn;
            if (tn > 0 && lane < 2 * KPW && 2 * s_lo + lane < NB)
                fpre = cvae_atomic_load_agent(p.flags + (long)in_ * NB + 2 * s_lo + lane);
        }
        if (!W2S && p.prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
    }
    if (p.prof && tid == 64 * ((p.exp >> 2) & 3))    // measurement: exp bits 2-3 pick the reporting wave
        for (int q = 0; q < 4; ++q) p.prof[(long)blockIdx.x * 4 + q] = pc[q];
}

// ------------------------------------------------------------------------------------------------------------------------
// Projection of a v6 pass: trj_out = scale_out(out_1(h_t)) (or the clamped out_1(h_t)) for every frame, reading the state where
// the recurrent kernel left it -- the limb triples of the exchange buffer -- and multiplying it with the (scale_out-folded)
// projection matrix in the same exact six-product form.  Block = (32-row tile of one slot, 32-column tile); the 4 waves split K
// and meet in LDS.  Reference: gru_vae.py:371/393 (out_1), :402-406 (scale_out), :408-412 (clamp).
// ------------------------------------------------------------------------------------------------------------------------
struct Out6Params {
    const float* hx;     // limb triples, [H/16][mtot/32]{ l0 | l1 | l2 } (2560 B per chunk and tile), slot s at tile s*Bp/32
    long mtot;
    const float* wo3;    // [Cop32/32][H/16][3 limbs][64 lanes][8 halves]: B operands (column 32n + lane&31, k = 16s + 8*(lane>>5) + e)
    const float* bo2;    // [Cop]
    int H, Bp, T, B, ncell, Co, clamp_from;
    float clamp_min;     // the floor of out[c >= clamp_from]: ln(1e-6) (gru_vae.py:412) or the Laplace variant's log-scale floor (:417)
    float* out[CVAE_MAX_CELLS];       // per cell [B][T][Co]
};

// wo3[n][s][m][lane][e] from wo2 [Cop][H] (scale_out . out_1 or out_1; rows >= Cop are zero)
__global__ void k_prep_wo3(const float* wo2, float* wo3, int H, int Cop, int NT) {
    const int nk = H >> 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)NT * nk * 512) {
        const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63), s = (int)((idx >> 9) % nk), n = (int)((idx >> 9) / nk);
        const int col = 32 * n + (lane & 31), k = 16 * s + 8 * (lane >> 5) + e;
        const float w = col < Cop ? wo2[(long)col * H + k] : 0.0f;
        unsigned short l0, l1, l2;
        cvae_split3_f16(w, l0, l1, l2);
        unsigned short* dst = (unsigned short*)wo3 + (((long)n * nk + s) * 3) * 512 + lane * 8 + e;
        dst[0] = l0;
        dst[512] = l1;
        dst[1024] = l2;
    }
}

template <int KPW>
__global__ __launch_bounds__(256) void k_outproj_v6(Out6Params p) {
    constexpr float S1 = 1.0f / 2048.0f;
    constexpr int RS = 36;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lc = lane & 31, kh = lane >> 5;
    const int nk = p.H >> 4, nt32 = p.Bp >> 5;
    const int n = blockIdx.y;                                  // column tile
    const long tile = (long)nt32 + blockIdx.x;                 // tile index counted from slot 0: slots 1..T
    float* red = (float*)CVAE_SMEM;                            // [4 waves][32 rows][RS]
    const long tstride = p.mtot >> 5;
    f32x16 a0 = cvae_zero16(), a1 = cvae_zero16(), a2 = cvae_zero16(), a3 = cvae_zero16();
    constexpr int RD = KPW < 4 ? KPW : 4;                      // steps in flight
    f32x4 ha[2 * RD], wb[3 * RD];
    f32x2 hb[RD];
    auto load = [&](int s) {
        const int sg = wave * KPW + s;
        const unsigned char* hp = (const unsigned char*)p.hx + ((long)sg * tstride + tile) * 2560;
        ha[2 * (s % RD)] = *(const f32x4*)(hp + kh * 512 + lc * 16);
        ha[2 * (s % RD) + 1] = *(const f32x4*)(hp + 1024 + kh * 512 + lc * 16);
        hb[s % RD] = *(const f32x2*)(hp + 2048 + kh * 256 + lc * 8);
        const float* wp = p.wo3 + (((long)n * nk + sg) * 3) * 256 + lane * 4;
#pragma unroll
        for (int m = 0; m < 3; ++m) wb[3 * (s % RD) + m] = *(const f32x4*)(wp + m * 256);
    };
    const bool has_k = wave * KPW < nk;
    if (has_k) {
#pragma unroll
        for (int s = 0; s < RD; ++s) load(s);
#pragma unroll
        for (int s = 0; s < KPW; ++s) {
            const f32x4 l0 = ha[2 * (s % RD)], l1 = ha[2 * (s % RD) + 1], l2 = cvae_bf8x8_to_h8(hb[s % RD]);
            const f32x4 w0 = wb[3 * (s % RD)], w1 = wb[3 * (s % RD) + 1], w2 = wb[3 * (s % RD) + 2];
            a0 = cvae_mfma_32x32x16_f16(l0, w0, a0);
            a1 = cvae_mfma_32x32x16_f16(l0, w1, a1);
            a2 = cvae_mfma_32x32x16_f16(l1, w1, a2);
            a3 = cvae_mfma_32x32x16_f16(l0, w2, a3);
            a1 = cvae_mfma_32x32x16_f16(l1, w0, a1);
            a2 = cvae_mfma_32x32x16_f16(l2, w0, a2);
            cvae_sched_fence();
            if (s + RD < KPW) load(s + RD);
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q)
        red[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh) * RS + lc] = a0[q] + (a1[q] + (a2[q] + a3[q]) * S1) * S1;
    __syncthreads();
    {   // 256 threads: row tid / 8, columns (tid % 8) + 8j of this tile
        const int r = tid >> 3;
        const long m = (long)blockIdx.x * 32 + r;              // row counted from slot 1
        const int t = (int)(m / p.Bp), b = (int)(m - (long)t * p.Bp);
        if (b < p.ncell * p.B) {
            const int cell = b / p.B, bb = b - cell * p.B;
            float* orow = p.out[cell] + ((long)bb * p.T + t) * p.Co;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int cl = (tid & 7) + 8 * jj, col = 32 * n + cl;
                if (col < p.Co) {
                    float v = red[(0 * 32 + r) * RS + cl] + red[(1 * 32 + r) * RS + cl] + red[(2 * 32 + r) * RS + cl] +
                              red[(3 * 32 + r) * RS + cl] + p.bo2[col];
                    if (p.clamp_from >= 0 && col >= p.clamp_from) v = fmaxf(v, p.clamp_min);
                    orow[col] = v;
                }
            }
        }
    }
}

// cvae_selftest_occupy: workgroups that hold `lds_words` of LDS (touched, so that the allocation is real) and stay on their CU
// for `cycles` shader cycles: the CU-side load of the residency tests -- the all-resident recurrent kernels are launched plainly
// (cvae_launch_coop), so a grid may have to wait for LDS / wave slots that another stream's kernels hold.
__global__ void k_selftest_occupy(long long cycles, int lds_words) {
    float* occupy_lds = (float*)CVAE_SMEM;
    for (int i = threadIdx.x; i < lds_words; i += blockDim.x) occupy_lds[i] = (float)i;
    __syncthreads();
    const long long t0 = cvae_clock();
    float acc = 0.f;
    int guard = 0;
    while (cvae_clock() - t0 < cycles && ++guard < (1 << 28)) {
        acc += occupy_lds[(threadIdx.x * 33 + guard) % (lds_words > 0 ? lds_words : 1)];
        cvae_sleep();
    }
    if (acc == -1.0f) occupy_lds[0] = acc;      // (keeps the loop)
}

// Self-test of the limb transport (cvae_selftest_limbs): split eight values the way a producer does (two halves and a bf8 byte
// each), decode the bytes the way a consumer does (cvae_bf8x8_to_h8), rebuild x' = l0 + l1/2^11 + l2/2^22.
__global__ void k_selftest_limbs(const float* x, float* y, long n) {
    const long i0 = 8 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
    if (i0 + 8 > n) return;
    unsigned short l0[8], l1[8];
    unsigned char b[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cvae_split3_f16b8(x[i0 + e], l0[e], l1[e], b[e]);
    unsigned w0 = 0, w1 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w0 |= (unsigned)b[e] << (8 * e);
        w1 |= (unsigned)b[4 + e] << (8 * e);
    }
    const f32x2 raw = (f32x2){__builtin_bit_cast(float, w0), __builtin_bit_cast(float, w1)};
    const f32x4 h8 = cvae_bf8x8_to_h8(raw);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float pair = h8[e >> 1];
        const unsigned bits = __builtin_bit_cast(unsigned, pair);
        const unsigned short hb = (unsigned short)((e & 1) ? bits >> 16 : bits & 0xffffu);
        y[i0 + e] = cvae_f16_bits_to_f32(l0[e]) + cvae_f16_bits_to_f32(l1[e]) * (1.0f / 2048.0f) +
                    cvae_f16_bits_to_f32(hb) * (1.0f / 4194304.0f);
    }
}
