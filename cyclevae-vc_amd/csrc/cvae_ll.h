// k_gru_steps_ll: the T dependent GRU steps of a pass with at most THREE batch rows (a single utterance, the encoder pair or the
// decoder triple of decode_gru-cyclevae_gauss.py:302-323) -- reference GRU_RNN.forward's Python loop, gru_vae.py:391-394.
//
// At this size a step is nothing but the hand-off: 3 x 1024 state values go from 256 producers to 256 consumers.  The dataflow
// kernels for larger batches (publish -> drain -> flag store -> flag poll -> operand load) spend three memory round trips per
// step on it; here a unit's state travels as ONE 16-byte word (h of rows 0..2, step tag) written by one dwordx4 write-through
// store, and consumers poll the data words themselves until every tag says "this step": one store and (normally) one load round
// trip per step, no drain, no flags.  Two slots (step parity) suffice: a block can only publish step t+1 after it has read every
// unit of step t, i.e. after every block has finished reading step t-1.
//
// Tags are (launch nonce, step), so the words need no zeroing.  Measured alternatives (profiles/r02_notes_small_batch.md): a block
// vote in the poll loop, and a two-level poll (one block per XCD polls memory and re-publishes into the XCD's L2) -- both slower.
//
// Arithmetic: plain fp32 FMAs on fp32 operands (the reference's own arithmetic, no limb splitting): block = 4 hidden units
// (16 gate columns: r, z, n_x, n_h), grid = H/4 blocks, thread (quad Q, member g) loads the words of units 16Q+4g..+3, the quad
// swaps them by DPP so that member g holds h[16Q .. 16Q+15] of every row, and accumulates gate g of the block's 4 units over
// those 16 k (64 register-resident weights); the 64 quads' partial sums are added through LDS in fixed order.  The input-side
// gate pre-activations come from one GEMM before the loop (gx, as for k_gru_steps_v2).
#pragma once
#include <cvae_intrin.h>

struct StepLLParams {
    float* hbuf;          // chunk-major fp32 trajectory [H/16][mtot][16]: slot 0 from the prologue, slots 1..T written here
    long mtot;
    float* xbuf;          // [2 slots][H units][4]: (h row 0, row 1, row 2, tag)
    int backoff;          // x 64 cycles of sleep between a step's publish and its first poll
    const unsigned* nonce_src;   // the workspace's launch counter (the pass prologue increments it): tags are
                                 // (counter & 0xffff) << 16 | step, T < 65536.  Per WORKSPACE, because the stale words a launch can
                                 // meet are those of the previous launch on the same workspace, whose counter differs by one
    const float* wrec2;   // [H/16][4][H/16][16][16]
    const float* gx;      // [B][Tp][3H]
    long gx_bstride;
    const float* bhn;
    int B, Bp, H, T;
    int* status;
    long long* prof;      // null, or 4 cycle sums of block 0: poll, FMAs, reduce, cell + publish
    int* dbg;             // with prof: workspace status word 1 += failed poll iterations of every block's wave 0
    const float* wyT;
    const float* dy;
    int Co;
};

template <int NR>   // rows carried (1..3); p.B <= NR
__global__ __launch_bounds__(256, 1) void k_gru_steps_ll(StepLLParams p) {
    constexpr int NO = 16 * NR, RS = NO + 1;                            // outputs per block: (gate, row, unit)
    const unsigned nonce = (*p.nonce_src & 0xffffu) << 16;
    const int tid = threadIdx.x, Q = tid >> 2, g = tid & 3, H = p.H, nch = H >> 4;
    const int jg = blockIdx.x >> 2, u0 = 4 * (blockIdx.x & 3);          // this block's units 16*jg + u0 .. +3
    float* red0 = (float*)CVAE_SMEM;                                    // [2 (step parity)][64 quads][RS]
    float* part = red0 + 2 * 64 * RS;                                   // [4 waves][NO]
    const cvae_buf xb = cvae_make_buf(p.xbuf, 2u * (unsigned)H * 16u);
    // Word q of a thread is unit 256*wave + 64*q + lane: a wave's load instruction reads ONE contiguous KiB (lane-strided words
    // made every instruction touch 64 lines).  A quad therefore owns the 16 units k(js, q) = 256*wave + 64*q + 4*(lane/4) + js.
    const int wave = tid >> 6, lane = tid & 63, kbase = 256 * wave + 4 * (lane >> 2);
    float w[4][4][4];                                                   // gate g: [unit][member js][word q] = W[g][unit][k(js, q)]
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kbase + 64 * q + js;
                w[u][js][q] = k < H ? p.wrec2[(((long)jg * 4 + g) * nch + (k >> 4)) * 256 + (u0 + u) * 16 + (k & 15)] : 0.f;
            }
    // cell threads.  NR = 1: lane 16*row + 4*unit of wave 0 (the quad's other lanes hold that (row, unit)'s z, n_x, n_h sums);
    // NR > 1: tid = 4*row + unit (one row: wave 0 finishes the step alone, one barrier per step: 2.31 -> 2.25 us; with two rows that
    // form measured slower, 3.35 -> 3.9 ms on the stage-6 pair).
    constexpr bool ONE_STAGE = NR == 1;
    const int crow = ONE_STAGE ? tid >> 4 : tid >> 2, cu = ONE_STAGE ? (tid >> 2) & 3 : tid & 3, j = 16 * jg + u0 + cu;
    const bool cell = (ONE_STAGE ? tid < 16 * NR && (tid & 3) == 0 : tid < 4 * NR) && crow < p.B;
    float hold = 0.f, bhn = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (cell) {
        hold = p.hbuf[((long)jg * p.mtot + crow) * 16 + u0 + cu];
        bhn = p.bhn[j];
        const float* gxp = p.gx + (long)crow * p.gx_bstride;
        g0 = gxp[j]; g1 = gxp[H + j]; g2 = gxp[2 * H + j];
        if (p.dy) cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, crow, g0, g1, g2);
    }
    long long pc[4] = {0, 0, 0, 0};
    unsigned iters_total = 0;
    const bool prof = p.prof && blockIdx.x == 0;
    for (int t = 0; t < p.T; ++t) {
        long long c0 = prof ? cvae_clock() : 0;
        // the input-side pre-activations of the NEXT step do not depend on the recurrence: requested FIRST, in front of the back-off
        // sleep and the poll, so that they are in flight while the step waits anyway.  (Requested behind the poll, as rounds 2-3
        // had it, the compiler's s_waitcnt vmcnt(0) at the head of the FMA phase -- the merge point of the t == 0 path -- made
        // wave 0 sit out their whole memory round trip every step, with the other three waves at the barrier behind it.)
        float n0 = 0.f, n1 = 0.f, n2 = 0.f;
        if (cell && t + 1 < p.T) {
            const float* gxp = p.gx + (long)crow * p.gx_bstride + (long)(t + 1) * 3 * H;
            n0 = gxp[j]; n1 = gxp[H + j]; n2 = gxp[2 * H + j];
        }
        float hv[NR][4];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) hv[r][q] = 0.f;
        if (t == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 256 * wave + 64 * q + lane;
                if (k < H)
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (r < p.B) hv[r][q] = p.hbuf[((long)(k >> 4) * p.mtot + r) * 16 + (k & 15)];
            }
        } else {
            const unsigned so = (unsigned)(t & 1) * (unsigned)H * 16u, vo = (unsigned)(256 * wave + lane) * 16u;
            // every WAVE polls the 16 lines of its own K share and goes on to its FMAs as soon as they are complete: no block
            // barrier inside the loop (tools/mb/mb_poll.hip: a block vote per iteration costs as much as the loads)
            unsigned spins = 0;
            for (int q = 0; q < p.backoff; ++q) cvae_sleep_64();
            for (;;) {
                cvae_compiler_fence();      // (the optimizer hoists even "volatile" buffer loads out of a loop without a barrier:
                                            //  tools/mb/mb_tear.hip polled once per lane until it had one)
                bool ok = true;
                f32x4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (256 * wave + 64 * q < H) v[q] = cvae_buf_poll_f4(xb, vo + 1024u * q, so);     // (wave-uniform; H % 64 == 0)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (256 * wave + 64 * q < H) {
#pragma unroll
                        for (int r = 0; r < NR; ++r) hv[r][q] = v[q][r];
                        const float tag = v[q][3];  // (a temporary: bit_cast of a vector ELEMENT reads element 0 with this clang)
                        ok = ok && __builtin_bit_cast(unsigned, tag) == nonce + (unsigned)t;
                    }
                if (cvae_wave_all(ok)) break;
                ++iters_total;
                if (++spins > (1u << 20)) {
                    p.status[0] = 6;
                    break;
                }
            }
        }
        if (prof) { const long long c1 = cvae_clock(); pc[0] += c1 - c0; c0 = c1; }
        float* red = red0 + (ONE_STAGE ? (t & 1) * 64 * RS : 0);      // (parity: wave 0 may still be reading the previous step's sums)
        float acc[NR][4];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[r][u] = 0.f;
#pragma unroll
        for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const float hb = js == 0 ? cvae_quad_bcast<0>(hv[r][q]) : js == 1 ? cvae_quad_bcast<1>(hv[r][q])
                                   : js == 2 ? cvae_quad_bcast<2>(hv[r][q]) : cvae_quad_bcast<3>(hv[r][q]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[r][u] = __builtin_fmaf(hb, w[u][js][q], acc[r][u]);   // (weights of k >= H are 0)
                }
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) red[Q * RS + g * 4 * NR + r * 4 + u] = acc[r][u];
        if (prof) { const long long c1 = cvae_clock(); pc[1] += c1 - c0; c0 = c1; }
        __syncthreads();
        if (ONE_STAGE) {
            // wave 0 finishes the step alone: lane = slice*16NR + 16*row + 4*unit + gate sums its slice of the 64 quads, the slices
            // are combined by two (one) cross-row shuffles, a quad then holds the four gate sums of its (row, unit)
            if (tid < 64) {
                constexpr int NOL = 16 * NR, NSL = 64 / NOL, QPS = 64 / NSL;
                const int sl = lane / NOL, wi = lane % NOL, a = wi & 3;
                const int o = a * 4 * NR + (wi >> 4) * 4 + ((wi >> 2) & 3);
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < QPS; ++i) sum += red[(sl * QPS + i) * RS + o];
                if (NSL == 4) sum += cvae_shfl(sum, lane ^ 16);
                if (NSL >= 2) sum += cvae_shfl(sum, lane ^ 32);
                const float s1 = cvae_quad_bcast<1>(sum), s2 = cvae_quad_bcast<2>(sum), s3 = cvae_quad_bcast<3>(sum);
                if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
                float hn = 0.f;
                if (cell) {
                    const float rg = cvae_sigmoid_fast(g0 + sum);
                    const float zg = cvae_sigmoid_fast(g1 + s1);
                    const float ng = cvae_tanh_fast(g2 + s2 + rg * (s3 + bhn));
                    hn = ng + zg * (hold - ng);
                    hold = hn;
                    p.hbuf[((long)jg * p.mtot + (long)(t + 1) * p.Bp + crow) * 16 + u0 + cu] = hn;
                }
                const float h1 = NR > 1 ? cvae_shfl(hn, (lane + 16) & 63) : 0.f;
                if (lane < 16 && (lane & 3) == 0 && t + 1 < p.T) {     // row 0's cell lanes publish their unit's word
                    const f32x4 wv = (f32x4){hn, h1, 0.f, __builtin_bit_cast(float, nonce + (unsigned)(t + 1))};
                    cvae_buf_store_f4_sc1(xb, (unsigned)(16 * jg + u0 + cu) * 16u, (unsigned)((t + 1) & 1) * (unsigned)H * 16u, wv);
                }
            }
        } else {
        {
            const int o = tid & 63, s = tid >> 6;
            if (o < NO) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) sum += red[(16 * s + i) * RS + o];
                part[s * NO + o] = sum;
            }
        }
        __syncthreads();
        if (prof) { const long long c1 = cvae_clock(); pc[2] += c1 - c0; c0 = c1; }
        if (tid < 64) {   // wave 0: lanes 4*row + unit finish the cell, lanes 0..3 publish their unit's word
            float hn = 0.f;
            if (cell) {
                float s[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int o = a * 4 * NR + crow * 4 + cu;
                    s[a] = (part[o] + part[NO + o]) + (part[2 * NO + o] + part[3 * NO + o]);
                }
                const float rg = cvae_sigmoid_fast(g0 + s[0]);
                const float zg = cvae_sigmoid_fast(g1 + s[1]);
                const float ng = cvae_tanh_fast(g2 + s[2] + rg * (s[3] + bhn));
                hn = ng + zg * (hold - ng);
                hold = hn;
                p.hbuf[((long)jg * p.mtot + (long)(t + 1) * p.Bp + crow) * 16 + u0 + cu] = hn;
            }
            const float h0 = cvae_shfl(hn, cu), h1 = NR > 1 ? cvae_shfl(hn, 4 + cu) : 0.f, h2 = NR > 2 ? cvae_shfl(hn, 8 + cu) : 0.f;
            if (tid < 4 && t + 1 < p.T) {
                const f32x4 wv = (f32x4){h0, h1, h2, __builtin_bit_cast(float, nonce + (unsigned)(t + 1))};
                cvae_buf_store_f4_sc1(xb, (unsigned)(16 * jg + u0 + tid) * 16u, (unsigned)((t + 1) & 1) * (unsigned)H * 16u, wv);
            }
        }
        }
        g0 = n0; g1 = n1; g2 = n2;
        if (prof) { const long long c1 = cvae_clock(); pc[3] += c1 - c0; c0 = c1; }
    }
    if (prof && tid == 0) {
        for (int q = 0; q < 4; ++q) p.prof[q] = pc[q];
    }
    if (p.prof && tid == 0) cvae_atomic_add_agent((unsigned*)p.dbg + 1, iters_total);
}
