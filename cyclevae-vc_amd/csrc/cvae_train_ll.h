// Training recurrences for at most THREE batch rows (the recipe's own batch_size_utt = 1, egs/one-to-one/run.sh:172, and the
// rec || cv pair stacked from it): the word-exchange form of k_gru_steps_ll (cvae_ll.h) for the train-mode step
// (reference gru_vae.py:378-382 forward, train_gru_cyclevae_gauss_batch.py:1419 backward).
//
// At this size a dependent step is nothing but its hand-off.  The tile kernels (cvae_train_x3.h) pay publish -> drain -> flag ->
// poll -> operand load and move 16 rows' worth of operands for one live row: 7.3 us per forward step and 8.6 us per reverse step
// at B = 1 (tools/train_phase_timing.py).  Here a unit's state travels as ONE 16-byte word (rows 0..2, step tag) that consumers
// poll directly; arithmetic is plain fp32 FMAs on fp32 operands and weights (the reference's own arithmetic).
//
// Forward: the operand of step t is [h_{t-1} ; o_{t-1}], o = gru_drop(h) = mask * h.  Only h is exchanged; every consumer
// multiplies it with the mask values of its own 16 k (loaded one step ahead: masks do not depend on the recurrence) -- exactly
// the products mask * h the producer would have sent.  Weights: the fp32 image of the per-step kernel (wrec_t: W_hh and the
// feedback fold F = W_ih[:, 9C:] . out_1.w), 128 values per thread, register-resident for the launch.
#pragma once
#include <cvae_intrin.h>

struct TrainFwdLLParams {
    float* xbuf;          // [2 slots][H units][4]: (h row 0, row 1, row 2, tag)
    unsigned nonce;       // (launch counter & 0xffff) << 16: tags are nonce + step, so stale words of an earlier launch never match
    int backoff;          // x 64 cycles of sleep between a step's publish and its first poll
    const float* wrec_t;  // [H/4][2*H/16][16 cols = 4*gate + unit][16 k]: chunks < H/16 act on h, the others (F) on o  (k_prep_wrec_train)
    const float* gi;      // [T*Bp][3H] time-major input-side pre-activations
    const float* bhn;     // [H]
    const float* gmask;   // [T][B][H]: dropout mask of the state fed to out_1, scaled by 1/(1-p)
    float* tape;          // [T*Bp][4H]: r, z, n, q = W_hn h + b_hn
    float* hrow;          // [(T+1)*Bp][H] row-major fp32: slot t+1 = h_t (slot 0: k_train_prologue)
    float* orow;          //                               slot t+1 = o_t
    const float* wyT;     // [Co][3H]
    const float* dy;      // [B][Co]
    int Co, B, Bp, H, T;
    int* status;
};

template <int NR>   // rows carried (1..3); p.B <= NR
__global__ __launch_bounds__(256, 1) void k_train_fwd_steps_ll(TrainFwdLLParams p) {
    constexpr int NO = 16 * NR, RS = NO + 1;                            // outputs per block: (gate, row, unit)
    const unsigned nonce = p.nonce;
    const int tid = threadIdx.x, Q = tid >> 2, g = tid & 3, H = p.H, nch = H >> 4;
    const int j0 = 4 * (int)blockIdx.x;                                 // this block's units j0 .. j0+3
    float* red0 = (float*)CVAE_SMEM;                                    // [2 (step parity)][64 quads][RS]
    float* part = red0 + 2 * 64 * RS;                                   // [4 waves][NO]
    const cvae_buf xb = cvae_make_buf(p.xbuf, 2u * (unsigned)H * 16u);
    // word q of a thread is unit 256*wave + 64*q + lane (one contiguous KiB per load instruction); after the quad exchange member
    // js of a quad has handed over unit k(js, q) = 256*wave + 64*q + 4*(lane/4) + js
    const int wave = tid >> 6, lane = tid & 63, kbase = 256 * wave + 4 * (lane >> 2);
    float wh[4][4][4], wo[4][4][4];                                     // gate g of [unit][member js][word q]: on h, on o
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int js = 0; js < 4; ++js)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = kbase + 64 * q + js;
                const float* w = p.wrec_t + (((long)blockIdx.x * 2 * nch + (k >> 4)) * 16 + g * 4 + u) * 16 + (k & 15);
                wh[u][js][q] = k < H ? w[0] : 0.f;
                wo[u][js][q] = k < H ? w[(long)nch * 256] : 0.f;
            }
    constexpr bool ONE_STAGE = NR == 1;
    const int crow = ONE_STAGE ? tid >> 4 : tid >> 2, cu = ONE_STAGE ? (tid >> 2) & 3 : tid & 3, j = j0 + cu;
    const bool cell = (ONE_STAGE ? tid < 16 * NR && (tid & 3) == 0 : tid < 4 * NR) && crow < p.B;
    float hold = 0.f, bhn = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, msk = 0.f;
    if (cell) {
        hold = p.hrow[(long)crow * H + j];
        bhn = p.bhn[j];
        const float* gip = p.gi + (long)crow * 3 * H;
        g0 = gip[j]; g1 = gip[H + j]; g2 = gip[2 * H + j];
        msk = p.gmask[(long)crow * H + j];
        cvae_t0_fix(p.wyT, p.dy, p.Co, H, j, crow, g0, g1, g2);
    }
    // mask values of this lane's own words for the operand of the NEXT step (o_t = mask_t * h_t), requested a step ahead
    float mk[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) mk[r][q] = 0.f;
    for (int t = 0; t < p.T; ++t) {
        // What the NEXT step needs and the recurrence does not produce (its gate inputs and mask for the cell lanes, the mask values
        // of this lane's own words for its operand o_t = mask_t * h_t) is requested FIRST, in front of the back-off sleep and the
        // poll: the loads are in flight while the step waits anyway and the poll's own wait covers them.  Requested behind the poll
        // (round 3) the compiler's s_waitcnt vmcnt(0) at the head of the FMA phase made every wave sit out their memory round trip.
        float n0 = 0.f, n1 = 0.f, n2 = 0.f, nmsk = 0.f;
        if (cell && t + 1 < p.T) {
            const float* gip = p.gi + ((long)(t + 1) * p.Bp + crow) * 3 * H;
            n0 = gip[j]; n1 = gip[H + j]; n2 = gip[2 * H + j];
            nmsk = p.gmask[((long)(t + 1) * p.B + crow) * H + j];
        }
        float mkn[NR][4];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) mkn[r][q] = 0.f;
        if (t + 1 < p.T) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = 256 * wave + 64 * q + lane;
                if (k < H)
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (r < p.B) mkn[r][q] = p.gmask[((long)t * p.B + r) * H + k];
            }
        }
        float hv[NR][4], ov[NR][4];
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
                        if (r < p.B) hv[r][q] = p.hrow[(long)r * H + k];
            }
        } else {
            const unsigned so = (unsigned)(t & 1) * (unsigned)H * 16u, vo = (unsigned)(256 * wave + lane) * 16u;
            unsigned spins = 0;
            for (int q = 0; q < p.backoff; ++q) cvae_sleep_64();
            for (;;) {
                cvae_compiler_fence();
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
                        const float tag = v[q][3];
                        ok = ok && __builtin_bit_cast(unsigned, tag) == nonce + (unsigned)t;
                    }
                if (cvae_wave_all(ok)) break;
                if (++spins > (1u << 20)) {
                    p.status[0] = 6;
                    break;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ov[r][q] = hv[r][q] * mk[r][q];          // (step 0: o_{-1} = 0, the prologue's dy carries y_in)
                mk[r][q] = mkn[r][q];
            }
        float* red = red0 + (ONE_STAGE ? (t & 1) * 64 * RS : 0);
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
                    const float ob = js == 0 ? cvae_quad_bcast<0>(ov[r][q]) : js == 1 ? cvae_quad_bcast<1>(ov[r][q])
                                   : js == 2 ? cvae_quad_bcast<2>(ov[r][q]) : cvae_quad_bcast<3>(ov[r][q]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        acc[r][u] = __builtin_fmaf(hb, wh[u][js][q], acc[r][u]);
                        acc[r][u] = __builtin_fmaf(ob, wo[u][js][q], acc[r][u]);
                    }
                }
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) red[Q * RS + g * 4 * NR + r * 4 + u] = acc[r][u];
        __syncthreads();
        if (ONE_STAGE) {
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
                float hn = 0.f;
                if (cell) {
                    const float rg = cvae_sigmoid(g0 + sum);
                    const float zg = cvae_sigmoid(g1 + s1);
                    const float qq = s3 + bhn;
                    const float ng = tanhf(g2 + s2 + rg * qq);
                    hn = ng + zg * (hold - ng);
                    hold = hn;
                    const long trow = (long)t * p.Bp + crow;
                    p.hrow[(trow + p.Bp) * H + j] = hn;
                    p.orow[(trow + p.Bp) * H + j] = hn * msk;
                    float* tp = p.tape + trow * 4 * H + j;
                    tp[0] = rg; tp[H] = zg; tp[2 * H] = ng; tp[3 * H] = qq;
                }
                if (lane < 16 && (lane & 3) == 0 && t + 1 < p.T) {     // row 0's cell lanes publish their unit's word
                    const f32x4 wv = (f32x4){hn, 0.f, 0.f, __builtin_bit_cast(float, nonce + (unsigned)(t + 1))};
                    cvae_buf_store_f4_sc1(xb, (unsigned)(j0 + cu) * 16u, (unsigned)((t + 1) & 1) * (unsigned)H * 16u, wv);
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
            if (tid < 64) {   // wave 0: lanes 4*row + unit finish the cell, lanes 0..3 publish their unit's word
                float hn = 0.f;
                if (cell) {
                    float s[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int o = a * 4 * NR + crow * 4 + cu;
                        s[a] = (part[o] + part[NO + o]) + (part[2 * NO + o] + part[3 * NO + o]);
                    }
                    const float rg = cvae_sigmoid(g0 + s[0]);
                    const float zg = cvae_sigmoid(g1 + s[1]);
                    const float qq = s[3] + bhn;
                    const float ng = tanhf(g2 + s[2] + rg * qq);
                    hn = ng + zg * (hold - ng);
                    hold = hn;
                    const long trow = (long)t * p.Bp + crow;
                    p.hrow[(trow + p.Bp) * H + j] = hn;
                    p.orow[(trow + p.Bp) * H + j] = hn * msk;
                    float* tp = p.tape + trow * 4 * H + j;
                    tp[0] = rg; tp[H] = zg; tp[2 * H] = ng; tp[3 * H] = qq;
                }
                const float h0 = cvae_shfl(hn, cu), h1 = NR > 1 ? cvae_shfl(hn, 4 + cu) : 0.f, h2 = NR > 2 ? cvae_shfl(hn, 8 + cu) : 0.f;
                if (tid < 4 && t + 1 < p.T) {
                    const f32x4 wv = (f32x4){h0, h1, h2, __builtin_bit_cast(float, nonce + (unsigned)(t + 1))};
                    cvae_buf_store_f4_sc1(xb, (unsigned)(j0 + tid) * 16u, (unsigned)((t + 1) & 1) * (unsigned)H * 16u, wv);
                }
            }
        }
        g0 = n0; g1 = n1; g2 = n2; msk = nmsk;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Reverse recurrence for at most three rows.  Same mathematics as k_train_bwd_steps (cvae_train_bwd.h):
//     dh_t = dhz_{t+1} + W_hh^T dgh_{t+1} + mask_t * (dovl_t + F^T dgi_{t+1}),   dgi = (drp, dzp, dnp), dgh = (drp, dzp, dq), dq = dnp r
// with fp32 operands, fp32 FMAs and no scaling (no exchange range to leave: this path never raises status 5).
// Exchange: per (row, unit) ONE word (drp, dzp, dnp, tag); the fourth gradient dq = dnp * r is rebuilt by every consumer from the
// taped reset gate of its own units (loaded a step ahead -- the tape does not depend on the reverse recurrence): the same fp32
// product the producer forms.  Block = 4 output units; a thread owns the words of 4 producer units (as in the forward kernel, no
// quad exchange: all four gradients of a unit feed all 8 sums of the block) and the 96 weights that go with them, taken from the
// fp32 image of the per-step kernel; the 256 partial sums per output meet in LDS in a fixed order.
// ------------------------------------------------------------------------------------------------------------------------
struct TrainBwdLLParams {
    float* xbuf;          // [2 slots][NR rows][H units][4]: (drp, dzp, dnp, tag)
    unsigned nonce;
    int backoff;
    const float* wrec_t;  // [H/4][2*H/16][16 cols][16 k] (k_prep_wrec_train): row (producer unit, gate), 16 consecutive k
    const float* dovl;    // [T*Bp][H]: W_o^T dyl_t
    const float* tape;    // [T*Bp][4H]: r, z, n, q
    const float* hrow;    // [(T+1)*Bp][H]: slot t = h_{t-1}
    const float* gmask;   // [T][B][H]
    float* dgi;           // [T*Bp][3H]
    float* dgh;           // [T*Bp][3H]
    int B, Bp, H, T;
    int* status;
};

template <int NR>
__global__ __launch_bounds__(256, 1) void k_train_bwd_steps_ll(TrainBwdLLParams p) {
    constexpr int NOUT = 8 * NR, RSB = NOUT + 1, NSL = 8;                // outputs per block: (row, path, unit); 8 slices of 32 threads
    const unsigned nonce = p.nonce;
    const int tid = threadIdx.x, H = p.H, nch = H >> 4;
    const int k0 = 4 * (int)blockIdx.x;                                  // this block's output units k0 .. k0+3
    float* red = (float*)CVAE_SMEM;                                      // [256 threads][RSB]
    float* part = red + 256 * RSB;                                       // [NSL][NOUT]
    const cvae_buf xb = cvae_make_buf(p.xbuf, 2u * (unsigned)NR * (unsigned)H * 16u);
    const int wave = tid >> 6, lane = tid & 63;
    // weights of this thread's producer units j(q) = 256*wave + 64*q + lane towards the block's 4 output units:
    // state path W_hh[gate][j][k0..k0+3] for the gradients (drp, dzp, dq), feedback path F[gate][j][k0..k0+3] for (drp, dzp, dnp)
    f32x4 ws[4][3], wf[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int jq = 256 * wave + 64 * q + lane;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ws[q][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            wf[q][a] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (jq < H) {
                const float* w = p.wrec_t + ((long)(jq >> 2) * 2 * nch + (k0 >> 4)) * 256 + (jq & 3) * 16 + (k0 & 15);
                ws[q][a] = *(const f32x4*)(w + (a == 2 ? 3 : a) * 64);               // h chunks: columns r, z, (0), n_h
                wf[q][a] = *(const f32x4*)(w + (long)nch * 256 + a * 64);             // o chunks: columns r, z, n_x
            }
        }
    }
    // cell threads: tid = 4*row + unit
    const int crow = tid >> 2, cu = tid & 3, k = k0 + cu;
    const bool cell = tid < 4 * NR && crow < p.B;
    float keep = 0.f;
    float ntr = 0.f, ntz = 0.f, ntn = 0.f, ntq = 0.f, nthp = 0.f, ntmask = 0.f, ntdov = 0.f;
    auto prefetch_cell = [&](int t) {
        if (cell && t >= 0) {
            const long rn = (long)t * p.Bp + crow;
            const float* tp = p.tape + rn * 4 * H + k;
            ntr = tp[0]; ntz = tp[H]; ntn = tp[2 * H]; ntq = tp[3 * H];
            nthp = p.hrow[rn * H + k];
            ntmask = p.gmask[((long)t * p.B + crow) * H + k];
            ntdov = p.dovl[rn * H + k];
        }
    };
    prefetch_cell(p.T - 1);
    // reset gates of this thread's producer units at the step whose gradients arrive next
    float rt[NR][4];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) rt[r][q] = 0.f;
    auto prefetch_r = [&](int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jq = 256 * wave + 64 * q + lane;
            if (jq < H)
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    if (r < p.B) rt[r][q] = p.tape[((long)t * p.Bp + r) * 4 * H + jq];
        }
    };
    prefetch_r(p.T - 1);
    float tr = 0.f, tz = 0.f, tn = 0.f, tq = 0.f, thp = 0.f, tmask = 0.f, tdov = 0.f;     // cell inputs of the running step
    float rtn[NR][4];
    for (int tt = 0; tt < p.T; ++tt) {
        const int t = p.T - 1 - tt;
        if (tt == 0) { tr = ntr; tz = ntz; tn = ntn; tq = ntq; thp = nthp; tmask = ntmask; tdov = ntdov; }
        // The next step's cell inputs (step t-1) and the reset gates of the gradients it will receive (those of step t) do not depend
        // on the recurrence: requested FIRST, in front of the back-off sleep and the poll (their latency disappears in the wait; the
        // poll's own s_waitcnt covers them).  They move into the registers of the running step after its cell math and BEFORE its
        // stores, so that no wait at the loop's end has the publish store's acknowledge in front of it (round 3 requested them
        // behind the FMAs: the copies ended up at the loop latch behind an s_waitcnt vmcnt(0) that also waited for the
        // write-through publish, and wave 0 then slept its back-off on top of that).
        prefetch_cell(t - 1);
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) rtn[r][q] = 0.f;
        if (t > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jq = 256 * wave + 64 * q + lane;
                if (jq < H)
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        if (r < p.B) rtn[r][q] = p.tape[((long)t * p.Bp + r) * 4 * H + jq];
            }
        }
        float sa[NR][4], sb[NR][4];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) { sa[r][u] = 0.f; sb[r][u] = 0.f; }
        if (tt > 0) {
            const unsigned so = (unsigned)(tt & 1) * (unsigned)NR * (unsigned)H * 16u, vo = (unsigned)(256 * wave + lane) * 16u;
            f32x4 v[NR][4];
            unsigned spins = 0;
            for (int q = 0; q < p.backoff; ++q) cvae_sleep_64();
            for (;;) {
                cvae_compiler_fence();
                bool ok = true;
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (256 * wave + 64 * q < H) v[r][q] = cvae_buf_poll_f4(xb, vo + 1024u * q, so + (unsigned)r * (unsigned)H * 16u);
#pragma unroll
                for (int r = 0; r < NR; ++r)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (256 * wave + 64 * q < H) {
                            const float tag = v[r][q][3];
                            ok = ok && __builtin_bit_cast(unsigned, tag) == nonce + (unsigned)tt;
                        }
                if (cvae_wave_all(ok)) break;
                if (++spins > (1u << 20)) {
                    p.status[0] = 6;
                    break;
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (256 * wave + 64 * q < H) {
                        const float g0 = v[r][q][0], g1 = v[r][q][1], g2 = v[r][q][2], gq = g2 * rt[r][q];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            sa[r][u] = __builtin_fmaf(g0, ws[q][0][u], sa[r][u]);
                            sa[r][u] = __builtin_fmaf(g1, ws[q][1][u], sa[r][u]);
                            sa[r][u] = __builtin_fmaf(gq, ws[q][2][u], sa[r][u]);
                            sb[r][u] = __builtin_fmaf(g0, wf[q][0][u], sb[r][u]);
                            sb[r][u] = __builtin_fmaf(g1, wf[q][1][u], sb[r][u]);
                            sb[r][u] = __builtin_fmaf(g2, wf[q][2][u], sb[r][u]);
                        }
                    }
        }
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) rt[r][q] = rtn[r][q];
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                red[tid * RSB + r * 8 + u] = sa[r][u];
                red[tid * RSB + r * 8 + 4 + u] = sb[r][u];
            }
        __syncthreads();
        if (tid < NSL * NOUT) {
            const int o = tid % NOUT, sl = tid / NOUT;
            float sum = 0.f;
#pragma unroll 8
            for (int i = 0; i < 256 / NSL; ++i) sum += red[(sl * (256 / NSL) + i) * RSB + o];
            part[sl * NOUT + o] = sum;
        }
        __syncthreads();
        if (tid < 64) {
            float v0 = 0.f, v1 = 0.f, v2 = 0.f;
            if (cell) {
                float fa = 0.f, fb = 0.f;
#pragma unroll
                for (int s = 0; s < NSL; ++s) {
                    fa += part[s * NOUT + crow * 8 + cu];
                    fb += part[s * NOUT + crow * 8 + 4 + cu];
                }
                const float dht = keep + fa + tmask * (tdov + fb);
                const float r = tr, z = tz, n = tn, qv = tq, hp = thp;
                const float dn = dht * (1.0f - z), dz = dht * (hp - n);
                v2 = dn * (1.0f - n * n);
                const float v3 = v2 * r;
                v0 = v2 * qv * r * (1.0f - r);
                v1 = dz * z * (1.0f - z);
                keep = dht * z;
                tr = ntr; tz = ntz; tn = ntn; tq = ntq; thp = nthp; tmask = ntmask; tdov = ntdov;     // (before the stores: see above)
                const long rowi = (long)t * p.Bp + crow;
                float* gi = p.dgi + rowi * 3 * H + k;
                float* gh = p.dgh + rowi * 3 * H + k;
                gi[0] = v0; gi[H] = v1; gi[2 * H] = v2;
                gh[0] = v0; gh[H] = v1; gh[2 * H] = v3;
            }
            if (tid < 4 * NR && t > 0) {      // every (row, unit) lane of the block publishes its word (dead rows: zeros)
                const f32x4 wv = (f32x4){v0, v1, v2, __builtin_bit_cast(float, nonce + (unsigned)(tt + 1))};
                cvae_buf_store_f4_sc1(xb, (unsigned)k * 16u, ((unsigned)((tt + 1) & 1) * (unsigned)NR + (unsigned)crow) * (unsigned)H * 16u, wv);
            }
        }
    }
}
