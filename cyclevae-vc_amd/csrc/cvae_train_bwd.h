// k_train_bwd_steps: the reverse recurrence of a train-mode pass (the BPTT part of what `batch_loss.backward()` does inside one
// GRU_RNN, reference train_gru_cyclevae_gauss_batch.py:1419 through gru_vae.py:365-393) as ONE persistent cooperative launch
// instead of 2T dependent launches.
//
// Per step t (from T-1 down) and batch row, with the tape of the forward (r, z, n, q = W_hn h + b_hn, h_{t-1}, dropout mask):
//     dh_t   = dhz_{t+1} + W_hh^T dgh_{t+1} + mask_t * ( W_o^T dyl_t + W_o^T W_iy^T dgi_{t+1} )
//     dn = dh_t (1-z), dz = dh_t (h_{t-1} - n);  dnp = dn (1-n^2), dq = dnp r, drp = dnp q r(1-r), dzp = dz z(1-z);  dhz_t = dh_t z
//     dgi_t = (drp, dzp, dnp), dgh_t = (drp, dzp, dq)
// The feedback path y_{t} -> input of step t+1 is FOLDED like in the forward kernels: F = W_ih[:, C9:] . out_1.w, so that one
// matrix product per step carries the gradient: [B x 4H] (drp, dzp, dnp, dq per unit) times [4H x 2H] (state path W_hh^T with a
// zero row for dnp, feedback path F^T with a zero row for dq).  W_o^T dyl_t for all t is one GEMM before the loop (`dovl`), the
// total d loss / d y_t that the weight-gradient GEMMs need is one GEMM after it.
//
// Decomposition: block = 8 hidden units (16 output columns: 8 state-path + 8 feedback-path sums) x 16-row tiles; 128 blocks per
// row-tile group at hu1024, the block's 16 x 4096 weights register-resident as fp16 pairs (256 registers per lane, the 4 waves
// split K).  Every block needs ALL 4H gate gradients of its rows per step: they are exchanged as fp16 pairs of (value * 2^8)
// (gradients are small: the scale keeps the leading limb a normal half down to 2.4e-7; |value| >= 256 sets status 5), in a
// tile-planar buffer with one slot per step, 2 KiB per (producer octet, row tile): what one load or store instruction touches is
// one contiguous KiB.  Hand-off as in k_gru_steps_v6: write-through publish, drain, per-(tile, octet) flag = number of steps
// published; consumers poll flags and stream the operands through a ring of 8 32-k steps with plain first-touch loads.
#pragma once
#include <cvae_intrin.h>

#define CVAE_BWD_GSCALE 256.0f

struct TrainBwdParams {
    const float* wbk;    // [H/8][4 waves][KPW][2 limbs][64 lanes][8 halves]: B operands (k_prep_wbk)
    float* gx;           // exchanged gate gradients: [T][H/8][Bp/16]{ hi [4 kq][16 rows][8 halves] | lo likewise } (2 KiB each)
    unsigned* flags;     // [Bp/16][H/8], zeroed before launch
    int* status;
    const float* dovl;   // [T*Bp][H]: W_o^T dyl_t
    const float* tape;   // [T*Bp][4H]: r, z, n, q
    const float* hrow;   // [(T+1)*Bp][H]: slot t = h_{t-1}
    const float* gmask;  // [T][B][H]
    float* dgi;          // [T*Bp][3H]
    float* dgh;          // [T*Bp][3H]
    float* dhz;          // [Bp][H]: carried z-path gradient when a block owns more than two row tiles
    int B, Bp, H, T, rts;
    int xmap;            // 1: XCD-aware block placement (cvae_block_map)
    int backoff;         // x 64 cycles before the first flag poll of a task (option train_bwd_backoff)
    int tile_lo, tile_n; // k_train_bwd_steps_x3: the row tiles of this launch (tile_n = 0: the whole pass)
    long long* prof;     // null, or 4 cycle sums of block 0: flag wait, loads + MFMA, reduce + cell + stores, publish
    float ovf;           // |value * 2^8| from which a gate gradient counts as outside the exchange range (60000; tests lower it)
};

// wbk[c][wave][s][limb][lane][e]: lane (col = lane & 15, kq = lane >> 4) holds K index k = 32*(wave*KPW + s) + 8*kq + e = 4*j + comp
// (producer unit j, comp: 0 drp, 1 dzp, 2 dnp, 3 dq) of column col:
//   col < 8, output unit ko = 8c + col      (state path):    comp 0: W_hh[j][ko], 1: W_hh[H+j][ko], 2: 0, 3: W_hh[2H+j][ko]
//   col >= 8, output unit ko = 8c + col - 8 (feedback path): comp 0: F[j][ko], 1: F[H+j][ko], 2: F[2H+j][ko], 3: 0   (F: k_prep_ffold)
__global__ void k_prep_wbk(const float* F, const float* whh, float* wbk, int H, int KPW) {
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
                v = F[(long)(comp * H + j) * H + 8 * c + col - 8];      // (k_prep_ffold)
            }
        }
        unsigned short hi, lo;
        cvae_split_f16(v, hi, lo);
        unsigned short* dst = (unsigned short*)wbk + ((((long)c * 4 + wave) * KPW + s) * 2) * 512 + lane * 8 + e;
        dst[0] = hi;
        dst[512] = lo;
    }
}

template <int KPW>   // 32-k steps per wave = 4H / 128
__global__ __launch_bounds__(256, 1) void k_train_bwd_steps(TrainBwdParams p) {
    constexpr int RD = KPW < 8 ? KPW : 8;
    constexpr int RS = 20;
    const int tid = threadIdx.x, wave = cvae_uniform(tid >> 6), lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int H = p.H, NB = H >> 3, nt16 = p.Bp >> 4;
    const int rts = p.rts, c = blockIdx.x % NB, ti = blockIdx.x / NB;
    float* red = (float*)CVAE_SMEM;                                   // [4 waves][16 rows][RS]
    unsigned short* pub = (unsigned short*)(red + 4 * 16 * RS);       // [2 limbs][4 kq][16 rows][8 halves]
    const cvae_buf gb = cvae_make_buf(p.gx, (unsigned)((long)p.T * NB * nt16 * 2048));
    f32x4 w0[KPW], w1[KPW];
#pragma unroll
    for (int s = 0; s < KPW; ++s) {
        const float* src = p.wbk + ((((long)c * 4 + wave) * KPW + s) * 2) * 256 + lane * 4;
        w0[s] = *(const f32x4*)src;
        w1[s] = *(const f32x4*)(src + 256);
    }
    const int row = (tid >> 3) & 15, u = tid & 7, k = 8 * c + u;
    const bool gate_thread = tid < 128;
    const int ntile = ti < nt16 ? (nt16 - ti + rts - 1) / rts : 0, ntask = p.T * ntile;
    float keep0 = 0.f, keep1 = 0.f;
    for (int kk = 0; kk < ntask; ++kk) {
        const int tt = kk / ntile, t = p.T - 1 - tt, i = ti + (kk % ntile) * rts;
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
        // what the cell backward of this thread's (row, unit) needs from the tape does not depend on the recurrence: requested
        // here, it arrives under the flag wait and the MFMA phase instead of behind them
        const int grow = i * 16 + row;
        const bool live = gate_thread && grow < p.B;
        const long rowi = (long)t * p.Bp + grow;
        float tr = 0.f, tz = 0.f, tn = 0.f, tq = 0.f, thp = 0.f, tmask = 0.f, tdov = 0.f;
        if (live) {
            const float* tp = p.tape + rowi * 4 * H + k;
            tr = tp[0]; tz = tp[H]; tn = tp[2 * H]; tq = tp[3 * H];
            thp = p.hrow[rowi * H + k];
            tmask = p.gmask[((long)t * p.B + grow) * H + k];
            tdov = p.dovl[rowi * H + k];
        }
        if (tt > 0) {
            unsigned spins = 0;
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
            f32x4 gc[2 * RD];
            auto load_g = [&](int s) {
                const unsigned so = ((unsigned)((t + 1) * NB + wave * KPW + s) * (unsigned)nt16 + (unsigned)i) * 2048u;
                gc[2 * (s % RD)] = cvae_buf_load_f4(gb, (unsigned)lane * 16u, so);
                gc[2 * (s % RD) + 1] = cvae_buf_load_f4(gb, (unsigned)lane * 16u, so + 1024u);
            };
#pragma unroll
            for (int s = 0; s < RD; ++s) load_g(s);
#pragma unroll
            for (int s = 0; s < KPW; ++s) {
                const f32x4 hi = gc[2 * (s % RD)], lo = gc[2 * (s % RD) + 1];
                a0 = cvae_mfma_16x16x32_f16(hi, w0[s], a0);
                a1 = cvae_mfma_16x16x32_f16(hi, w1[s], a1);
                a2 = cvae_mfma_16x16x32_f16(lo, w0[s], a2);
                cvae_sched_fence();
                if (s + RD < KPW) load_g(s + RD);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) red[(wave * 16 + kq * 4 + q) * RS + lr] = a0[q] + (a1[q] + a2[q]) * (1.0f / 2048.0f);
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
                unsigned short hi, lo;
                cvae_split_f16(sv, hi, lo);
                const int kl = 4 * u + cm;
                pub[((kl >> 3) * 16 + row) * 8 + (kl & 7)] = hi;
                pub[512 + ((kl >> 3) * 16 + row) * 8 + (kl & 7)] = lo;
            }
        }
        __syncthreads();
        if (tid < 64) {   // wave 0: 2 KiB = the LDS image, lane-linear per limb
            const unsigned so = ((unsigned)(t * NB + c) * (unsigned)nt16 + (unsigned)i) * 2048u;
            cvae_buf_store_f4_sc1(gb, (unsigned)tid * 16u, so, *(const f32x4*)(pub + tid * 8));
            cvae_buf_store_f4_sc1(gb, (unsigned)tid * 16u, so + 1024u, *(const f32x4*)(pub + 512 + tid * 8));
            cvae_drain_vmem();
            cvae_wave_barrier();
            if (tid == 0) cvae_atomic_store_agent(p.flags + (long)i * NB + c, (unsigned)(tt + 1));
        }
    }
}
