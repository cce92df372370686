// libcyclevae_hip.so -- C ABI (include/cyclevae_hip.h) over the kernels in cvae_kernels.h.
// Host orchestration only: carves the caller's buffers, enqueues kernels on the caller's stream.
#include <cvae_intrin.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <new>
#include <utility>
#include <vector>

#include "cvae_kernels.h"
#include "cvae_train_kernels.h"
#include "cyclevae_hip.h"

namespace {

thread_local char g_err[512] = "";
// The library keeps NO process-wide mutable state (ABI 6): what used to be process-wide settings -- where a timed-out hand-off spin is
// reported, this rank's place in a data-parallel job's batch for the Philox streams, the named options, the side stream and its
// events, the profiling brackets, the record of which MFMA-order images a train image holds -- lives in the caller's cvae_ctx
// (cvae_ctx_create: one per (device, stream) user; not thread-safe per handle, independent across handles).  Every entry point
// takes the handle and installs it for the duration of the call (tl_ctx below, thread-local plumbing: never read outside a call).

// Tuning / diagnostic switches (cvae_set_option): the library reads NO environment variable.
enum OptId {
    OPT_V6_LIMBS_H64, OPT_NO_LL, OPT_MAX_RT, OPT_LL_BACKOFF, OPT_EXP, OPT_OLD_OUTPROJ, OPT_GEMM_FORCE, OPT_GEMM_LOG, OPT_TRAIN_OLD_GEMM,
    OPT_GEMM_TRACE, OPT_TRAIN_PER_STEP, OPT_TRAIN_PROF, OPT_TRAIN_BACKOFF, OPT_TRAIN_FP32_MFMA, OPT_TRAIN_BWD_PER_STEP, OPT_TRAIN_KERNEL, OPT_X3_TILE, OPT_BWD_OVERFLOW_AT, OPT_GEMM_MAX_SPLIT, OPT_BWD_KS, OPT_BWD_WIDE, OPT_COOP_LAUNCH, OPT_V6_LIMBS_H2048, OPT_V6_W2S_H64, OPT_STEP_COL_TILES, OPT_T0_IN_KERNEL, OPT_TRAIN_PROFILE, OPT_TRAIN_XMAP, OPT_LL_WIDE_ROWS, OPT_TRAIN_BP16, OPT_BWD_SPLIT_LAUNCH, OPT_WGRAD_ORDER, OPT_SIDE_TILE_CAP, OPT_MASKS_ON_SIDE, OPT_V6_BACKOFF, OPT_TRAIN_BWD_BACKOFF, OPT_LL_ROW_PAD, OPT_GEMM_MIN_DEPTH, OPT_GEMM_OCC_MODEL, OPT_GEMM_NT_FIT, OPT_TRAIN_BWD_GEOM, OPT_BWD_W3_L1_H64, OPT_TRAIN_FWD_GEOM, OPT_BWD_W3_TWO_TILES, OPT_TRAIN_FWD_BACKOFF, OPT_COUNT
};
struct OptEntry { const char* name; long dflt; };
const OptEntry g_opt[OPT_COUNT] = {      // names and DEFAULTS (immutable); the values live in the context
    {"v6_limbs_h64", 3},          // 2: the two-limb code path of k_gru_steps_v6 at H = 64 (what runs at H = 2048), for the emulator tests
    {"no_ll", 0},                 // 1: passes of <= 3 rows take the dataflow kernel instead of k_gru_steps_ll
    {"max_rt", 0},                // > 0: cap on the row tiles handled concurrently (tests: several row tiles per block on small problems)
    {"ll_backoff", -1},          // >= 0: s_sleep units before the first poll of a step in k_gru_steps_ll (-1: the swept default)
    {"exp", 0},                   // measurement switches of the dataflow kernels (Step6Params::exp)
    {"old_outproj", 0},           // 1: projection of a v6 pass from the fp32 state copy instead of the limb triples
    {"gemm_force", 0},            // measurement: TM*10000 + TN*100 + ks forces the tile / split of every training GEMM
    {"gemm_log", 0},              // measurement: every training GEMM bracketed by HIP events and printed to stderr
    {"train_old_gemm", 0},        // 1: the simple GEMM kernels kept as unaligned-operand fallbacks, everywhere
    {"gemm_trace", 0},            // 1: print when a GEMM takes a fallback kernel
    {"train_per_step", 0},        // 1: forward training recurrence as T launches
    {"train_prof", 0},            // 1: phase cycle sums of block 0 of the training recurrences (cvae_train_debug_counters)
    {"train_backoff", 32},       // s_sleep units before the first poll of a step, pair-form forward training recurrence
    {"train_fp32_mfma", 0},       // 1: forward training recurrence on v_mfma_f32_16x16x4_f32
    {"train_bwd_per_step", 0},    // 1: reverse training recurrence as 2T launches (fp32 products)
    {"train_kernel", 0},          // training recurrences: 0 exact fp32 operands (fp16 triples), 1 fp16 pairs, 2 fp32-input MFMA
    {"x3_tile", 0},               // exact-operand forward training recurrence: 0 pick by tiles per block, 16 / 32 force that row tile
    {"bwd_overflow_at", 60000},   // |gate gradient * 2^8| that raises status 5 in the persistent reverse recurrences (tests lower it)
    {"gemm_max_split", 16},      // cap on the contraction split of the training GEMMs (1: never split)
    {"bwd_ks", 8},                // K slices of the per-step backward product k_bwd_step_gemm (1..32)
    {"bwd_wide", 0},              // 1: four column tiles per block in k_bwd_step_gemm where the shape allows (measured: no gain)
    {"coop_launch", 0},           // 1: the all-resident recurrent kernels go through hipLaunchCooperativeKernel (cvae_launch_coop)
    {"v6_limbs_h2048", 3},        // 2: k_gru_steps_v6 at H = 2048 on fp16 PAIRS (faster, 22-23 bit operands) instead of exact triples
    {"v6_w2s_h64", 0},            // 1: the streamed-third-limb form of k_gru_steps_v6 (what runs at H = 2048) at H = 64, for the emulator tests
    {"step_col_tiles", 0},        // per-step forward training kernel: 0 pick (two 16-column tiles per block when every CU still gets a block), 1 / 2 force
    {"t0_in_kernel", 0},          // 1: k_gru_steps_v6 forms the frame-0 feedback correction itself (cvae_t0_fix) instead of reading the prologue's gx0
    {"train_profile", 0},         // 1: HIP events around the training recurrences and GEMMs, summed per class (cvae_train_profile_collect)
    {"train_xmap", 0},            // bit 0 / 1: XCD-aware block placement in the exact forward / reverse training recurrences
    {"ll_wide_rows", 0},          // 1: word-exchange training passes keep the 32-row padding of the tile kernels (round 3's layout, for A/B)
    {"train_bp16", 1},            // training passes of 4..16 rows pad to ONE 16-row tile (B = 8: 16.5 -> 13.7 ms per step); 0: 32 rows = two 16-row tiles, one dead (round 3)
    {"bwd_split_launch", 1},      // exact reverse recurrence: a pass with more than two row tiles per block runs as one launch per two tiles per block (0: one launch)
    {"wgrad_order", -1},          // side-stream weight-gradient GEMMs of a backward pass: 0 all start right behind its reverse recurrence (beside the dgrad chain), 1 all behind the dgrad chain (under the NEXT recurrence), 2 the light ones at once and the two big contractions behind the chain; -1: 2 for passes of >= 64 rows, else 0
    {"side_tile_cap", 2},         // > 0: GEMMs on the side stream use tiles of at most 32*cap x 32*cap (several 64 x 64 workgroups fit on a CU beside a block of the reverse recurrence: B=64 24.07-24.13 -> 23.94-24.05 ms; 0: no cap)
    {"masks_on_side", 1},         // train-mode forward with a side stream set: the recurrence's dropout mask is drawn on it, beside the front-end GEMMs (0: on the launch stream)
    {"v6_backoff", -1},           // >= 0: x 64 cycles before the first flag poll of a step in k_gru_steps_v6 blocks with one row tile (-1: swept per front-end width)
    {"train_bwd_backoff", -1},    // x 64 cycles before the first flag poll of a task of the exact reverse training recurrences; -1: 32 when a block has ONE tile (B = 8: reverse recurrences 5.3 -> 4.35 ms per step, round 6), 0 with two or more (swept in round 5: slower)
    {"ll_row_pad", 0},            // rows per frame of the time-major buffers of a word-exchange training pass (<= 3 rows): 0 = exactly B (every GEMM of a one-utterance pass over T rows; round 5: 5.28 -> 4.87 ms per step), 4 = round 4's layout
    {"gemm_min_depth", 128},      // a split contraction keeps at least this many k per slice (256 until round 5: one utterance 4.86 -> 4.76 ms)
    {"gemm_occ_model", 1},        // tile picker of the training GEMMs: workgroups per CU from the kernels' register use (0: at most four)
    {"gemm_nt_fit", 1},           // tile picker: constants fitted to the round-5 sweep for the launch stream's k_gemm_nt2 GEMMs (0: the shared ones)
    {"train_bwd_geom", -1},       // exact reverse training recurrence: 1 = 16 units x 16-row tiles, zero rows of [W_hh^T | F^T] dropped (k_train_bwd_steps_w3, round 6), 0 = 8 units x 16-row tiles (k_train_bwd_steps_x3); -1: the 16-unit form for passes of at least four 16-row tiles (two tiles per block: 128 rows at hu1024 29.0K instead of 2 x 23.3K cycles per step; 64 rows on half the chip, see bwd_w3_two_tiles), else the 8-unit form (ONE tile per 16-unit block leaves the hand-off exposed: 22.1K vs 23.2K cycles per 64 rows alone, and no room for a co-resident GEMM)
    {"bwd_w3_l1_h64", 0},         // 1: k_train_bwd_steps_w3 at H = 64 keeps the second limbs of two fragments per wave in LDS (what runs at H = 1024), for the emulator tests
    {"train_fwd_geom", -1},       // exact forward training recurrence: 1 = 16 units x 16-row tiles with the zero column tiles of [W_hh | F] dropped (k_train_fwd_steps_w3, round 6), 0 = the 8-unit kernels (k_train_fwd_steps_x3 / x3h); -1: the 16-unit form for passes of at least four 16-row tiles (64 rows: one tile per block behind a first-poll back-off, 128 rows: two tiles per block), else the 8-unit form
    {"bwd_w3_two_tiles", 1},      // the 16-unit reverse recurrence gives a block two tiles whenever the pass has them: a 64-row pass then runs on 128 blocks = HALF the chip (1.12 instead of 0.84 ms), and the side stream's weight-gradient GEMMs -- which cannot share a CU with a 16-unit block -- get the other 128 CUs to themselves, uncapped tiles: B=64 step 23.2-23.3 -> 22.65-22.86 ms same box; 0: one tile per block on every CU
    {"train_fwd_backoff", -1},    // x 64 cycles before the first flag poll of a task of the exact forward training recurrences (16-row-tile kernels); -1: swept value when a block has ONE tile (nothing else covers the hand-off and early polls slow the publishes they wait for), 0 with two or more
};

// hipEvent pairs recorded around the recurrent kernel when CVAE_FLAG_PROFILE is set
struct ProfEvents {
    std::vector<hipEvent_t> start, stop;
    std::vector<int> rows, cin;      // of the bracketed launch: stacked batch rows and input channels (which instantiation / geometry)
    size_t used = 0;
};
// the same for the training step, per kernel class (option train_profile; cvae_train_profile_collect): 0 forward recurrence,
// 1 reverse recurrence, 2 forward / data-gradient GEMMs (gemm_nt), 3 weight-gradient contractions (gemm_tn)
enum { TPROF_FWD = 0, TPROF_BWD = 1, TPROF_GEMM = 2, TPROF_WGRAD = 3, TPROF_CLASSES = 4 };
struct TrainProfEvents {
    std::vector<hipEvent_t> start, stop;
    std::vector<int> cls;
    std::vector<double> flop;
    size_t used = 0;
};
struct SideDone { const void* scratch; hipEvent_t ev; };

}  // namespace

// The handle behind every entry point (include/cyclevae_hip.h: cvae_ctx_create / cvae_ctx_destroy).
struct cvae_ctx {
    int32_t* status_sink = nullptr;                                  // cvae_set_status_sink
    long draw_row0 = 0, draw_rows = 0, draw_frames = 0, draw_parts = 1;   // cvae_set_draw_origin / cvae_set_draw_parts
    long opt[OPT_COUNT];                                             // cvae_set_option
    ProfEvents prof;
    TrainProfEvents tprof;
    hipStream_t side = nullptr;                                      // cvae_set_side_stream
    SideDone side_done[8] = {};
    hipEvent_t side_ready = nullptr, side_join = nullptr, side_last = nullptr;   // side_last: behind ALL side work enqueued so far
    int side_cap = 0;                                                // tile cap of the side-stream GEMMs being launched (launch_wgrad_side)
    hipEvent_t mask_fork = nullptr, mask_join = nullptr;                         // train-mode forward: the feedback mask drawn on the side stream
    unsigned side_evict = 0;
    // which MFMA-order weight images a train image of THIS context holds (cvae_net_prepare_train_v), by address; an address the
    // context has not prepared holds none (the old process-wide registry assumed "all" for unknown addresses and grew
    // without bound); the oldest entries go when the table is full
    std::map<const void*, std::pair<int, unsigned long>> train_var;
    unsigned long train_var_tick = 0;
    std::atomic<unsigned> ll_train_launch{1u};                       // tag nonce of the word-exchange training kernels (cvae_train_ll.h)
};

namespace {

thread_local cvae_ctx* tl_ctx = nullptr;       // the context of the entry point running on this thread (installed by CVAE_ENTER)
inline cvae_ctx& cx() { return *tl_ctx; }
inline long opt(OptId i) { return tl_ctx->opt[i]; }
struct CtxScope {
    cvae_ctx* prev;
    explicit CtxScope(cvae_ctx* c) : prev(tl_ctx) {
        tl_ctx = c;
        g_cvae_coop_launch = (int)c->opt[OPT_COOP_LAUNCH];
    }
    ~CtxScope() { tl_ctx = prev; }
};

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// first statement of every entry point that takes the handle
#define CVAE_ENTER(c)                                                          \
    if (!(c)) return fail(-1, "null context (cvae_ctx_create)");               \
    CtxScope cvae_scope_(c)
#define CVAE_ENTER_SZ(c)                                                       \
    if (!(c)) {                                                                \
        (void)fail(-1, "null context (cvae_ctx_create)");                      \
        return 0;                                                              \
    }                                                                          \
    CtxScope cvae_scope_(c)

#define CVAE_HIP_OK(expr)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail(-3, "%s failed: %s", #expr, hipGetErrorString(e_));     \
    } while (0)

inline long up(long x, long m) { return (x + m - 1) / m * m; }
// clamp_lat_dim argument of the passes: -1 none; L: out[..., L:] >= ln(1e-6) (clamp_vae, gru_vae.py:412); L | CVAE_CLAMP_LAPLACE: the
// log-scale floor of the Laplace variant (clamp_vae_laplace, gru_vae.py:417)
inline int clamp_dim(int v) { return v < 0 ? -1 : (v & ~CVAE_CLAMP_LAPLACE); }
inline float clamp_floor(int v) { return v >= 0 && (v & CVAE_CLAMP_LAPLACE) ? -7.2543288692621097f : -13.815510557964274f; }
inline unsigned nblk(long n, int per) { return (unsigned)((n + per - 1) / per); }

struct Dims {
    int C, Cp, Co, Cop, H, H3, ks, R, pad, c1, c2, tot, Kfe, nch, KFW;
};

int make_dims(const cvae_net_desc* d, Dims* o) {
    if (!d) return fail(-1, "null net descriptor");
    if (d->layers != 2) return fail(-1, "layers (reference dilation_size) must be 2, got %d", d->layers);
    if (d->kernel_size < 1 || d->kernel_size % 2 == 0) return fail(-1, "kernel_size must be odd, got %d", d->kernel_size);
    if (d->hidden < 16 || d->hidden % 16) return fail(-1, "hidden must be a positive multiple of 16, got %d", d->hidden);
    if (d->in_dim < 1 || d->out_dim < 1) return fail(-1, "bad in_dim/out_dim %d/%d", d->in_dim, d->out_dim);
    o->C = d->in_dim;
    o->Cp = (int)up(d->in_dim, 8);   // multiple of 8: eight consecutive fp16 halves of a window are one 16-byte MFMA operand
    o->Co = d->out_dim;
    o->Cop = (int)up(d->out_dim, 16);
    o->H = d->hidden;
    o->H3 = 3 * d->hidden;
    o->ks = d->kernel_size;
    o->R = o->ks * o->ks;
    o->pad = (o->R - 1) / 2;
    o->c1 = o->ks * o->C;
    o->c2 = o->R * o->C;
    o->tot = o->c2 + o->Co;
    o->Kfe = (int)up((long)o->R * o->Cp, 16);
    o->nch = o->H / 16;
    o->KFW = (o->Kfe / 16 + 3) / 4;   // front-end 16-k chunks per wave in the fused recurrent kernel
    return 0;
}

// k_gru_steps_v6 is instantiated for H = 1024 (16 16-k steps per wave, three limbs), H = 64 (one step, three limbs; tests) and
// H = 2048 (32 steps, TWO limbs: the hu2048 stress configuration, BASELINE configs[4]); front-end steps per wave as listed
inline int exact3_kpw(const Dims& m) { return m.H / 64; }
inline int v6_limbs(const Dims& m) {
    if (m.H == 64) {   // tests: the two-limb code path at a size the host-fiber emulator can run
        if (opt(OPT_V6_LIMBS_H64) == 2) return 2;
    }
    return m.H == 2048 && opt(OPT_V6_LIMBS_H2048) == 2 ? 2 : 3;
}
inline bool v6_w2s(const Dims& m) { return v6_limbs(m) == 3 && (m.H == 2048 || (m.H == 64 && opt(OPT_V6_W2S_H64))); }
inline bool exact3_ok(const Dims& m) {
    return (m.H == 1024 && (m.KFW == 8 || m.KFW == 6)) || (m.H == 64 && m.KFW >= 1 && m.KFW <= 3) ||
           (m.H == 2048 && (m.KFW == 8 || m.KFW == 11));
}

// prepared image: offsets in floats, every block 64-float aligned
struct Prep {
    long afold, afold3, afold_h, afold_t, cfold, wrec, wrec2, wrec_h, wrec_t, wrec_l2b, bhn, wyT, wo, bo, wo2, bo2, wo3, sin_w, sin_b, sout_w,
        sout_b, total;
};

Prep prep_layout(const Dims& m, bool sin, bool sout) {
    Prep p;
    long o = 0;
    auto take = [&](long n) { long r = o; o += up(n, 64); return r; };
    p.afold = take((long)m.H3 * m.Kfe);
    p.afold3 = take((long)m.nch * 4 * m.KFW * 3 * 256);
    p.afold_h = take((long)m.nch * 4 * ((m.KFW + 1) / 2) * 6 * 256);   // front-end weights as fp16 pairs (LDS image of v5)
    p.cfold = take(m.H3);
    p.wrec = take((long)(m.H / 4) * m.nch * 256);
    p.wrec2 = take((long)m.nch * 4 * m.nch * 256);
    p.wrec_h = take((long)m.nch * 4 * (m.H / 32) * 2 * 256);   // fp16-pair image of wrec2 for k_gru_steps_v5
    // fp16-triple images of k_gru_steps_v6 (exact fp32 operands): recurrent B operands per (octet, wave, 16-k step, limb) and
    // the front-end LDS image; only for the sizes that kernel is built for
    p.wrec_t = exact3_ok(m) ? take((long)(m.H / 8) * 4 * exact3_kpw(m) * 3 * 256) : -1;
    p.afold_t = exact3_ok(m) ? take((long)(m.H / 8) * 4 * m.KFW * 3 * 256) : -1;
    p.wrec_l2b = exact3_ok(m) ? take((long)(m.H / 8) * 4 * exact3_kpw(m) * 128) : -1;   // third weight limbs as bf8 bytes (streamed form)
    p.bhn = take(m.H);
    p.wyT = take((long)m.H3 * m.Co);
    p.wo = take((long)m.Cop * m.H);
    p.bo = take(m.Cop);
    p.wo2 = take((long)m.Cop * m.H);
    p.bo2 = take(m.Cop);
    p.wo3 = exact3_ok(m) ? take((long)((m.Cop + 31) / 32) * (m.H / 16) * 3 * 256) : -1;   // projection B operands as limb triples
    p.sin_w = sin ? take((long)m.C * m.C) : -1;
    p.sin_b = sin ? take(m.C) : -1;
    p.sout_w = sout ? take((long)m.Co * m.Co) : -1;
    p.sout_b = sout ? take(m.Co) : -1;
    p.total = o;
    return p;
}

// pass workspace: offsets in floats.  Brows = total batch rows of the pass (cells stacked along the batch axis)
struct Work {
    long status, xnp, xs, xs_plane, xt, xt_slack, gx, hbuf, hs, y, dy, gx0, prof, flags, total;
    int Bp, Tp;
    long mtot;
};

Work work_layout(const Dims& m, int Brows, int T) {
    Work w;
    // batch rows are padded to whole row tiles: one 16-row tile for the word-exchange kernel (at most 3 rows), 32-row tiles
    // (k_gru_steps_v6: exact operands) from 4 rows on wherever that kernel exists -- ONE arithmetic width for every batch size,
    // a pass of 4..16 rows runs a half-empty tile rather than the 22-bit pair kernel -- else 16-row tiles up to 16 rows
    w.Bp = Brows <= 3 ? 16 : (exact3_ok(m) || Brows > 16 ? (int)up(Brows, 32) : 16);
    w.Tp = T + 2 * m.pad;
    w.mtot = (long)(T + 1) * w.Bp;
    long o = 0;
    auto take = [&](long n) { long r = o; o += up(n, 64); return r; };
    w.status = take(64);  // int32[4] status + barrier counter at word 8
    w.xnp = take((long)Brows * w.Tp * m.Cp + 64L * m.KFW + 64);
    w.xs_plane = up((long)Brows * w.Tp * m.Cp + 64L * m.KFW + 64, 8);   // halves per plane (hi, lo) of the fp16-pair copy
    w.xs = take(w.xs_plane);                                              // 2 planes x 2 bytes = xs_plane floats
    // k_gru_steps_v6's input: limb triples (5 bytes per element) + zero slack for the K padding of the last frames (16-bit words)
    w.xt_slack = (64L * m.KFW / 8 + 2) * 640 + 64;
    w.xt = exact3_ok(m) && w.Bp % 32 == 0 ? take(((long)w.Bp * w.Tp * m.Cp * 5 / 2 + w.xt_slack) / 2 + 8) : -1;
    w.gx = take((long)Brows * w.Tp * m.H3);
    w.hbuf = take((long)m.nch * w.mtot * 16);
    w.hs = take((long)m.nch * w.mtot * 24);     // exchanged state as fp16 pairs (64 B per row and 16 units) or triples (96 B)
    w.y = take((long)T * w.Bp * m.Cop);
    w.dy = take((long)w.Bp * m.Co);
    w.gx0 = take((long)w.Bp * m.H3);   // frame-0 feedback correction of the exact-operand kernel (prologue role gx0)
    w.prof = take(2048);  // long long[<=256 blocks][4] step-timing counters
    w.flags = take((long)(w.Bp / 16) * m.nch);
    w.total = o;
    return w;
}

bool prof_begin(hipStream_t st, int rows = 0, int cin = 0) {
    if (cx().prof.used == cx().prof.start.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
        cx().prof.start.push_back(a);
        cx().prof.stop.push_back(b);
        cx().prof.rows.push_back(0);
        cx().prof.cin.push_back(0);
    }
    cx().prof.rows[cx().prof.used] = rows;
    cx().prof.cin[cx().prof.used] = cin;
    return hipEventRecord(cx().prof.start[cx().prof.used], st) == hipSuccess;
}
void prof_end(hipStream_t st) {
    (void)hipEventRecord(cx().prof.stop[cx().prof.used], st);
    cx().prof.used++;
}

struct TrainProf {     // RAII bracket; inactive unless the option is set
    hipStream_t st;
    bool on;
    TrainProf(hipStream_t st_, int cls, double flop) : st(st_), on(false) {
        if (!opt(OPT_TRAIN_PROFILE)) return;
        if (cx().tprof.used == cx().tprof.start.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            cx().tprof.start.push_back(a); cx().tprof.stop.push_back(b); cx().tprof.cls.push_back(0); cx().tprof.flop.push_back(0.0);
        }
        cx().tprof.cls[cx().tprof.used] = cls;
        cx().tprof.flop[cx().tprof.used] = flop;
        on = hipEventRecord(cx().tprof.start[cx().tprof.used], st) == hipSuccess;
    }
    void end() {
        if (!on) return;
        (void)hipEventRecord(cx().tprof.stop[cx().tprof.used], st);
        cx().tprof.used++;
        on = false;
    }
    ~TrainProf() { end(); }
};

// a few words to zero in front of a chain of kernels: as a KERNEL (hipMemsetAsync of 32 bytes is followed by ~15 us of idle
// stream before the next kernel starts, rocprofv3 trace on MI355X; a kernel chains back to back)
__global__ void k_clear_words(int32_t* p, int n) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = 0;
}
inline hipError_t clear_words(void* p, int n, hipStream_t st) {
    hipLaunchKernelGGL((k_clear_words), dim3(1), dim3(64), 0, st, (int32_t*)p, n);
    return hipGetLastError();
}

int cu_count() {
    static int cached[16] = {};          // (asked several times per pass: the layouts depend on it)
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 16 && cached[dev] > 0) return cached[dev];
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (dev >= 0 && dev < 16) cached[dev] = n;
    return n;
}

// One cell of a pass: passes that share weights and have no mutual dependence (rec || cv of a cycle) run as ONE
// pass whose batch is the cells' batches stacked; the recurrence then has extra independent row tiles to interleave.
struct Cell {
    const cvae_pass_input* in;
    const float* y_in;   // [B][Co]
    const float* h_in;   // [B][H] or null
    float* trj_out;      // [B][T][Co]
    float* y_last;       // [B][Co] or null
    float* h_last;       // [B][H] or null
};

int run_pass(const Dims& m, const cvae_net_desc* d, const float* P, const Cell* cells, int ncell, int B, int T,
             int clamp_lat_dim, float* ws, int* status, int flags, hipStream_t st, bool clear_status = false) {
    const Prep pl = prep_layout(m, d->has_scale_in != 0, d->has_scale_out != 0);
    const int Brows = ncell * B;
    const Work wl = work_layout(m, Brows, T);
    if (cx().status_sink) status = cx().status_sink;    // host-visible sticky word: a time-out is seen without reading the workspace back
    for (int c = 0; c < ncell; ++c) {
        const cvae_pass_input* in = cells[c].in;
        const int w_in = in->seg0.width + (in->lat ? in->lat_dim : in->seg1.width);
        if (w_in != m.C) return fail(-1, "pass input width %d != in_dim %d", w_in, m.C);
    }
    unsigned* bar = (unsigned*)(ws + wl.status) + 8;  // status words themselves are sticky: zeroed by the entry point
    float* xnp = ws + wl.xnp;
    float* gx = ws + wl.gx;
    float* hbuf = ws + wl.hbuf;
    float* y = ws + wl.y;
    float* dy = ws + wl.dy;
    unsigned* hflags = (unsigned*)(ws + wl.flags);
    const int nrt = wl.Bp / 16;
    if (ncell > CVAE_MAX_CELLS) return fail(-1, "at most %d stacked cells per pass", CVAE_MAX_CELLS);
    const int cus = cu_count();
    // k_gru_steps_v6 (exact fp32 operands as fp16 triples): 32-row tiles, 8-unit octets, every block resident
    const bool use_exact3 = (flags & CVAE_FLAG_PERSISTENT) && (flags & CVAE_FLAG_EXACT3) && !(flags & CVAE_FLAG_GENERIC_STEP) &&
                            !(flags & CVAE_FLAG_HOISTED_FRONTEND) && T > 1 && exact3_ok(m) && wl.Bp % 32 == 0 &&
                            cus >= m.H / 8 && (long)m.nch * wl.mtot * 80 < (1L << 31);

    // k_gru_steps_ll (at most three rows: a step is one store + one polled load per unit, plain fp32 FMAs): every block resident
    const bool use_ll = (flags & CVAE_FLAG_PERSISTENT) && (flags & CVAE_FLAG_EXACT3) && !(flags & CVAE_FLAG_GENERIC_STEP) && T > 1 &&
                        Brows <= 3 && T < 65536 && m.H % 64 == 0 && m.H <= 1024 && cus >= m.H / 4 && !opt(OPT_NO_LL);

    bool gx0_ready = false;
    {   // one prologue launch: assemble + scale_in + padding, slot-0 init, frame-0 feedback correction, zeroing
        ProParams pp;
        memset(&pp, 0, sizeof(pp));
        for (int c = 0; c < ncell; ++c) {
            const cvae_pass_input* in = cells[c].in;
            pp.cell[c].seg0 = CvaeSeg{in->seg0.ptr, in->seg0.width, in->seg0.row_stride};
            pp.cell[c].seg1 = CvaeSeg{in->seg1.ptr, in->seg1.width, in->seg1.row_stride};
            pp.cell[c].lat = in->lat;
            pp.cell[c].eps = in->eps;
            pp.cell[c].seed = in->seed;
            pp.cell[c].draw = in->draw_id;
            pp.cell[c].y_in = cells[c].y_in;
            pp.cell[c].h_in = cells[c].h_in;
            pp.cell[c].frames = in->frames;
            pp.cell[c].n_draws = in->n_draws;
            pp.cell[c].ctx_before = in->ctx_before; pp.cell[c].ctx_after = in->ctx_after;
            pp.cell[c].draw_frame0 = (long)in->draw_frame0; pp.cell[c].eps_stride = (long)in->eps_draw_stride;
            if (in->frames < 0 || in->frames > T) return fail(-1, "cell %d: frames %d outside [0, T=%d]", c, in->frames, T);
            if (in->ctx_before < 0 || in->ctx_after < 0 || in->draw_frame0 < 0 || in->eps_draw_stride < 0)
                return fail(-1, "cell %d: negative window context / draw origin", c);
            if ((in->ctx_before || in->ctx_after || in->draw_frame0) && B != 1)
                return fail(-1, "cell %d: windows of a longer utterance (ctx_before / ctx_after / draw_frame0) are single-row cells, B=%d", c, B);
            if (in->lat) pp.L = in->lat_dim;
        }
        bool many_draws = false;
        for (int c = 0; c < ncell; ++c) many_draws = many_draws || (cells[c].in->lat && cells[c].in->n_draws > 1);
        pp.ncell = ncell;
        pp.frame0 = (uint64_t)cx().draw_row0 * (uint64_t)T;
        pp.sin_w = d->has_scale_in ? P + pl.sin_w : nullptr;
        pp.sin_b = d->has_scale_in ? P + pl.sin_b : nullptr;
        pp.wo = P + pl.wo; pp.bo = P + pl.bo;
        pp.B = B; pp.T = T; pp.C = m.C; pp.Cp = m.Cp; pp.pad = m.pad; pp.Co = m.Co; pp.H = m.H; pp.Bp = wl.Bp;
        pp.nslack = 64 * m.KFW + 64;
        pp.mtot = wl.mtot;
        pp.xnp = xnp; pp.hbuf = hbuf; pp.dy = dy;
        // (k_gru_steps_v6 reads the fp32 buffers themselves and splits in registers: no limb copies)
        pp.hx = use_exact3 ? ws + wl.hs : nullptr;    // (the pair buffer's space: same size, never both in one pass)
        pp.xt = use_exact3 ? ws + wl.xt : nullptr;
        pp.nxt_slack = (int)wl.xt_slack;
        pp.hs = (!use_exact3 && (flags & CVAE_FLAG_SPLIT_F16)) ? ws + wl.hs : nullptr;
        pp.xs = pp.hs ? ws + wl.xs : nullptr;
        pp.xs_plane = wl.xs_plane;
        // bar (8 words) ... flags are not adjacent: zero the flags here, the barrier words with the status block
        pp.zero_words = hflags; pp.nzero = nrt * m.nch;
        pp.zero_status = clear_status ? status : nullptr;     // (the recurrent kernel, the first writer of these words, runs behind the prologue)
        pp.ll_counter = use_ll ? (unsigned*)(ws + wl.status) + 16 : nullptr;
        pp.nA = (use_exact3 ? wl.Bp / 32 : Brows) * wl.Tp;      // v6: one block per (32-row tile, padded frame)
        pp.nH = (int)nblk((long)wl.Bp * m.H, 1024);
        pp.nD = (int)nblk((long)Brows * m.Co, 64);
        bool any_h_in = false;
        for (int c = 0; c < ncell; ++c) any_h_in = any_h_in || cells[c].h_in != nullptr;
        gx0_ready = use_exact3 && !any_h_in && m.H3 % 4 == 0 && !opt(OPT_T0_IN_KERNEL);
        pp.nG = gx0_ready ? (int)nblk((long)Brows * m.H3 / 4, 256) : 0;
        pp.gx0 = ws + wl.gx0; pp.wyT = P + pl.wyT;
        if (use_exact3)
            hipLaunchKernelGGL((k_prologue), dim3(pp.nA + pp.nH + pp.nD + pp.nG + 1), dim3(256),
                               (size_t)32 * (m.C + 1) * sizeof(float) + (size_t)(m.Cp / 8) * 1280 + (size_t)32 * pp.L * sizeof(float), st, pp);
        else if (many_draws)     // 256 threads per block: the draws of a frame are summed in parallel slices
            hipLaunchKernelGGL((k_prologue), dim3(pp.nA + pp.nH + pp.nD + 1), dim3(256), (size_t)(m.C + 1024 + 256) * sizeof(float), st, pp);
        else
            hipLaunchKernelGGL((k_prologue), dim3(pp.nA + pp.nH + pp.nD + 1), dim3(64), m.C * sizeof(float), st, pp);
    }
    // (the barrier counter of the any-H persistent kernel is zeroed where that kernel is launched: the dataflow kernels do
    // not use it, and a memset is a launch of its own)
    const size_t step_lds = 4 * 64 * 20 * sizeof(float);
    const bool want_persistent = (flags & CVAE_FLAG_PERSISTENT) && T > 1;
    const bool small = (long)m.nch * wl.mtot * 64 < (1L << 31);
    const bool tuned_ok = want_persistent && !(flags & CVAE_FLAG_GENERIC_STEP) && small &&
                          (m.H == 1024 || m.H == 64) && cus >= m.nch;
    int RT = m.nch > 0 ? cus / m.nch : 1;
    RT = RT < 1 ? 1 : (RT > nrt ? nrt : RT);
    if (opt(OPT_MAX_RT) >= 1 && opt(OPT_MAX_RT) < RT) RT = (int)opt(OPT_MAX_RT);   // tests: several row tiles per block on small problems
    const size_t lds2 = (4 * 16 * 84 + 16 * 16) * sizeof(float);
    const bool prof = (flags & CVAE_FLAG_PROFILE) && prof_begin(st, ncell * B, m.C);
    bool launched = false;
    // ---- at most three rows: input-side GEMM for all frames, then the word-exchange kernel
    if (use_ll) {
        const int M = Brows * wl.Tp, N = m.H3;
        hipLaunchKernelGGL((k_gemm_nt<4, 4, 2, 2, false>), dim3(nblk(N, 128), nblk(M, 128)), dim3(256), 0, st,
                           (const float*)xnp, (long)m.Cp, 0L, P + pl.afold, (long)m.Kfe, P + pl.cfold, gx, (long)m.H3,
                           M, N, m.Kfe);
        StepLLParams q;
        q.hbuf = hbuf; q.mtot = wl.mtot; q.xbuf = ws + wl.hs; q.wrec2 = P + pl.wrec2; q.gx = gx; q.gx_bstride = (long)wl.Tp * m.H3;
        q.bhn = P + pl.bhn; q.B = Brows; q.Bp = wl.Bp; q.H = m.H; q.T = T; q.status = status;
        q.prof = (flags & CVAE_FLAG_STEP_TIMING) ? (long long*)(ws + wl.prof) : nullptr;
        q.wyT = P + pl.wyT; q.dy = dy; q.Co = m.Co;
        q.dbg = (int*)(ws + wl.status);
        q.backoff = opt(OPT_LL_BACKOFF) >= 0 ? (int)opt(OPT_LL_BACKOFF) : (Brows == 1 ? 22 : 20);   // swept per row count (round 4, after the prefetch reordering: profiles/r04_notes.md; round 2: 18 / 16)
        q.nonce_src = (const unsigned*)(ws + wl.status) + 16;
        const dim3 gl(m.H / 4);
        const size_t ldsl = (size_t)(2 * 64 * 49 + 4 * 48) * sizeof(float);
        hipError_t e = Brows == 1 ? cvae_launch_coop(k_gru_steps_ll<1>, gl, dim3(256), ldsl, st, q)
                     : Brows == 2 ? cvae_launch_coop(k_gru_steps_ll<2>, gl, dim3(256), ldsl, st, q)
                                  : cvae_launch_coop(k_gru_steps_ll<3>, gl, dim3(256), ldsl, st, q);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(-3, "small-batch recurrent kernel failed to launch: %s", hipGetErrorString(e));
        }
        launched = true;
    }
    // ---- exact-operand fused kernel (fp16 triples, six MFMAs per product)
    if (use_exact3) {
        Step6Params q;
        q.hbuf = hbuf; q.mtot = wl.mtot; q.hx = ws + wl.hs; q.wrec3 = P + pl.wrec_t; q.afold3 = P + pl.afold_t; q.w2s = P + pl.wrec_l2b;
        q.cfold = P + pl.cfold; q.bhn = P + pl.bhn; q.xt = ws + wl.xt; q.Tp = wl.Tp; q.Cp = m.Cp;
        q.B = Brows; q.Bp = wl.Bp; q.H = m.H; q.T = T; q.flags = hflags; q.status = status;
        q.prof = (flags & CVAE_FLAG_STEP_TIMING) ? (long long*)(ws + wl.prof) : nullptr;
        q.wyT = P + pl.wyT; q.dy = dy; q.Co = m.Co;
        q.gx0 = gx0_ready ? ws + wl.gx0 : nullptr;
        {   // the fp32 state copy is only read by the raw projection (y_last), k_hlast and the fallback projection kernels
            bool need = v6_limbs(m) != 3 || opt(OPT_OLD_OUTPROJ);
            for (int c = 0; c < ncell; ++c) need = need || cells[c].y_last != nullptr || cells[c].h_last != nullptr;
            q.want_f32 = (need || (opt(OPT_EXP) & 16)) ? 1 : 0;     // (exp bit 4: always write it, for A/B measurements)
        }
        const int NB = m.H / 8, nrt32 = wl.Bp / 32;
        int RT6 = cus / NB;
        RT6 = RT6 < 1 ? 1 : (RT6 > nrt32 ? nrt32 : RT6);
        if (opt(OPT_MAX_RT) >= 1 && opt(OPT_MAX_RT) < RT6) RT6 = (int)opt(OPT_MAX_RT);
        q.rts = RT6;
        q.exp = (int)opt(OPT_EXP);   // measurement switches only
        // One row tile per block (B <= 64 at hu1024): nothing can be published before the other blocks' front-ends are through, and
        // 256 waves polling through that window slow the publishes and operand loads they wait for.  The decoder's front-end is a
        // quarter shorter than the encoder's (KFW 6 vs 8), so its blocks arrive at the poll earlier: swept on MI355X
        // (tools/ab_eval_exp.sh, round 5), 64-cycle units -- decoder pass 421.7 (0) / 405.8 (4) / 397.1 (8) / 401.0 (12) / 409.0 us
        // (16), encoder pass 412.3 / 411.0 / 416.6 / 423.9 / 434.5 us.
        q.backoff = opt(OPT_V6_BACKOFF) >= 0 ? (int)opt(OPT_V6_BACKOFF) : (m.H == 1024 ? (m.KFW <= 6 ? 8 : 2) : 0);
        const size_t lds6 = (size_t)(4 * 32 * 40 + 32 * 8 + 384 + 4 * m.KFW * v6_limbs(m) * 256) * sizeof(float);
        const dim3 g6(NB * RT6);
        hipError_t e = hipErrorUnknown;
        if (m.H == 1024 && m.KFW == 8) e = cvae_launch_coop(k_gru_steps_v6<16, 8>, g6, dim3(256), lds6, st, q);
        else if (m.H == 1024 && m.KFW == 6) e = cvae_launch_coop(k_gru_steps_v6<16, 6>, g6, dim3(256), lds6, st, q);
        else if (m.H == 2048 && v6_limbs(m) == 2 && m.KFW == 8) e = cvae_launch_coop(k_gru_steps_v6<32, 8, 2>, g6, dim3(256), lds6, st, q);
        else if (m.H == 2048 && v6_limbs(m) == 2 && m.KFW == 11) e = cvae_launch_coop(k_gru_steps_v6<32, 11, 2>, g6, dim3(256), lds6, st, q);
        else if (m.H == 2048 && m.KFW == 8) e = cvae_launch_coop(k_gru_steps_v6<32, 8, 3, true>, g6, dim3(256), lds6, st, q);
        else if (m.H == 2048 && m.KFW == 11) e = cvae_launch_coop(k_gru_steps_v6<32, 11, 3, true>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && v6_w2s(m) && m.KFW == 3) e = cvae_launch_coop(k_gru_steps_v6<1, 3, 3, true>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && v6_w2s(m) && m.KFW == 2) e = cvae_launch_coop(k_gru_steps_v6<1, 2, 3, true>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && v6_w2s(m) && m.KFW == 1) e = cvae_launch_coop(k_gru_steps_v6<1, 1, 3, true>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && v6_limbs(m) == 2 && m.KFW == 3) e = cvae_launch_coop(k_gru_steps_v6<1, 3, 2>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && v6_limbs(m) == 2 && m.KFW == 2) e = cvae_launch_coop(k_gru_steps_v6<1, 2, 2>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && v6_limbs(m) == 2 && m.KFW == 1) e = cvae_launch_coop(k_gru_steps_v6<1, 1, 2>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && m.KFW == 3) e = cvae_launch_coop(k_gru_steps_v6<1, 3>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && m.KFW == 2) e = cvae_launch_coop(k_gru_steps_v6<1, 2>, g6, dim3(256), lds6, st, q);
        else if (m.H == 64 && m.KFW == 1) e = cvae_launch_coop(k_gru_steps_v6<1, 1>, g6, dim3(256), lds6, st, q);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(-3, "exact-operand recurrent kernel failed to launch: %s", hipGetErrorString(e));
        }
        launched = true;
    }
    // ---- fused kernel: front-end + recurrence in one cooperative launch (no gx buffer, no GEMM launch)
    if (!launched && tuned_ok && !(flags & CVAE_FLAG_HOISTED_FRONTEND) && (m.KFW == 8 || m.KFW == 6 || m.KFW <= 2)) {
        Step3Params q;
        q.hbuf = hbuf; q.mtot = wl.mtot; q.wrec2 = P + pl.wrec2; q.afold2 = nullptr; q.cfold = P + pl.cfold;
        q.xnp = xnp; q.Tp = wl.Tp; q.Cp = m.Cp; q.bhn = P + pl.bhn; q.B = Brows; q.Bp = wl.Bp; q.H = m.H; q.T = T;
        q.flags = hflags; q.status = status;
        q.prof = (flags & CVAE_FLAG_STEP_TIMING) ? (long long*)(ws + wl.prof) : nullptr;
        q.wyT = P + pl.wyT; q.dy = dy; q.Co = m.Co;
        hipError_t e = hipErrorUnknown;
        q.rts = RT;
        q.hs = ws + wl.hs; q.wrec_h = P + pl.wrec_h; q.xs = ws + wl.xs; q.xs_plane = wl.xs_plane;
        if ((flags & CVAE_FLAG_SPLIT_F16) && m.H % 32 == 0) {
            // v5: v4 with the recurrent product as three fp16 MFMAs on (hi, lo) pairs
            Step3Params q5 = q;
            q5.afold2 = P + pl.afold_h;
            q5.exp = (int)opt(OPT_EXP);   // measurement switches only
            const size_t lds5 = lds2 + (size_t)4 * ((m.KFW + 1) / 2) * 6 * 256 * sizeof(float);
            const dim3 g5(m.nch * RT);
            if (m.H == 1024 && m.KFW == 8) e = cvae_launch_coop(k_gru_steps_v5<16, 8>, g5, dim3(256), lds5, st, q5);
            else if (m.H == 1024 && m.KFW == 6) e = cvae_launch_coop(k_gru_steps_v5<16, 6>, g5, dim3(256), lds5, st, q5);
            else if (m.H == 64 && m.KFW == 2) e = cvae_launch_coop(k_gru_steps_v5<1, 2>, g5, dim3(256), lds5, st, q5);
            else if (m.H == 64 && m.KFW == 1) e = cvae_launch_coop(k_gru_steps_v5<1, 1>, g5, dim3(256), lds5, st, q5);
            if (e == hipSuccess) launched = true; else (void)hipGetLastError();
        }
        if (!launched) {
            // v4: front-end weights in LDS, double-buffered h operands
            Step3Params q4 = q;
            q4.afold2 = P + pl.afold3;
            q4.exp = (int)opt(OPT_EXP);   // measurement switches only
            const size_t lds4 = lds2 + (size_t)4 * m.KFW * 3 * 256 * sizeof(float);
            const dim3 g4(m.nch * RT);
            if (m.H == 1024 && m.KFW == 8) e = cvae_launch_coop(k_gru_steps_v4<16, 8>, g4, dim3(256), lds4, st, q4);
            else if (m.H == 1024 && m.KFW == 6) e = cvae_launch_coop(k_gru_steps_v4<16, 6>, g4, dim3(256), lds4, st, q4);
            else if (m.H == 64 && m.KFW == 2) e = cvae_launch_coop(k_gru_steps_v4<1, 2>, g4, dim3(256), lds4, st, q4);
            else if (m.H == 64 && m.KFW == 1) e = cvae_launch_coop(k_gru_steps_v4<1, 1>, g4, dim3(256), lds4, st, q4);
            if (e == hipSuccess) launched = true; else (void)hipGetLastError();
        }
    }
    if (!launched) {
        // gx[b*Tp + t] = afold . xnp[b, t:t+R, :] + cfold : one GEMM over overlapping rows (lda = Cp)
        const int M = Brows * wl.Tp, N = m.H3;
        hipLaunchKernelGGL((k_gemm_nt<4, 4, 2, 2, false>), dim3(nblk(N, 128), nblk(M, 128)), dim3(256), 0, st,
                           (const float*)xnp, (long)m.Cp, 0L, P + pl.afold, (long)m.Kfe, P + pl.cfold, gx, (long)m.H3,
                           M, N, m.Kfe);
    }
    // ---- tuned 2-D kernel with a hoisted front-end GEMM: block = (16 hidden units) x (row tiles i0, i0+RT, ...)
    if (!launched && tuned_ok) {
        Step2Params q;
        q.hbuf = hbuf; q.mtot = wl.mtot; q.wrec2 = P + pl.wrec2; q.gx = gx; q.gx_bstride = (long)wl.Tp * m.H3;
        q.bhn = P + pl.bhn; q.B = Brows; q.Bp = wl.Bp; q.H = m.H; q.T = T; q.flags = hflags; q.status = status;
        q.prof = (flags & CVAE_FLAG_STEP_TIMING) ? (long long*)(ws + wl.prof) : nullptr;
        q.wyT = P + pl.wyT; q.dy = dy; q.Co = m.Co;
        hipError_t e = m.H == 1024 ? cvae_launch_coop(k_gru_steps_v2<16>, dim3(m.nch, RT), dim3(256), lds2, st, q)
                                   : cvae_launch_coop(k_gru_steps_v2<1>, dim3(m.nch, RT), dim3(256), lds2, st, q);
        if (e == hipSuccess) launched = true; else (void)hipGetLastError();
    }
    if (!launched) {
        StepParams sp;
        sp.hbuf = hbuf; sp.mtot = wl.mtot; sp.wrec = P + pl.wrec; sp.gx = gx; sp.gx_bstride = (long)wl.Tp * m.H3;
        sp.bhn = P + pl.bhn; sp.B = Brows; sp.Bp = wl.Bp; sp.H = m.H; sp.T = T; sp.t0 = 0; sp.bar = bar; sp.status = status;
        sp.nwg = (unsigned)(m.H / 4);
        sp.prof = nullptr;
        sp.wyT = P + pl.wyT; sp.dy = dy; sp.Co = m.Co;
        // every block of a persistent launch must be resident: one 256-thread block per CU is always admitted
        if (want_persistent && (cus <= 0 || (int)sp.nwg <= cus)) {
            CVAE_HIP_OK(hipMemsetAsync(bar, 0, 8 * sizeof(unsigned), st));
            hipError_t e = hipSuccess;
            e = cvae_launch_coop(k_gru_steps<true>, dim3(sp.nwg), dim3(256), step_lds, st, sp);
            if (e == hipSuccess) launched = true; else (void)hipGetLastError();
        }
        if (!launched) {
            for (int t = 0; t < T; ++t) {
                sp.t0 = t;
                hipLaunchKernelGGL((k_gru_steps<false>), dim3(sp.nwg), dim3(256), step_lds, st, sp);
            }
        }
    }
    if (prof) prof_end(st);

    bool want_raw = false;
    for (int c = 0; c < ncell; ++c) want_raw = want_raw || cells[c].y_last != nullptr;
    const int ntn = m.Cop / 16;
    if (!want_raw && use_exact3 && v6_limbs(m) == 3 && !opt(OPT_OLD_OUTPROJ)) {
        // the v6 pass left the state as limb triples in the exchange buffer: project from there, same exact arithmetic
        Out6Params op;
        op.hx = ws + wl.hs; op.mtot = wl.mtot; op.wo3 = P + pl.wo3; op.bo2 = P + pl.bo2; op.H = m.H; op.Bp = wl.Bp; op.T = T;
        op.B = B; op.ncell = ncell; op.Co = m.Co; op.clamp_from = d->has_scale_out ? -1 : clamp_dim(clamp_lat_dim);
        op.clamp_min = clamp_floor(clamp_lat_dim);
        for (int c = 0; c < CVAE_MAX_CELLS; ++c) op.out[c] = c < ncell ? cells[c].trj_out : nullptr;
        const dim3 g((unsigned)((long)T * wl.Bp / 32), (unsigned)((m.Cop + 31) / 32));
        const size_t lds = (size_t)4 * 32 * 36 * sizeof(float);
        if (m.H == 1024) hipLaunchKernelGGL((k_outproj_v6<16>), g, dim3(256), lds, st, op);
        else if (m.H == 2048) hipLaunchKernelGGL((k_outproj_v6<32>), g, dim3(256), lds, st, op);
        else hipLaunchKernelGGL((k_outproj_v6<1>), g, dim3(256), lds, st, op);
    } else if (!want_raw && (ntn == 1 || ntn == 4 || ntn == 8)) {
        // fused projection: scale_out folded in, clamp, written straight into [B][T][Co]
        OutParams op;
        op.hbuf = hbuf; op.mtot = wl.mtot; op.wo2 = P + pl.wo2; op.bo2 = P + pl.bo2; op.H = m.H; op.Bp = wl.Bp; op.T = T;
        op.B = B; op.ncell = ncell; op.Co = m.Co; op.clamp_from = d->has_scale_out ? -1 : clamp_dim(clamp_lat_dim);
        op.clamp_min = clamp_floor(clamp_lat_dim);
        for (int c = 0; c < CVAE_MAX_CELLS; ++c) op.out[c] = c < ncell ? cells[c].trj_out : nullptr;
        const unsigned nb = (unsigned)((long)T * wl.Bp / 16);
        const size_t lds = (size_t)4 * 16 * (m.Cop + 4) * sizeof(float);
        if (ntn == 1) hipLaunchKernelGGL((k_outproj<1>), dim3(nb), dim3(256), lds, st, op);
        else if (ntn == 4) hipLaunchKernelGGL((k_outproj<4>), dim3(nb), dim3(256), lds, st, op);
        else hipLaunchKernelGGL((k_outproj<8>), dim3(nb), dim3(256), lds, st, op);
    } else {
        // y[t*Bp + b] = out_1(h_t): A = hbuf slots 1..T (chunk-major), rows offset by Bp; then the epilogue kernel
        const int M = T * wl.Bp, N = m.Co;
        hipLaunchKernelGGL((k_gemm_nt<2, 4, 4, 1, true>), dim3(nblk(N, 64), nblk(M, 128)), dim3(256), 0, st,
                           (const float*)(hbuf + (long)wl.Bp * 16), 0L, wl.mtot, P + pl.wo, (long)m.H, P + pl.bo, y,
                           (long)m.Cop, M, N, m.H);
        for (int c = 0; c < ncell; ++c) {
            EpiParams ep;
            ep.y = y; ep.ldy = m.Cop;
            ep.sout_w = d->has_scale_out ? P + pl.sout_w : nullptr;
            ep.sout_b = d->has_scale_out ? P + pl.sout_b : nullptr;
            ep.clamp_from = d->has_scale_out ? -1 : clamp_dim(clamp_lat_dim);
            ep.clamp_min = clamp_floor(clamp_lat_dim);
            ep.B = B; ep.Bp = wl.Bp; ep.T = T; ep.Co = m.Co; ep.b0 = c * B;
            ep.trj_out = cells[c].trj_out; ep.y_last = cells[c].y_last;
            ep.t_last = (cells[c].in && cells[c].in->frames > 0 && cells[c].in->frames < T ? cells[c].in->frames : T) - 1;
            hipLaunchKernelGGL((k_epilogue), dim3(B * T), dim3(64), m.Co * sizeof(float), st, ep);
        }
    }
    for (int c = 0; c < ncell; ++c)
        if (cells[c].h_last)
            // the state a cell carries on is the one behind ITS last valid frame (slot `frames`), not behind the pass's T steps: a
            // cell shorter than the pass keeps stepping over padding (or, for a window with ctx_after > 0, over frames that belong
            // to the next window), and continuing from slot T restarted the next window from the wrong state
            hipLaunchKernelGGL((k_hlast), dim3(nblk((long)B * m.H, 256)), dim3(256), 0, st, (const float*)hbuf, wl.mtot,
                               cells[c].h_last, B, wl.Bp, m.H,
                               cells[c].in && cells[c].in->frames > 0 && cells[c].in->frames < T ? cells[c].in->frames : T, c * B);
    CVAE_HIP_OK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

const char* cvae_last_error_string(void) { return g_err; }
int cvae_abi_version(void) { return CVAE_ABI_VERSION; }

cvae_ctx* cvae_ctx_create(void) {
    cvae_ctx* c = new (std::nothrow) cvae_ctx();
    if (!c) {
        (void)fail(-3, "out of memory");
        return nullptr;
    }
    for (int i = 0; i < OPT_COUNT; ++i) c->opt[i] = g_opt[i].dflt;
    return c;
}

int cvae_ctx_destroy(cvae_ctx* ctx) {
    if (!ctx) return 0;
    if (tl_ctx == ctx) return fail(-1, "cvae_ctx_destroy from inside a call on the same context");
    // the events the context created (profiling brackets, side-stream joins); the caller has synchronised whatever used them
    for (hipEvent_t e : ctx->prof.start) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->prof.stop) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->tprof.start) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->tprof.stop) (void)hipEventDestroy(e);
    for (SideDone& d : ctx->side_done)
        if (d.ev) (void)hipEventDestroy(d.ev);
    for (hipEvent_t e : {ctx->side_ready, ctx->side_join, ctx->side_last, ctx->mask_fork, ctx->mask_join})
        if (e) (void)hipEventDestroy(e);
    delete ctx;
    return 0;
}

int cvae_set_status_sink(cvae_ctx* ctx, int32_t* sink) {
    CVAE_ENTER(ctx);
    cx().status_sink = sink;
    return 0;
}

int cvae_set_draw_origin(cvae_ctx* ctx, int64_t row0, int64_t global_rows, int64_t frames_per_row) {
    CVAE_ENTER(ctx);
    if (row0 < 0 || global_rows < 0 || frames_per_row < 0 || (global_rows > 0 && row0 >= global_rows))
        return fail(-1, "bad draw origin: row0 %lld of %lld rows", (long long)row0, (long long)global_rows);
    cx().draw_row0 = (long)row0;
    cx().draw_rows = (long)global_rows;
    cx().draw_frames = (long)frames_per_row;
    return 0;
}

int cvae_selftest_limbs(cvae_ctx* ctx, const float* x, float* y, int64_t n, void* stream) {
    CVAE_ENTER(ctx);
    if (!x || !y || n < 0 || n % 8) return fail(-1, "selftest: n must be a non-negative multiple of 8");
    if (n) hipLaunchKernelGGL((k_selftest_limbs), dim3(nblk(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n);
    CVAE_HIP_OK(hipGetLastError());
    return 0;
}

int cvae_selftest_occupy(cvae_ctx* ctx, int blocks, size_t lds_bytes, int64_t cycles, void* stream) {
    CVAE_ENTER(ctx);
    if (blocks < 1 || cycles < 0 || lds_bytes > 160 * 1024) return fail(-1, "selftest_occupy: bad argument");
    hipLaunchKernelGGL((k_selftest_occupy), dim3(blocks), dim3(256), lds_bytes, (hipStream_t)stream, (long long)cycles, (int)(lds_bytes / 4));
    CVAE_HIP_OK(hipGetLastError());
    return 0;
}

int cvae_set_option(cvae_ctx* ctx, const char* name, int64_t value) {
    CVAE_ENTER(ctx);
    if (!name) return fail(-1, "null option name");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(g_opt[i].name, name)) {
            ctx->opt[i] = (long)value;
            return 0;
        }
    return fail(-1, "unknown option '%s'", name);
}

int cvae_get_option(cvae_ctx* ctx, const char* name, int64_t* value) {
    CVAE_ENTER(ctx);
    if (!name || !value) return fail(-1, "null argument");
    for (int i = 0; i < OPT_COUNT; ++i)
        if (!strcmp(g_opt[i].name, name)) {
            *value = ctx->opt[i];
            return 0;
        }
    return fail(-1, "unknown option '%s'", name);
}

int cvae_reset_options(cvae_ctx* ctx) {
    CVAE_ENTER(ctx);
    for (int i = 0; i < OPT_COUNT; ++i) ctx->opt[i] = g_opt[i].dflt;
    return 0;
}

int cvae_set_draw_parts(cvae_ctx* ctx, int32_t parts) {
    CVAE_ENTER(ctx);
    if (parts < 1) return fail(-1, "bad draw parts %d", (int)parts);
    cx().draw_parts = parts;
    return 0;
}

size_t cvae_net_prepared_bytes(cvae_ctx* ctx, const cvae_net_desc* d) {
    CVAE_ENTER_SZ(ctx);
    Dims m;
    if (make_dims(d, &m)) return 0;
    return (size_t)prep_layout(m, d->has_scale_in != 0, d->has_scale_out != 0).total * sizeof(float);
}

size_t cvae_net_prepare_scratch_bytes(cvae_ctx* ctx, const cvae_net_desc* d) {
    CVAE_ENTER_SZ(ctx);
    Dims m;
    if (make_dims(d, &m)) return 0;
    return ((size_t)m.c2 * m.R * m.C + m.c2 + 64) * sizeof(double);
}

int cvae_net_prepare(cvae_ctx* ctx, const cvae_net_desc* d, const cvae_net_weights* w, void* prepared, size_t prepared_bytes,
                     void* scratch, size_t scratch_bytes, void* stream) {
    CVAE_ENTER(ctx);
    Dims m;
    if (int rc = make_dims(d, &m)) return rc;
    if (!w || !prepared || !scratch) return fail(-1, "null argument");
    if (!w->conv0_w || !w->conv0_b || !w->conv1_w || !w->conv1_b || !w->w_ih || !w->w_hh || !w->b_ih || !w->b_hh ||
        !w->out_w || !w->out_b)
        return fail(-1, "missing weight pointer");
    if (d->has_scale_in && (!w->scale_in_w || !w->scale_in_b)) return fail(-1, "has_scale_in without scale_in weights");
    if (d->has_scale_out && (!w->scale_out_w || !w->scale_out_b)) return fail(-1, "has_scale_out without scale_out weights");
    if (prepared_bytes < cvae_net_prepared_bytes(ctx, d)) return fail(-2, "prepared buffer too small");
    if (scratch_bytes < cvae_net_prepare_scratch_bytes(ctx, d)) return fail(-2, "prepare scratch too small");
    hipStream_t st = (hipStream_t)stream;
    const Prep pl = prep_layout(m, d->has_scale_in != 0, d->has_scale_out != 0);
    float* P = (float*)prepared;
    double* mfull = (double*)scratch;
    double* bprime = mfull + (size_t)m.c2 * m.R * m.C;

    CVAE_HIP_OK(hipMemsetAsync(P, 0, (size_t)pl.total * sizeof(float), st));
    hipLaunchKernelGGL((k_prep_mfull), dim3(nblk((long)m.c2 * m.R * m.C, 256)), dim3(256), 0, st, w->conv0_w, w->conv1_w,
                       mfull, m.C, m.ks);
    hipLaunchKernelGGL((k_prep_bprime), dim3(nblk(m.c2, 128)), dim3(128), 0, st, w->conv0_b, w->conv1_w, w->conv1_b,
                       bprime, m.C, m.ks);
    hipLaunchKernelGGL((k_prep_afold), dim3(nblk((long)m.H3 * m.Kfe, 256)), dim3(256), 0, st, w->w_ih,
                       (const double*)mfull, P + pl.afold, m.C, m.Cp, m.ks, m.tot, m.Kfe, m.H3);
    hipLaunchKernelGGL((k_prep_afold_h), dim3(nblk((long)m.nch * 4 * ((m.KFW + 1) / 2) * 3 * 512, 256)), dim3(256), 0, st,
                       (const float*)(P + pl.afold), P + pl.afold_h, m.H, m.Kfe, (m.KFW + 1) / 2);
    hipLaunchKernelGGL((k_prep_afold3), dim3(nblk((long)m.nch * 4 * m.KFW * 3 * 256, 256)), dim3(256), 0, st,
                       (const float*)(P + pl.afold), P + pl.afold3, m.H, m.Kfe, m.KFW);
    hipLaunchKernelGGL((k_prep_cfold), dim3(nblk(m.H3, 128)), dim3(128), 0, st, w->w_ih, w->b_ih, w->b_hh, w->out_b,
                       (const double*)bprime, P + pl.cfold, m.c2, m.Co, m.tot, m.H);
    hipLaunchKernelGGL((k_prep_wrec), dim3(nblk((long)(m.H / 4) * m.nch * 256, 256)), dim3(256), 0, st, w->w_ih, w->w_hh,
                       w->out_w, P + pl.wrec, m.c2, m.Co, m.tot, m.H);
    auto copy2d = [&](float* dst, long dld, const float* src, long sld, int rows, int cols) {
        hipLaunchKernelGGL((k_copy2d), dim3(nblk((long)rows * cols, 256)), dim3(256), 0, st, dst, dld, src, sld, rows, cols);
    };
    hipLaunchKernelGGL((k_prep_wrec2), dim3(nblk((long)m.nch * 4 * m.nch * 256, 256)), dim3(256), 0, st, w->w_ih, w->w_hh,
                       w->out_w, P + pl.wrec2, m.c2, m.Co, m.tot, m.H);
    if (m.H % 32 == 0)
        hipLaunchKernelGGL((k_prep_wrec_h), dim3(nblk((long)m.nch * 4 * (m.H / 32) * 512, 256)), dim3(256), 0, st,
                           (const float*)(P + pl.wrec2), P + pl.wrec_h, m.H);
    if (exact3_ok(m)) {
        hipLaunchKernelGGL((k_prep_wrec3), dim3(nblk((long)(m.H / 8) * 4 * exact3_kpw(m) * 512, 256)), dim3(256), 0, st,
                           (const float*)(P + pl.wrec2), P + pl.wrec_t, m.H, exact3_kpw(m));
        hipLaunchKernelGGL((k_prep_afold3l), dim3(nblk((long)(m.H / 8) * 4 * m.KFW * 512, 256)), dim3(256), 0, st,
                           (const float*)(P + pl.afold), P + pl.afold_t, m.H, m.Kfe, m.KFW);
        hipLaunchKernelGGL((k_prep_wrec3_l2b), dim3(nblk((long)(m.H / 8) * 4 * exact3_kpw(m) * 512, 256)), dim3(256), 0, st,
                           (const float*)(P + pl.wrec2), (unsigned char*)(P + pl.wrec_l2b), m.H, exact3_kpw(m));
    }
    copy2d(P + pl.bhn, m.H, w->b_hh + 2 * m.H, m.H, 1, m.H);
    hipLaunchKernelGGL((k_copy2d_t), dim3(nblk((long)m.H3 * m.Co, 256)), dim3(256), 0, st, P + pl.wyT, w->w_ih + m.c2,
                       (long)m.tot, m.H3, m.Co);
    copy2d(P + pl.wo, m.H, w->out_w, m.H, m.Co, m.H);
    copy2d(P + pl.bo, m.Co, w->out_b, m.Co, 1, m.Co);
    hipLaunchKernelGGL((k_prep_wo2), dim3(nblk((long)m.Cop * m.H + m.Cop, 256)), dim3(256), 0, st, w->out_w, w->out_b,
                       d->has_scale_out ? w->scale_out_w : (const float*)nullptr,
                       d->has_scale_out ? w->scale_out_b : (const float*)nullptr, P + pl.wo2, P + pl.bo2, m.Co, m.Cop, m.H);
    if (exact3_ok(m))
        hipLaunchKernelGGL((k_prep_wo3), dim3(nblk((long)((m.Cop + 31) / 32) * (m.H / 16) * 512, 256)), dim3(256), 0, st,
                           (const float*)(P + pl.wo2), P + pl.wo3, m.H, m.Cop, (m.Cop + 31) / 32);
    if (d->has_scale_in) {
        copy2d(P + pl.sin_w, m.C, w->scale_in_w, m.C, m.C, m.C);
        copy2d(P + pl.sin_b, m.C, w->scale_in_b, m.C, 1, m.C);
    }
    if (d->has_scale_out) {
        copy2d(P + pl.sout_w, m.Co, w->scale_out_w, m.Co, m.Co, m.Co);
        copy2d(P + pl.sout_b, m.Co, w->scale_out_b, m.Co, 1, m.Co);
    }
    CVAE_HIP_OK(hipGetLastError());
    return 0;
}

size_t cvae_pass_workspace_bytes(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T) {
    CVAE_ENTER_SZ(ctx);
    Dims m;
    if (make_dims(d, &m) || B < 1 || T < 1) return 0;
    return (size_t)work_layout(m, B, T).total * sizeof(float);
}

int cvae_gru_rnn_forward(cvae_ctx* ctx, const cvae_net_desc* d, const void* prepared, const cvae_pass_input* in, const float* y_in,
                         const float* h_in, int B, int T, int clamp_lat_dim, float* trj_out, float* y_last,
                         float* h_last, void* workspace, size_t workspace_bytes, int flags, void* stream) {
    CVAE_ENTER(ctx);
    Dims m;
    if (int rc = make_dims(d, &m)) return rc;
    if (B < 1 || T < 1) return fail(-1, "empty batch: B=%d T=%d", B, T);
    if (!prepared || !in || !y_in || !trj_out || !workspace) return fail(-1, "null argument");
    if (!in->seg0.ptr || (in->seg1.width > 0 && !in->lat && !in->seg1.ptr)) return fail(-1, "null input segment");
    if (workspace_bytes < cvae_pass_workspace_bytes(ctx, d, B, T)) return fail(-2, "workspace too small");
    const Cell cell{in, y_in, h_in, trj_out, y_last, h_last};
    return run_pass(m, d, (const float*)prepared, &cell, 1, B, T, clamp_lat_dim, (float*)workspace, (int*)workspace,
                    flags, (hipStream_t)stream, true);
}

int cvae_gru_rnn_forward_stacked(cvae_ctx* ctx, const cvae_net_desc* d, const void* prepared, int ncell, const cvae_pass_input* in,
                                 const float* const* y_in, int B, int T, int clamp_lat_dim, float* const* trj_out,
                                 void* workspace, size_t workspace_bytes, int flags, void* stream) {
    CVAE_ENTER(ctx);
    Dims m;
    if (int rc = make_dims(d, &m)) return rc;
    if (B < 1 || T < 1 || ncell < 1 || ncell > CVAE_MAX_CELLS) return fail(-1, "bad sizes: ncell=%d B=%d T=%d", ncell, B, T);
    if (!prepared || !in || !y_in || !trj_out || !workspace) return fail(-1, "null argument");
    if (workspace_bytes < cvae_pass_workspace_bytes(ctx, d, ncell * B, T)) return fail(-2, "workspace too small");
    Cell cells[CVAE_MAX_CELLS];
    for (int c = 0; c < ncell; ++c) {
        if (!in[c].seg0.ptr || (in[c].seg1.width > 0 && !in[c].lat && !in[c].seg1.ptr) || !y_in[c] || !trj_out[c])
            return fail(-1, "cell %d: null pointer", c);
        cells[c] = Cell{&in[c], y_in[c], nullptr, trj_out[c], nullptr, nullptr};
    }
    return run_pass(m, d, (const float*)prepared, cells, ncell, B, T, clamp_lat_dim, (float*)workspace, (int*)workspace, flags,
                    (hipStream_t)stream, true);
}

int cvae_gru_rnn_forward_stacked_carry(cvae_ctx* ctx, const cvae_net_desc* d, const void* prepared, int ncell, const cvae_pass_input* in,
                                       const float* const* y_in, const float* const* h_in, int B, int T, int clamp_lat_dim,
                                       float* const* trj_out, float* const* h_last, void* workspace, size_t workspace_bytes,
                                       int flags, void* stream) {
    CVAE_ENTER(ctx);
    Dims m;
    if (int rc = make_dims(d, &m)) return rc;
    if (B < 1 || T < 1 || ncell < 1 || ncell > CVAE_MAX_CELLS) return fail(-1, "bad sizes: ncell=%d B=%d T=%d", ncell, B, T);
    if (!prepared || !in || !y_in || !trj_out || !workspace) return fail(-1, "null argument");
    if (workspace_bytes < cvae_pass_workspace_bytes(ctx, d, ncell * B, T)) return fail(-2, "workspace too small");
    Cell cells[CVAE_MAX_CELLS];
    for (int c = 0; c < ncell; ++c) {
        const float* hi = h_in ? h_in[c] : nullptr;
        if (!in[c].seg0.ptr || (in[c].seg1.width > 0 && !in[c].lat && !in[c].seg1.ptr) || !trj_out[c])
            return fail(-1, "cell %d: null pointer", c);
        if (!y_in[c] && !hi) return fail(-1, "cell %d: no y_in and no state to continue from", c);
        cells[c] = Cell{&in[c], y_in[c], hi, trj_out[c], nullptr, h_last ? h_last[c] : nullptr};
    }
    return run_pass(m, d, (const float*)prepared, cells, ncell, B, T, clamp_lat_dim, (float*)workspace, (int*)workspace, flags,
                    (hipStream_t)stream, true);
}

int cvae_sample(cvae_ctx* ctx, const float* lat, int rows, int lat_dim, const float* eps, uint64_t seed, uint64_t draw_id, float* z,
                float* eps_out, void* stream) {
    CVAE_ENTER(ctx);
    if (!lat || !z || rows < 0 || lat_dim < 1) return fail(-1, "bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL((k_sample), dim3(nblk((long)rows * lat_dim, 256)), dim3(256), 0, (hipStream_t)stream, lat, rows,
                       lat_dim, eps, seed, draw_id, z, eps_out, (uint64_t)cx().draw_row0 * (uint64_t)cx().draw_frames);
    CVAE_HIP_OK(hipGetLastError());
    return 0;
}

int cvae_sample_laplace(cvae_ctx* ctx, const float* lat, int rows, int lat_dim, const float* eps, uint64_t seed, uint64_t draw_id,
                        float* z, float* eps_out, void* stream) {
    CVAE_ENTER(ctx);
    if (!lat || !z || rows < 0 || lat_dim < 1) return fail(-1, "bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL((k_sample_laplace), dim3(nblk((long)rows * lat_dim, 256)), dim3(256), 0, (hipStream_t)stream, lat, rows,
                       lat_dim, eps, seed, draw_id, z, eps_out, (uint64_t)cx().draw_row0 * (uint64_t)cx().draw_frames);
    CVAE_HIP_OK(hipGetLastError());
    return 0;
}

int cvae_sample_laplace_backward(cvae_ctx* ctx, const float* dz, const float* lat, const float* z, int rows, int lat_dim, float* dlat,
                                 void* stream) {
    CVAE_ENTER(ctx);
    if (!dz || !lat || !z || !dlat || rows < 0 || lat_dim < 1) return fail(-1, "bad argument");
    if (rows == 0) return 0;
    hipLaunchKernelGGL((k_sample_laplace_bwd), dim3(nblk((long)rows * lat_dim, 256)), dim3(256), 0, (hipStream_t)stream, dz, lat, z,
                       rows, lat_dim, dlat);
    CVAE_HIP_OK(hipGetLastError());
    return 0;
}

// cycle workspace = [status 64 floats][per-pass workspace (max of enc/dec)][5 trajectories for the cycle in flight]
static long cycle_layout(const Dims& me, const Dims& md, int B, int T, long* pass_off, long* traj_off) {
    const long pe = work_layout(me, B, T).total, pd = work_layout(md, 2 * B, T).total;
    long o = 64;
    *pass_off = o;
    o += pe > pd ? pe : pd;
    *traj_off = o;
    o += up((long)B * T * me.Co, 64) * 2 + up((long)B * T * md.Co, 64) * 3;
    return o;
}

size_t cvae_cycle_workspace_bytes(cvae_ctx* ctx, const cvae_net_desc* enc, const cvae_net_desc* dec, int B, int T, int n_cyc) {
    CVAE_ENTER_SZ(ctx);
    Dims me, md;
    if (make_dims(enc, &me) || make_dims(dec, &md) || B < 1 || T < 1 || n_cyc < 1) return 0;
    long po, to;
    return (size_t)cycle_layout(me, md, B, T, &po, &to) * sizeof(float);
}

static int cycle_forward_impl(const cvae_net_desc* enc, const void* enc_prepared, const cvae_net_desc* dec,
                       const void* dec_prepared, const float* x, const float* cvx, int stdim, const float* code_src,
                       const float* code_trg, int ncode, const float* y_in_enc, const float* y_in_dec, int B, int T,
                       int n_cyc, int lat_dim, const float* eps, uint64_t seed, float* out_lat, float* out_rec,
                       float* out_cv, float* out_latcv, float* out_reccyc, void* workspace, size_t workspace_bytes,
                       int flags, void* stream, const cvae_cycle_state* sin, const cvae_cycle_state* sout) {
    Dims me, md;
    if (int rc = make_dims(enc, &me)) return rc;
    if (int rc = make_dims(dec, &md)) return rc;
    if (B < 1 || T < 1 || n_cyc < 1) return fail(-1, "empty work: B=%d T=%d n_cyc=%d", B, T, n_cyc);
    if (!enc_prepared || !dec_prepared || !x || !cvx || !code_src || !code_trg || !y_in_enc || !y_in_dec || !workspace)
        return fail(-1, "null argument");
    if (me.Co != 2 * lat_dim) return fail(-1, "encoder out_dim %d != 2*lat_dim %d", me.Co, 2 * lat_dim);
    if (md.C != ncode + lat_dim) return fail(-1, "decoder in_dim %d != ncode+lat_dim %d", md.C, ncode + lat_dim);
    if (me.C != stdim + md.Co) return fail(-1, "encoder in_dim %d != stdim+decoder out_dim %d", me.C, stdim + md.Co);
    if (workspace_bytes < cvae_cycle_workspace_bytes(tl_ctx, enc, dec, B, T, n_cyc)) return fail(-2, "workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    long po, to;
    cycle_layout(me, md, B, T, &po, &to);
    float* pws = ws + po;
    const long ne = (long)B * T * me.Co, nd = (long)B * T * md.Co;
    float* t_lat = ws + to;
    float* t_latcv = t_lat + up(ne, 64);
    float* t_rec = t_latcv + up(ne, 64);
    float* t_cv = t_rec + up(nd, 64);
    float* t_reccyc = t_cv + up(nd, 64);
    const float* prev_reccyc = nullptr;
    int* status = (int*)workspace;
    const long neps = (long)B * T * lat_dim;
    if (sin && (!sin->y_enc || !sin->y_dec || !sin->h_enc || !sin->h_dec)) return fail(-1, "incomplete input cycle state");
    if (sout && (!sout->y_enc || !sout->y_dec || !sout->h_enc || !sout->h_dec)) return fail(-1, "incomplete output cycle state");
    // state of pass `slot` (encoder: 0 lat, 1 latcv; decoder: 0 rec, 1 cv, 2 reccyc) of cycle i
    auto ye = [&](const cvae_cycle_state* s, int i, int slot) { return s ? s->y_enc + ((long)i * 2 + slot) * B * me.Co : nullptr; };
    auto he = [&](const cvae_cycle_state* s, int i, int slot) { return s ? s->h_enc + ((long)i * 2 + slot) * B * me.H : nullptr; };
    auto yd = [&](const cvae_cycle_state* s, int i, int slot) { return s ? s->y_dec + ((long)i * 3 + slot) * B * md.Co : nullptr; };
    auto hd = [&](const cvae_cycle_state* s, int i, int slot) { return s ? s->h_dec + ((long)i * 3 + slot) * B * md.H : nullptr; };

    for (int i = 0; i < n_cyc; ++i) {
        float* lat = out_lat ? out_lat + i * ne : t_lat;
        float* latcv = out_latcv ? out_latcv + i * ne : t_latcv;
        float* rec = out_rec ? out_rec + i * nd : t_rec;
        float* cv = out_cv ? out_cv + i * nd : t_cv;
        // rec_cyc of cycle i feeds cycle i+1's encoder, which has consumed it before this cycle's last pass rewrites it
        float* reccyc = out_reccyc ? out_reccyc + i * nd : t_reccyc;
        cvae_pass_input in, in2;
        memset(&in, 0, sizeof(in));
        int rc;
        // lat = E(x) or E([x[:,:,:stdim] ; rec_cyc_{i-1}])      (train...:1334 / :1328)
        if (i == 0) {
            in.seg0 = cvae_seg{x, me.C, me.C};
        } else {
            in.seg0 = cvae_seg{x, stdim, me.C};
            in.seg1 = cvae_seg{prev_reccyc, md.Co, md.Co};
        }
        {
            const Cell c{&in, sin ? ye(sin, i, 0) : y_in_enc, he(sin, i, 0), lat, ye(sout, i, 0), he(sout, i, 0)};
            if ((rc = run_pass(me, enc, (const float*)enc_prepared, &c, 1, B, T, lat_dim, pws, status, flags, st, i == 0))) return rc;
        }
        // rec = D([code_src ; z1]) and cv = D([code_trg ; z2]) share the decoder and do not depend on each other
        // (train...:1335-1336): one stacked pass, 2B rows
        memset(&in, 0, sizeof(in));
        in.seg0 = cvae_seg{code_src, ncode, ncode};
        in.lat = lat; in.lat_dim = lat_dim;
        in.eps = eps ? eps + (i * 3 + 0) * neps : nullptr;
        in.seed = seed; in.draw_id = (uint64_t)(i * 3 + 0);
        in2 = in;
        in2.seg0 = cvae_seg{code_trg, ncode, ncode};
        in2.eps = eps ? eps + (i * 3 + 1) * neps : nullptr;
        in2.draw_id = (uint64_t)(i * 3 + 1);
        {
            const Cell c2[2] = {{&in, sin ? yd(sin, i, 0) : y_in_dec, hd(sin, i, 0), rec, yd(sout, i, 0), hd(sout, i, 0)},
                                {&in2, sin ? yd(sin, i, 1) : y_in_dec, hd(sin, i, 1), cv, yd(sout, i, 1), hd(sout, i, 1)}};
            if ((rc = run_pass(md, dec, (const float*)dec_prepared, c2, 2, B, T, -1, pws, status, flags, st))) return rc;
        }
        // latcv = E([cvx ; cv])                                   (train...:1337)
        memset(&in, 0, sizeof(in));
        in.seg0 = cvae_seg{cvx, stdim, stdim};
        in.seg1 = cvae_seg{cv, md.Co, md.Co};
        {
            const Cell c{&in, sin ? ye(sin, i, 1) : y_in_enc, he(sin, i, 1), latcv, ye(sout, i, 1), he(sout, i, 1)};
            if ((rc = run_pass(me, enc, (const float*)enc_prepared, &c, 1, B, T, lat_dim, pws, status, flags, st))) return rc;
        }
        // rec_cyc = D([code_src ; z3])                            (train...:1338)
        memset(&in, 0, sizeof(in));
        in.seg0 = cvae_seg{code_src, ncode, ncode};
        in.lat = latcv; in.lat_dim = lat_dim;
        in.eps = eps ? eps + (i * 3 + 2) * neps : nullptr;
        in.seed = seed; in.draw_id = (uint64_t)(i * 3 + 2);
        {
            const Cell c{&in, sin ? yd(sin, i, 2) : y_in_dec, hd(sin, i, 2), reccyc, yd(sout, i, 2), hd(sout, i, 2)};
            if ((rc = run_pass(md, dec, (const float*)dec_prepared, &c, 1, B, T, -1, pws, status, flags, st))) return rc;
        }
        prev_reccyc = reccyc;
    }
    return 0;
}

int cvae_cycle_forward(cvae_ctx* ctx, const cvae_net_desc* enc, const void* enc_prepared, const cvae_net_desc* dec,
                       const void* dec_prepared, const float* x, const float* cvx, int stdim, const float* code_src,
                       const float* code_trg, int ncode, const float* y_in_enc, const float* y_in_dec, int B, int T,
                       int n_cyc, int lat_dim, const float* eps, uint64_t seed, float* out_lat, float* out_rec,
                       float* out_cv, float* out_latcv, float* out_reccyc, void* workspace, size_t workspace_bytes,
                       int flags, void* stream) {
    CVAE_ENTER(ctx);
    return cycle_forward_impl(enc, enc_prepared, dec, dec_prepared, x, cvx, stdim, code_src, code_trg, ncode, y_in_enc, y_in_dec,
                              B, T, n_cyc, lat_dim, eps, seed, out_lat, out_rec, out_cv, out_latcv, out_reccyc, workspace,
                              workspace_bytes, flags, stream, nullptr, nullptr);
}

int cvae_cycle_forward_carry(cvae_ctx* ctx, const cvae_net_desc* enc, const void* enc_prepared, const cvae_net_desc* dec,
                             const void* dec_prepared, const float* x, const float* cvx, int stdim, const float* code_src,
                             const float* code_trg, int ncode, const float* y_in_enc, const float* y_in_dec, int B, int T,
                             int n_cyc, int lat_dim, const float* eps, uint64_t seed, float* out_lat, float* out_rec,
                             float* out_cv, float* out_latcv, float* out_reccyc, void* workspace, size_t workspace_bytes,
                             int flags, void* stream, const cvae_cycle_state* state_in, const cvae_cycle_state* state_out) {
    CVAE_ENTER(ctx);
    return cycle_forward_impl(enc, enc_prepared, dec, dec_prepared, x, cvx, stdim, code_src, code_trg, ncode, y_in_enc, y_in_dec,
                              B, T, n_cyc, lat_dim, eps, seed, out_lat, out_rec, out_cv, out_latcv, out_reccyc, workspace,
                              workspace_bytes, flags, stream, state_in, state_out);
}

int cvae_step_timing(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T, const void* workspace, double out[8], void* stream) {
    CVAE_ENTER(ctx);
    Dims m;
    if (int rc = make_dims(d, &m)) return rc;
    if (!workspace || !out || B < 1 || T < 1) return fail(-1, "bad argument");
    const Work wl = work_layout(m, B, T);
    const int nrt = wl.Bp / 16;
    const int nwg = m.nch * (nrt < 4 ? nrt : 4) > 256 ? 256 : m.nch * (nrt < 4 ? nrt : 4);
    std::vector<long long> h((size_t)nwg * 4);
    CVAE_HIP_OK(hipMemcpyAsync(h.data(), (const float*)workspace + wl.prof, h.size() * sizeof(long long),
                               hipMemcpyDeviceToHost, (hipStream_t)stream));
    CVAE_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    for (int q = 0; q < 4; ++q) {
        double sum = 0, mx = 0;
        for (int g = 0; g < nwg; ++g) {
            const double v = (double)h[(size_t)g * 4 + q];
            sum += v;
            mx = v > mx ? v : mx;
        }
        out[q] = sum / nwg;
        out[4 + q] = mx;
    }
    return 0;
}

int cvae_profile_collect_launches(cvae_ctx* ctx, double* ms, int* rows, int* cin, int cap) {
    CVAE_ENTER(ctx);
    const int n = (int)cx().prof.used < cap ? (int)cx().prof.used : cap;
    for (int i = 0; i < n; ++i) {
        float t = 0.f;
        CVAE_HIP_OK(hipEventSynchronize(cx().prof.stop[i]));
        CVAE_HIP_OK(hipEventElapsedTime(&t, cx().prof.start[i], cx().prof.stop[i]));
        if (ms) ms[i] = t;
        if (rows) rows[i] = cx().prof.rows[i];
        if (cin) cin[i] = cx().prof.cin[i];
    }
    cx().prof.used = 0;
    return n;
}

int cvae_profile_collect(cvae_ctx* ctx, double* total_ms, int* launches) {
    CVAE_ENTER(ctx);
    double tot = 0.0;
    for (size_t i = 0; i < cx().prof.used; ++i) {
        float ms = 0.f;
        CVAE_HIP_OK(hipEventSynchronize(cx().prof.stop[i]));
        CVAE_HIP_OK(hipEventElapsedTime(&ms, cx().prof.start[i], cx().prof.stop[i]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = (int)cx().prof.used;
    cx().prof.used = 0;
    return 0;
}

int cvae_train_profile_collect(cvae_ctx* ctx, double total_ms[4], int launches[4], double flop[4]) {
    CVAE_ENTER(ctx);
    for (int c = 0; c < TPROF_CLASSES; ++c) {
        if (total_ms) total_ms[c] = 0.0;
        if (launches) launches[c] = 0;
        if (flop) flop[c] = 0.0;
    }
    for (size_t i = 0; i < cx().tprof.used; ++i) {
        float ms = 0.f;
        CVAE_HIP_OK(hipEventSynchronize(cx().tprof.stop[i]));
        CVAE_HIP_OK(hipEventElapsedTime(&ms, cx().tprof.start[i], cx().tprof.stop[i]));
        const int c = cx().tprof.cls[i];
        if (total_ms) total_ms[c] += ms;
        if (launches) launches[c] += 1;
        if (flop) flop[c] += cx().tprof.flop[i];
    }
    cx().tprof.used = 0;
    return 0;
}

int cvae_workspace_status(cvae_ctx* ctx, const void* workspace, int32_t status_out[4], void* stream) {
    CVAE_ENTER(ctx);
    if (!workspace || !status_out) return fail(-1, "null argument");
    CVAE_HIP_OK(hipMemcpyAsync(status_out, workspace, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CVAE_HIP_OK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

}  // extern "C"

#include "cvae_train.inc"
#include "cvae_stage4.inc"
#include "cvae_stage6.inc"
