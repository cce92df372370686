/*
 * cyclevae_hip.h -- C ABI of libcyclevae_hip.so: the CycleVAE-VC encoder -> latent -> decoder hot path
 * as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (patrickltobing/cyclevae-vc) has no FFI: its boundary for this path is the Python module
 * src/nets/gru_vae.py, found through PYTHONPATH (egs/one-to-one/path.sh:11).  The drop-in module
 * cyclevae-vc_amd/gru_vae.py keeps that module's names and binds the entry points below with ctypes
 * (see INTEGRATION.md).  Each entry point states the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous float32 unless stated; shapes in comments
 *   - the caller owns all memory, including the `prepared` weight image and the `workspace`;
 *     the library allocates nothing and keeps no global state besides a thread-local error string and the process-wide
 *     settings of cvae_set_status_sink / cvae_set_draw_origin / cvae_set_draw_parts / cvae_set_side_stream / cvae_set_option;
 *     it reads NO environment variable
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises the device
 *   - return value: 0 = ok, negative = error (cvae_last_error_string() describes it); never throws
 *   - hidden size must be a multiple of 16; kernel_size odd; conv layers (reference `dilation_size`) == 2
 */
#ifndef CYCLEVAE_HIP_H
#define CYCLEVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVAE_ABI_VERSION 6

/* Shape of one reference GRU_RNN (src/nets/gru_vae.py:282-320). */
typedef struct cvae_net_desc {
    int32_t in_dim;        /* Cin                                   */
    int32_t out_dim;       /* Cout (encoder: 2*lat_dim)             */
    int32_t hidden;        /* H, multiple of 16                     */
    int32_t kernel_size;   /* 3 in the recipe                       */
    int32_t layers;        /* reference `dilation_size`; must be 2  */
    int32_t has_scale_in;  /* scale_in_flag  (gru_vae.py:296-297)   */
    int32_t has_scale_out; /* scale_out_flag (gru_vae.py:317-318)   */
} cvae_net_desc;

/* Raw weights in the reference's state_dict layout; NULL where the layer is absent. */
typedef struct cvae_net_weights {
    const float* scale_in_w;  /* scale_in.weight      [Cin,Cin,1]        */
    const float* scale_in_b;  /* scale_in.bias        [Cin]              */
    const float* conv0_w;     /* conv.conv.0.weight   [ks*Cin,Cin,ks]    */
    const float* conv0_b;     /* conv.conv.0.bias     [ks*Cin]           */
    const float* conv1_w;     /* conv.conv.1.weight   [ks^2*Cin,ks*Cin,ks] */
    const float* conv1_b;     /* conv.conv.1.bias     [ks^2*Cin]         */
    const float* w_ih;        /* gru.weight_ih_l0     [3H, ks^2*Cin+Cout] */
    const float* w_hh;        /* gru.weight_hh_l0     [3H, H]            */
    const float* b_ih;        /* gru.bias_ih_l0       [3H]               */
    const float* b_hh;        /* gru.bias_hh_l0       [3H]               */
    const float* out_w;       /* out_1.weight         [Cout,H,1]         */
    const float* out_b;       /* out_1.bias           [Cout]             */
    const float* scale_out_w; /* scale_out.weight     [Cout,Cout,1]      */
    const float* scale_out_b; /* scale_out.bias       [Cout]             */
} cvae_net_weights;

/* One row-segment of a pass input: element (b,t,c) lives at ptr[(b*T+t)*row_stride + c], c < width. */
typedef struct cvae_seg {
    const float* ptr;
    int32_t width;
    int32_t row_stride;
} cvae_seg;

/*
 * Input of one GRU_RNN pass = [seg0 ; seg1] along the feature axis (the torch.cat calls at
 * train_gru_cyclevae_gauss_batch.py:1328-1338).  If `lat` is non-NULL, seg1 is ignored and replaced by the
 * reparameterised draw z = mu + exp(log_var/2)*eps of lat[B,T,2*lat_dim] (sampling_vae_batch,
 * gru_vae.py:85-98): eps is read from `eps`[B,T,lat_dim] if non-NULL, else drawn on device with
 * Philox4x32-10 keyed by (seed, draw_id, frame b*T+t, dim / 4): one block yields the four N(0,1) values of a dim quad (Box-Muller,
 * cosine and sine branch of word pairs (0,1) and (2,3)).
 */
typedef struct cvae_pass_input {
    cvae_seg seg0, seg1;
    const float* lat;
    int32_t lat_dim;
    const float* eps;
    uint64_t seed;
    uint64_t draw_id;
    int32_t frames;  /* valid frames of this input (0 = all T): frames beyond are zero AFTER normalisation, exactly what the conv
                        padding of a shorter utterance run alone would see; lets utterances of different length share a pass */
    int32_t n_draws; /* > 1 with `lat`: z uses the MEAN of n_draws eps (draw ids draw_id .. draw_id + n_draws - 1, or
                        eps[n_draws][B][T][lat_dim]): the n_smpl_dec latent mean of decode_gru-cyclevae_gauss.py:304-305 */
    /* ABI 5 -- a pass over a WINDOW of a longer utterance (single-row cells, B = 1): the pointers address the window's first
       frame, and ctx_before / ctx_after frames of the same arrays in front of it / behind its last valid frame are real data that
       the dilated conv front-end sees instead of zero padding (the reference convolves an utterance in one piece,
       gru_vae.py:353-357; its +-4-frame reach must not notice where a window was cut).  draw_frame0: index of the window's frame 0
       in the Philox frame numbering of the whole utterance.  eps_draw_stride: floats between two draws in `eps` (0: B*T*lat_dim).
       All zero: the window is the utterance (what every earlier ABI did). */
    int32_t ctx_before, ctx_after;
    int64_t draw_frame0;
    int64_t eps_draw_stride;
} cvae_pass_input;

const char* cvae_last_error_string(void);
int cvae_abi_version(void);

/*
 * ABI 6 -- the handle.  The library keeps NO process-wide mutable state: everything that configures it lives in a context the
 * caller creates, and EVERY entry point below takes that context as its first argument ("no global state besides
 * a per-handle descriptor; one handle per (device, stream); not thread-safe per handle, independent across handles").  A context
 * holds: the status sink (cvae_set_status_sink), this rank's draw origin / parts (cvae_set_draw_origin, cvae_set_draw_parts), the
 * named options (cvae_set_option), the side stream and its join events (cvae_set_side_stream), the profiling brackets
 * (cvae_profile_collect*, cvae_train_profile_collect) and the record of which MFMA-order weight images its train images hold
 * (cvae_net_prepare_train_v: a train image must be used through the context that prepared it).  Two modules on two devices, or
 * two threads, use two contexts and never see each other's settings.  The only thread-local datum is the error string behind
 * cvae_last_error_string.  cvae_ctx_create returns NULL when out of memory; cvae_ctx_destroy(NULL) is a no-op; destroy a context
 * only after the streams it enqueued on have been synchronised (it owns HIP events).  A NULL context fails with -1 (size queries: 0).
 * (The reference has nothing to mirror here: it has no FFI; its per-process state is Python module state.)
 */
typedef struct cvae_ctx cvae_ctx;
cvae_ctx* cvae_ctx_create(void);
int cvae_ctx_destroy(cvae_ctx* ctx);

/*
 * Settings of a context (ABI <= 5 kept them process-wide).
 *
 * cvae_set_status_sink: `sink` = int32[4] the DEVICE can write and the HOST can read without a copy (pinned host memory), or
 * NULL.  When set, the persistent kernels report a timed-out hand-off spin there (sink[0] != 0) instead of in the workspace's
 * status words, so the host-side module can notice it before using any result, without a device synchronisation.  The word is
 * sticky: the caller clears it.  (The reference has nothing to replace here: its GRU loop is a Python loop, gru_vae.py:391-394.)
 *
 * cvae_set_draw_origin: place of this process in a data-parallel job: row0 = global index of its first batch
 * row, global_rows = batch rows of the whole job (0 = this process alone), frames_per_row = T of the windows fed to
 * cvae_sample (whose `rows` are flattened frames; 0 = no offset there).  The on-device Philox streams (latent draws, dropout
 * masks) are keyed by GLOBAL row, so a row sees the same eps / masks on whichever rank it lands and results do not depend on
 * the number of ranks.  Defaults 0, 0, 0 reproduce the single-process numbering.
 *
 * cvae_set_draw_parts: the batch of the following train-mode passes is `parts` stacked copies of this process's rows (the two
 * decoder passes rec and cv of train...:1335-1336 run as ONE launch of 2B rows, stage4.chain_loss(stack_rec_cv=True)): copy c
 * of local row b is keyed as row c*global_rows + row0 + b of a parts*global_rows-row job, so the dropout masks stay
 * independent of the number of ranks.  Default 1; ignored when the batch is not a multiple of parts.
 */
int cvae_set_status_sink(cvae_ctx* ctx, int32_t* sink);
/*
 * cvae_status_latch (ABI 4): ONE stream-ordered launch that moves the status word from the sink into a DEVICE word the caller
 * owns: latch[0] = max(latch[0], sink[0]); sink[0] = 0.  A training loop that does not synchronise every step calls it at the end
 * of each step and gates the update on `latch` (cvae_adam_step): the latch stays raised -- every later step's update is skipped
 * -- until the caller, having SEEN the code through a stream-ordered copy, clears it with a stream-ordered memset.  The host
 * never writes the sink while steps are in flight (stage4.Stage4Step).  No sink set: no-op.
 */
int cvae_status_latch(cvae_ctx* ctx, int32_t* latch, void* stream);
int cvae_set_draw_origin(cvae_ctx* ctx, int64_t row0, int64_t global_rows, int64_t frames_per_row);
int cvae_set_draw_parts(cvae_ctx* ctx, int32_t parts);

/*
 * Tuning / diagnostic switches of a context, by name (every configuration the library has besides the `flags` arguments; the
 * reference has no counterpart).  Unknown names fail.  cvae_reset_options restores the defaults.
 *   name                default  meaning
 *   "max_rt"            0        > 0: cap on the row tiles a grid handles concurrently (tests: several row tiles per block)
 *   "no_ll"             0        1: passes of <= 3 rows run the dataflow kernel instead of the word-exchange kernel k_gru_steps_ll
 *   "ll_backoff"        -1       >= 0: s_sleep units before the first poll of a step in k_gru_steps_ll (-1: swept default)
 *   "v6_limbs_h64"      3        2: the two-limb form of k_gru_steps_v6 (what H = 2048 runs) at H = 64, for the emulator tests
 *   "old_outproj"       0        1: projection of an exact-operand pass from the fp32 state copy instead of the limb triples
 *   "exp"               0        measurement switches of the dataflow kernels (bit layout: Step6Params::exp)
 *   "train_kernel"      0        training recurrences: 0 exact fp32 operands (three fp16 limbs), 1 fp16 pairs (22 bits), 2 fp32-input MFMA
 *   "bwd_overflow_at"   60000    |gate gradient * 2^8| from which the persistent reverse recurrences raise status 5 (tests lower it)
 *   "x3_tile"           0        exact-operand forward training recurrence: 16 / 32 force that row-tile geometry (0: by tiles per block)
 *   "train_per_step"    0        1: forward training recurrence as T launches
 *   "train_bwd_per_step" 0       1: reverse training recurrence as 2T launches (fp32 products; the fallback of a range overflow)
 *   "train_fp32_mfma"   0        1: same as train_kernel = 2 for the forward recurrence (kept for the tests)
 *   "train_backoff"     32       s_sleep units before the first poll of a step (pair-form forward training recurrence)
 *   "train_prof"        0        1: phase cycle sums of block 0 of the training recurrences (cvae_train_debug_counters)
 *   "train_old_gemm"    0        1: the simple GEMM kernels (the unaligned-operand fallbacks) everywhere
 *   "gemm_max_split"    16       cap on the contraction split the tile picker may choose for a training GEMM (1: never split)
 *   "bwd_ks"            8        K slices of the per-step reverse product (the any-H path, e.g. H = 2048); 1..32
 *   "bwd_wide"          0        1: four column tiles per block in that product (measured at hu2048: no gain)
 *   "v6_limbs_h2048"    3        2: k_gru_steps_v6 at H = 2048 on fp16 PAIRS (22-23 bit operands, 1.5x faster) instead of exact triples
 *                                with the third weight limb streamed from L2
 *   "v6_w2s_h64"        0        1: that streamed form at H = 64, for the emulator tests
 *   "step_col_tiles"    0        per-step forward training kernel (any-H path): 16-column tiles per block, 0 = pick, 1 / 2 = force
 *   "t0_in_kernel"      0        1: k_gru_steps_v6 forms the frame-0 feedback correction itself instead of reading the prologue's
 *   "coop_launch"       0        1: the all-resident recurrent kernels are launched with hipLaunchCooperativeKernel (residency
 *                                checked by the runtime at every launch, ~27 us of idle GPU around each one on MI355X);
 *                                0: residency checked once per kernel through the occupancy query, then plain launches
 *   "gemm_force"        0        measurement: TM*10000 + TN*100 + ks forces tile and contraction split of every training GEMM
 *   "gemm_log"          0        measurement: every training GEMM bracketed by HIP events and printed to stderr (synchronises)
 *   "gemm_trace"        0        1: print when a GEMM takes a fallback kernel
 *   "train_xmap"        0        bit 0 / bit 1: XCD-aware block placement in the exact-operand forward / reverse training recurrence
 *   "ll_wide_rows"      0        1: training passes of <= 3 rows pad their time-major buffers to 32 rows per frame (round 3) instead of B
 *   "ll_row_pad"        0        rows per frame of the time-major buffers of a training pass of <= 3 rows (the word-exchange kernels address
 *                                rows by stride only): 0 = exactly B, so that every GEMM of a one-utterance pass runs over T rows; 4: round 4
 *   "gemm_min_depth"    128      training GEMMs: a split contraction keeps at least this many k per slice (256 until round 5)
 *   "gemm_nt_fit"       1        tile picker: constants fitted to the round-5 sweep (tools/gemm_sweep.sh) for the launch stream's GEMMs; 0: the shared ones
 *   "gemm_occ_model"    1        tile picker of the training GEMMs counts the workgroups a CU really holds (registers of each tile's
 *                                kernel); 0: at most four per CU whatever the tile (round 4)
 *   "train_bwd_backoff" 0        x 64 cycles before the first flag poll of a task of the exact reverse training recurrence (measured: no gain)
 *   "train_bp16"        1        training passes of 4..16 rows are padded to one 16-row tile; 0: to 32 rows (two 16-row tiles, one dead: round 3)
 *   "bwd_split_launch"  1        exact reverse recurrence: passes with more than two row tiles per block run as one launch per two tiles
 *                                per block (rows are independent); 0: one launch per pass (round 3)
 *   "train_profile"     0        1: HIP events on the launch stream around the training recurrences and GEMMs (cvae_train_profile_collect)
 *   "v6_backoff"        -1       >= 0: units of 64 cycles a block of k_gru_steps_v6 with one row tile sleeps before the first flag poll
 *                                of a step (-1: swept per front-end width: 8 for the decoder's KFW = 6, 2 for the encoder's 8)
 *   "masks_on_side"     1        train-mode forward with a side stream set: the dropout mask of the recurrence's feedback operand is
 *                                drawn on the side stream, beside the prologue and the front-end GEMMs (0: on the launch stream)
 *   "wgrad_order"       -1       where the side-stream weight-gradient GEMMs of a backward pass start: 0 all right behind its reverse
 *                                recurrence (beside the data-gradient chain), 1 all behind that chain (under the NEXT pass's recurrence),
 *                                2 the light ones at once and the two big contractions behind the chain; -1: 2 for passes of >= 64
 *                                rows, else 0 (profiles/r05_notes_training.md)
 *   "side_tile_cap"     2        > 0 caps the tiles of side-stream GEMMs at 32*cap x 32*cap (small tiles fit on a CU
 *                                beside a block of the reverse recurrence); 0: no cap
 */
int cvae_set_option(cvae_ctx* ctx, const char* name, int64_t value);
int cvae_get_option(cvae_ctx* ctx, const char* name, int64_t* value);
int cvae_reset_options(cvae_ctx* ctx);

/*
 * Self-test of the operand transport of the exact-operand kernels (no reference counterpart: the reference multiplies fp32
 * values directly): y[i] = l0 + l1/2^11 + l2/2^22 where (l0, l1, l2) are the two halves and the bf8 byte a producer publishes for
 * x[i] and l2 has gone through the consumer's packed decode.  n a multiple of 8.  y == x bit for bit for |x| >= 2^-16 (and 0),
 * |y - x| <= 2^-40 below; tests also compare the device result bit for bit with the host build of the same code.
 */
int cvae_selftest_limbs(cvae_ctx* ctx, const float* x, float* y, int64_t n, void* stream);
/* Test aid (ABI 4): `blocks` workgroups of 256 threads that each hold `lds_bytes` of LDS and stay resident for `cycles` shader
 * cycles on `stream`: CU-side contention for the all-resident recurrent kernels, which are launched plainly after a one-time
 * occupancy check (tests/test_gpu_parity.py::test_hand_off_under_cu_contention). */
int cvae_selftest_occupy(cvae_ctx* ctx, int blocks, size_t lds_bytes, int64_t cycles, void* stream);

/* Bytes of the caller-owned prepared-weights image / prepare-time scratch for a net. */
size_t cvae_net_prepared_bytes(cvae_ctx* ctx, const cvae_net_desc* d);
size_t cvae_net_prepare_scratch_bytes(cvae_ctx* ctx, const cvae_net_desc* d);

/*
 * Build the device weight image used by the forward kernels (call again whenever the weights change):
 * folds scale-free conv0*conv1*W_ih[:, :ks^2*Cin] into one ks^2-tap matrix, folds the autoregressive
 * feedback W_ih[:, ks^2*Cin:] * out_1 into the recurrent matrix, and lays both out for MFMA fragment loads.
 * Replaces nothing the reference does at run time; it is the load-time half of GRU_RNN.forward
 * (gru_vae.py:353-357 convs, :365/:392 gate matmuls, :371/:393 projection).
 */
int cvae_net_prepare(cvae_ctx* ctx, const cvae_net_desc* d, const cvae_net_weights* w, void* prepared, size_t prepared_bytes,
                     void* scratch, size_t scratch_bytes, void* stream);

/* Bytes of workspace one pass of (B,T) needs. */
size_t cvae_pass_workspace_bytes(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T);

#define CVAE_FLAG_PERSISTENT 1 /* run the T recurrent steps as ONE launch of an all-resident grid (blocks hand over through flags) */
#define CVAE_FLAG_HOISTED_FRONTEND 32 /* with PERSISTENT: keep the front-end as a separate GEMM launch + gx buffer (tests, A/B) */
#define CVAE_FLAG_SPLIT_F16 256 /* with PERSISTENT: matrix products of the recurrent kernel as three fp16 MFMAs on (hi, lo) pairs,
                                   x = hi + lo/2048 (22-bit operands, f32 accumulate); without it the all-fp32-MFMA kernel runs.
                                   (bits 16, 64, 128 selected kernel generations that no longer exist: ignored) */
#define CVAE_CLAMP_LAPLACE (1 << 30) /* OR into a clamp_lat_dim argument: the Laplace variant's floor instead of ln(1e-6) */
#define CVAE_FLAG_EXACT3 512 /* with PERSISTENT: every matrix product of the recurrent kernel on EXACT fp32 operands, each carried
                                as three fp16 limbs (x = l0 + l1/2^11 + l2/2^22), six f16 MFMAs per product, f32 accumulate
                                (k_gru_steps_v6; H = 1024 or 64, more than 16 batch rows); takes precedence over SPLIT_F16 */
#define CVAE_FLAG_GENERIC_STEP 4 /* with PERSISTENT: use the any-H recurrent kernel even where a tuned one exists (tests) */
#define CVAE_FLAG_STEP_TIMING 8  /* debugging: the tuned recurrent kernel accumulates per-phase cycle counters */
#define CVAE_FLAG_PROFILE 2    /* bracket the recurrent kernel(s) of each pass with hipEvents (see cvae_profile_collect) */

/*
 * One GRU_RNN.forward in eval mode (gru_vae.py:322-455, live branch: res/noise/softmax/... flags off).
 *   in      : the pass input, B*T rows of Cin = seg0.width + (lat ? lat_dim : seg1.width) features
 *   y_in    : [B,Cout]  initial feedback (gru_vae.py:365)
 *   h_in    : [B,H] or NULL (zeros)      (gru_vae.py:364-367)
 *   clamp_lat_dim : >=0 -> clamp trj_out[..., clamp_lat_dim:] to >= ln(1e-6) (clamp_vae, gru_vae.py:410-412);
 *                   L | CVAE_CLAMP_LAPLACE -> to >= -7.2543288692621097, the log-scale floor of the Laplace variant
 *                   (clamp_vae_laplace, gru_vae.py:415-417); ignored when the net has scale_out.  The same
 *                   encoding holds for every clamp_lat_dim argument below (train-mode forward / backward included)
 *   trj_out : [B,T,Cout]; y_last: [B,Cout] raw last projection; h_last: [B,H]   (gru_vae.py:452-453)
 *   status  : device int32[4]; status[0] != 0 after completion means a grid barrier timed out
 */
int cvae_gru_rnn_forward(cvae_ctx* ctx, const cvae_net_desc* d, const void* prepared, const cvae_pass_input* in,
                         const float* y_in, const float* h_in, int B, int T, int clamp_lat_dim,
                         float* trj_out, float* y_last, float* h_last,
                         void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * Up to 32 INDEPENDENT inputs through the same net as one pass (rows stacked along the batch axis: the recurrence is per row, so
 * stacking changes no result and divides the number of dependent steps).  Stage 6 runs E(src) || E(trg) and then its three
 * decoder passes this way (decode_gru-cyclevae_gauss.py:303-323), for one utterance pair or for several pairs at once (a step
 * costs the same chip-wide hand-off for 1 row and for 32: one row tile of the dataflow kernels).  in[c], y_in[c] [B][Cout], trj_out[c] [B][T][Cout] per cell;
 * h = 0, no y_last / h_last.  Workspace: cvae_pass_workspace_bytes(d, ncell * B, T).
 */
int cvae_gru_rnn_forward_stacked(cvae_ctx* ctx, const cvae_net_desc* d, const void* prepared, int ncell, const cvae_pass_input* in,
                                 const float* const* y_in, int B, int T, int clamp_lat_dim, float* const* trj_out,
                                 void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * The same with carried state (ABI 5), for passes over consecutive windows of the stacked utterances (cvae_pass_input::ctx_*):
 * h_in[c] [B][H] or NULL (state 0); h_last[c] [B][H] or NULL.  y_in[c] == NULL with h_in[c] != NULL means "this window CONTINUES
 * the recurrence that left h_in[c]": the feedback of its first frame is out_1(h_in) through the folded recurrent matrix, exactly
 * what frame t of an unbroken pass does with h_{t-1} -- windows then reproduce the unbroken pass bit for bit (same kernel, same
 * per-row arithmetic).  The reference has one Python loop per pass (gru_vae.py:391-394) and nothing to mirror here; this entry
 * exists so that the decoder launch of window w can run beside the encoder launch of window w+1 (stage6.convert_pairs).
 */
int cvae_gru_rnn_forward_stacked_carry(cvae_ctx* ctx, const cvae_net_desc* d, const void* prepared, int ncell, const cvae_pass_input* in,
                                       const float* const* y_in, const float* const* h_in, int B, int T, int clamp_lat_dim,
                                       float* const* trj_out, float* const* h_last, void* workspace, size_t workspace_bytes,
                                       int flags, void* stream);

/*
 * sampling_vae_batch (gru_vae.py:85-98) on device: z[n,l] = lat[n,l] + exp(lat[n,L+l]/2) * eps[n,l],
 * n < rows.  eps NULL -> Philox draw keyed (seed, draw_id, n, l).  eps_out (optional) receives the eps used.
 */
int cvae_sample(cvae_ctx* ctx, const float* lat, int rows, int lat_dim, const float* eps, uint64_t seed, uint64_t draw_id,
                float* z, float* eps_out, void* stream);

/*
 * SURVEY 8(f) row 4, first variant -- the Laplace posterior of the sibling recipes: sampling_vae_laplace (gru_vae.py:101-112, the
 * log-scale branch) on device: z[n,l] = lat[n,l] - exp(lat[n,L+l]) * sign(eps) * log1p(-2|eps|), eps ~ U(-0.4999, 0.5) as the
 * reference draws it (`uniform_(-0.4999, 0.5)`); eps NULL -> Philox keyed (seed, draw_id, n, l).  The backward gives
 * d lat = [dz ; dz * (z - mu)].  The matching clamp of GRU_RNN.forward is CVAE_CLAMP_LAPLACE in clamp_lat_dim; loss_vae_laplace
 * (:130-139) is torch ops in the drop-in module.
 */
int cvae_sample_laplace(cvae_ctx* ctx, const float* lat, int rows, int lat_dim, const float* eps, uint64_t seed, uint64_t draw_id,
                        float* z, float* eps_out, void* stream);
int cvae_sample_laplace_backward(cvae_ctx* ctx, const float* dz, const float* lat, const float* z, int rows, int lat_dim, float* dlat,
                                 void* stream);

/* Bytes of workspace the fused cycle chain needs. */
size_t cvae_cycle_workspace_bytes(cvae_ctx* ctx, const cvae_net_desc* enc, const cvae_net_desc* dec, int B, int T, int n_cyc);

/*
 * The n_cyc reconversion loop in eval form (train_gru_cyclevae_gauss_batch.py:1326-1338 with do=False):
 * per cycle  lat = E(x | [x[:,:,:stdim]; rec_cyc_prev]),  rec = D([code_src; z1]),  cv = D([code_trg; z2]),
 *            latcv = E([cvx; cv]),  rec_cyc = D([code_src; z3]).
 *   x [B,T,Cin_enc]; cvx [B,T,stdim]; code_src/code_trg [B,T,ncode]; y_in_enc [B,2L]; y_in_dec [B,Cout_dec]
 *   eps: NULL (Philox from seed) or [n_cyc,3,B,T,L] in draw order (rec, cv, rec_cyc)
 *   outputs (each may be NULL to skip the copy-out): lat/latcv [n_cyc,B,T,2L]; rec/cv/reccyc [n_cyc,B,T,Cout_dec]
 */
int cvae_cycle_forward(cvae_ctx* ctx, const cvae_net_desc* enc, const void* enc_prepared,
                       const cvae_net_desc* dec, const void* dec_prepared,
                       const float* x, const float* cvx, int stdim,
                       const float* code_src, const float* code_trg, int ncode,
                       const float* y_in_enc, const float* y_in_dec,
                       int B, int T, int n_cyc, int lat_dim, const float* eps, uint64_t seed,
                       float* out_lat, float* out_rec, float* out_cv, float* out_latcv, float* out_reccyc,
                       void* workspace, size_t workspace_bytes, int flags, void* stream);

/*
 * The same chain continued from / returning the state of every pass: the windowed form of the loop
 * (train_gru_cyclevae_gauss_batch.py:1299-1311: each of the 5 passes of a cycle restarts from ITS OWN (y_last, h) of the previous
 * window; the conv front-end is re-padded with zeros per window).  Pass order inside a cycle: encoder slots {lat, latcv},
 * decoder slots {rec, cv, reccyc}.  state_in NULL = fresh window (y_in_enc / y_in_dec, h = 0); state_out NULL = not wanted.
 * y values are the RAW last projections (pre-clamp / pre-scale_out, gru_vae.py:452).
 */
typedef struct cvae_cycle_state {
    float* y_enc; /* [n_cyc][2][B][2*lat_dim]  */
    float* y_dec; /* [n_cyc][3][B][Cout_dec]   */
    float* h_enc; /* [n_cyc][2][B][H_enc]      */
    float* h_dec; /* [n_cyc][3][B][H_dec]      */
} cvae_cycle_state;
int cvae_cycle_forward_carry(cvae_ctx* ctx, const cvae_net_desc* enc, const void* enc_prepared,
                             const cvae_net_desc* dec, const void* dec_prepared,
                             const float* x, const float* cvx, int stdim,
                             const float* code_src, const float* code_trg, int ncode,
                             const float* y_in_enc, const float* y_in_dec,
                             int B, int T, int n_cyc, int lat_dim, const float* eps, uint64_t seed,
                             float* out_lat, float* out_rec, float* out_cv, float* out_latcv, float* out_reccyc,
                             void* workspace, size_t workspace_bytes, int flags, void* stream,
                             const cvae_cycle_state* state_in, const cvae_cycle_state* state_out);

/*
 * Measurement aid for bench.py: with CVAE_FLAG_PROFILE every pass records a hipEvent pair on `stream` around its
 * recurrent kernel (the dominant kernel: k_gru_steps).  This call waits for the recorded pairs, returns their
 * summed elapsed time and count, and clears the list.  The events are the only thing the library ever allocates.
 */
int cvae_profile_collect(cvae_ctx* ctx, double* total_ms, int* launches);
/* The same brackets launch by launch (ABI 4): fills ms / rows / cin (stacked batch rows and input channels of the pass each
 * bracket belongs to: which instantiation and geometry of the recurrent kernel ran) for up to `cap` launches, returns how many,
 * forgets them.  bench.py groups them into the per-instantiation rooflines. */
int cvae_profile_collect_launches(cvae_ctx* ctx, double* ms, int* rows, int* cin, int cap);

/*
 * Debugging aid: after a cvae_gru_rnn_forward with CVAE_FLAG_PERSISTENT|CVAE_FLAG_STEP_TIMING on the tuned kernel,
 * out[0..3] = mean over blocks and out[4..7] = max over blocks of the cycle sums spent in
 * {operand loads + MFMA, reduce + gates + publish, store drain, barrier wait} over the T steps.  Synchronises.
 */
int cvae_step_timing(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T, const void* workspace, double out[8], void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training (stage 4, train_gru_cyclevae_gauss_batch.py:1326-1420): train-mode pass with a tape, BPTT backward, Adam.
 * Recurrences: one persistent launch per pass and direction at H = 1024 / 64 (exact fp32 operands as fp16 triples by default, option
 * "train_kernel"), per-step launches at other sizes; GEMMs on the fp32-input MFMA.
 * ------------------------------------------------------------------------------------------------------------------ */

/* Gradient outputs in the reference's state_dict layout (scale_in / scale_out are frozen, train...:369-372). */
typedef struct cvae_net_grads {
    float *conv0_w, *conv0_b, *conv1_w, *conv1_b, *w_ih, *w_hh, *b_ih, *b_hh, *out_w, *out_b;
} cvae_net_grads;

size_t cvae_train_image_bytes(cvae_ctx* ctx, const cvae_net_desc* d);
/* Weight image for the train-mode kernels; rebuild after every optimiser step.  gru_drop_p: the dropout probability the passes run
 * on this image will be called with (the `do_prob` of the reference module, gru_vae.py:312-316; 0 when dropout is off): the
 * exact-operand forward recurrence multiplies the feedback weights with the MASKED state, so the mask's scale 1/(1-p) is folded
 * into that weight image here.  cvae_gru_rnn_forward_train must be given the same p_drop (supplied masks: 0 or 1/(1-p)). */
int cvae_net_prepare_train(cvae_ctx* ctx, const cvae_net_desc* d, const cvae_net_weights* w, void* image, size_t image_bytes, float gru_drop_p,
                           void* stream);
/* The same with a choice of the MFMA-order weight images to build (ABI 4): `variants` = OR of 1 (exact-operand tile kernels), 2
 * (fp16-pair kernels), 4 (fp32-MFMA persistent forward); 0 = none of them -- enough for passes of at most three rows (word-exchange
 * kernels) and for the per-step launch paths.  cvae_net_prepare_train = variants 7.  cvae_train_variants_needed(d, B, T): what a
 * pass of that shape needs under the current options.  A pass whose image lacks what it needs fails with -4 (the library keeps a
 * host-side record per image address); the unused images are ~300 MB of writes and ~0.15 ms of kernels per net at hu1024. */
int cvae_net_prepare_train_v(cvae_ctx* ctx, const cvae_net_desc* d, const cvae_net_weights* w, void* image, size_t image_bytes, float gru_drop_p,
                             int variants, void* stream);
int cvae_train_variants_needed(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T);
size_t cvae_train_tape_bytes(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T);     /* per pass, kept until its backward */
size_t cvae_train_scratch_bytes(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T);  /* shared by all passes */

/*
 * GRU_RNN.forward with do=True (gru_vae.py:353-355 conv_drop, :378-382 gru_drop on the state fed to out_1; the carried h
 * is un-dropped).  x [B,T,Cin] contiguous.  cmask [B,T,ks^2*Cin] / gmask [T,B,H]: dropout masks already scaled by
 * 1/(1-p), or NULL to draw them with Philox from `seed`.  Activations needed by the backward are written to `tape`.
 */
int cvae_gru_rnn_forward_train(cvae_ctx* ctx, const cvae_net_desc* d, const void* image, const float* x, const float* y_in, const float* h_in,
                               int B, int T, int clamp_lat_dim, const float* cmask, const float* gmask, uint64_t seed,
                               float p_drop, float* trj_out, float* y_last, float* h_last, void* tape, size_t tape_bytes,
                               void* scratch, size_t scratch_bytes, void* stream);

/*
 * Backward of that pass (the autograd the reference gets from `batch_loss.backward()`, train...:1419): dout [B,T,Cout] is
 * d loss / d trj_out; y_last / h_last are treated as detached (train...:1301).  Writes dx [B,T,Cin] (may be NULL) and the
 * parameter gradients (accumulate != 0 adds to what is there).
 */
/*
 * cvae_set_side_stream (process-wide, NULL = off, the default): when cvae_gru_rnn_backward is called with accumulate != 0, the
 * four weight-gradient contractions of the recurrent / projection weights and their bias sums -- results nothing in the same
 * backward chain reads -- are enqueued on this second stream, so that they overlap the NEXT pass's reverse recurrence (a
 * latency-bound kernel that leaves most of every CU idle).  The caller then must (1) give consecutive backward calls of a net
 * different `scratch` buffers (the library makes a call wait for the side work that last used its scratch), (2) keep the `tape` of
 * a pass alive until the join, (3) call cvae_join_side_stream(stream) before anything on `stream` reads the gradient buffers.
 * The reference has no counterpart: autograd runs its backward on one stream.
 */
int cvae_set_side_stream(cvae_ctx* ctx, void* stream);
int cvae_join_side_stream(cvae_ctx* ctx, void* stream);

int cvae_gru_rnn_backward(cvae_ctx* ctx, const cvae_net_desc* d, const void* image, const float* dout, int B, int T, int clamp_lat_dim,
                          const void* tape, void* scratch, size_t scratch_bytes, float* dx, const cvae_net_grads* g,
                          int accumulate, void* stream);

/* With the option "train_profile" set, every launch (or per-step launch sequence) of the training recurrences and every training
 * GEMM (its split-contraction reduction included) is bracketed by a HIP event pair on the stream it is launched on.  This call waits
 * for them and returns, per class {0 forward recurrence, 1 reverse recurrence, 2 forward / data-gradient GEMMs, 3 weight-gradient
 * contractions}, the summed duration (ms), the number of brackets and the summed 2*M*N*K of the GEMM classes; then forgets them.
 * Measurement only (an event record leaves ~6 us of idle stream on either side of a launch). */
int cvae_train_profile_collect(cvae_ctx* ctx, double total_ms[4], int launches[4], double flop[4]);

/* Debugging aid: with the option "train_prof" set, block 0 of the persistent training forward recurrence
 * accumulates shader-cycle sums per phase {poll, loads+MFMA, reduce+cell math, publish} in out[0..3] (out[4..7] unused). */
int cvae_train_debug_counters(cvae_ctx* ctx, const cvae_net_desc* d, int B, int T, const void* scratch, long long out[8], void* stream);

/* torch.optim.Adam semantics (no weight decay), `step` counted from 1 (train...:377, :1420), over one flat buffer.
 * gate: NULL, or an int32 the DEVICE can read (normally the status sink of cvae_set_status_sink): when it is non-zero at the time
 * the kernel runs -- a hand-off timed out or a gate gradient left the exchange range during this step -- the update is skipped and
 * parameters and moments keep their values, so a bad step can never corrupt the optimiser state. */
int cvae_adam_step(cvae_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                   float beta2, float eps, int step, const int32_t* gate, void* stream);
/* The same update with the step counter ON THE DEVICE (ABI 4): state = int32[4] in device memory, state[0] = number of updates
 * applied so far (the caller zeroes it once, or writes the `step` of a resumed optimiser), state[1..2] scratch for the bias
 * corrections.  The counter advances only when the gate lets the update through, so a loop that does not synchronise every step
 * keeps Adam's bias correction right across skipped steps -- on every data-parallel rank alike, since `gate` is then the
 * MAX-reduced latch of cvae_status_latch. */
int cvae_adam_step_counted(cvae_ctx* ctx, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                           float beta2, float eps, int32_t* state, const int32_t* gate, void* stream);

/*
 * Stage-4 glue between the passes of a training step (no GEMM: element-wise, one launch each).
 *
 * cvae_sample_cat: the decoder input of train...:1335-1338, [code ; sampling_vae_batch(lat)] (gru_vae.py:85-98 + torch.cat), for
 * `parts` = 1 pass or for the two passes rec || cv stacked along the batch axis (parts = 2: rows [0,B) use code0 / eps0 / draw0,
 * rows [B,2B) code1 / eps1 / draw1).  lat [B][T][2L]; code [B][T][ncode]; eps NULL = Philox keyed (seed, draw, global frame, dim)
 * like cvae_sample; out [parts*B][T][ncode+L]; eps_out [parts][B][T][L] receives the eps used (the backward reads it).
 * cvae_sample_cat_backward: dout [parts*B][T][ncode+L] -> dlat [B][T][2L] (written, not accumulated).
 *
 * cvae_stage4_loss: one cycle's terms of the batch loss of train...:1363-1410 and their gradients.  Per frame (b, t) with weight
 * w[b][t] (1/n_j over the first n_j = min(flen_acc[j], T) frames of a selected utterance, else 0: the mean over frames, summed
 * over utterances):  w * ( K |rec - tgt|_1 + K |reccyc - tgt|_1 + kl_scale KL(lat) + latcv_w[b] KL(latcv) ),
 * K = 10/ln10 sqrt2 (gru_vae.py:525-527), KL = 0.5 sum(exp(s) + mu^2 - s - 1) (gru_vae.py:117-123), tgt = x[b][t][stdim:stdim+D].
 * kl_scale / latcv_w carry the reference's :1393 quirk (KL(lat) counted twice for more than one selected utterance, KL(latcv) of
 * the last selected utterance only); reccyc / latcv NULL = half cycle (:283-287).  Writes the four gradient arrays (d loss / d
 * trajectory, same shapes), frame_loss [B*T] (scratch) and loss[0] (+= when accumulate != 0) summed in a fixed order.
 */
int cvae_sample_cat(cvae_ctx* ctx, const float* lat, const float* code0, const float* code1, const float* eps0, const float* eps1, uint64_t seed,
                    uint64_t draw0, uint64_t draw1, int B, int T, int lat_dim, int ncode, int parts, float* out, float* eps_out,
                    void* stream);
int cvae_sample_cat_backward(cvae_ctx* ctx, const float* dout, const float* lat, const float* eps, int B, int T, int lat_dim, int ncode, int parts,
                             float* dlat, void* stream);
int cvae_stage4_loss(cvae_ctx* ctx, const float* rec, const float* reccyc, const float* lat, const float* latcv, const float* x, int x_stride, int stdim,
                     const float* w, const float* latcv_w, float kl_scale, int B, int T, int D, int lat_dim, float* d_rec,
                     float* d_reccyc, float* d_lat, float* d_latcv, float* frame_loss, float* loss, int accumulate, void* stream);
/*
 * The script's own per-utterance loss calls (train...:1366-1372) as one launch each way (ABI 4), for the UNCHANGED script flow:
 * cvae_mcd_l1 = TWFSEloss(x, y, twf=None, GV=False, L2=False) (gru_vae.py:525-533): per frame mcd_i = (10/ln10) sqrt2 sum_d |x - y|,
 * out3 = (sum, mean, unbiased std), frame_mcd [frames] kept for the backward; x / y rows of D floats with the given row strides.
 * cvae_mcd_l1_backward: dx [frames][D] from the three upstream gradients g3 (device).  cvae_kl_gauss = loss_vae (gru_vae.py:117-123):
 * mean over frames of 0.5 sum_l (exp(s) + mu^2 - s - 1), param rows [mu | s] of 2 * lat_dim floats; cvae_kl_gauss_backward: dparam
 * [frames][2 * lat_dim] contiguous.  As torch ops these are ~8 launches forward and as many autograd nodes backward per call, five
 * calls per utterance and cycle.
 */
int cvae_mcd_l1(cvae_ctx* ctx, const float* x, long x_stride, const float* y, long y_stride, int frames, int D, float* frame_mcd, float* out3, void* stream);
int cvae_mcd_l1_backward(cvae_ctx* ctx, const float* x, long x_stride, const float* y, long y_stride, int frames, int D, const float* frame_mcd,
                         const float* out3, const float* g3, float* dx, void* stream);
int cvae_kl_gauss(cvae_ctx* ctx, const float* param, long stride, int frames, int lat_dim, float* out1, void* stream);
int cvae_kl_gauss_backward(cvae_ctx* ctx, const float* param, long stride, int frames, int lat_dim, const float* g1, float* dparam, void* stream);

/*
 * Stage-6 post-processing next to the decoder output, f64 on the device like the reference's numpy on
 * the host.  GV post-filter, decode_gru-cyclevae_gauss.py:417-420:
 *   mean_d = mean_t c[t][d];  out[t][0] = c[t][0] (+ dpow[t]);  out[t][d] = sqrt(gv_trg[d-1]/cvgv[d-1]) * (c[t][d]-mean_d) + mean_d
 * c [T][D] fp32 (the decoder trajectory); dpow NULL or [T] (the power correction of mod_pow, feature_extract_vc.py:131-138,
 * whose SPTK mc2e is out of scope); gv_trg, cvgv [D-1]; out [T][D] f64; out_var NULL or [D-1] = np.var(out[:,1:], 0)
 * (decode...:421); work: 2*D doubles of device scratch.
 */
int cvae_gv_postfilter(cvae_ctx* ctx, const float* c, int T, int D, const double* dpow, const double* gv_trg, const double* cvgv, double* out,
                       double* out_var, double* work, void* stream);

/*
 * Energy of the impulse response of every frame's mel-cepstrum, f64: SPTK's mc2e, which the reference reaches through
 * pysptk.mc2e in mod_pow (feature_extract_vc.py:131-138; decode_gru-cyclevae_gauss.py:406: the power correction
 * dpow = log(mc2e(mcep) / mc2e(cvmcep)) / 2 added to coefficient 0, i.e. the `dpow` argument of cvae_gv_postfilter):
 * c' = freqt(mc, irlen - 1, -alpha), h = c2ir(c', irlen), e = sum h^2.  mc [T][ld] device memory, float32 (is_f64 = 0) or
 * float64; e_out [T].  pysptk is not in the reference tree nor in this image: restated from SPTK's published freqt / c2ir.
 */
int cvae_mc2e(cvae_ctx* ctx, const void* mc, int is_f64, long ld, int T, int D, double alpha, int irlen, double* e_out, void* stream);

/*
 * Frame-wise mel-cepstral distortion of two ALIGNED sequences over coefficients d0..D-1, f64 (gru_vae.py:523 L2 / :525 L1;
 * the per-frame values dtw_c.calc_mcd is called for at decode...:377-378 with d0 = 0 "mcdpow" and d0 = 1 "mcd").
 * a, b [rows][ld] fp32; frames [rows] f64; stats NULL or [4] = sum, mean, population std (np.std), sample std (torch.std).
 */
int cvae_mcd_aligned(cvae_ctx* ctx, const float* a, long lda, const float* b, long ldb, int rows, int D, int d0, int l2, double* frames,
                     double* stats, void* stream);

/*
 * Dynamic time warping of org [T1][D] onto the time axis of trg [T2][D], f64, on the device -- the role of dtw_c.dtw_org_to_trg at
 * decode_gru-cyclevae_gauss.py:334-364, :424 and train...:679-688 (there: a D2H copy per utterance and a host library).  dtw_c's
 * source is not in the reference tree: PARITY UNPINNED; the algorithm is the textbook one, every choice documented at
 * oracle/cyclevae_oracle.py::dtw_org_to_trg (local cost mel-cd, or cosine distance with mcd = 0; steps (1,1), (1,0), (0,1) with unit
 * weights; free of windows; target frame j takes the path's org frame of smallest local cost).  Outputs: aligned [T2][D], twf [T2]
 * (int64 org index per target frame), frames [T2] (their local costs), mean_out [1].  work: cvae_dtw_work_bytes(T1, T2) of device memory.
 */
size_t cvae_dtw_work_bytes(cvae_ctx* ctx, int T1, int T2);
int cvae_dtw_org_to_trg(cvae_ctx* ctx, const double* org, const double* trg, int T1, int T2, int D, int mcd, double* aligned, long long* twf,
                        double* frames, double* mean_out, void* work, size_t work_bytes, void* stream);

/* Copy status words (int32[4]) of a workspace to the host; synchronises `stream`.  status[0]!=0 = barrier timeout. */
int cvae_workspace_status(cvae_ctx* ctx, const void* workspace, int32_t status_out[4], void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CYCLEVAE_HIP_H */
