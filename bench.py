#!/usr/bin/env python
"""Headline benchmark: mcep frames/s of the hu1024/ld32/cyc2 CycleVAE eval chain on MI355X.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one synthetic batch: the n_cyc=2 reconversion chain (4 encoder + 6
decoder GRU_RNN passes, reference train_gru_cyclevae_gauss_batch.py:1326-1338 in eval form, latent draws on
device) on x[B=64 per GPU, T=80, 54].  Inputs and weights are resident in HBM before the timed region.  Utterance
rows are independent, so N GPUs shard the batch with no data-path collective (weak scaling).
Rank 0 prints ONE JSON line; `roofline` and `cpu_baseline` are described in DESIGN.md section "Measurement".
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cyclevae-vc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic MACs per frame per pass
MAC_ENC, MAC_DEC = 5166220, 4397100
# what the dominant kernel (k_gru_steps_v5 / v4: front-end + recurrence of one pass) computes, in the reference's terms:
# conv0 + conv1 + W_ih[:, :9C].x_conv + W_ih[:, 9C:].y + W_hh.h  (everything of a pass but scale_in, out_1, scale_out)
MAC_KERN_ENC = 26244 + 236196 + 1492992 + 196608 + 3145728
MAC_KERN_DEC = 10404 + 93636 + 940032 + 153600 + 3145728
PEAK_F32_MFMA_TFLOPS = 157.3                                      # MI355X_MICROARCH.md chip table


def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def _emit(res):
    """The ONE JSON line, as the last line of stdout (whatever C libraries still hold in their stdio buffers goes out first)."""
    _flush_c_stdio()
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--profile-every", type=int, default=4,
                    help="HIP events bracket the dominant kernel's launches on every N-th timed step (1: every launch)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("eval", "train"), default="eval",
                    help="eval: the headline cyc2 eval chain (BASELINE configs[1]); train: one stage-4 step (configs[2])")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="utterance rows per GPU (default 64 eval, 8 train)")
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-paths", action="store_true", help="skip the conversion-only / single-utterance extras (profiling runs)")
    ap.add_argument("--no-persistent", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel's launches with HIP events")
    ap.add_argument("--config", choices=("headline", "stress"), default="headline",
                    help="headline: hu1024/ld32/cyc2 (BASELINE configs[1]); stress: the eval chain at hu2048/ld64/cyc4 (the forward of configs[4])")
    ap.add_argument("--headline-only", action="store_true", help="time only the default (exact-operand) kernel (profiling runs)")
    ap.add_argument("--no-train-leg", action="store_true", help="eval mode: skip the stage-4 training-step leg (BASELINE configs[2])")
    ap.add_argument("--train-batch", type=int, default=64, help="utterances per GPU of the training-step leg")
    ap.add_argument("--train-steps", type=int, default=8)
    ap.add_argument("--train-warmup", type=int, default=2)
    ap.add_argument("--train-kernel", choices=("exact3", "pair", "fp32"), default="exact3",
                    help="operand form of the training recurrences (library option train_kernel)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: still init_process_group('nccl') (RCCL) and issue every collective of the N > 1 path -- the "
                         "barrier, the max-over-ranks time, the flat gradient all-reduce, the status MAX-reduce -- in a group of one")
    ap.add_argument("--no-other-flows", action="store_true", help="train leg: skip the unfused / unchanged-script flows")
    ap.add_argument("--step-option", action="append", default=[], metavar="NAME=VALUE",
                    help="attribute of stage4.Stage4Step set after construction, e.g. prep_dec_on_side=0; measurement runs only")
    ap.add_argument("--lib-option", action="append", default=[], metavar="NAME=VALUE",
                    help="library tuning / diagnostic switch (cvae_set_option), e.g. exp=2; measurement runs only")
    args = ap.parse_args()

    import shard
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-run this very command as N ranks (one process per GPU,
        # RCCL rendezvous on 127.0.0.1) and pass rank 0's JSON line through
        sys.exit(shard.spawn_ranks(__file__, sys.argv[1:], args.gpus))
    world, rank, local = shard.launched_world(args.gpus)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local)          # before the process group: RCCL binds the communicator to the current device
    dev = torch.device("cuda", local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:          # (no launcher: --force-dist on one GPU)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        dist.init_process_group("nccl", rank=rank, world_size=world)
        # RCCL writes a version banner through C stdio when the communicator comes up; into a pipe that is buffered until the
        # process exits and would land BEHIND the JSON line.  Bring the communicator up now and push the banner out, on every rank.
        dist.barrier()
        _flush_c_stdio()
    import _cabi
    import gru_vae
    import synth

    if args.no_persistent:
        gru_vae.set_kernel(persistent=False)

    for kv in args.lib_option:
        gru_vae._lib().set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if args.batch_per_gpu is None:
        args.batch_per_gpu = 64 if args.mode == "eval" else 8
    if args.mode == "train":
        res = train_leg(args, world, rank, dev, args.batch_per_gpu, args.steps, args.warmup, stress=args.config == "stress")
        if use_dist:
            dist.destroy_process_group()
        if rank == 0:
            _emit(res)
        return
    if args.config == "stress":
        return bench_stress(args, world, rank, dev)
    B, T, L, NCYC = args.batch_per_gpu, args.frames, 32, 2
    P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="bench/rank%d" % rank)
    W = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="bench/rank0")    # every rank holds the same weights

    def mod(sd, i, o, enc):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, kernel_size=3, dilation_size=2,
                            scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev).eval()

    enc, dec = mod(W.enc, 54, 64, True), mod(W.dec, 34, 50, False)
    chain = gru_vae.CycleChain(enc, dec, lat_dim=L, n_cyc=NCYC)
    gru_vae.set_draw_origin(rank * B, world * B, T)       # latent draws keyed by GLOBAL row: results independent of N
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    inputs = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    lib = gru_vae._lib()

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    flags_env = not args.no_kernel_events

    per_launch = {}

    def timed_leg(kernel, warm):
        """W warm-up chains, then EXACTLY K timed chains on the named recurrent kernel; HIP events recorded by the library on
        this stream bracket every launch of the dominant kernel inside the timed region."""
        gru_vae._force_kernel = kernel
        with torch.no_grad():
            for _ in range(warm):
                chain(*inputs, seed=1234)
            sync_all()
            lib.profile_collect()
            t0 = time.perf_counter()
            for k in range(args.steps):
                # the event pairs bracket the kernel's launches of every `--profile-every`-th timed step: an event record costs
                # ~6 us of idle stream on either side of a launch (rocprofv3 trace), 2.3 % of the step when every launch carries one
                gru_vae._flags_extra = _cabi.FLAG_PROFILE if flags_env and k % max(1, args.profile_every) == 0 else 0
                chain(*inputs, seed=1000 + k)
            sync_all()
            dt_ = time.perf_counter() - t0
            gru_vae._flags_extra = 0
        launches = lib.profile_collect_launches()
        ms, n = sum(l[0] for l in launches), len(launches)
        per_launch[kernel] = launches
        assert chain.status()[0] == 0, "a hand-off spin timed out during the bench (%s)" % kernel
        gru_vae._force_kernel = None
        if use_dist:
            import shard
            dt_ = shard.max_over_ranks(dt_, dist, dev, force=args.force_dist)
        return dt_, ms, n

    # headline: the exact-operand kernel (k_gru_steps_v6).  The two other forms of the same kernel are timed in the same run
    # and reported as co-equal lines: split2 (22-bit fp16 pairs, k_gru_steps_v5) and fp32 (v_mfma_f32_16x16x4_f32, v4).
    dt, kern_ms, kern_n = timed_leg("exact3", args.warmup)
    legs = {}
    if not args.headline_only:
        for name in ("split2", "fp32"):
            legs[name] = timed_leg(name, max(1, args.warmup))
    frames_per_step = B * T * world
    value = frames_per_step * args.steps / dt

    # ---- second leg of the default run: the stage-4 training step (BASELINE configs[2]), every rank takes part (gradient all-reduce)
    train_res = None
    if not args.no_train_leg and not args.no_persistent:
        try:
            train_res = train_leg(args, world, rank, dev, args.train_batch, args.train_steps, args.train_warmup)
        except Exception as e:      # the headline line must survive a failure of the second leg
            train_res = {"error": "%s: %s" % (type(e).__name__, e)}
            log("training-step leg failed: %s" % train_res["error"])

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    flop_frame = 2 * (NCYC * 2 * MAC_ENC + NCYC * 3 * MAC_DEC)
    res = {
        "metric": "mcep_frames_per_sec_hu1024_ld32_cyc2", "value": value, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "data": "synthetic",
        "dtype": "f32 (every matrix product on exact fp32 operands: each operand carried as three fp16 limbs x = l0 + l1/2^11 + "
                 "l2/2^22, six v_mfma_f32_32x32x16_f16 per product, f32 accumulate; dropped terms < 2^-33 of a product; gates, "
                 "carried state, projection and outputs f32)",
        "config": {"workload": "cyc2 eval chain: 4 encoder + 6 decoder GRU_RNN passes over x[B,T,54] (BASELINE configs[1])",
                   "batch_per_gpu": B, "frames": T, "hidden_units": 1024, "lat_dim": L, "n_cyc": NCYC,
                   "latent_draws": "on-device Philox", "sharding": "batch rows, %d/GPU, no collective" % B,
                   "recurrence": "per-step launches" if args.no_persistent else "one launch per pass (every block resident, hand-over through flags)"},
        "whole_job": {"algorithmic_flop_per_frame": flop_frame, "tflops": value * flop_frame / 1e12,
                      "frac_of_f32_mfma_peak": value * flop_frame / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)},
    }
    # ---- roofline of the dominant kernel (front-end + T-step recurrence of one pass, one launch per pass).
    # Launches per step: 8 on the persistent path (4 encoder passes, 2 single decoder passes, 2 launches that run rec||cv
    # stacked over 2B rows).  achieved = ALGORITHMIC fp32 flops of all timed launches / their summed HIP-event time.
    flop_per_step = 2.0 * B * T * (NCYC * 2 * MAC_KERN_ENC + NCYC * 3 * MAC_KERN_DEC)
    # MFMA instructions one (row tile, time step, block) executes per wave, and the shape / pipe cycles of that instruction
    # (MI355X_MICROARCH cycle table), per kernel; enc / dec differ in the front-end share
    KERN = {
        "exact3": dict(name="k_gru_steps_v6", insn="v_mfma_f32_32x32x16_f16", flop=2.0 * 32 * 32 * 16, cyc=32, rows=32, blocks=128,
                       per_wave=(96 + 48, 96 + 36), peak=2500.0,
                       operands="exact fp32 (three fp16 limbs per operand, six MFMAs per product)"),
        "split2": dict(name="k_gru_steps_v5", insn="v_mfma_f32_16x16x32_f16", flop=2.0 * 16 * 16 * 32, cyc=16, rows=16, blocks=64,
                       per_wave=(96 + 36, 96 + 27), peak=2500.0,
                       operands="22 significant bits (fp16 pairs hi + lo/2048, three MFMAs per product): NARROWER than fp32"),
        "fp32": dict(name="k_gru_steps_v4", insn="v_mfma_f32_16x16x4_f32", flop=2.0 * 16 * 16 * 4, cyc=32, rows=16, blocks=64,
                     per_wave=(256 + 96, 256 + 72), peak=PEAK_F32_MFMA_TFLOPS, operands="fp32 operands on the fp32-input MFMA"),
    }

    # HBM-side bytes per launch of the dominant kernel: the PMC counters cannot be read inside this run; the figure is the one
    # tools/prof_round.sh measured on this very command (two rocprofv3 --pmc passes, FETCH_SIZE doubled as the guide prescribes for
    # gfx950), committed as profiles/traffic.json next to the per-kernel counter summaries it was derived from
    traffic = {}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for kn in tj.get("kernel", []):
            traffic[kn.split("<")[0].replace("void ", "").strip()] = tj
    except (OSError, ValueError):
        pass

    def per_inst(kernel):
        """The same figure per geometry of the kernel: encoder passes (front-end over 54 channels: KFW 8), single decoder passes
        (34 channels: KFW 6) and the stacked rec || cv decoder launches (2B rows)."""
        out = {}
        for name, cin_, rows_, mac in (("encoder_pass_%d_rows" % B, 54, B, MAC_KERN_ENC), ("decoder_pass_%d_rows" % B, 34, B, MAC_KERN_DEC),
                                       ("decoder_rec_cv_stacked_%d_rows" % (2 * B), 34, 2 * B, MAC_KERN_DEC)):
            sel = [l[0] for l in per_launch.get(kernel, []) if l[1] == rows_ and l[2] == cin_]
            if not sel:
                continue
            avg = sum(sel) / len(sel)
            ach = 2.0 * rows_ * T * mac / (avg * 1e-3) / 1e12
            out[name] = {"launches_timed": len(sel), "avg_launch_ms": avg, "achieved": ach, "frac": ach / PEAK_F32_MFMA_TFLOPS}
        return out

    def roof(kernel, dt_, ms, n):
        if not (n > 0 and ms > 0):
            return None
        k = KERN[kernel]
        tj = traffic.get(k["name"]) if (B == 64 and T == 80) else None
        lps = n / float(len(range(0, args.steps, max(1, args.profile_every))))     # launches of one step
        avg_ms = ms / n
        ach = (flop_per_step / lps) / (avg_ms * 1e-3) / 1e12
        tiles = (B + k["rows"] - 1) // k["rows"]
        insn_per_step = 4 * T * k["blocks"] * tiles * (4 * k["per_wave"][0] + 6 * k["per_wave"][1])   # 4 waves per block
        exec_tf = insn_per_step * k["flop"] / lps / (avg_ms * 1e-3) / 1e12
        # matrix-pipe occupancy: pipe cycles of one SIMD's instructions per launch / launch duration in shader cycles is not
        # known without the clock; the counter-based figure is in profiles/ (SQ_VALU_MFMA_BUSY_CYCLES)
        return {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                "fp32_equivalent_frac": ach / PEAK_F32_MFMA_TFLOPS,
                "peak_is": "dense fp32-input MFMA (157.3 TFLOP/s), the governing roofline of SURVEY 8(d); achieved = ALGORITHMIC "
                           "fp32 flops / HIP-event time of the kernel's launches",
                "traffic": tj["k_gru_steps_hbm_bytes_per_launch"] if tj else None,
                "traffic_is": ("fabric-side bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc in separate passes on this "
                               "command, profiles/traffic.json + profiles/r04_v6_pmc_*.md; Infinity-Cache hits included): every XCD's L2 "
                               "pulls the state and input window of both row tiles once per step") if tj else None,
                "kernel": "%s (front-end + T-step recurrence of one pass, one launch of an all-resident grid)" % k["name"],
                "operand_width": k["operands"],
                "executed": {"instruction": k["insn"], "tflops": exec_tf, "dense_peak_tflops": k["peak"],
                             "frac_of_executed_instruction_peak": exec_tf / k["peak"]},
                "avg_launch_ms": avg_ms, "launches_timed": n, "launches_per_step": lps,
                "launches_timed_are": "all launches of every %d-th timed step (HIP events recorded by the library on the launch stream)"
                                      % max(1, args.profile_every),
                "share_of_step_time": avg_ms * lps * args.steps / (1e3 * dt_) if world == 1 else None,
                "algorithmic_flop_per_launch": flop_per_step / lps,
                "per_instantiation": per_inst(kernel)}

    res["roofline"] = roof("exact3", dt, kern_ms, kern_n)
    res["other_kernels"] = {}
    for name, (dt_k, ms_k, n_k) in legs.items():
        res["other_kernels"][name] = {"value": frames_per_step * args.steps / dt_k, "unit": "frames/s",
                                      "ms_per_step": 1e3 * dt_k / args.steps, "roofline": roof(name, dt_k, ms_k, n_k)}

    # ---- sub-paths SURVEY 8(d) asks to report next to config 2 (rank 0, N=1 only; not part of `value`)
    if world == 1 and not args.no_sub_paths:
        def timed(fn, n):
            with torch.no_grad():
                for _ in range(2):
                    fn()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
            return (time.perf_counter() - t2) / n

        code_trg = inputs[3]

        def conversion_only():      # what stage 6 ships: E(x) -> z -> D([code_trg; z])   (decode...:303-311)
            lat = enc(inputs[0], inputs[4], clamp_vae=True, lat_dim=L)[0]
            z = gru_vae.sampling_vae_batch(lat, lat_dim=L)
            return dec(torch.cat((code_trg, z), 2), inputs[5])[0]

        tc = timed(conversion_only, 20)
        PU = synth.CycleVAEProblem(B=1, T=637, bias_scale=0.0, tag="bench/utt")
        xu, yu, cu, ydu = tt(PU.x[0]), tt(PU.y_in_enc), tt(PU.code_trg[0]), tt(PU.y_in_dec)

        def one_utterance():        # single 637-frame utterance through the 2-D path, 300-draw latent mean (decode...:303-311)
            lat = enc(xu, yu, clamp_vae=True, lat_dim=L)[0]
            z = torch.mean(gru_vae.sampling_vae_batch(lat.unsqueeze(0).repeat(300, 1, 1), lat_dim=L), 0)
            return dec(torch.cat((cu, z), 1), ydu)[0]

        tu = timed(one_utterance, 5)
        import stage6
        PV = synth.CycleVAEProblem(B=1, T=660, bias_scale=0.0, tag="bench/utt_trg")
        xv = tt(PV.x[0])

        def stage6_pair():          # the whole network path of decode...:302-323 for one (source, target) pair: two stacked launches
            return stage6.convert_pair(enc, dec, xu, xv, yu, ydu, ydu, L, n_smpl_dec=300)

        tp = timed(stage6_pair, 5)
        tpw = timed(lambda: stage6.convert_pair(enc, dec, xu, xv, yu, ydu, ydu, L, n_smpl_dec=300, window=224), 5)
        tp5 = timed(lambda: stage6.convert_pairs(enc, dec, [(xu, xv)] * 5, yu, ydu, ydu, L, n_smpl_dec=300), 5)
        tp10 = timed(lambda: stage6.convert_pairs(enc, dec, [(xu, xv)] * 10, yu, ydu, ydu, L, n_smpl_dec=300), 5)
        tpl = timed(lambda: stage6.convert_list(enc, dec, [[(xu, xv)]] * 8, yu, ydu, ydu, L, n_smpl_dec=300), 3) / 8
        tpl10 = timed(lambda: stage6.convert_list(enc, dec, [[(xu, xv)] * 10] * 6, yu, ydu, ydu, L, n_smpl_dec=300), 3) / 6
        seq_w = 4.0 * ((196608 + 3145728 + 65536) + (153600 + 3145728 + 51200))     # bytes of weights every frame needs, enc + dec
        res["sub_paths"] = {"conversion_only_B%dxT%d" % (B, T): {"frames_per_s": B * T / tc, "ms": 1e3 * tc, "passes": "1 encoder + 1 decoder"},
                            "single_utterance_T637_300draws": {
                                "frames_per_s": 637 / tu, "ms": 1e3 * tu, "us_per_dependent_step": 1e6 * tu / 1274,
                                "sequential_weight_bytes_per_s": seq_w * 637 / tu,
                                "passes": "1 encoder + 1 decoder at B=1 through the module API (2-D input): 1274 dependent steps, each a "
                                          "chip-wide hand-off (latency bound); weights stay register-resident, the bytes/s figure is "
                                          "what a weight-streaming implementation would have to move (SURVEY 8(d))"},
                            "stage6_pair_T637_T660_300draws": {
                                "converted_frames_per_s": 637 / tp, "ms": 1e3 * tp,
                                "passes": "decode...:302-323 for one utterance pair (2 encoder + 3 decoder passes) as two stacked launches "
                                          "(stage6.convert_pair), 300-draw latent means in the prologue"},
                            "stage6_pair_as_wavefront_of_windows": {
                                "converted_frames_per_s": 637 / tpw, "ms": 1e3 * tpw,
                                "passes": "the same pair cut into 224-frame windows (stage6.convert_pair(window=224)): passes with carried state "
                                          "whose conv front-end sees the neighbouring frames (ABI 5), the decoder launch of window w beside the "
                                          "encoder launch of window w+1 -- a pass-level wavefront, bit-identical to the unbroken pair"},
                            "stage6_list_of_pairs_pipelined": {
                                "converted_frames_per_s": 637 / tpl, "ms_per_pair": 1e3 * tpl,
                                "passes": "a list of eight such pairs, one pair per call (stage6.convert_list): the encoder launch of pair g+1 "
                                          "runs side by side with the decoder launch of pair g on a second stream -- two hand-off-bound "
                                          "recurrences co-resident on every CU; bit-identical to one convert_pair per pair"},
                            "stage6_list_of_ten_pair_calls_pipelined": {
                                "converted_frames_per_s": 10 * 637 / tpl10, "ms_per_call": 1e3 * tpl10,
                                "passes": "a list of six ten-pair calls through stage6.convert_list: a pass of <= 32 rows is ONE row tile of the "
                                          "dataflow kernel = 128 blocks, half the chip, so the encoder launch of call g+1 and the decoder launch "
                                          "of call g are resident together on disjoint CUs; bit-identical to call after call"},
                            "stage6_five_pairs_per_call": {
                                "converted_frames_per_s": 5 * 637 / tp5, "ms": 1e3 * tp5,
                                "passes": "the same for five utterance pairs at once (10 encoder rows, 15 decoder rows per stacked launch): "
                                          "a dependent step costs the same hand-off for one row and for thirty-two"},
                            "stage6_ten_pairs_per_call": {
                                "converted_frames_per_s": 10 * 637 / tp10, "ms": 1e3 * tp10,
                                "passes": "ten pairs per call: 20 encoder rows, 30 decoder rows = one 32-row tile of the dataflow kernel, "
                                          "the most a call takes"}}

    # ---- parity in the same run + CPU baseline (rank 0, N=1 only)
    if world == 1:
        from oracle import torch_stock as ts
        from oracle import cyclevae_oracle as orc
        ncpu = os.cpu_count() or 1
        log("gpu: %.0f frames/s, %.3f ms/step; host has %d logical cpus" % (value, 1e3 * dt / args.steps, ncpu))
        ce, cd = ts.StockGRURNN(W.enc, 54, 64, 1024), ts.StockGRURNN(W.dec, 34, 50, 1024)
        c = lambda a: torch.from_numpy(np.ascontiguousarray(a))

        def cpu_chain(nrow, nfr):
            a = [c(getattr(P, n)[:nrow, :nfr]) for n in ("x", "cvx", "code_src", "code_trg")]
            a += [c(P.y_in_enc[:nrow]), c(P.y_in_dec[:nrow]), c(P.eps[:, :, :nrow, :nfr])]
            t1 = time.perf_counter()
            r = ts.cycle_chain(ce, cd, *a, NCYC, L)
            return r, time.perf_counter() - t1

        # thread count: tiny per-frame GEMMs do not scale to every hyper-thread; pick the fastest of a few
        # candidates on an 8-frame slice of the same batch, then time the real sample with it
        best_thr, best_t = 1, None
        for thr in sorted(set(min(ncpu, k) for k in (8, 16, 32, 64, 128))):
            torch.set_num_threads(thr)
            cpu_chain(B, 4)
            tcal = cpu_chain(B, 8)[1]
            log("cpu calibration: %d threads -> %.3f s for B=%d,T=8" % (thr, tcal, B))
            if best_t is None or tcal < best_t:
                best_thr, best_t = thr, tcal
            elif tcal > 1.3 * best_t:
                break
        torch.set_num_threads(best_thr)
        nrow = min(B, 32)       # more than 16 rows: the batch runs the same 32-row-tile kernel as the timed region
        r = cpu_chain(nrow, T)[0]

        def mcd_of(kernel):
            gru_vae._force_kernel = kernel
            with torch.no_grad():
                g = chain(*[v[:nrow] for v in inputs], eps=tt(P.eps[:, :, :nrow]))
            gru_vae._force_kernel = None
            out = {}
            for k in ("rec", "cv", "reccyc"):
                a = g[k].cpu().numpy().reshape(-1, 50)
                b = np.stack([v.numpy() for v in r[k]]).reshape(-1, 50)
                out[k] = [float(np.mean(orc.mcd_frames(a, b))), float(np.mean(orc.mcd_frames(a[:, 1:], b[:, 1:])))]
            return out

        mcd = mcd_of("exact3")
        res["mcd_db_vs_cpu"] = {"rows": nrow, "per_output_dims0_49_and_1_49": mcd,
                                "max": max(max(v) for v in mcd.values()), "budget": 0.01}
        log("mcd vs cpu: %s" % res["mcd_db_vs_cpu"]["max"])
        for name in res["other_kernels"]:
            m2 = mcd_of(name)
            res["other_kernels"][name]["mcd_db_vs_cpu_max"] = max(max(v) for v in m2.values())
        if not args.no_cpu_baseline:
            # bounded sample: the full B x T chain if one run fits ~6 s, else fewer frames of the same batch
            est = best_t * T / 8.0
            nfr = T if est <= 6.0 else max(8, int(T * 6.0 / est))
            reps = 5 if est <= 3.0 else 3
            cpu_chain(B, nfr)
            times = [cpu_chain(B, nfr)[1] for _ in range(reps)]
            med = sorted(times)[len(times) // 2]
            res["cpu_baseline"] = {"value": B * nfr / med, "unit": "frames/s", "cores": best_thr, "kind": "port",
                                   "sample": "the same cyc2 chain on B=%d rows x T=%d frames of the bench batch, stock torch.nn "
                                             "Conv1d/GRU composed like the reference (oracle/torch_stock.py), fp32, %d threads "
                                             "(fastest of a calibration sweep; host has %d logical cpus), median of %d after 1 "
                                             "warm-up" % (B, nfr, best_thr, ncpu, reps),
                                   "ms_per_step": 1e3 * med}
            # the numpy restatement of the reference (oracle/cyclevae_oracle.py, the parity checker) timed beside it on a
            # shorter slice of the same batch (SURVEY 8(d) asks for both); whatever BLAS threading numpy comes with
            nf2 = max(4, min(T, 16))
            t2 = time.perf_counter()
            orc.cycle_chain(W.enc, W.dec, P.x[:, :nf2], P.cvx[:, :nf2], P.code_src[:, :nf2], P.code_trg[:, :nf2], P.y_in_enc,
                            P.y_in_dec, P.eps[:, :, :, :nf2], NCYC, L)
            t2 = time.perf_counter() - t2
            res["cpu_baseline"]["numpy_restatement"] = {"value": B * nf2 / t2, "unit": "frames/s",
                                                        "sample": "one run of the same chain on B=%d rows x T=%d frames" % (B, nf2)}
    if train_res is not None:
        res["train_step"] = train_res
    if args.force_dist:
        res["config"]["collectives"] = ("--force-dist: process group 'nccl' (RCCL) of ONE rank; barrier + MAX all-reduce of the elapsed time "
                                        "around the timed region, flat gradient all-reduce + status MAX-reduce in every training step")
    if use_dist:
        dist.destroy_process_group()
    _emit(res)


def bench_stress(args, world, rank, dev):
    """The eval chain at the dims of BASELINE configs[4]: hu2048 / ld64 / n_cyc = 4 (8 encoder 54->128 + 12 decoder 66->50 passes) on
    x[B=64 per GPU, T=80, 54].  Recurrent kernel: k_gru_steps_v6<32, ., 3, W2S> -- 8-unit x 32-row blocks on all 256 CUs, both row
    tiles of the batch in every block, exact fp32 operands as fp16 triples; l0 and l1 of a block's 32 columns x 2048 k fill 256
    registers per lane, the third weight limbs are streamed from L2 every step as bf8 bytes.  The fp16-PAIR form (22-23 bits,
    library option v6_limbs_h2048=2) is timed in the same run as `other_kernels.pairs`."""
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    import _cabi
    import gru_vae
    import synth
    from oracle import cyclevae_oracle as orc

    B, T, L, NCYC, H = args.batch_per_gpu, args.frames, 64, 4, 2048
    mac_enc, mac_dec = 16882828, 17036588                         # SURVEY 8(d), per frame and pass
    mac_k_enc = mac_enc - 2916 - H * 2 * L                        # without scale_in and out_1 (the projection kernel)
    mac_k_dec = mac_dec - H * 50 - 2500                           # without out_1 and scale_out
    P = synth.CycleVAEProblem(B=B, T=T, lat_dim=L, hidden=H, n_cyc=NCYC, bias_scale=0.0, tag="stressbench/rank%d" % rank)
    W = synth.CycleVAEProblem(B=1, T=1, lat_dim=L, hidden=H, n_cyc=NCYC, bias_scale=0.0, tag="stressbench/rank0")

    def mod(sd, i, o, enc):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=H, kernel_size=3, dilation_size=2, scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev).eval()

    enc, dec = mod(W.enc, 54, 2 * L, True), mod(W.dec, 2 + L, 50, False)
    chain = gru_vae.CycleChain(enc, dec, lat_dim=L, n_cyc=NCYC)
    gru_vae.set_draw_origin(rank * B, world * B, T)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    inputs = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    lib = gru_vae._lib()

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(limbs):
        """warm-up + EXACTLY args.steps chains with the recurrent kernel on `limbs` fp16 limbs per operand (3: exact fp32, 2: pairs)"""
        lib.set_option("v6_limbs_h2048", limbs)
        with torch.no_grad():
            for _ in range(args.warmup):
                chain(*inputs, seed=1234, outputs=False)
            sync_all()
            lib.profile_collect()
            gru_vae._flags_extra = _cabi.FLAG_PROFILE
            t0 = time.perf_counter()
            for k in range(args.steps):
                chain(*inputs, seed=1000 + k, outputs=False)
            sync_all()
            dt_ = time.perf_counter() - t0
            gru_vae._flags_extra = 0
        ms_, n_ = lib.profile_collect()
        assert chain.status()[0] == 0
        if use_dist:
            import shard
            dt_ = shard.max_over_ranks(dt_, dist, dev, force=args.force_dist)
        return dt_, ms_, n_

    dt, kern_ms, kern_n = timed(3)
    dt2, kern_ms2, kern_n2 = timed(2)
    lib.set_option("v6_limbs_h2048", 3)
    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return
    value = B * T * world * args.steps / dt
    flop_frame = 2 * (NCYC * 2 * mac_enc + NCYC * 3 * mac_dec)
    flop_k = 2.0 * B * T * (NCYC * 2 * mac_k_enc + NCYC * 3 * mac_k_dec)
    lps = kern_n / float(args.steps)
    avg_ms = kern_ms / max(1, kern_n)
    ach = (flop_k / lps) / (avg_ms * 1e-3) / 1e12 if kern_n else 0.0
    tiles = (B + 31) // 32
    insn = 4 * T * 256 * tiles * (NCYC * 2 * (192 + 48) + NCYC * 3 * (192 + 66))   # per wave and tile-step: 32 steps x 6 + front-end 8|11 x 6
    exec_tf = insn * 2.0 * 32 * 32 * 16 / lps / (avg_ms * 1e-3) / 1e12 if kern_n else 0.0
    res = {"metric": "mcep_frames_per_sec_hu2048_ld64_cyc4", "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "data": "synthetic",
           "dtype": "f32 (every matrix product of the recurrent kernel on exact fp32 operands: three fp16 limbs, six v_mfma_f32_32x32x16_f16 "
                    "per product, f32 accumulate; the third limbs of the recurrent weights are streamed from L2 as bf8 bytes -- l0 and l1 of a "
                    "block's 32 columns x 2048 k fill the registers; gates, carried state, projection and outputs f32)",
           "config": {"workload": "cyc4 eval chain: 8 encoder (54->128) + 12 decoder (66->50) GRU_RNN passes, the forward of BASELINE configs[4]",
                      "batch_per_gpu": B, "frames": T, "hidden_units": H, "lat_dim": L, "n_cyc": NCYC, "sharding": "batch rows, no collective"},
           "whole_job": {"algorithmic_flop_per_frame": flop_frame, "tflops": value * flop_frame / 1e12,
                         "frac_of_f32_mfma_peak": value * flop_frame / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)},
           "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                        "fp32_equivalent_frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                        "peak_is": "dense fp32-input MFMA (157.3 TFLOP/s); achieved = ALGORITHMIC fp32 flops / HIP-event time of the kernel's launches",
                        "kernel": "k_gru_steps_v6<32, 8|11, 3, streamed third weight limb> (front-end + T-step recurrence of one pass, one launch "
                                  "of an all-resident grid)",
                        "operand_width": "exact fp32 (three fp16 limbs per operand, six MFMAs per product)",
                        "executed": {"instruction": "v_mfma_f32_32x32x16_f16", "tflops": exec_tf, "dense_peak_tflops": 2500.0,
                                     "frac_of_executed_instruction_peak": exec_tf / 2500.0},
                        "avg_launch_ms": avg_ms, "launches_timed": kern_n, "launches_per_step": lps},
           "other_kernels": {"pairs": {"value": B * T * world * args.steps / dt2, "unit": "frames/s", "ms_per_step": 1e3 * dt2 / args.steps,
                                       "frac_of_f32_mfma_peak": B * T * world * args.steps / dt2 * flop_frame / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
                                       "avg_launch_ms": kern_ms2 / max(1, kern_n2),
                                       "operand_width": "22-23 significant bits (fp16 pairs, three MFMAs per product): NARROWER than fp32; "
                                                        "library option v6_limbs_h2048=2"}},
           "cpu_baseline": None}
    if world == 1:
        nrow, rows = 32, [0, 13, 31]
        with torch.no_grad():
            g = chain(*[v[:nrow] for v in inputs], eps=tt(P.eps[:, :, :nrow]))
        t1 = time.perf_counter()
        r = orc.cycle_chain(W.enc, W.dec, P.x[rows], P.cvx[rows], P.code_src[rows], P.code_trg[rows], P.y_in_enc[rows], P.y_in_dec[rows],
                            P.eps[:, :, rows], NCYC, L)
        tcpu = time.perf_counter() - t1
        m = max(float(np.mean(orc.mcd_frames(g[k][:, rows].cpu().numpy().reshape(-1, 50), np.stack(r[k]).reshape(-1, 50))))
                for k in ("rec", "cv", "reccyc"))
        res["mcd_db_vs_cpu"] = {"rows": len(rows), "max": m, "budget": 0.01}
        res["cpu_baseline"] = {"value": len(rows) * T / tcpu, "unit": "frames/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": "the numpy restatement (oracle/cyclevae_oracle.py) on %d rows x %d frames of the same chain, one run, "
                                         "numpy's BLAS threading" % (len(rows), T)}
    if use_dist:
        dist.destroy_process_group()
    _emit(res)


TRAIN_KERNELS = {
    "exact3": (0, "f32 (forward and reverse recurrence: every matrix product on EXACT fp32 operands carried as three fp16 limbs, six f16 "
                  "MFMAs per product, f32 accumulate; all other GEMMs fp32-input MFMA)"),
    "pair": (1, "f32 accumulate; forward and reverse recurrence on fp16-PAIR operands (22 bits: narrower than fp32), three f16 MFMAs per "
                "product; all other GEMMs fp32-input MFMA"),
    "fp32": (2, "f32 (forward recurrence on the fp32-input MFMA, reverse recurrence as 2T fp32 launches; all GEMMs fp32-input MFMA)"),
}


def train_leg(args, world, rank, dev, B, steps, warmup, stress=False):
    """One step = the stage-4 step (cyc2 chain in train mode with dropout 0.5, loss, backward, gradient all-reduce when N > 1, Adam)
    on a fresh 80-frame window of B utterances per GPU (reference train_gru_cyclevae_gauss_batch.py:1326-1420; BASELINE configs[2],
    with stress=True the dims of configs[4]).  Every rank calls it; rank 0 gets the result dict, the others None."""
    import torch.distributed as dist
    import gru_vae
    import stage4
    import synth

    T = args.frames
    L, NCYC, H = (64, 4, 2048) if stress else (32, 2, 1024)
    mac_enc, mac_dec = (16882828, 17036588) if stress else (MAC_ENC, MAC_DEC)      # SURVEY 8(d), per frame and pass
    kw = dict(lat_dim=L, hidden=H, n_cyc=NCYC) if stress else {}
    P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="trainbench/rank%d" % rank, **kw)
    W = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="trainbench/rank0", **kw)
    lib = gru_vae._lib()
    kern_id, kern_dtype = TRAIN_KERNELS[args.train_kernel]
    use_dist = world > 1 or args.force_dist

    def set_kernel(kid):
        lib.set_option("train_kernel", kid)
        lib.set_option("train_fp32_mfma", 1 if kid == 2 else 0)
        lib.set_option("train_bwd_per_step", 1 if kid == 2 else 0)

    set_kernel(kern_id)

    def mod(sd, i, o, enc):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=H, kernel_size=3, dilation_size=2, do_prob=0.5,
                            scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev).train()

    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

    def sync_all():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(kernel, n_steps=None, **step_kw):
        """warmup + `steps` timed stage-4 steps on fresh modules with the recurrences in the named operand form"""
        n_steps = steps if n_steps is None else n_steps
        set_kernel(TRAIN_KERNELS[kernel][0])
        st_ = stage4.Stage4Step(mod(W.enc, 54, 2 * L, True), mod(W.dec, 2 + L, 50, False), lat_dim=L, n_cyc=NCYC, lr=1e-4,
                                dist=dist if use_dist else None, force_collectives=args.force_dist, **step_kw)
        for kv in args.step_option:
            setattr(st_, kv.split("=")[0], int(kv.split("=")[1]))
        for _ in range(warmup):
            st_(*data)
        sync_all()
        st_.time_allreduce = use_dist
        lib.profile_collect()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            loss_ = st_(*data)
        sync_all()
        dt_ = time.perf_counter() - t0
        if use_dist:
            import shard
            dt_ = shard.max_over_ranks(dt_, dist, dev, force=args.force_dist)
        return st_, dt_ / n_steps, float(loss_.item())

    try:
        gru_vae.set_draw_origin(rank * B, world * B, T)       # dropout masks / draws keyed by GLOBAL row: results independent of N
        data = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")] + [None]   # eps: Philox
        step, dt, final_loss = timed(args.train_kernel)
        dt *= steps
        ar_ms = [a.elapsed_time(b) for a, b in step.allreduce_ms]
        kernels = train_kernel_rooflines(args, lib, step, data, B, T, H, NCYC, stress) if rank == 0 or use_dist else None
        other = None
        if not stress and args.train_kernel == "exact3" and not args.headline_only:
            _, dt_p, _ = timed("pair")
            other = {"pair": {"value": B * T * world / dt_p, "unit": "frames/s", "ms_per_step": 1e3 * dt_p,
                              "dtype": TRAIN_KERNELS["pair"][1]}}
            set_kernel(kern_id)
        # the flows a user gets WITHOUT editing the training script, timed beside the fused step on the same batch:
        # only path.sh:11 changed = ten separate passes with per-pass autograd, the script's per-utterance loss loop with its host
        # read-backs, torch.optim.Adam; and the same with the loss vectorised (stage4.loss_terms)
        flows = None
        if world == 1 and not args.headline_only and not args.no_other_flows:
            nf = max(2, steps // 2)
            flows = {}
            for name, kwf, what in (
                    ("dropin_unchanged_script", dict(fused=False, stack_rec_cv=False, overlap_wgrad=False, script_loss=True),
                     "only path.sh:11 changed: what train...:1326-1420 executes -- ten GRU_RNN passes with per-pass autograd, the script's "
                     "per-utterance loss loop incl. its .item() read-backs (stage4.script_loss_loop), torch.optim.Adam"),
                    ("dropin_unfused", dict(fused=False, stack_rec_cv=False, overlap_wgrad=False),
                     "the same ten passes + torch.optim.Adam with the loss vectorised over utterances (stage4.loss_terms)")):
                _, dt_f, loss_f = timed(args.train_kernel, nf, **kwf)
                flows[name] = {"value": B * T / dt_f, "unit": "frames/s", "ms_per_step": 1e3 * dt_f, "steps": nf, "final_loss": loss_f,
                               "what": what}
        # the recipe's own utterance batches (run.sh:172-173: batch_size_utt = 1, alternative 8) on the same step, 1 GPU only:
        # passes of at most three rows run the word-exchange recurrences (cvae_train_ll.h)
        small = None
        if world == 1 and not stress and args.train_kernel == "exact3" and not args.headline_only and B > 8:
            small, full_data = {}, data
            for bs in (1, 8):
                Pb = synth.CycleVAEProblem(B=bs, T=T, bias_scale=0.0, tag="trainbench/b%d" % bs)
                gru_vae.set_draw_origin(0, bs, T)
                data = [tt(getattr(Pb, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")] + [None]
                _, dt_b, _ = timed(args.train_kernel)
                small["utterances_%d" % bs] = {"value": bs * T / dt_b, "unit": "frames/s", "ms_per_step": 1e3 * dt_b}
                if bs == 1 and not args.no_other_flows:
                    # the recipe's own configuration through the UNCHANGED script flow (only path.sh:11 swapped)
                    _, dt_u, _ = timed(args.train_kernel, max(2, steps // 2), fused=False, stack_rec_cv=False, overlap_wgrad=False,
                                       script_loss=True)
                    small["utterances_1"]["dropin_unchanged_script_ms_per_step"] = 1e3 * dt_u
                if bs == 1:
                    # what bounds a one-utterance step: its dependent steps (16 recurrent launches x T) times the measured
                    # chip-wide hand-off of the word-exchange kernels -- not the matrix pipe
                    dep = (2 * NCYC + 3 * NCYC) * 2 * T
                    small["utterances_1"]["roofline"] = {
                        "bound": "latency", "dependent_steps": dep, "us_per_dependent_step_floor": 0.41,
                        "floor_ms": dep * 0.41e-3, "frac": dep * 0.41e-3 / (1e3 * dt_b),
                        "floor_is": "forward + reverse recurrence steps of the ten passes x the 0.41 us cross-XCD store -> polled-load "
                                    "round trip (tools/mb/mb_pingpong.hip); the MFMA roofline does not govern an 80-row problem"}
            data = full_data
            gru_vae.set_draw_origin(rank * B, world * B, T)
        if rank != 0:
            return None
        value = B * T * world * steps / dt
        flop = 3.0 * 2 * (NCYC * 2 * mac_enc + NCYC * 3 * mac_dec)     # forward + dgrad + wgrad
        tf = value * flop / 1e12
        res = {
            "metric": "stage4_train_frames_per_sec_hu%d_ld%d_cyc%d" % (H, L, NCYC), "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "data": "synthetic",
            "dtype": "f32 (all GEMMs fp32 MFMA; per-step recurrence launches)" if stress else kern_dtype,
            "config": {"workload": "stage-4 step: cyc%d chain (train mode, dropout 0.5) + loss + backward + Adam (BASELINE configs[%d])"
                                   % (NCYC, 4 if stress else 2),
                       "utterances_per_gpu": B, "frames": T, "hidden_units": H, "lat_dim": L, "n_cyc": NCYC,
                       "rec_cv_stacked": step.stack_rec_cv, "weight_gradient_gemms_on_side_stream": bool(step.overlap_wgrad),
                       "glue": "cvae_sample_cat + cvae_stage4_loss + flat cvae_adam_step_counted (device-gated)" if step.fused else "torch ops + torch.optim.Adam",
                       "latent_draws_and_dropout_masks": "on-device Philox, keyed by global row",
                       "host_sync_per_step": "one (status word read after the update, like the reference's loss.item())",
                       "gradient_allreduce": ("one flat fp32 bucket per step (RCCL), gradients are views of it (no copies)"
                                              + (" -- group of ONE rank (--force-dist)" if world == 1 else "")) if use_dist else "none (1 GPU)"},
            "allreduce": {"ms_per_step_rank0": sum(ar_ms) / len(ar_ms), "bytes": 4 * step.grads.flat.numel(),
                          "timed_by": "HIP events around dist.all_reduce on rank 0"} if ar_ms else None,
            "final_loss": final_loss, "steps_repeated_with_fp32_reverse_recurrence": step.fallbacks,
            "other_kernels": other,
            "other_flows": flows,
            "other_batch_sizes": small,
            "whole_job": {"algorithmic_flop_per_frame": flop, "tflops": tf, "frac_of_f32_mfma_peak": tf / (PEAK_F32_MFMA_TFLOPS * world)},
            # no single kernel dominates a training step (forward recurrences, reverse recurrences, weight-gradient GEMMs):
            # the headline roofline figure is the WHOLE step's algorithmic fp32 work (forward + dgrad + wgrad = 282.3 MFLOP per frame at
            # hu1024 cyc2) against the fp32-input MFMA peak; `kernels` carries the per-kernel figures of the three dominant ones
            "roofline": {"bound": "mfma", "achieved": tf / world, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": tf / (PEAK_F32_MFMA_TFLOPS * world), "traffic": None,
                         "kernel": "whole stage-4 step (all kernels + host glue), wall-clocked",
                         "algorithmic_flop_per_step_per_gpu": flop * B * T,
                         "kernels": kernels},
            "cpu_baseline": None}
        if world == 1 and not args.no_cpu_baseline:
            res.update(train_check_and_cpu_baseline(P, W, B, T, L, NCYC, H, dev, mod, tt, stress))
        return res
    finally:
        set_kernel(0)
        gru_vae.set_draw_origin(0, 0, 0)


def train_kernel_rooflines(args, lib, step, data, B, T, H, NCYC, stress):
    """Per-kernel roofline of the training step's dominant kernels: two extra (untimed) steps in which the library brackets every
    launch of the training recurrences and every training GEMM with HIP events on the stream it is launched on (option
    train_profile; cvae_train_profile_collect sums durations, launches and GEMM flops per kernel class).  achieved = ALGORITHMIC
    flops of those launches / their summed durations; traffic = fabric-side bytes per launch from the committed rocprofv3 --pmc
    passes over this leg (profiles/traffic_train.json), null where there is none."""
    NPROF = 2
    lib.set_option("train_profile", 1)
    try:
        lib.train_profile_collect()
        for _ in range(NPROF):
            step(*data)
        torch.cuda.synchronize()
        prof = lib.train_profile_collect()
    finally:
        lib.set_option("train_profile", 0)
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_train.json"))) if (B == 64 and T == 80 and not stress) else {}
    except (OSError, ValueError):
        tj = {}
    # algorithmic MACs per frame and pass of a recurrence: W_hh.h (3H x H) + the feedback W_ih[:, 9C:].y (3H x Cout) + out_1 inside
    # the loop (Cout x H); the reverse recurrence carries the same products transposed
    co = {True: 2 * (64 if stress else 32), False: 50}
    mac_rec = lambda enc: 3 * H * H + 3 * H * co[enc] + co[enc] * H
    n_pass = {True: 2 * NCYC, False: 3 * NCYC}
    flop_rec = 2.0 * B * T * sum(n_pass[e] * mac_rec(e) for e in (True, False))     # one step's forward (= reverse) recurrences
    names = {"fwd_recurrence": "k_train_fwd_steps_x3h<8,4> (64-row passes) + k_train_fwd_steps_x3<16> (stacked 128-row passes)",
             "bwd_recurrence": "k_train_bwd_steps_x3<32>", "forward_and_dgrad_gemms": "k_gemm_nt2<TM,TN> (+ split-contraction sums)",
             "wgrad_gemms": "k_gemm_tn2<TM,TN> (+ split-contraction sums), side stream"}
    if stress:
        names["fwd_recurrence"], names["bwd_recurrence"] = "T x k_gru_step_train (any-H path)", "T x (k_gru_step_bwd + k_bwd_step_gemm)"
    out = {}
    for name, (ms, n, fl) in prof.items():
        if n <= 0 or ms <= 0:
            continue
        flop = flop_rec * NPROF if name.endswith("recurrence") else fl
        ach = flop / (ms * 1e-3) / 1e12
        t = tj.get(name, {})
        out[name] = {"kernel": names[name], "bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": ach / PEAK_F32_MFMA_TFLOPS, "launches_per_step": n / float(NPROF), "avg_launch_ms": ms / n,
                     "kernel_ms_per_step": ms / NPROF, "algorithmic_flop_per_step": flop / NPROF,
                     "traffic": t.get("bytes_per_launch"), "traffic_is": t.get("what"),
                     "timed_by": "HIP events on the launch stream around every launch, %d untimed steps (cvae_train_profile_collect); "
                                 "kernels of different classes overlap across the two streams, so the classes do not add up to the step"
                                 % NPROF}
    return out


def train_check_and_cpu_baseline(P, W, B, T, L, NCYC, H, dev, mod, tt, stress=False):
    """The same step on the host cores -- stock-torch autograd through the checker's train-mode pass (oracle/torch_stock.py) -- and
    the GPU step checked against it at the TIMED geometry.

    hu1024: the checker runs the step on ALL B utterances of the bench batch (identical dropout masks and eps on both sides): its
    loss and the eval-mode trajectories AFTER the update are what the GPU step is checked against (`loss_check`,
    `mcd_db_vs_cpu_after_step`); `cpu_baseline` is timed on 8 utterances of the batch.
    stress (hu2048 / ld64 / cyc4): the GPU runs all B rows through the timed kernels with THREE rows selected for the loss
    (select_utt_idx, the generator's own mechanism, train...:1363: the others are computed and ignored), the checker runs those three
    rows; the same three-row step is the timed CPU sample."""
    import gru_vae
    import stage4
    from oracle import cyclevae_oracle as orc
    from oracle import torch_stock as ts
    ncpu = os.cpu_count() or 1
    thr = min(ncpu, 16)
    torch.set_num_threads(thr)
    rows = [0, min(13, B - 1), B - 1][:min(3, B)] if stress else list(range(B))
    rows = sorted(set(rows))
    nb_time = len(rows) if stress else min(B, 8)
    cin_e, cout_e, cin_d = 54, 2 * L, 2 + L
    c = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    gen = torch.Generator().manual_seed(1)
    mk = lambda shape: (torch.rand(shape, generator=gen) >= 0.5).float() * 2.0
    masks = {"enc": [(mk((B, T, 9 * cin_e)), mk((T, B, H))) for _ in range(2 * NCYC)],
             "dec": [(mk((B, T, 9 * cin_d)), mk((T, B, H))) for _ in range(3 * NCYC)]}
    sub = lambda r: {k: [(a[r].contiguous(), b[:, r].contiguous()) for a, b in v] for k, v in masks.items()}
    inp = lambda r: [c(P.x[r]), c(P.cvx[r]), c(P.code_src[r]), c(P.code_trg[r]), c(P.y_in_enc[r]), c(P.y_in_dec[r]), c(P.eps[:, :, r])]

    def fresh():
        leaf = {k: {n: torch.from_numpy(v.copy()).requires_grad_(n in stage4.TRAINABLE) for n, v in sd.items()}
                for k, sd in (("enc", W.enc), ("dec", W.dec))}
        return leaf, torch.optim.Adam([leaf[k][n] for k in leaf for n in stage4.TRAINABLE], lr=1e-4)

    def cpu_step(leaf, opt, cin, msk):
        t1 = time.perf_counter()
        opt.zero_grad()
        l_ = stage4.chain_loss(lambda kind, xin, y_in, clamp, m_: ts.train_forward_t(leaf[kind], xin, y_in, m_[0], m_[1], clamp),
                               *cin, L, NCYC, msk)
        l_.backward()
        opt.step()
        return time.perf_counter() - t1, float(l_.item())

    leaf, opt = fresh()
    chk_in, chk_masks = inp(rows), sub(rows)
    t_first, cpu_loss = cpu_step(leaf, opt, chk_in, chk_masks)      # first step from the initial weights: the one the GPU is checked against
    after = {k: {n: v.detach().numpy() for n, v in leaf[k].items()} for k in leaf}
    ce, cd = ts.StockGRURNN(after["enc"], cin_e, cout_e, H), ts.StockGRURNN(after["dec"], cin_d, 50, H)
    ev_rows = rows if stress else rows[:32]
    cpu_eval = ts.cycle_chain(ce, cd, *inp(ev_rows), NCYC, L)
    if stress:
        tc = cpu_step(leaf, opt, chk_in, chk_masks)[0] if t_first < 60.0 else t_first
        n_timed = 1
    else:
        tr = list(range(nb_time))
        leaf2, opt2 = fresh()
        tin, tmasks = inp(tr), sub(tr)
        cpu_step(leaf2, opt2, tin, tmasks)
        tc = sorted(cpu_step(leaf2, opt2, tin, tmasks)[0] for _ in range(3))[1]
        n_timed = 3
    # the GPU side of the check: fresh modules, the whole bench batch at the timed geometry, masks and eps injected
    enc, dec = mod(W.enc, cin_e, cout_e, True), mod(W.dec, cin_d, 50, False)
    step = stage4.Stage4Step(enc, dec, lat_dim=L, n_cyc=NCYC, lr=1e-4)
    gmasks = {k: [(a.to(dev), b.to(dev)) for a, b in v] for k, v in masks.items()}
    gin = [v.to(dev) for v in inp(list(range(B)))]
    gpu_loss = float(step(*gin, masks=gmasks, select_utt_idx=rows if stress else None).item())
    del gmasks
    enc.eval(); dec.eval()
    ein = [v.to(dev) for v in inp(ev_rows)]
    with torch.no_grad():
        g = gru_vae.CycleChain(enc, dec, lat_dim=L, n_cyc=NCYC)(*ein[:6], eps=ein[6])
    mcd = {}
    for k in ("rec", "cv", "reccyc"):
        a = g[k].cpu().numpy().reshape(-1, 50)
        b = np.stack([v.numpy() for v in cpu_eval[k]]).reshape(-1, 50)
        mcd[k] = float(np.mean(orc.mcd_frames(a, b)))
    log("train leg check (%d utterances in the loss, %d rows on the GPU): loss gpu %.6f cpu %.6f, post-step MCD %.2e dB"
        % (len(rows), B, gpu_loss, cpu_loss, max(mcd.values())))
    return {"cpu_baseline": {"value": nb_time * T / tc, "unit": "frames/s", "cores": thr, "kind": "port",
                             "sample": "the same step (forward, loss, backward, Adam) on %d utterances x %d frames of the bench batch, "
                                       "stock-torch autograd through oracle/torch_stock.py, fp32, %d threads (host has %d logical "
                                       "cpus), %s after 1 warm-up" % (nb_time, T, thr, ncpu, "median of 3" if n_timed == 3 else "one step"),
                             "ms_per_step": 1e3 * tc},
            "loss_check": {"utterances_in_the_loss": len(rows), "rows_through_the_gpu_kernels": B, "gpu": gpu_loss, "cpu": cpu_loss,
                           "rel_diff": abs(gpu_loss - cpu_loss) / abs(cpu_loss),
                           "what": "loss of the first step from the initial weights at the TIMED geometry, identical dropout masks and eps on both "
                                   "sides" + ("; the GPU step runs all %d rows and selects rows %s for the loss (select_utt_idx), the checker "
                                              "runs those rows" % (B, rows) if stress else "")},
            "mcd_db_vs_cpu_after_step": {"utterances": len(ev_rows), "per_output": mcd, "max": max(mcd.values()), "budget": 0.01,
                                         "what": "eval-mode cyc%d chain with the weights AFTER that step (GPU: cvae_adam_step_counted, CPU: "
                                                 "torch.optim.Adam), same eps" % NCYC}}


if __name__ == "__main__":
    main()
