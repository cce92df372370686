"""The headline leg: mcep frames/s of the hu1024 / ld32 / cyc2 eval chain (BASELINE configs[1]), its roofline, the CPU baseline, the
sub-paths and -- appended as `train_step` -- the stage-4 training step (configs[2])."""
import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .report import MAC_DEC, MAC_ENC, MAC_KERN_DEC, MAC_KERN_ENC, PEAK_F32_MFMA_TFLOPS, emit, log



def eval_leg(args, world, rank, dev, use_dist):
    import _cabi
    import gru_vae
    import synth
    from .train import train_leg
    dist = None
    if use_dist:
        import torch.distributed as dist
    B, T, L, NCYC = args.batch_per_gpu, args.frames, 32, 2
    emu = getattr(args, "emu", False)
    kw = {}
    cin_e, cout_e, cin_d, cout_d, HID = 54, 64, 34, 50, 1024
    if emu:      # (tests/emu_bench_backend.py: a small model on the host-fiber build, only to run this file's N > 1 path without GPUs)
        d_ = args.emu_dims
        L, HID = d_["lat_dim"], d_["hidden"]
        kw = dict(in_dim=d_["in_dim"], out_dim=d_["out_dim"], lat_dim=L, hidden=HID)
        cin_e, cout_e, cin_d, cout_d = d_["in_dim"], 2 * L, 2 + L, d_["out_dim"]
    P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="bench/rank%d" % rank, **kw)
    W = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="bench/rank0", **kw)    # every rank holds the same weights

    def mod(sd, i, o, enc):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=HID, kernel_size=3, dilation_size=2,
                            scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev).eval()

    enc, dec = mod(W.enc, cin_e, cout_e, True), mod(W.dec, cin_d, cout_d, False)
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
    per_rank_ms = {}

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
            # every rank's own time next to the MAX the value is computed from (which rank was the slow one)
            per_rank_ms[kernel] = [1e3 * v / args.steps for v in shard.gather_over_ranks(dt_, dist, dev, force=args.force_dist)]
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
                   "batch_per_gpu": B, "frames": T, "hidden_units": HID, "lat_dim": L, "n_cyc": NCYC,
                   "latent_draws": "on-device Philox", "sharding": "batch rows, %d/GPU, no collective" % B,
                   "recurrence": "per-step launches" if args.no_persistent else "one launch per pass (every block resident, hand-over through flags)"},
        "whole_job": {"algorithmic_flop_per_frame": flop_frame, "tflops": value * flop_frame / 1e12,
                      "frac_of_f32_mfma_peak": value * flop_frame / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)},
    }
    if emu:
        res["config"]["backend"] = ("EMULATOR (tests/emu, host fibers) on a hidden-%d model: exercises this file's control flow and "
                                    "collectives on CPU ranks; the numbers say nothing about the product" % HID)
    if use_dist:
        # every rank's own ms per step of the headline leg (the value divides by the MAX); the backend the ranks rendezvoused on
        pr = per_rank_ms.get("exact3")
        res["ranks"] = {"world_size": world, "backend": dist.get_backend(), "ms_per_step_per_rank": pr,
                        "frames_per_rank_and_step": B * T,
                        # the job's rate (every rank waits for the slowest) over the sum of what each rank delivered on its own clock;
                        # NOT a scaling efficiency against the one-GPU run -- the driver computes that from the per-N values
                        "frac_of_linear": (world * B * T / (1e-3 * max(pr))) / sum(B * T / (1e-3 * v) for v in pr) if pr else None}
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
                               "command, profiles/traffic.json + the round's profiles/rNN_v6_pmc_*.md named in it; Infinity-Cache hits included): every XCD's L2 "
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

    # ---- single-GPU batch sweep of the headline chain: how the kernel's fraction moves with tiles per block
    if world == 1 and args.batch_sweep:
        sweep = {}
        for Bs in [int(v) for v in args.batch_sweep.split(",") if v.strip()]:
            Ps = synth.CycleVAEProblem(B=Bs, T=T, bias_scale=0.0, tag="bench/sweep%d" % Bs)
            ins = [tt(getattr(Ps, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
            nrep = 6
            with torch.no_grad():
                for _ in range(2):
                    chain(*ins, seed=77)
                torch.cuda.synchronize()
                lib.profile_collect()
                t2 = time.perf_counter()
                for k in range(nrep):
                    gru_vae._flags_extra = _cabi.FLAG_PROFILE if k % 2 == 0 else 0
                    chain(*ins, seed=2000 + k)
                torch.cuda.synchronize()
                dts = (time.perf_counter() - t2) / nrep
                gru_vae._flags_extra = 0
            ln = lib.profile_collect_launches()
            assert chain.status()[0] == 0, "a hand-off spin timed out during the batch sweep (B=%d)" % Bs
            kms = sum(l[0] for l in ln)
            # algorithmic flops of the profiled launches: per launch 2 * rows * T * MAC(front-end width)
            kfl = sum(2.0 * l[1] * T * (MAC_KERN_ENC if l[2] == 54 else MAC_KERN_DEC) for l in ln)
            fl = 2.0 * Bs * T * (NCYC * 2 * MAC_ENC + NCYC * 3 * MAC_DEC)
            sweep["B%d" % Bs] = {"ms_per_chain": 1e3 * dts, "frames_per_s": Bs * T / dts,
                                 "whole_chain_frac_of_f32_mfma_peak": fl / dts / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                 "kernel_frac": (kfl / (kms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS) if kms > 0 else None,
                                 "kernel_launches_timed": len(ln),
                                 "row_tiles_per_block_encoder_pass": max(1, Bs // 64)}
            del ins
        res.setdefault("sub_paths", {})["batch_sweep"] = {
            "what": "the same cyc2 eval chain at other batch sizes per GPU, %d frames; kernel_frac = algorithmic flops / HIP-event time of "
                    "k_gru_steps_v6's launches over the fp32-input MFMA peak (the headline's roofline.frac at B=64)" % T, **sweep}

    # ---- parity in the same run + CPU baseline (rank 0, N=1 only)
    if world == 1:
        from oracle import torch_stock as ts
        from oracle import cyclevae_oracle as orc
        ncpu = os.cpu_count() or 1
        log("gpu: %.0f frames/s, %.3f ms/step; host has %d logical cpus" % (value, 1e3 * dt / args.steps, ncpu))
        ce, cd = ts.StockGRURNN(W.enc, cin_e, cout_e, HID), ts.StockGRURNN(W.dec, cin_d, cout_d, HID)
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
                a = g[k].cpu().numpy().reshape(-1, cout_d)
                b = np.stack([v.numpy() for v in r[k]]).reshape(-1, cout_d)
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
    emit(res)
