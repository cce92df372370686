"""The stage-4 training-step leg (BASELINE configs[2]; --config stress: the dims of configs[4]): timing, per-class rooflines, the other
flows (unfused / unchanged script), the in-run loss check and the CPU baseline."""
import json
import os
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .report import MAC_DEC, MAC_ENC, MAC_KERN_DEC, MAC_KERN_ENC, PEAK_F32_MFMA_TFLOPS, emit, log


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
    # the same two steps with NOTHING beside the recurrences (weight gradients on the launch stream, behind each pass): what the two
    # recurrence kernels take when they have the chip to themselves -- in the step the side stream's GEMMs share their matrix pipe
    alone = {}
    if getattr(step, "overlap_wgrad", False):
        import gru_vae
        prev = gru_vae.set_backward_overlap(False)
        step.overlap_wgrad = False
        lib.set_option("train_profile", 1)
        try:
            step(*data)                      # (first step of the other flow: allocations)
            torch.cuda.synchronize()
            lib.train_profile_collect()
            for _ in range(NPROF):
                step(*data)
            torch.cuda.synchronize()
            alone = lib.train_profile_collect()
        finally:
            lib.set_option("train_profile", 0)
            step.overlap_wgrad = True
            gru_vae.set_backward_overlap(prev)
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
    names = {"fwd_recurrence": "k_train_fwd_steps_w3<8,4,11> (64-row passes: one tile per block behind a first-poll back-off; stacked 128-row passes: two tiles per block)",
             "bwd_recurrence": "k_train_bwd_steps_w3<8,4,12> (two tiles per block: 64-row passes on half the chip, stacked 128-row passes on all of it)", "forward_and_dgrad_gemms": "k_gemm_nt2<TM,TN> (+ split-contraction sums)",
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
                     **({"kernel_ms_per_step_with_nothing_beside_it": alone[name][0] / NPROF,
                         "frac_with_nothing_beside_it": flop / (alone[name][0] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}
                        if name.endswith("recurrence") and name in alone and alone[name][0] > 0 else {}),
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
