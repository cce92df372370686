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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("eval", "train"), default="eval",
                    help="eval: the headline cyc2 eval chain (BASELINE configs[1]); train: one stage-4 step (configs[2])")
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="utterance rows per GPU (default 64 eval, 8 train)")
    ap.add_argument("--frames", type=int, default=80)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sub-paths", action="store_true", help="skip the conversion-only / single-utterance extras (profiling runs)")
    ap.add_argument("--no-persistent", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: re-run this very command as N ranks (one process per GPU,
        # RCCL rendezvous on 127.0.0.1) and pass rank 0's JSON line through
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "--gpus %d but the launcher started %d ranks" % (args.gpus, world)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback for the product path)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if args.no_persistent:
        os.environ["CYCLEVAE_NO_PERSISTENT"] = "1"

    import _cabi
    import gru_vae
    import synth

    if args.batch_per_gpu is None:
        args.batch_per_gpu = 64 if args.mode == "eval" else 8
    if args.mode == "train":
        return bench_train(args, world, rank, dev)
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
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    inputs = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    lib = gru_vae._lib()

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            chain(*inputs, seed=1234)
        sync_all()
        lib.profile_collect()
        # ---- timed region: exactly K steps; HIP events (recorded by the library on this stream) bracket every
        # launch of the dominant kernel inside the same region
        flags_env = os.environ.get("CYCLEVAE_PROFILE", "1") != "0"
        if flags_env:
            gru_vae._flags_extra = _cabi.FLAG_PROFILE
        t0 = time.perf_counter()
        for k in range(args.steps):
            chain(*inputs, seed=1000 + k)
        sync_all()
        dt = time.perf_counter() - t0
        gru_vae._flags_extra = 0
    kern_ms, kern_n = lib.profile_collect()
    assert chain.status()[0] == 0, "grid barrier timed out during the bench"
    # ---- the same K steps once more on the all-fp32 MFMA kernel (k_gru_steps_v4), reported next to the headline
    with torch.no_grad():
        gru_vae._force_fp32_mfma = True
        for _ in range(max(1, args.warmup)):
            chain(*inputs, seed=1234)
        sync_all()
        if flags_env:
            gru_vae._flags_extra = _cabi.FLAG_PROFILE
        t1 = time.perf_counter()
        for k in range(args.steps):
            chain(*inputs, seed=1000 + k)
        sync_all()
        dt32 = time.perf_counter() - t1
        gru_vae._flags_extra = 0
        gru_vae._force_fp32_mfma = False
    kern32_ms, kern32_n = lib.profile_collect()

    if world > 1:
        import shard
        dt = shard.max_over_ranks(dt, dist, dev)
        dt32 = shard.max_over_ranks(dt32, dist, dev)
    frames_per_step = B * T * world
    value = frames_per_step * args.steps / dt

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    res = {
        "metric": "mcep_frames_per_sec_hu1024_ld32_cyc2", "value": value, "unit": "frames/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "data": "synthetic",
        "dtype": "f32 (GEMM operands of the recurrent kernel as fp16 pairs hi + lo/2048 = 22 bits, three f16 MFMAs per product, "
                 "f32 accumulate; gates, carried state, projection and outputs f32)",
        "config": {"workload": "cyc2 eval chain: 4 encoder + 6 decoder GRU_RNN passes over x[B,T,54] (BASELINE configs[1])",
                   "batch_per_gpu": B, "frames": T, "hidden_units": 1024, "lat_dim": L, "n_cyc": NCYC,
                   "latent_draws": "on-device Philox", "sharding": "batch rows, %d/GPU, no collective" % B,
                   "recurrence": "per-step launches" if args.no_persistent else "one cooperative launch per pass"},
        "whole_job": {"algorithmic_flop_per_frame": 2 * (NCYC * 2 * MAC_ENC + NCYC * 3 * MAC_DEC),
                      "tflops": value * 2 * (NCYC * 2 * MAC_ENC + NCYC * 3 * MAC_DEC) / 1e12,
                      "frac_of_f32_mfma_peak": value * 2 * (NCYC * 2 * MAC_ENC + NCYC * 3 * MAC_DEC) / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)},
    }
    # ---- roofline of the dominant kernel (k_gru_steps: the T-step recurrence of one pass, one launch per pass)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("k_gru_steps_hbm_bytes_per_launch")
        except Exception:
            traffic = None
    if kern_n > 0 and kern_ms > 0:
        # launches per step: 8 on the persistent path (4 encoder passes, 2 single decoder passes, 2 launches that run
        # rec||cv stacked over 2B rows), 10 otherwise; achieved = algorithmic flops of all launches / their summed time
        launches_per_step = kern_n / float(args.steps)
        flop_per_step = 2.0 * B * T * (NCYC * 2 * MAC_KERN_ENC + NCYC * 3 * MAC_KERN_DEC)
        avg_ms = kern_ms / kern_n
        ach = (flop_per_step / launches_per_step) / (avg_ms * 1e-3) / 1e12
        # executed MFMA work of the split kernel per (row tile, step, block), f16 16x16x32 instructions per wave: 96 for the
        # recurrent product (three per 32 k and column tile) + 36 (encoder) / 27 (decoder) for the front-end; x 4 waves,
        # 64 blocks per row tile, 4 encoder + 6 decoder passes per chain.  No fp32 MFMA is left in this kernel.
        tiles = (B + 15) // 16
        exec_f32 = 0.0
        exec_f16 = 2.0 * 16 * 16 * 32 * 4 * T * 64 * tiles * (4 * (96 + 36) + 6 * (96 + 27))
        res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                           "kernel": "k_gru_steps_v5 (front-end + T-step recurrence of one pass, one cooperative launch; "
                                     "recurrent product as split-fp16 MFMA)",
                           "peak_is": "dense fp32 MFMA, the governing roofline of SURVEY 8(d); achieved = ALGORITHMIC fp32 flops / time",
                           "avg_launch_ms": avg_ms, "launches_timed": kern_n, "launches_per_step": launches_per_step,
                           "share_of_step_time": kern_ms / (1e3 * dt) if world == 1 else None,
                           "algorithmic_flop_per_launch": flop_per_step / launches_per_step,
                           "executed": {"f16_mfma_tflops": exec_f16 / launches_per_step / (avg_ms * 1e-3) / 1e12,
                                        "f16_dense_peak_tflops": 2500.0,
                                        "f32_mfma_tflops": exec_f32 / launches_per_step / (avg_ms * 1e-3) / 1e12}}
        if kern32_n > 0 and kern32_ms > 0:
            avg32 = kern32_ms / kern32_n
            ach32 = (flop_per_step / (kern32_n / float(args.steps))) / (avg32 * 1e-3) / 1e12
            res["all_fp32_mfma_path"] = {"value": frames_per_step * args.steps / dt32, "unit": "frames/s",
                                         "ms_per_step": 1e3 * dt32 / args.steps, "kernel": "k_gru_steps_v4 (CYCLEVAE_FP32_MFMA=1)",
                                         "roofline_achieved": ach32, "roofline_frac": ach32 / PEAK_F32_MFMA_TFLOPS,
                                         "avg_launch_ms": avg32}
    else:
        res["roofline"] = None

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
        res["sub_paths"] = {"conversion_only_B%dxT%d" % (B, T): {"frames_per_s": B * T / tc, "ms": 1e3 * tc, "passes": "1 encoder + 1 decoder"},
                            "single_utterance_T637_300draws": {"frames_per_s": 637 / tu, "ms": 1e3 * tu,
                                                               "passes": "1 encoder + 1 decoder at B=1 (per-step hand-off latency bound)"}}

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
        nrow = 4
        with torch.no_grad():
            g = chain(*[v[:nrow] for v in inputs], eps=tt(P.eps[:, :, :nrow]))
        r = cpu_chain(nrow, T)[0]
        mcd = {}
        for k in ("rec", "cv", "reccyc"):
            a = g[k].cpu().numpy().reshape(-1, 50)
            b = np.stack([v.numpy() for v in r[k]]).reshape(-1, 50)
            mcd[k] = [float(np.mean(orc.mcd_frames(a, b))), float(np.mean(orc.mcd_frames(a[:, 1:], b[:, 1:])))]
        res["mcd_db_vs_cpu"] = {"rows": nrow, "per_output_dims0_49_and_1_49": mcd,
                                "max": max(max(v) for v in mcd.values()), "budget": 0.01}
        log("mcd vs cpu: %s" % res["mcd_db_vs_cpu"]["max"])
        if res.get("all_fp32_mfma_path") is not None:
            gru_vae._force_fp32_mfma = True
            with torch.no_grad():
                g32 = chain(*[v[:nrow] for v in inputs], eps=tt(P.eps[:, :, :nrow]))
            gru_vae._force_fp32_mfma = False
            m32 = 0.0
            for k in ("rec", "cv", "reccyc"):
                a = g32[k].cpu().numpy().reshape(-1, 50)
                b = np.stack([v.numpy() for v in r[k]]).reshape(-1, 50)
                m32 = max(m32, float(np.mean(orc.mcd_frames(a, b))), float(np.mean(orc.mcd_frames(a[:, 1:], b[:, 1:]))))
            res["all_fp32_mfma_path"]["mcd_db_vs_cpu_max"] = m32
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
    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def bench_train(args, world, rank, dev):
    """One step = the stage-4 step (cyc2 chain in train mode with dropout 0.5, loss, backward, gradient all-reduce when N > 1,
    torch.optim.Adam) on a fresh 80-frame window of B utterances per GPU (reference train_gru_cyclevae_gauss_batch.py:1326-1420)."""
    import torch.distributed as dist
    import gru_vae
    import stage4
    import synth

    B, T, L, NCYC = args.batch_per_gpu, args.frames, 32, 2
    P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="trainbench/rank%d" % rank)
    W = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="trainbench/rank0")

    def mod(sd, i, o, enc):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, kernel_size=3, dilation_size=2, do_prob=0.5,
                            scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev).train()

    step = stage4.Stage4Step(mod(W.enc, 54, 64, True), mod(W.dec, 34, 50, False), lat_dim=L, n_cyc=NCYC, lr=1e-4,
                             dist=dist if world > 1 else None)
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    data = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec", "eps")]

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(*data)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(*data)
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        import shard
        dt = shard.max_over_ranks(dt, dist, dev)
    if rank == 0:
        value = B * T * world * args.steps / dt
        flop = 3.0 * 2 * (NCYC * 2 * MAC_ENC + NCYC * 3 * MAC_DEC)     # forward + dgrad + wgrad
        print(json.dumps({
            "metric": "stage4_train_frames_per_sec_hu1024_ld32_cyc2", "value": value, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "data": "synthetic",
            "dtype": "f32 (forward recurrence: GEMM operands as fp16 pairs, 22 bits, f32 accumulate; all other GEMMs fp32 MFMA)",
            "config": {"workload": "stage-4 step: cyc2 chain (train mode, dropout 0.5) + loss + backward + Adam (BASELINE configs[2])",
                       "utterances_per_gpu": B, "frames": T, "hidden_units": 1024, "lat_dim": L, "n_cyc": NCYC,
                       "gradient_allreduce": "one flat fp32 bucket per step (RCCL)" if world > 1 else "none (1 GPU)"},
            "final_loss": float(loss.item()),
            "whole_job": {"algorithmic_flop_per_frame": flop, "tflops": value * flop / 1e12,
                          "frac_of_f32_mfma_peak": value * flop / 1e12 / (PEAK_F32_MFMA_TFLOPS * world)},
            "roofline": None, "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
