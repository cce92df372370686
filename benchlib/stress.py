"""bench.py --config stress: the eval chain at the dims of BASELINE configs[4] (hu2048 / ld64 / n_cyc = 4)."""
import json
import os
import time

import numpy as np
import torch

from .report import MAC_DEC, MAC_ENC, MAC_KERN_DEC, MAC_KERN_ENC, PEAK_F32_MFMA_TFLOPS, emit, log


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
    emit(res)
