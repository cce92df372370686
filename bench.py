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

from benchlib.report import emit as _emit, flush_c_stdio as _flush_c_stdio  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--profile-every", type=int, default=5,
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
    ap.add_argument("--batch-sweep", default="", metavar="B,B,...",
                    help="eval mode, one GPU: also time the same chain at these batch sizes per GPU (sub_paths.batch_sweep: ms per chain and the "
                         "dominant kernel's roofline fraction per batch size), e.g. 64,128,256,512")
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
    # CYCLEVAE_BENCH_BACKEND=emu (tests only, tests/emu_bench_backend.py): the host-fiber build of the library behind the module and
    # gloo instead of RCCL, so that the CPU suite runs THIS file's N > 1 path on two ranks.  Device selection only: everything below
    # is the code the GPUs run.
    args.emu = os.environ.get("CYCLEVAE_BENCH_BACKEND") == "emu"
    if args.emu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import emu_bench_backend
        dev = emu_bench_backend.install()
        args.emu_dims = emu_bench_backend.DIMS
    else:
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
        dist.init_process_group("gloo" if args.emu else "nccl", rank=rank, world_size=world)
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
        from benchlib.train import train_leg
        res = train_leg(args, world, rank, dev, args.batch_per_gpu, args.steps, args.warmup, stress=args.config == "stress")
        if use_dist:
            dist.destroy_process_group()
        if rank == 0:
            _emit(res)
        return
    if args.config == "stress":
        from benchlib.stress import bench_stress
        return bench_stress(args, world, rank, dev)
    from benchlib.eval import eval_leg
    return eval_leg(args, world, rank, dev, use_dist)


if __name__ == "__main__":
    main()
