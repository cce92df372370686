"""World-size-2 gloo test of the N>1 path on CPU: rows are sharded with shard_rows, each rank runs the (emulated)
library on its shard only, and the gathered result equals the single-process result on the whole batch -- i.e. the
path needs no data-path collective.  Also checks the max-over-ranks timing reduction bench.py uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import shard
import synth
from oracle import cyclevae_oracle as orc


def test_shard_rows_partition():
    for n in (1, 2, 7, 64, 512):
        for world in (1, 2, 3, 8):
            blocks = [shard.shard_rows(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) - min(sizes) <= 1
            assert sizes == [len(c) for c in np.array_split(np.arange(n), world)]


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emu_util import NpNet, emu_lib, ptr
    lib = emu_lib()
    P = synth.CycleVAEProblem(B=5, T=6, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="dist")
    lo, hi = shard.shard_rows(5, world, rank)
    enc, dec = NpNet(lib, P.enc, 6, 8, 32), NpNet(lib, P.dec, 6, 4, 32)
    B, T = hi - lo, 6
    c = lambda a: np.ascontiguousarray(a[lo:hi])
    x, cvx, cs, ct = c(P.x), c(P.cvx), c(P.code_src), c(P.code_trg)
    ye, yd = c(P.y_in_enc).reshape(B, 8), c(P.y_in_dec).reshape(B, 4)
    eps = np.ascontiguousarray(P.eps[:, :, lo:hi])
    rec = np.zeros((2, B, T, 4), np.float32)
    ws = np.zeros(lib.cycle_workspace_bytes(enc.d, dec.d, B, T, 2) // 4, np.float32)
    lib.cycle_forward(enc.d, ptr(enc.prepared), dec.d, ptr(dec.prepared), ptr(x), ptr(cvx), 2, ptr(cs), ptr(ct), 2,
                      ptr(ye), ptr(yd), B, T, 2, 4, ptr(eps), 0, None, None, None, None, ptr(rec), ptr(ws), ws.nbytes, 0)
    # gather only to CHECK; the product path never exchanges rows
    parts = [None] * world
    dist.all_gather_object(parts, (lo, hi, rec))
    slow = shard.max_over_ranks(0.25 * (rank + 1), dist)
    if rank == 0:
        full = np.concatenate([p[2] for p in sorted(parts, key=lambda p: p[0])], axis=1)
        np.save(out_path, full)
        assert abs(slow - 0.25 * world) < 1e-12
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_chain_equals_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "rec.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    P = synth.CycleVAEProblem(B=5, T=6, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="dist")
    ref = orc.cycle_chain(P.enc, P.dec, P.x, P.cvx, P.code_src, P.code_trg, P.y_in_enc, P.y_in_dec, P.eps, 2, 4)
    assert got.shape == (2, 5, 6, 4)
    assert float(np.max(np.abs(got - np.stack(ref["reccyc"])))) <= 3e-4


def _grad_worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import torch_stock as ts
    from train_util import TRAINABLE
    P = synth.CycleVAEProblem(B=4, T=6, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=1, bias_scale=0.1, tag="dpgrad")
    lo, hi = shard.shard_rows(4, world, rank)
    leaf = {n: torch.from_numpy(v.copy()).requires_grad_(n in TRAINABLE) for n, v in P.enc.items()}
    ones_c = torch.ones(hi - lo, 6, 54)
    ones_g = torch.ones(6, hi - lo, 32)
    out = ts.train_forward_t(leaf, torch.from_numpy(P.x[lo:hi]), torch.from_numpy(P.y_in_enc[lo:hi]), ones_c, ones_g, 4)
    out.sum().backward()                       # SUM loss over this rank's rows
    params = [leaf[n] for n in TRAINABLE]
    n = shard.allreduce_gradients(params, dist)
    # the copy-free form the stage-4 step uses: gradients accumulate INTO views of one flat buffer, which is all-reduced as is
    leaf2 = {k: torch.from_numpy(v.copy()).requires_grad_(k in TRAINABLE) for k, v in P.enc.items()}
    fg = shard.FlatGradients([leaf2[k] for k in TRAINABLE])
    for _ in range(2):                         # a second step must start from zeros again and still write through the views
        fg.zero()
        ts.train_forward_t(leaf2, torch.from_numpy(P.x[lo:hi]), torch.from_numpy(P.y_in_enc[lo:hi]), ones_c, ones_g, 4).sum().backward()
    assert all(leaf2[k].grad.data_ptr() >= fg.flat.data_ptr() for k in TRAINABLE)
    assert fg.allreduce(dist) == n
    for k in TRAINABLE:
        assert torch.allclose(leaf2[k].grad, leaf[k].grad, rtol=1e-5, atol=1e-6), k
    if rank == 0:
        np.savez(out_path, n=n, **{k.replace(".", "_"): leaf[k].grad.numpy() for k in TRAINABLE})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gradient_allreduce_equals_full_batch(tmp_path):
    """Sum-loss gradients all-reduced over 2 ranks == gradients of the whole batch on one process."""
    from oracle import torch_stock as ts
    from train_util import TRAINABLE
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "g.npz")
    mp.spawn(_grad_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    P = synth.CycleVAEProblem(B=4, T=6, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=1, bias_scale=0.1, tag="dpgrad")
    leaf = {n: torch.from_numpy(v.copy()).requires_grad_(n in TRAINABLE) for n, v in P.enc.items()}
    ts.train_forward_t(leaf, torch.from_numpy(P.x), torch.from_numpy(P.y_in_enc), torch.ones(4, 6, 54), torch.ones(6, 4, 32), 4).sum().backward()
    assert int(got["n"]) == sum(leaf[k].numel() for k in TRAINABLE)
    for k in TRAINABLE:
        ref = leaf[k].grad.numpy()
        assert float(np.max(np.abs(got[k.replace(".", "_")] - ref))) <= 1e-4 * max(1.0, float(np.abs(ref).max())), k


def _product_grad_worker(rank, world, port, out_path):
    """Each rank: the PRODUCT's train-mode pass and its backward (host build of the library) on its shard of the rows, dropout masks
    drawn on "device" by Philox keyed by GLOBAL row (cvae_set_draw_origin), parameter gradients accumulated into views of the flat
    buffer, one all-reduce."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import _cabi
    from emu_util import emu_lib, ptr
    from train_util import TRAINABLE
    lib = emu_lib()
    Bg, T = 6, 5
    P = synth.CycleVAEProblem(B=Bg, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="dpprod")
    lo, hi = shard.shard_rows(Bg, world, rank)
    B = hi - lo
    sd = {k: np.ascontiguousarray(v, np.float32) for k, v in P.enc.items()}
    d = lib.desc(6, 8, 64, 3, 2, True, False)
    image = np.zeros(lib.train_image_bytes(d) // 4, np.float32)
    lib.net_prepare_train(d, {f: ptr(sd[k]) for f, k in _cabi.STATE_KEYS.items() if k in sd}, ptr(image), image.nbytes, gru_drop_p=0.5)
    params = [torch.nn.Parameter(torch.from_numpy(sd[k].copy())) for k in TRAINABLE]
    fg = shard.FlatGradients(params)
    fields = ("conv0_w", "conv0_b", "conv1_w", "conv1_b", "w_ih", "w_hh", "b_ih", "b_hh", "out_w", "out_b")   # order of TRAINABLE
    x = np.ascontiguousarray(P.x[lo:hi], np.float32)
    y_in = np.ascontiguousarray(P.y_in_enc[lo:hi].reshape(B, 8), np.float32)
    cot = np.ascontiguousarray(synth.normal("dpprod/cot", (Bg, T, 8))[lo:hi], np.float32)
    out, yl, hl = np.zeros((B, T, 8), np.float32), np.zeros((B, 8), np.float32), np.zeros((B, 64), np.float32)
    tape = np.zeros(lib.train_tape_bytes(d, B, T) // 4, np.float32)
    scr = np.zeros(lib.train_scratch_bytes(d, B, T) // 4, np.float32)
    lib.set_draw_origin(lo, Bg, T)
    try:
        lib.forward_train(d, ptr(image), ptr(x), ptr(y_in), None, B, T, 4, None, None, 4242, 0.5, ptr(out), ptr(yl), ptr(hl),
                          ptr(tape), tape.nbytes, ptr(scr), scr.nbytes)
        fg.zero()
        lib.backward(d, ptr(image), ptr(cot), B, T, 4, ptr(tape), ptr(scr), scr.nbytes, None,
                     {f: p.grad.data_ptr() for f, p in zip(fields, params)}, accumulate=True)
    finally:
        lib.set_draw_origin(0, 0, 0)
    n = fg.allreduce(dist)
    parts = [None] * world
    dist.all_gather_object(parts, (lo, out))
    if rank == 0:
        np.savez(out_path, n=n, flat=fg.flat.numpy(), out=np.concatenate([p[1] for p in sorted(parts, key=lambda p: p[0])], 0))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_product_gradients_equal_the_full_batch(tmp_path):
    """Data-parallel training as the product runs it: two ranks, each with the library's own train-mode forward / backward on its
    rows, Philox dropout masks keyed by global row, gradients summed by ONE all-reduce of the flat buffer == the gradients (and
    outputs) of a single process holding all rows.  N-invariance of the masks is what makes the two runs the same function."""
    import _cabi
    from emu_util import emu_lib, ptr
    from train_util import TRAINABLE
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out_path = str(tmp_path / "pg.npz")
    mp.spawn(_product_grad_worker, args=(2, port, out_path), nprocs=2, join=True)
    got = np.load(out_path)
    lib = emu_lib()
    Bg, T = 6, 5
    P = synth.CycleVAEProblem(B=Bg, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="dpprod")
    sd = {k: np.ascontiguousarray(v, np.float32) for k, v in P.enc.items()}
    d = lib.desc(6, 8, 64, 3, 2, True, False)
    image = np.zeros(lib.train_image_bytes(d) // 4, np.float32)
    lib.net_prepare_train(d, {f: ptr(sd[k]) for f, k in _cabi.STATE_KEYS.items() if k in sd}, ptr(image), image.nbytes, gru_drop_p=0.5)
    fields = ("conv0_w", "conv0_b", "conv1_w", "conv1_b", "w_ih", "w_hh", "b_ih", "b_hh", "out_w", "out_b")
    grads = {f: np.zeros(sd[k].shape, np.float32) for f, k in zip(fields, TRAINABLE)}
    x = np.ascontiguousarray(P.x, np.float32)
    y_in = np.ascontiguousarray(P.y_in_enc.reshape(Bg, 8), np.float32)
    cot = np.ascontiguousarray(synth.normal("dpprod/cot", (Bg, T, 8)), np.float32)
    out, yl, hl = np.zeros((Bg, T, 8), np.float32), np.zeros((Bg, 8), np.float32), np.zeros((Bg, 64), np.float32)
    tape = np.zeros(lib.train_tape_bytes(d, Bg, T) // 4, np.float32)
    scr = np.zeros(lib.train_scratch_bytes(d, Bg, T) // 4, np.float32)
    lib.forward_train(d, ptr(image), ptr(x), ptr(y_in), None, Bg, T, 4, None, None, 4242, 0.5, ptr(out), ptr(yl), ptr(hl),
                      ptr(tape), tape.nbytes, ptr(scr), scr.nbytes)
    lib.backward(d, ptr(image), ptr(cot), Bg, T, 4, ptr(tape), ptr(scr), scr.nbytes, None, {f: ptr(g) for f, g in grads.items()})
    ref = np.concatenate([grads[f].reshape(-1) for f in fields])
    assert int(got["n"]) == ref.size
    assert float(np.abs(got["out"] - out).max()) <= 1e-5          # same masks on whichever rank a row lands
    assert float(np.abs(got["flat"] - ref).max()) <= 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.timeout(600)
def test_bench_py_itself_on_two_emulator_ranks():
    """The REAL bench.py as `python bench.py --gpus 2` runs it on an 8-GPU node -- self-spawn under torch.distributed.run on 127.0.0.1
    (shard.spawn_ranks), world size checked against --gpus, process group, barrier, every rank's timed region, the MAX over ranks,
    the per-rank times, rank 0's single JSON line -- on two CPU ranks: CYCLEVAE_BENCH_BACKEND=emu puts the host-fiber build of the
    library behind the drop-in module and gloo in place of RCCL (tests/emu_bench_backend.py; device selection only).  The first
    multi-GPU lease has to produce the scaling curve with no code change; this is what can be checked without one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["CYCLEVAE_BENCH_BACKEND"] = "emu"
    import emu_util
    emu_util.build_emu()             # (once, before two ranks race for it)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-per-gpu", "2",
                        "--frames", "4", "--no-cpu-baseline", "--no-sub-paths", "--no-train-leg", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=560, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["steps"] == 2 and res["warmup"] == 1
    assert res["metric"] == "mcep_frames_per_sec_hu1024_ld32_cyc2" and "EMULATOR" in res["config"]["backend"]
    ranks = res["ranks"]
    assert ranks["world_size"] == 2 and ranks["backend"] == "gloo" and len(ranks["ms_per_step_per_rank"]) == 2
    assert all(v > 0 for v in ranks["ms_per_step_per_rank"]) and 0.0 < ranks["frac_of_linear"] <= 1.0 + 1e-9
    # value = frames of BOTH ranks over the slowest rank's time
    slow = max(ranks["ms_per_step_per_rank"])
    assert abs(res["value"] - 2 * 2 * 4 / (1e-3 * slow)) <= 0.05 * res["value"]
    assert abs(res["ms_per_step"] - slow) <= 0.05 * slow


def test_a_launcher_with_the_wrong_number_of_ranks_is_refused():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # a launcher that started the wrong number of ranks is refused
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r2 = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import shard; shard.launched_world(2)" %
                         os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "cyclevae-vc_amd")],
                        env=env2, capture_output=True, text=True)
    assert r2.returncode != 0 and "launcher started 3 ranks" in r2.stderr
