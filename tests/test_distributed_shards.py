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
