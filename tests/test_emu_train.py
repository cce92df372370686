"""Training entry points (train-mode forward with dropout, BPTT backward, Adam) of the REAL library on the host-fiber
emulator, against gradients recorded from the reference (tests/golden/train_h*.npz: reference modules in train mode, the
dropout masks they drew captured by hooks, autograd gradients of sum(out * cotangent)).

Tolerance: fp32 re-association; gradients are sums over up to T*B terms, so they are compared relative to each tensor's
largest entry: |d| <= 1e-4 * max(1, max|ref|).
"""
import numpy as np
import pytest

import _cabi
import synth
from emu_util import emu_lib, ptr

GRAD_KEYS = {"conv0_w": "conv.conv.0.weight", "conv0_b": "conv.conv.0.bias", "conv1_w": "conv.conv.1.weight",
             "conv1_b": "conv.conv.1.bias", "w_ih": "gru.weight_ih_l0", "w_hh": "gru.weight_hh_l0", "b_ih": "gru.bias_ih_l0",
             "b_hh": "gru.bias_hh_l0", "out_w": "out_1.weight", "out_b": "out_1.bias"}


def rel_err(a, ref):
    assert a.shape == ref.shape, (a.shape, ref.shape)
    assert np.all(np.isfinite(a))
    return float(np.max(np.abs(a.astype(np.float64) - ref))) / max(1.0, float(np.max(np.abs(ref))))


class TrainNet(object):
    def __init__(self, lib, sd, i, o, h):
        self.lib = lib
        self.sd = {k: np.ascontiguousarray(v, np.float32) for k, v in sd.items()}
        self.d = lib.desc(i, o, h, 3, 2, "scale_in.weight" in sd, "scale_out.weight" in sd)
        self.image = np.zeros(lib.train_image_bytes(self.d) // 4, np.float32)
        lib.net_prepare_train(self.d, {f: ptr(self.sd[k]) for f, k in _cabi.STATE_KEYS.items() if k in self.sd},
                              ptr(self.image), self.image.nbytes, gru_drop_p=0.5)

    def run(self, x, y_in, h_in, cmask, gmask, cot, clamp, accumulate_into=None):
        lib, d = self.lib, self.d
        x = np.ascontiguousarray(x, np.float32)
        B, T, _ = x.shape
        Co, H = d.out_dim, d.hidden
        y_in = np.ascontiguousarray(y_in.reshape(B, Co), np.float32)
        h_in = None if h_in is None else np.ascontiguousarray(h_in.reshape(B, H), np.float32)
        cm = None if cmask is None else np.ascontiguousarray(cmask, np.float32)
        gm = None if gmask is None else np.ascontiguousarray(gmask, np.float32)
        out, yl, hl = np.full((B, T, Co), np.nan, np.float32), np.full((B, Co), np.nan, np.float32), np.full((B, H), np.nan, np.float32)
        tape = np.zeros(lib.train_tape_bytes(d, B, T) // 4, np.float32)
        scr = np.zeros(lib.train_scratch_bytes(d, B, T) // 4, np.float32)
        lib.forward_train(d, ptr(self.image), ptr(x), ptr(y_in), ptr(h_in), B, T, clamp, ptr(cm), ptr(gm), 11, 0.5, ptr(out),
                          ptr(yl), ptr(hl), ptr(tape), tape.nbytes, ptr(scr), scr.nbytes)
        grads = accumulate_into or {f: np.full(self.sd[k].shape, np.nan, np.float32) for f, k in GRAD_KEYS.items()}
        dx = np.full(x.shape, np.nan, np.float32)
        cot = np.ascontiguousarray(cot, np.float32)
        lib.backward(d, ptr(self.image), ptr(cot), B, T, clamp, ptr(tape), ptr(scr), scr.nbytes, ptr(dx),
                     {f: ptr(v) for f, v in grads.items()}, accumulate=accumulate_into is not None)
        return out, yl, hl, dx, grads


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.mark.parametrize("tag,hid,B,T", [("train_h32", 32, 3, 10), ("train_h64", 64, 18, 7)])
def test_train_pass_forward_and_backward_match_reference(lib, golden, tag, hid, B, T):
    g = golden(tag)
    P = synth.CycleVAEProblem(B=B, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=hid, n_cyc=1, bias_scale=0.1, tag=tag)
    x_dec = np.concatenate([P.code_src, synth.normal(tag + "/z", (B, T, 4))], 2)
    x2 = synth.features(tag + "/x2", B, T, P.mu, P.sigma)
    enc, dec = TrainNet(lib, P.enc, 6, 8, hid), TrainNet(lib, P.dec, 6, 4, hid)
    cases = [("enc", enc, P.x, P.y_in_enc, None, 4), ("dec", dec, x_dec, P.y_in_dec, None, -1),
             ("enc2", enc, x2, g["enc_y_last"], g["enc_h_last"], 4)]
    for name, net, x, y_in, h_in, clamp in cases:
        o = net.d.out_dim
        cot = synth.normal(tag + "/cot_" + name.replace("2", ""), (B, T, o))
        out, yl, hl, dx, grads = net.run(x, y_in, h_in, g[name + "_cmask"], g[name + "_gmask"], cot, clamp)
        assert rel_err(out, g[name + "_out"]) <= 5e-5, name
        assert rel_err(yl[:, None], g[name + "_y_last"]) <= 5e-5 and rel_err(hl[None], g[name + "_h_last"]) <= 5e-5
        assert rel_err(dx, g[name + "_dx"]) <= 1e-4, name
        for f, k in GRAD_KEYS.items():
            assert rel_err(grads[f], g[name + "_g_" + k]) <= 1e-4, (name, k)


def test_gradient_accumulation_and_philox_masks(lib, golden):
    g = golden("train_h32")
    P = synth.CycleVAEProblem(B=3, T=10, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=1, bias_scale=0.1, tag="train_h32")
    enc = TrainNet(lib, P.enc, 6, 8, 32)
    cot = synth.normal("train_h32/cot_enc", (3, 10, 8))
    _, _, _, _, g1 = enc.run(P.x, P.y_in_enc, None, g["enc_cmask"], g["enc_gmask"], cot, 4)
    acc = {f: v.copy() for f, v in g1.items()}
    enc.run(P.x, P.y_in_enc, None, g["enc_cmask"], g["enc_gmask"], cot, 4, accumulate_into=acc)
    for f in g1:
        assert rel_err(acc[f], 2.0 * g1[f].astype(np.float64)) <= 1e-5, f
    # library-drawn masks: deterministic in the seed, about half the units dropped, outputs finite
    o1 = enc.run(P.x, P.y_in_enc, None, None, None, cot, 4)[0]
    o2 = enc.run(P.x, P.y_in_enc, None, None, None, cot, 4)[0]
    assert np.array_equal(o1, o2) and np.all(np.isfinite(o1))
    assert rel_err(o1, g["enc_out"]) > 1e-3      # different masks than the reference drew


def test_adam_step_matches_torch(lib):
    import torch
    rng = np.random.RandomState(3)
    p0 = rng.randn(1000).astype(np.float32)
    pt = torch.nn.Parameter(torch.from_numpy(p0.copy()))
    opt = torch.optim.Adam([pt], lr=1e-4)
    p, m, v = p0.copy(), np.zeros(1000, np.float32), np.zeros(1000, np.float32)
    for step in range(1, 4):
        gr = rng.randn(1000).astype(np.float32)
        pt.grad = torch.from_numpy(gr.copy())
        opt.step()
        lib.adam_step(ptr(p), ptr(gr), ptr(m), ptr(v), 1000, 1e-4, 0.9, 0.999, 1e-8, step)
        assert float(np.max(np.abs(p - pt.detach().numpy()))) <= 2e-7


@pytest.mark.parametrize("env", [{"max_rt": 1}, {"train_per_step": 1}, {"train_per_step": 1, "step_col_tiles": 2}, {"train_old_gemm": 1},
                                 {"train_fp32_mfma": 1}, {"train_fp32_mfma": 1, "max_rt": 1},
                                 {"train_bwd_per_step": 1}, {}, {"train_bwd_geom": 1}, {"train_bwd_geom": 1, "max_rt": 1}, {"train_bwd_geom": 1, "bwd_w3_l1_h64": 1},
                                 {"train_fwd_geom": 1}, {"train_fwd_geom": 1, "max_rt": 1, "train_bwd_geom": 1}, {"train_fwd_geom": 1, "bwd_w3_l1_h64": 1}, {"train_kernel": 1}, {"train_kernel": 1, "max_rt": 1}, {"x3_tile": 16}, {"x3_tile": 32},
                                 {"x3_tile": 16, "max_rt": 1}])
def test_train_recurrence_variants_agree(lib, golden, options, env):
    """Persistent train recurrences (split-fp16 default, all-fp32 MFMA form) with one / two row tiles per block, and the
    per-step fallback and the simple GEMM kernels kept as unaligned-operand fallbacks, against the reference.  The backward
    recurrence is the persistent k_train_bwd_steps by default (one launch, folded feedback path, gate gradients exchanged as
    scaled fp16 pairs; option max_rt = 1: two row tiles per block) or 2T per-step launches (option train_bwd_per_step)."""
    options(**env)
    g = golden("train_h64")
    P = synth.CycleVAEProblem(B=18, T=7, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="train_h64")
    enc = TrainNet(lib, P.enc, 6, 8, 64)
    cot = synth.normal("train_h64/cot_enc", (18, 7, 8))
    out, yl, hl, dx, grads = enc.run(P.x, P.y_in_enc, None, g["enc_cmask"], g["enc_gmask"], cot, 4)
    assert rel_err(out, g["enc_out"]) <= 5e-5 and rel_err(dx, g["enc_dx"]) <= 1e-4
    for f, k in GRAD_KEYS.items():
        assert rel_err(grads[f], g["enc_g_" + k]) <= 1e-4, k


def test_adam_step_skipped_when_gate_is_raised(lib):
    """cvae_adam_step with a gate word: non-zero = a kernel of the step reported a failure, the update must leave parameters and
    moments untouched (stage4.Stage4Step passes the status sink: a bad step can never corrupt the optimiser state)."""
    rng = np.random.RandomState(5)
    p0, gr = rng.randn(777).astype(np.float32), rng.randn(777).astype(np.float32)
    p, m, v = p0.copy(), np.zeros(777, np.float32), np.zeros(777, np.float32)
    gate = np.array([5, 0, 0, 0], np.int32)
    lib.adam_step(ptr(p), ptr(gr), ptr(m), ptr(v), 777, 1e-3, 0.9, 0.999, 1e-8, 1, gate=ptr(gate))
    assert np.array_equal(p, p0) and not m.any() and not v.any()
    gate[0] = 0
    lib.adam_step(ptr(p), ptr(gr), ptr(m), ptr(v), 777, 1e-3, 0.9, 0.999, 1e-8, 1, gate=ptr(gate))
    assert np.abs(p - p0).max() > 5e-4 and m.any() and v.any()


@pytest.mark.parametrize("parts", [1, 2])
def test_sample_cat_and_its_backward_match_torch(lib, parts):
    """cvae_sample_cat / cvae_sample_cat_backward: [code ; mu + exp(s/2) eps] of train...:1335-1338 (gru_vae.py:96 + torch.cat), for
    one pass and for rec || cv stacked, against torch autograd; Philox draws are deterministic in (seed, draw) and standard normal."""
    import torch
    import stage4
    B, T, L, nc = 3, 5, 4, 2
    lat = (0.5 * synth.normal("scat/lat", (B, T, 2 * L))).astype(np.float32)
    codes = [np.ascontiguousarray(synth.normal("scat/c%d" % q, (B, T, nc)), np.float32) for q in range(parts)]
    eps = [np.ascontiguousarray(synth.normal("scat/e%d" % q, (B, T, L)), np.float32) for q in range(parts)]
    out, eps_used = np.full((parts * B, T, nc + L), np.nan, np.float32), np.full((parts, B, T, L), np.nan, np.float32)
    lib.sample_cat(ptr(lat), [ptr(c) for c in codes], [ptr(e) for e in eps], 0, list(range(parts)), B, T, L, nc, ptr(out), ptr(eps_used))
    lt = torch.from_numpy(lat).requires_grad_(True)
    ref = stage4.torch_dec_input(lt, [torch.from_numpy(c) for c in codes], [torch.from_numpy(e) for e in eps], L, (0, 0))
    assert np.abs(out - ref.detach().numpy()).max() <= 1e-6 and np.array_equal(eps_used, np.stack(eps))
    cot = np.ascontiguousarray(synth.normal("scat/cot", out.shape), np.float32)
    (ref * torch.from_numpy(cot)).sum().backward()
    dlat = np.full(lat.shape, np.nan, np.float32)
    lib.sample_cat_backward(ptr(cot), ptr(lat), ptr(eps_used), B, T, L, nc, parts, ptr(dlat))
    assert np.abs(dlat - lt.grad.numpy()).max() <= 2e-6
    # Philox: same (seed, draw) -> same eps; different draw -> different; roughly N(0,1)
    big = np.zeros((1, 400, 2 * L), np.float32)
    cb = np.zeros((1, 400, nc), np.float32)
    o1, e1 = np.zeros((1, 400, nc + L), np.float32), np.zeros((1, 1, 400, L), np.float32)
    e2, e3 = np.zeros_like(e1), np.zeros_like(e1)
    lib.sample_cat(ptr(big), [ptr(cb)], [None], 77, [3], 1, 400, L, nc, ptr(o1), ptr(e1))
    lib.sample_cat(ptr(big), [ptr(cb)], [None], 77, [3], 1, 400, L, nc, ptr(o1), ptr(e2))
    lib.sample_cat(ptr(big), [ptr(cb)], [None], 77, [4], 1, 400, L, nc, ptr(o1), ptr(e3))
    assert np.array_equal(e1, e2) and not np.array_equal(e1, e3)
    assert abs(float(e1.mean())) < 0.1 and abs(float(e1.std()) - 1.0) < 0.1 and np.allclose(o1[0, :, nc:], e3[0, 0], atol=1e-6)


@pytest.mark.parametrize("half,select,flen_acc", [(False, None, None), (False, [0, 2], [5, 3, 9]), (True, [1], [4, 4, 4]), (False, [1], [2, 6, 1])])
def test_stage4_loss_kernel_matches_the_torch_loss(lib, half, select, flen_acc):
    """cvae_stage4_loss (value and the four gradients, one launch per cycle) against stage4.loss_terms + autograd: ragged flen_acc,
    select_utt_idx, the :1393 quirk (KL(lat) twice for more than one selected utterance, KL(latcv) of the last one), half cycle."""
    import torch
    import stage4
    B, T, D, L, std = 3, 6, 5, 4, 2
    x = np.ascontiguousarray(synth.normal("s4l/x", (B, T, std + D)), np.float32)
    tr = {k: np.ascontiguousarray(0.7 * synth.normal("s4l/" + k, (B, T, D if k in ("rec", "cv", "reccyc") else 2 * L)), np.float32)
          for k in ("lat", "rec", "cv", "latcv", "reccyc")}
    tt = {k: torch.from_numpy(v).requires_grad_(True) for k, v in tr.items()}
    ref = stage4.loss_terms([tt], torch.from_numpy(x), std, L, flen_acc, select, half)
    ref.backward()
    w, last, n_sel = stage4.frame_weights(B, T, flen_acc, select)
    w, last = np.asarray(w, np.float32), np.asarray(last, np.float32)
    full = (not half) and n_sel > 0
    g = {k: np.full(tr[k].shape, np.nan, np.float32) for k in ("rec", "reccyc", "lat", "latcv")}
    fl, loss = np.zeros(B * T, np.float32), np.array([123.0], np.float32)
    lib.stage4_loss(ptr(tr["rec"]), ptr(tr["reccyc"]) if full else None, ptr(tr["lat"]), ptr(tr["latcv"]) if full else None, ptr(x),
                    std + D, std, ptr(w), ptr(last), 2.0 if (full and n_sel > 1) else 1.0, B, T, D, L, ptr(g["rec"]),
                    ptr(g["reccyc"]) if full else None, ptr(g["lat"]), ptr(g["latcv"]) if full else None, ptr(fl), ptr(loss), False)
    assert abs(float(loss[0]) - float(ref)) <= 2e-6 * abs(float(ref))
    for k in ("rec", "lat") + (("reccyc", "latcv") if full else ()):
        assert np.abs(g[k] - tt[k].grad.numpy()).max() <= 2e-6, k
    lib.stage4_loss(ptr(tr["rec"]), None, ptr(tr["lat"]), None, ptr(x), std + D, std, ptr(w), ptr(last), 1.0, B, T, D, L, ptr(g["rec"]),
                    None, ptr(g["lat"]), None, ptr(fl), ptr(loss), True)          # accumulate: adds a second cycle's terms
    assert float(loss[0]) > float(ref) or n_sel == 0


@pytest.mark.parametrize("B,T,max_rt,tile,geom", [(70, 4, 1, 16, 0), (70, 4, 1, 32, 1), (40, 5, 0, 16, 1), (130, 3, 1, 32, 0)])
def test_exact_operand_train_recurrence_several_tiles(lib, options, B, T, max_rt, tile, geom):
    """k_train_fwd_steps_x3 with one, three and (130 rows, one tile group) five-tile blocks -- the last falls back to the pair kernel
    (at most four tiles per block keep h in registers) -- against the stock-torch checker and against the pair-form kernel:
    outputs, h_last, dx and every parameter gradient."""
    import torch
    from oracle import torch_stock as ts
    P = synth.CycleVAEProblem(B=B, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="x3t%d" % B)
    cm = (synth.uniform01("x3t/c%d" % B, (B, T, 54)) >= 0.5).astype(np.float32) * 2.0
    gm = (synth.uniform01("x3t/g%d" % B, (T, B, 64)) >= 0.5).astype(np.float32) * 2.0
    cot = synth.normal("x3t/cot%d" % B, (B, T, 8))
    h_in = (0.3 * synth.normal("x3t/h%d" % B, (1, B, 64))).astype(np.float32)
    out_r, _, h_r, Pr, xr = ts.train_forward(P.enc, P.x, P.y_in_enc, h_in, cm, gm, 4)
    (out_r * torch.from_numpy(cot)).sum().backward()
    res = {}
    for kern in (0, 1):
        # (geom 1: the 16-unit forward and reverse recurrences, five / three tiles per block)
        options(train_kernel=kern, max_rt=max_rt, x3_tile=tile, train_bwd_geom=geom, train_fwd_geom=geom)
        enc = TrainNet(lib, P.enc, 6, 8, 64)
        res[kern] = enc.run(P.x, P.y_in_enc, h_in, cm, gm, cot, 4)
    for kern, (out, yl, hl, dx, grads) in res.items():
        assert rel_err(out, out_r.detach().numpy()) <= 2e-5 and rel_err(hl, h_r.detach().numpy()) <= 2e-5, kern
        assert rel_err(dx, xr.grad.numpy()) <= 1e-4, kern
        for f, k in GRAD_KEYS.items():
            assert rel_err(grads[f], Pr[k].grad.numpy()) <= 1e-4, (kern, k)
    assert rel_err(res[0][0], res[1][0].astype(np.float64)) <= 1e-5


@pytest.mark.parametrize("row_pad", [4, 0])
@pytest.mark.parametrize("B,T", [(1, 6), (2, 5), (3, 4)])
def test_word_exchange_train_recurrence_for_up_to_three_rows(lib, options, B, T, row_pad):
    """k_train_fwd_steps_ll / k_train_bwd_steps_ll (cvae_train_ll.h: at most three rows, the recipe's batch_size_utt = 1 and the
    rec || cv pair stacked from it) against the stock-torch checker and against the tile kernels (option no_ll): outputs, carried
    state, dx and every parameter gradient, with a carried-in state."""
    import torch
    from oracle import torch_stock as ts
    P = synth.CycleVAEProblem(B=B, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="llt%d" % B)
    cm = (synth.uniform01("llt/c%d" % B, (B, T, 54)) >= 0.5).astype(np.float32) * 2.0
    gm = (synth.uniform01("llt/g%d" % B, (T, B, 64)) >= 0.5).astype(np.float32) * 2.0
    cot = synth.normal("llt/cot%d" % B, (B, T, 8))
    h_in = (0.3 * synth.normal("llt/h%d" % B, (1, B, 64))).astype(np.float32)
    out_r, y_r, h_r, Pr, xr = ts.train_forward(P.enc, P.x, P.y_in_enc, h_in, cm, gm, 4)
    (out_r * torch.from_numpy(cot)).sum().backward()
    res = {}
    for no_ll in (0, 1):
        options(no_ll=no_ll, ll_row_pad=row_pad)      # (row_pad 0: the time-major buffers hold exactly B rows per frame)
        enc = TrainNet(lib, P.enc, 6, 8, 64)
        res[no_ll] = enc.run(P.x, P.y_in_enc, h_in, cm, gm, cot, 4)
    for no_ll, (out, yl, hl, dx, grads) in res.items():
        assert rel_err(out, out_r.detach().numpy()) <= 2e-5 and rel_err(hl, h_r.detach().numpy()) <= 2e-5, no_ll
        assert rel_err(dx, xr.grad.numpy()) <= 1e-4, no_ll
        for f, k in GRAD_KEYS.items():
            assert rel_err(grads[f], Pr[k].grad.numpy()) <= 1e-4, (no_ll, k)
    assert not np.array_equal(res[0][0], res[1][0])        # they really are different kernels
    assert rel_err(res[0][0], res[1][0].astype(np.float64)) <= 1e-5


def test_per_step_forward_kernel_two_column_tiles_eight_row_tiles(lib, options):
    """k_gru_step_train<2, 8> (what a stacked rec || cv pass of 128 rows runs at hu2048: two 16-column tiles per block, eight
    16-row tiles per trip over the weights) and <2, 4> with a ragged last trip, forced at H = 64, against the stock-torch checker."""
    import torch
    from oracle import torch_stock as ts
    for B, T in ((130, 3), (70, 3)):
        P = synth.CycleVAEProblem(B=B, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="ps%d" % B)
        cm = (synth.uniform01("ps/c%d" % B, (B, T, 54)) >= 0.5).astype(np.float32) * 2.0
        gm = (synth.uniform01("ps/g%d" % B, (T, B, 64)) >= 0.5).astype(np.float32) * 2.0
        cot = synth.normal("ps/cot%d" % B, (B, T, 8))
        h_in = (0.3 * synth.normal("ps/h%d" % B, (1, B, 64))).astype(np.float32)
        out_r, _, h_r, Pr, xr = ts.train_forward(P.enc, P.x, P.y_in_enc, h_in, cm, gm, 4)
        (out_r * torch.from_numpy(cot)).sum().backward()
        options(train_per_step=1, step_col_tiles=2)
        enc = TrainNet(lib, P.enc, 6, 8, 64)
        out, yl, hl, dx, grads = enc.run(P.x, P.y_in_enc, h_in, cm, gm, cot, 4)
        assert rel_err(out, out_r.detach().numpy()) <= 2e-5 and rel_err(hl, h_r.detach().numpy()) <= 2e-5
        assert rel_err(dx, xr.grad.numpy()) <= 1e-4
        for f, k in GRAD_KEYS.items():
            assert rel_err(grads[f], Pr[k].grad.numpy()) <= 1e-4, k


def test_train_image_variants_are_built_on_demand_and_checked(lib):
    """cvae_net_prepare_train_v: a train image prepared with variants = 0 (no MFMA-order image: what a net needs that only sees passes
    of at most three rows) runs such a pass exactly like the full image, and a pass that needs the exact-operand tile kernels is REFUSED
    on it (-4) instead of reading weight images that were never written.  cvae_train_variants_needed names what a shape needs."""
    hid = 64
    P = synth.CycleVAEProblem(B=18, T=7, in_dim=6, out_dim=4, lat_dim=4, hidden=hid, n_cyc=1, bias_scale=0.1, tag="variants")
    full = TrainNet(lib, P.enc, 6, 8, hid)
    bare = TrainNet(lib, P.enc, 6, 8, hid)
    bare.image[:] = np.nan                               # whatever is not written must not be read
    lib.net_prepare_train(bare.d, {f: ptr(bare.sd[k]) for f, k in _cabi.STATE_KEYS.items() if k in bare.sd}, ptr(bare.image),
                          bare.image.nbytes, gru_drop_p=0.5, variants=0)
    assert lib.train_variants_needed(full.d, 18, 7) & 1            # 18 rows: the exact-operand tile kernels
    cu_ok = lib.train_variants_needed(full.d, 2, 7) == 0           # <= 3 rows: word-exchange kernels, no MFMA-order image ...
    cm = (synth.uniform01("variants/c", (18, 7, 9 * 6)) >= 0.5).astype(np.float32) * 2.0
    gm = (synth.uniform01("variants/g", (7, 18, hid)) >= 0.5).astype(np.float32) * 2.0
    cot = synth.normal("variants/cot", (18, 7, 8))
    if cu_ok:                                                      # (... when the grid of hid/4 blocks fits the emulated device)
        a = full.run(P.x[:2], P.y_in_enc[:2], None, cm[:2], gm[:, :2], cot[:2], 4)
        b = bare.run(P.x[:2], P.y_in_enc[:2], None, cm[:2], gm[:, :2], cot[:2], 4)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[3], b[3])
        for f in GRAD_KEYS:
            assert np.array_equal(a[4][f], b[4][f]), f
    with pytest.raises(_cabi.CvaeError, match="-4"):
        bare.run(P.x, P.y_in_enc, None, cm, gm, cot, 4)
    out = full.run(P.x, P.y_in_enc, None, cm, gm, cot, 4)[0]
    assert np.all(np.isfinite(out))
