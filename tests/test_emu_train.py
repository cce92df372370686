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
                              ptr(self.image), self.image.nbytes)

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


@pytest.mark.parametrize("env", [{"max_rt": 1}, {"train_per_step": 1}, {"train_old_gemm": 1},
                                 {"train_fp32_mfma": 1}, {"train_fp32_mfma": 1, "max_rt": 1},
                                 {"train_bwd_per_step": 1}, {}])
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
