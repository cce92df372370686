"""CPU tests of the REAL library code (cyclevae-vc_amd/csrc/*) compiled for the host and run on fibers
(tests/emu): kernels, launch sequence and the C ABI are exactly the product's; only the intrinsics header is
swapped.  Compared with the oracle at sizes the emulator finishes in seconds.  Tolerance: fp32 re-association
plus the load-time folds (conv0*conv1*W_ih and W_ih_y*out_1 are pre-multiplied in fp64, rounded once):
max|d| <= 5e-5 on single passes, 3e-4 on the 10-pass chain.
"""
import numpy as np
import pytest

import _cabi
import synth
from emu_util import NpNet, emu_lib, ptr
from oracle import cyclevae_oracle as orc


def maxabs(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def tiny(B=2, T=12, hidden=32, tag="tiny", in_dim=6, out_dim=4, lat=4):
    return synth.CycleVAEProblem(B=B, T=T, in_dim=in_dim, out_dim=out_dim, lat_dim=lat, hidden=hidden, n_cyc=2, bias_scale=0.1, tag=tag)


def test_exports_and_abi(lib):
    for name in _cabi.EXPORTS:
        assert hasattr(lib.lib, name), name
    assert lib.lib.cvae_abi_version() == _cabi.ABI_VERSION


def test_bad_arguments_fail_loudly(lib):
    with pytest.raises(_cabi.CvaeError):
        lib.prepared_bytes(lib.desc(6, 8, 40))          # hidden not a multiple of 16
    with pytest.raises(_cabi.CvaeError):
        lib.prepared_bytes(lib.desc(6, 8, 32, 3, 3))    # layers != 2
    with pytest.raises(_cabi.CvaeError):
        lib.pass_workspace_bytes(lib.desc(6, 8, 32), 0, 5)   # empty batch


@pytest.mark.parametrize("flags", [0, _cabi.FLAG_PERSISTENT])
def test_encoder_pass_matches_oracle_and_golden(lib, golden, flags):
    P = tiny()
    net = NpNet(lib, P.enc, 6, 8, 32)
    trj, yl, hl = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=flags)
    o_trj, o_yl, o_hl = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, clamp_vae=True, lat_dim=4)
    assert maxabs(trj, o_trj) <= 5e-5 and maxabs(yl, o_yl) <= 5e-5 and maxabs(hl, o_hl) <= 5e-5
    g = golden("tiny_ops")
    assert maxabs(trj, g["lat"]) <= 5e-5 and maxabs(hl, g["lat_h"]) <= 5e-5


def test_decoder_pass_with_fused_sampling(lib, golden):
    P = tiny()
    g = golden("tiny_ops")
    net = NpNet(lib, P.dec, 6, 4, 32)
    lat = np.ascontiguousarray(g["lat"])
    eps = np.ascontiguousarray(P.eps[0, 0])
    trj, yl, hl = net.forward(P.code_src, P.y_in_dec, lat=lat, lat_dim=4, eps=eps)
    assert maxabs(trj, g["rec"]) <= 5e-5 and maxabs(yl, g["rec_y"]) <= 5e-5 and maxabs(hl, g["rec_h"]) <= 5e-5


def test_two_segment_input_and_state_carry(lib):
    P = tiny(B=3, T=20, tag="carry")
    net = NpNet(lib, P.enc, 6, 8, 32)
    xa = np.ascontiguousarray(P.x[:, :, :2])
    xb = np.ascontiguousarray(P.x[:, :, 2:])
    whole = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4)[0]
    split = net.forward(xa, P.y_in_enc, clamp_lat_dim=4, seg1=xb)[0]
    assert maxabs(whole, split) == 0.0
    # windows with (y_last, h) carried (train_gru_cyclevae_gauss_batch.py:1301-1311)
    a, ay, ah = net.forward(P.x[:, :10], P.y_in_enc, clamp_lat_dim=4)
    b, by, bh = net.forward(P.x[:, 10:], ay, h_in=ah, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT)
    oa, oay, oah = orc.gru_rnn_forward(P.enc, P.x[:, :10], P.y_in_enc, clamp_vae=True, lat_dim=4)
    ob, oby, obh = orc.gru_rnn_forward(P.enc, P.x[:, 10:], oay, h_in=oah, clamp_vae=True, lat_dim=4)
    assert maxabs(a, oa) <= 5e-5 and maxabs(b, ob) <= 5e-5 and maxabs(by, oby) <= 5e-5 and maxabs(bh, obh) <= 5e-5
    # a y_in that is NOT out_1(h_in) must be honoured too (frame-0 correction path)
    y_odd = (ay + 0.37).astype(np.float32)
    c = net.forward(P.x[:, 10:], y_odd, h_in=ah, clamp_lat_dim=4)[0]
    oc = orc.gru_rnn_forward(P.enc, P.x[:, 10:], y_odd, h_in=oah, clamp_vae=True, lat_dim=4)[0]
    assert maxabs(c, oc) <= 5e-5


def test_clamp_exercised(lib):
    P = tiny()
    sd = dict(P.enc)
    sd["out_1.bias"] = sd["out_1.bias"] - np.float32(20.0)
    net = NpNet(lib, sd, 6, 8, 32)
    trj = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4)[0]
    o = orc.gru_rnn_forward(sd, P.x, P.y_in_enc, clamp_vae=True, lat_dim=4)[0]
    assert np.any(trj[:, :, 4:] == orc.LOG_VAR_FLOOR) and maxabs(trj, o) <= 5e-5
    noclamp = net.forward(P.x, P.y_in_enc, clamp_lat_dim=-1)[0]
    assert np.any(noclamp[:, :, 4:] < orc.LOG_VAR_FLOOR)


def test_ragged_sizes(lib):
    """B not a multiple of 16 (17 -> two row tiles), T=1, wide hidden split over all four waves."""
    P = tiny(B=17, T=3, hidden=64, tag="ragged")
    net = NpNet(lib, P.enc, 6, 8, 64)
    trj, yl, hl = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT)
    o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, clamp_vae=True, lat_dim=4)
    assert maxabs(trj, o[0]) <= 5e-5 and maxabs(hl, o[2]) <= 5e-5
    t1 = net.forward(P.x[:, :1], P.y_in_enc, clamp_lat_dim=4)[0]
    assert maxabs(t1, orc.gru_rnn_forward(P.enc, P.x[:, :1], P.y_in_enc, clamp_vae=True, lat_dim=4)[0]) <= 5e-5


def test_sampling_kernel(lib):
    P = tiny()
    lat = np.ascontiguousarray(synth.normal("smp/lat", (24, 8)))
    eps = np.ascontiguousarray(synth.normal("smp/eps", (24, 4)))
    z = np.zeros((24, 4), np.float32)
    lib.sample(ptr(lat), 24, 4, ptr(eps), 0, 0, ptr(z))
    assert maxabs(z, orc.sampling_vae_batch(lat, eps, 4)) <= 2e-6
    # Philox path: deterministic in (seed, draw), N(0,1)-looking, different across draws
    big = np.zeros((4096, 8), np.float32)
    e1, e2, e3 = (np.zeros((4096, 4), np.float32) for _ in range(3))
    zz = np.zeros((4096, 4), np.float32)
    lib.sample(ptr(big), 4096, 4, None, 7, 0, ptr(zz), ptr(e1))
    lib.sample(ptr(big), 4096, 4, None, 7, 0, ptr(zz), ptr(e2))
    lib.sample(ptr(big), 4096, 4, None, 7, 1, ptr(zz), ptr(e3))
    assert np.array_equal(e1, e2) and not np.array_equal(e1, e3)
    assert np.array_equal(zz, e3)           # mu = 0, log_var = 0  ->  z == eps
    assert abs(e1.mean()) < 0.05 and abs(e1.std() - 1.0) < 0.05


def test_cycle_chain_matches_golden(lib, golden):
    P = tiny()
    g = golden("tiny_chain")
    enc, dec = NpNet(lib, P.enc, 6, 8, 32), NpNet(lib, P.dec, 6, 4, 32)
    B, T, L = 2, 12, 4
    outs = {k: np.full((2, B, T, c), np.nan, np.float32) for k, c in (("lat", 8), ("rec", 4), ("cv", 4), ("latcv", 8), ("reccyc", 4))}
    ws = np.zeros(lib.cycle_workspace_bytes(enc.d, dec.d, B, T, 2) // 4, np.float32)
    ye, yd = np.ascontiguousarray(P.y_in_enc.reshape(B, 8)), np.ascontiguousarray(P.y_in_dec.reshape(B, 4))
    eps = np.ascontiguousarray(P.eps)
    for flags in (0, _cabi.FLAG_PERSISTENT):
        lib.cycle_forward(enc.d, ptr(enc.prepared), dec.d, ptr(dec.prepared), ptr(P.x), ptr(P.cvx), 2, ptr(P.code_src),
                          ptr(P.code_trg), 2, ptr(ye), ptr(yd), B, T, 2, L, ptr(eps), 0, ptr(outs["lat"]), ptr(outs["rec"]),
                          ptr(outs["cv"]), ptr(outs["latcv"]), ptr(outs["reccyc"]), ptr(ws), ws.nbytes, flags)
        assert lib.workspace_status(ptr(ws))[0] == 0
        for k in outs:
            assert maxabs(outs[k], g[k]) <= 3e-4, k
    # outputs the caller does not want may be NULL; the chain still runs on workspace buffers
    only = np.full((2, B, T, 4), np.nan, np.float32)
    lib.cycle_forward(enc.d, ptr(enc.prepared), dec.d, ptr(dec.prepared), ptr(P.x), ptr(P.cvx), 2, ptr(P.code_src),
                      ptr(P.code_trg), 2, ptr(ye), ptr(yd), B, T, 2, L, ptr(eps), 0, None, None, None, None, ptr(only),
                      ptr(ws), ws.nbytes, 0)
    assert maxabs(only, g["reccyc"]) <= 3e-4


@pytest.mark.parametrize("B,T", [(3, 5), (17, 4), (64, 3), (80, 2)])
def test_tuned_persistent_kernels_h64(lib, B, T):
    """The persistent recurrences against the oracle and each other.  The any-H kernel (k_gru_steps<persist>, grid barrier)
    repeats the per-step launches bit for bit; k_gru_steps_v2 (2-D blocks, per-chunk dataflow flags, hoisted front-end GEMM)
    has the same MFMA / reduction order but hardware-exp gates, so it agrees to rounding; k_gru_steps_v4 (all-fp32 MFMA)
    additionally computes the front-end inside the step; k_gru_steps_v5 is covered by test_split_f16_recurrence_h64."""
    P = tiny(B=B, T=T, hidden=64, tag="v1_%d_%d" % (B, T))
    net = NpNet(lib, P.enc, 6, 8, 64)
    v4 = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT)
    v2 = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT | _cabi.FLAG_HOISTED_FRONTEND)
    gen = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT | _cabi.FLAG_GENERIC_STEP)
    step = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=0)
    o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, clamp_vae=True, lat_dim=4)
    for b, c, d, e, f in zip(gen, step, o, v2, v4):
        assert np.array_equal(b, c)
        assert maxabs(b, d) <= 5e-5 and maxabs(e, d) <= 5e-5 and maxabs(e, b) <= 5e-6
        assert maxabs(f, d) <= 5e-5 and maxabs(f, b) <= 5e-6      # fused front-end: same sums, different order


@pytest.mark.parametrize("B,T,max_rt", [(3, 6, None), (40, 4, None), (33, 3, "1"), (50, 3, "1")])
def test_split_f16_recurrence_h64(lib, options, B, T, max_rt):
    """k_gru_steps_v5: weights and exchanged state as (hi, lo) fp16 pairs, x = hi + lo/2048 (22 bits), the recurrent product as
    three fp16 MFMAs with fp32 accumulation.  Against the oracle (fp32) the difference is a few 1e-6 -- two orders above the
    all-fp32 kernel, three below the 1e-3 chain budget; one / two / three-plus row tiles per block (hold carried in registers
    or re-read from the pair buffer)."""
    if max_rt:
        options(max_rt=int(max_rt))
    P = tiny(B=B, T=T, hidden=64, tag="split_%d_%d" % (B, T))
    net = NpNet(lib, P.enc, 6, 8, 64)
    h_in = (0.3 * synth.normal("split_h/%d" % B, (1, B, 64))).astype(np.float32)
    sp = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT | _cabi.FLAG_SPLIT_F16)
    f32 = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT)
    o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, h_in=h_in, clamp_vae=True, lat_dim=4)
    worst = 0.0
    for a, b, d in zip(sp, f32, o):
        assert maxabs(b, d) <= 5e-5
        worst = max(worst, maxabs(a, d))
    assert worst <= 2e-5, worst
    assert any(not np.array_equal(a, b) for a, b in zip(sp, f32))     # it really is the other kernel


def test_stacked_cells_equal_separate_passes(lib, golden):
    """rec || cv run as one decoder pass over 2B stacked rows inside cvae_cycle_forward; per-row arithmetic must not
    depend on where a row sits, so the chain equals five separate passes (checked through the golden above) and a
    hidden=64 chain with B=20 (cells straddle row tiles: 40 rows -> 3 tiles) matches the oracle."""
    P = synth.CycleVAEProblem(B=20, T=6, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=2, bias_scale=0.1, tag="stack")
    enc, dec = NpNet(lib, P.enc, 6, 8, 64), NpNet(lib, P.dec, 6, 4, 64)
    B, T, L = 20, 6, 4
    outs = {k: np.full((2, B, T, c), np.nan, np.float32) for k, c in (("lat", 8), ("rec", 4), ("cv", 4), ("latcv", 8), ("reccyc", 4))}
    ws = np.zeros(lib.cycle_workspace_bytes(enc.d, dec.d, B, T, 2) // 4, np.float32)
    ye, yd = np.ascontiguousarray(P.y_in_enc.reshape(B, 8)), np.ascontiguousarray(P.y_in_dec.reshape(B, 4))
    eps = np.ascontiguousarray(P.eps)
    ref = orc.cycle_chain(P.enc, P.dec, P.x, P.cvx, P.code_src, P.code_trg, P.y_in_enc, P.y_in_dec, P.eps, 2, 4)
    # (with EXACT3: k_gru_steps_v6 on 32-row tiles and its projection k_outproj_v6 straight from the limb triples)
    for flags in (_cabi.FLAG_PERSISTENT, 0, _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3):
        lib.cycle_forward(enc.d, ptr(enc.prepared), dec.d, ptr(dec.prepared), ptr(P.x), ptr(P.cvx), 2, ptr(P.code_src),
                          ptr(P.code_trg), 2, ptr(ye), ptr(yd), B, T, 2, L, ptr(eps), 0, ptr(outs["lat"]), ptr(outs["rec"]),
                          ptr(outs["cv"]), ptr(outs["latcv"]), ptr(outs["reccyc"]), ptr(ws), ws.nbytes, flags)
        assert lib.workspace_status(ptr(ws))[0] == 0
        for k in outs:
            assert maxabs(outs[k], np.stack(ref[k])) <= 3e-4, k


@pytest.mark.parametrize("max_rt,B", [(1, 33), (1, 20), (2, 64)])
def test_several_row_tiles_per_block(lib, options, max_rt, B):
    """Blocks that own 2 or 3 row tiles (what happens at hu1024 with stacked passes or B > 64): the software pipeline of
    k_gru_steps_v4 (early operand request, carried h registers, flag probes), the tile loops of v5 and v2, vs the oracle."""
    options(max_rt=int(max_rt))
    T = 5
    P = tiny(B=B, T=T, hidden=64, tag="mt_%d_%d" % (max_rt, B))
    net = NpNet(lib, P.enc, 6, 8, 64)
    o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, clamp_vae=True, lat_dim=4)
    for flags in (_cabi.FLAG_PERSISTENT, _cabi.FLAG_PERSISTENT | _cabi.FLAG_SPLIT_F16,
                  _cabi.FLAG_PERSISTENT | _cabi.FLAG_HOISTED_FRONTEND):
        got = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=flags)
        for a, d in zip(got, o):
            assert maxabs(a, d) <= 5e-5, flags


@pytest.mark.parametrize("B,T,max_rt", [(17, 6, None), (40, 4, None), (70, 3, "1"), (64, 3, "1"), (33, 5, None)])
def test_exact3_recurrence_h64(lib, options, B, T, max_rt):
    """k_gru_steps_v6: every fp32 operand (weights, exchanged state, normalised input) as THREE fp16 limbs (exact), six
    f16 MFMAs per product, 32-row x 8-unit blocks.  It must sit as close to the fp32 oracle as the all-fp32-MFMA kernel does
    (rounding of different summation orders only) and closer than the 22-bit pair kernel; one / two / three row tiles per
    block (carried h in registers or re-read from the triple buffer), ragged last tile, h_in and an off-manifold y_in."""
    if max_rt:
        options(max_rt=int(max_rt))
    P = tiny(B=B, T=T, hidden=64, tag="ex3_%d_%d" % (B, T))
    net = NpNet(lib, P.enc, 6, 8, 64)
    h_in = (0.3 * synth.normal("ex3_h/%d" % B, (1, B, 64))).astype(np.float32)
    y_in = (P.y_in_enc + 0.21).astype(np.float32)
    ex = net.forward(P.x, y_in, h_in=h_in, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3)
    f32 = net.forward(P.x, y_in, h_in=h_in, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT)
    o = orc.gru_rnn_forward(P.enc, P.x, y_in, h_in=h_in, clamp_vae=True, lat_dim=4)
    for a, b, d in zip(ex, f32, o):
        assert maxabs(a, d) <= 5e-6, maxabs(a, d)
        assert maxabs(a, b) <= 3e-6, maxabs(a, b)
    assert any(not np.array_equal(a, b) for a, b in zip(ex, f32))     # it really is the other kernel


def test_cycle_chain_carry_form(lib):
    """cvae_cycle_forward_carry: two 6-frame windows of the same utterances, the second continued from the first one's state
    of every pass (reference train...:1299-1311), against the oracle run pass by pass with the same carries."""
    P = tiny(B=3, T=12, tag="carrychain")
    enc, dec = NpNet(lib, P.enc, 6, 8, 32), NpNet(lib, P.dec, 6, 4, 32)
    B, L, n = 3, 4, 2
    ws = np.zeros(lib.cycle_workspace_bytes(enc.d, dec.d, B, 6, n) // 4, np.float32)
    ye, yd = np.ascontiguousarray(P.y_in_enc.reshape(B, 8)), np.ascontiguousarray(P.y_in_dec.reshape(B, 4))
    shapes = ((n, 2, B, 8), (n, 3, B, 4), (n, 2, B, 32), (n, 3, B, 32))
    state = None
    ref_state = {}
    for w, flags in ((0, _cabi.FLAG_PERSISTENT), (1, 0)):
        sl = slice(6 * w, 6 * w + 6)
        xs = [np.ascontiguousarray(a[:, sl]) for a in (P.x, P.cvx, P.code_src, P.code_trg)]
        eps = np.ascontiguousarray(P.eps[:, :, :, sl])
        outs = {k: np.full((n, B, 6, c), np.nan, np.float32) for k, c in (("lat", 8), ("rec", 4), ("cv", 4), ("latcv", 8), ("reccyc", 4))}
        new = tuple(np.full(s_, np.nan, np.float32) for s_ in shapes)
        lib.cycle_forward_carry(enc.d, ptr(enc.prepared), dec.d, ptr(dec.prepared), ptr(xs[0]), ptr(xs[1]), 2, ptr(xs[2]), ptr(xs[3]), 2,
                                ptr(ye), ptr(yd), B, 6, n, L, ptr(eps), 0, ptr(outs["lat"]), ptr(outs["rec"]), ptr(outs["cv"]),
                                ptr(outs["latcv"]), ptr(outs["reccyc"]), ptr(ws), ws.nbytes, flags, 0,
                                None if state is None else tuple(ptr(a) for a in state), tuple(ptr(a) for a in new))
        assert lib.workspace_status(ptr(ws))[0] == 0
        # oracle, pass by pass with the same carries
        prev = None
        for i in range(n):
            def run(sd, xin, slot, clamp, y0):
                yc, hc = ref_state.get((i, slot), (y0, None))
                o, yl, hl = orc.gru_rnn_forward(sd, xin, yc, h_in=hc, clamp_vae=clamp, lat_dim=L)
                ref_state[(i, slot)] = (yl, hl)
                return o
            e_in = xs[0] if i == 0 else np.concatenate([xs[0][:, :, :2], prev], 2)
            lat = run(P.enc, e_in, "lat", True, P.y_in_enc)
            rec = run(P.dec, np.concatenate([xs[2], orc.sampling_vae_batch(lat, eps[i, 0], L)], 2), "rec", False, P.y_in_dec)
            cv = run(P.dec, np.concatenate([xs[3], orc.sampling_vae_batch(lat, eps[i, 1], L)], 2), "cv", False, P.y_in_dec)
            latcv = run(P.enc, np.concatenate([xs[1], cv], 2), "latcv", True, P.y_in_enc)
            prev = run(P.dec, np.concatenate([xs[2], orc.sampling_vae_batch(latcv, eps[i, 2], L)], 2), "reccyc", False, P.y_in_dec)
            for k, v in (("lat", lat), ("rec", rec), ("cv", cv), ("latcv", latcv), ("reccyc", prev)):
                assert maxabs(outs[k][i], v) <= 3e-4, (w, i, k)
            assert maxabs(new[2][i, 1], ref_state[(i, "latcv")][1][0]) <= 3e-4 and maxabs(new[1][i, 2], ref_state[(i, "reccyc")][0][:, 0]) <= 3e-4
        state = new


@pytest.mark.parametrize("B,T,max_rt", [(40, 4, None), (70, 3, "1")])
def test_v6_two_limb_form_h64(lib, options, B, T, max_rt):
    """k_gru_steps_v6<..., LIMBS = 2>: the 32-row x 8-unit kernel on (l0, l1) pairs only -- what runs at H = 2048 (the hu2048 stress
    configuration), where three limbs of a block's weights cannot be register-resident.  option v6_limbs_h64 = 2 selects that code path
    at H = 64 so that the emulator can run it; accuracy class of k_gru_steps_v5 (22-bit operands)."""
    options(v6_limbs_h64=2)
    if max_rt:
        options(max_rt=int(max_rt))
    P = tiny(B=B, T=T, hidden=64, tag="v6l2_%d_%d" % (B, T))
    net = NpNet(lib, P.enc, 6, 8, 64)
    h_in = (0.3 * synth.normal("v6l2_h/%d" % B, (1, B, 64))).astype(np.float32)
    two = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3)
    options(v6_limbs_h64=3)
    three = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=_cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3)
    o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, h_in=h_in, clamp_vae=True, lat_dim=4)
    for a, b, d in zip(two, three, o):
        assert maxabs(a, d) <= 2e-5 and maxabs(b, d) <= 5e-6
    assert any(not np.array_equal(a, b) for a, b in zip(two, three))


@pytest.mark.parametrize("B,T,max_rt", [(40, 5, None), (70, 3, 1)])
def test_v6_streamed_third_limb_form_h64(lib, options, B, T, max_rt):
    """k_gru_steps_v6<..., LIMBS = 3, W2S>: exact three-limb operands with the third limbs of the recurrent weights streamed as bf8
    bytes through a ring instead of register-resident (what runs at H = 2048, where l0 and l1 alone fill the registers), shorter
    operand rings, one S2 accumulator chain.  Option v6_w2s_h64 selects it at H = 64: same accuracy class as the resident form
    (both multiply fp32-exact operands; the streamed third weight limb is a bf8 byte), several tiles per block included."""
    if max_rt:
        options(max_rt=max_rt)
    P = tiny(B=B, T=T, hidden=64, tag="v6w2s_%d_%d" % (B, T))
    net = NpNet(lib, P.enc, 6, 8, 64)
    h_in = (0.3 * synth.normal("v6w2s_h/%d" % B, (1, B, 64))).astype(np.float32)
    fl = _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3
    resident = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=fl)
    options(v6_w2s_h64=1)
    streamed = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=fl)
    o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, h_in=h_in, clamp_vae=True, lat_dim=4)
    for a, b, d in zip(streamed, resident, o):
        assert maxabs(a, d) <= 5e-6 and maxabs(b, d) <= 5e-6
        assert maxabs(a, b) <= 2e-6


@pytest.mark.parametrize("B,T,hidden", [(1, 9, 64), (2, 7, 64), (3, 6, 64), (3, 4, 128)])
def test_word_exchange_kernel_small_batches(lib, options, B, T, hidden):
    """k_gru_steps_ll: passes of at most three rows (single utterance, encoder pair, decoder triple of decode...:302-323) exchange
    the state as (h row 0..2, step tag) words and accumulate in plain fp32.  Must match the oracle with h_in / y_in carries, agree
    with the dataflow kernel it replaces, and leave y_last / h_last right."""
    P = tiny(B=B, T=T, hidden=hidden, tag="ll_%d_%d_%d" % (B, T, hidden))
    net = NpNet(lib, P.enc, 6, 8, hidden)
    h_in = (0.3 * synth.normal("ll_h/%d/%d" % (B, hidden), (1, B, hidden))).astype(np.float32)
    fl = _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3
    new = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=fl)
    options(no_ll=1)
    old = net.forward(P.x, P.y_in_enc, h_in=h_in, clamp_lat_dim=4, flags=fl)
    options(no_ll=0)
    o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, h_in=h_in, clamp_vae=True, lat_dim=4)
    for a, b, d in zip(new, old, o):
        assert maxabs(a, d) <= 5e-6 and maxabs(b, d) <= 5e-5
    assert any(not np.array_equal(a, b) for a, b in zip(new, old))      # it is another kernel
    dec = NpNet(lib, P.dec, 6, 4, hidden)
    lat = np.ascontiguousarray(new[0])
    eps = np.ascontiguousarray(P.eps[0, 0])
    rec = dec.forward(P.code_src, P.y_in_dec, lat=lat, lat_dim=4, eps=eps, flags=fl)
    z = orc.sampling_vae_batch(lat, eps, 4)
    o_rec = orc.gru_rnn_forward(P.dec, np.concatenate([P.code_src, z], 2), P.y_in_dec)
    for a, d in zip(rec, o_rec):
        assert maxabs(a, d) <= 5e-6


def test_thirty_stacked_cells_with_ragged_lengths(lib):
    """cvae_gru_rnn_forward_stacked with 30 single-row cells (ten utterance pairs of stage 6: 3N decoder rows in ONE 32-row tile of
    the dataflow kernel) of different lengths, two of them with a fused 3-draw latent mean: every cell must equal its own pass
    run alone (shorter cells see zeros after normalisation beyond their last frame, like the conv padding alone)."""
    hidden, T, ncell = 64, 14, 30
    P = tiny(B=ncell, T=T, hidden=hidden, tag="stack30")
    net = NpNet(lib, P.enc, 6, 8, hidden)
    fl = _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3
    frames = [T - (c * 5) % 9 for c in range(ncell)]
    xs = [np.ascontiguousarray(P.x[c, :frames[c]]) for c in range(ncell)]
    y_in = np.ascontiguousarray(P.y_in_enc.reshape(ncell, -1)[:1])
    outs = [np.full((T, 8), np.nan, np.float32) for _ in range(ncell)]
    pins = [lib.pass_input((ptr(xs[c]), 6, 6), frames=frames[c]) for c in range(ncell)]
    ws = np.zeros(lib.pass_workspace_bytes(net.d, ncell, T) // 4, np.float32)
    lib.gru_rnn_forward_stacked(net.d, ptr(net.prepared), pins, [ptr(y_in)] * ncell, 1, T, 4, [ptr(o) for o in outs], ptr(ws), ws.nbytes, fl)
    assert lib.workspace_status(ptr(ws))[0] == 0
    for c in range(ncell):
        alone = orc.gru_rnn_forward(P.enc, xs[c][None], y_in[None], clamp_vae=True, lat_dim=4)[0][0]
        assert maxabs(outs[c][:frames[c]], alone) <= 5e-5, c
    # 31 rows of a decoder with the latent mean of 3 draws fused into the prologue (cells 0 and 30 share one latent, as cvmcep /
    # cvmcep_src do)
    dec = NpNet(lib, P.dec, 6, 4, hidden)
    ncd = 31
    lat = [np.ascontiguousarray(outs[c % ncell][:frames[c % ncell]]) for c in range(ncd)]
    eps = [np.ascontiguousarray(synth.normal("stack30/eps/%d" % (c % ncell), (3, 1, T, 4))[:, :, :frames[c % ncell]].reshape(3, frames[c % ncell], 4))
           for c in range(ncd)]
    codes = np.eye(2, dtype=np.float32)
    yd = np.ascontiguousarray(P.y_in_dec.reshape(ncell, -1)[:1])
    douts = [np.full((T, 4), np.nan, np.float32) for _ in range(ncd)]
    # (eps layout [n_draws][B=1][T][L] with the pass's T: pad the shorter cells)
    eps_p = []
    for c in range(ncd):
        e = np.zeros((3, T, 4), np.float32)
        e[:, :frames[c % ncell]] = eps[c]
        eps_p.append(e)
    pins = [lib.pass_input((ptr(codes[c & 1]), 2, 0), lat=ptr(lat[c]), lat_dim=4, eps=ptr(eps_p[c]), frames=frames[c % ncell], n_draws=3)
            for c in range(ncd)]
    ws = np.zeros(lib.pass_workspace_bytes(dec.d, ncd, T) // 4, np.float32)
    lib.gru_rnn_forward_stacked(dec.d, ptr(dec.prepared), pins, [ptr(yd)] * ncd, 1, T, -1, [ptr(o) for o in douts], ptr(ws), ws.nbytes, fl)
    assert lib.workspace_status(ptr(ws))[0] == 0
    for c in (0, 7, 29, 30):
        f = frames[c % ncell]
        z = np.mean(np.stack([orc.sampling_vae_batch(lat[c][None], eps[c][k][None], 4)[0] for k in range(3)]), 0)
        code = np.tile(codes[c & 1], (f, 1))
        alone = orc.gru_rnn_forward(P.dec, np.concatenate([code, z], 1)[None], yd[None])[0][0]
        assert maxabs(douts[c][:f], alone) <= 5e-5, c
    with pytest.raises(_cabi.CvaeError):
        lib.gru_rnn_forward_stacked(dec.d, ptr(dec.prepared), pins + pins[:2], [ptr(yd)] * 33, 1, T, -1, [ptr(o) for o in douts + douts[:2]],
                                    ptr(ws), ws.nbytes, fl)


def _windowed_stacked(lib, net, cells, y_in, L, T_total, clamp, fl, out_dim, H):
    """cells: list of dicts(pin=lambda start, frames -> pass_input, n=frames of the utterance).  Runs consecutive windows of L frames
    through cvae_gru_rnn_forward_stacked_carry and returns the per-cell outputs [n, out_dim]."""
    outs = [np.full((c["n"], out_dim), np.nan, np.float32) for c in cells]
    hs = [np.zeros((1, H), np.float32) for _ in cells]
    for start in range(0, T_total, L):
        alive = [i for i, c in enumerate(cells) if c["n"] > start]
        fr = [min(L, cells[i]["n"] - start) for i in alive]
        T = max(fr)
        pins = [cells[i]["pin"](start, f) for i, f in zip(alive, fr)]
        wout = [np.full((T, out_dim), np.nan, np.float32) for _ in alive]
        hl = [np.full((1, H), np.nan, np.float32) for _ in alive]
        ws = np.zeros(lib.pass_workspace_bytes(net.d, len(alive), T) // 4, np.float32)
        lib.gru_rnn_forward_stacked_carry(net.d, ptr(net.prepared), pins, [ptr(y_in) if start == 0 else None] * len(alive),
                                          [None if start == 0 else ptr(hs[i]) for i in alive], 1, T, clamp, [ptr(o) for o in wout],
                                          [ptr(h) for h in hl], ptr(ws), ws.nbytes, fl)
        assert lib.workspace_status(ptr(ws))[0] == 0
        for k, i in enumerate(alive):
            outs[i][start:start + fr[k]] = wout[k][:fr[k]]
            hs[i] = hl[k]
    return outs


@pytest.mark.parametrize("ncell", [2, 5])
def test_windows_of_an_utterance_reproduce_the_unbroken_pass_bit_for_bit(lib, ncell):
    """ABI 5: a pass over consecutive WINDOWS of the stacked utterances (cvae_pass_input::ctx_before / ctx_after: the conv front-end
    sees the neighbouring frames instead of zero padding; y_in = NULL + h_in: the window continues the recurrence; draw_frame0 /
    eps_draw_stride: the window's draws are the utterance's) must give exactly the unbroken pass -- for the word-exchange kernel
    (2 cells) and the dataflow kernel (5 cells), an encoder and a decoder with the 3-draw latent mean fused into its prologue,
    explicit eps and Philox draws.  This is what lets stage 6 run the decoder of window w beside the encoder of window w+1."""
    hidden, L = 64, 8
    lens = [23, 17, 20, 9, 23][:ncell]
    Tt = max(lens)
    P = tiny(B=ncell, T=Tt, hidden=hidden, tag="win%d" % ncell)
    enc, dec = NpNet(lib, P.enc, 6, 8, hidden), NpNet(lib, P.dec, 6, 4, hidden)
    fl = _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3
    xs = [np.ascontiguousarray(P.x[c, :lens[c]]) for c in range(ncell)]
    y_e = np.ascontiguousarray(P.y_in_enc.reshape(ncell, -1)[:1])
    y_d = np.ascontiguousarray(P.y_in_dec.reshape(ncell, -1)[:1])
    # unbroken encoder pass
    whole = [np.full((Tt, 8), np.nan, np.float32) for _ in range(ncell)]
    ws = np.zeros(lib.pass_workspace_bytes(enc.d, ncell, Tt) // 4, np.float32)
    lib.gru_rnn_forward_stacked(enc.d, ptr(enc.prepared), [lib.pass_input((ptr(xs[c]), 6, 6), frames=lens[c]) for c in range(ncell)],
                                [ptr(y_e)] * ncell, 1, Tt, 4, [ptr(o) for o in whole], ptr(ws), ws.nbytes, fl)
    cells = [{"n": lens[c], "pin": (lambda start, f, c=c: lib.pass_input((xs[c].ctypes.data + start * 6 * 4, 6, 6), frames=f, ctx_before=start,
                                                                         ctx_after=lens[c] - start - f))} for c in range(ncell)]
    win = _windowed_stacked(lib, enc, cells, y_e, L, Tt, 4, fl, 8, hidden)
    for c in range(ncell):
        assert np.array_equal(win[c], whole[c][:lens[c]]), ("encoder", c, float(np.abs(win[c] - whole[c][:lens[c]]).max()))
    # decoder on [code ; mean of 3 draws]: explicit eps (stride of the whole utterance), then Philox draws
    lat = [np.ascontiguousarray(whole[c][:lens[c]]) for c in range(ncell)]
    codes = np.eye(2, dtype=np.float32)
    eps = [np.ascontiguousarray(synth.normal("win/eps/%d" % c, (3, lens[c], 4))) for c in range(ncell)]
    for philox in (False, True):
        def dpin(c, start, f, n_total):
            return lib.pass_input((ptr(codes[c & 1]), 2, 0), lat=lat[c].ctypes.data + start * 8 * 4, lat_dim=4,
                                  eps=None if philox else eps[c].ctypes.data + start * 4 * 4, seed=77, draw_id=10 * c, frames=f, n_draws=3,
                                  ctx_before=start, ctx_after=lens[c] - start - f, draw_frame0=start, eps_draw_stride=lens[c] * 4)
        dwhole = [np.full((Tt, 4), np.nan, np.float32) for _ in range(ncell)]
        ws = np.zeros(lib.pass_workspace_bytes(dec.d, ncell, Tt) // 4, np.float32)
        lib.gru_rnn_forward_stacked(dec.d, ptr(dec.prepared), [dpin(c, 0, lens[c], lens[c]) for c in range(ncell)], [ptr(y_d)] * ncell, 1, Tt, -1,
                                    [ptr(o) for o in dwhole], ptr(ws), ws.nbytes, fl)
        dcells = [{"n": lens[c], "pin": (lambda start, f, c=c: dpin(c, start, f, lens[c]))} for c in range(ncell)]
        dwin = _windowed_stacked(lib, dec, dcells, y_d, L, Tt, -1, fl, 4, hidden)
        for c in range(ncell):
            assert np.isfinite(dwin[c]).all()
            assert np.array_equal(dwin[c], dwhole[c][:lens[c]]), ("decoder", philox, c, float(np.abs(dwin[c] - dwhole[c][:lens[c]]).max()))
        if not philox:      # and the unbroken pass is the oracle's
            z = np.mean(np.stack([orc.sampling_vae_batch(lat[0][None], eps[0][k][None], 4)[0] for k in range(3)]), 0)
            alone = orc.gru_rnn_forward(P.dec, np.concatenate([np.tile(codes[0], (lens[0], 1)), z], 1)[None], y_d[None])[0][0]
            assert maxabs(dwhole[0][:lens[0]], alone) <= 5e-5
    # a cell must either start (y_in) or continue (h_in)
    with pytest.raises(_cabi.CvaeError):
        ws = np.zeros(lib.pass_workspace_bytes(enc.d, 1, 4) // 4, np.float32)
        o = np.zeros((4, 8), np.float32)
        lib.gru_rnn_forward_stacked_carry(enc.d, ptr(enc.prepared), [lib.pass_input((ptr(xs[0]), 6, 6))], [None], [None], 1, 4, 4, [ptr(o)], [None],
                                          ptr(ws), ws.nbytes, fl)
    # a window with context needs single-row cells
    with pytest.raises(_cabi.CvaeError):
        bad = lib.pass_input((ptr(P.x), 6, 6), ctx_before=2)
        ws = np.zeros(lib.pass_workspace_bytes(enc.d, 2, 4) // 4, np.float32)
        o = np.zeros((2, 4, 8), np.float32)
        lib.gru_rnn_forward_stacked(enc.d, ptr(enc.prepared), [bad], [ptr(y_e)], 2, 4, 4, [ptr(o)], ptr(ws), ws.nbytes, fl)


@pytest.mark.parametrize("ncell", [2, 5])
def test_unfinished_cell_shorter_than_the_pass_carries_the_state_behind_its_own_frames(lib, ncell):
    """a window cell with frames < T and ctx_after > 0 -- an UNFINISHED row stacked beside a longer finishing row --
    keeps stepping to T over frames of its next window; its carried state must be the one behind ITS `frames` (slot frames of the
    trajectory, not slot T), so that continuing from frame `frames` reproduces the unbroken pass bit for bit.  Cells: 0 = 8 frames of
    an 18-frame utterance (continues), 1 = a 10-frame utterance that finishes -> T = 10.  Both kernels (word exchange: 2 cells; row
    tiles: 5 cells)."""
    hidden = 64
    lens = [18, 10, 18, 17, 18][:ncell]
    first = [8, 10, 8, 8, 9][:ncell]           # frames of the first window: every cell but 1 is unfinished and shorter than T = 10
    # (five cells: four continue, so that both windows run the row-tile kernel -- the two kernels differ in the last bit)
    Tt = max(lens)
    P = tiny(B=ncell, T=Tt, hidden=hidden, tag="short%d" % ncell)
    enc = NpNet(lib, P.enc, 6, 8, hidden)
    fl = _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3
    xs = [np.ascontiguousarray(P.x[c, :lens[c]]) for c in range(ncell)]
    y_e = np.ascontiguousarray(P.y_in_enc.reshape(ncell, -1)[:1])
    whole = [np.full((Tt, 8), np.nan, np.float32) for _ in range(ncell)]
    ws = np.zeros(lib.pass_workspace_bytes(enc.d, ncell, Tt) // 4, np.float32)
    lib.gru_rnn_forward_stacked(enc.d, ptr(enc.prepared), [lib.pass_input((ptr(xs[c]), 6, 6), frames=lens[c]) for c in range(ncell)],
                                [ptr(y_e)] * ncell, 1, Tt, 4, [ptr(o) for o in whole], ptr(ws), ws.nbytes, fl)

    def pin(c, start, f):
        return lib.pass_input((xs[c].ctypes.data + start * 6 * 4, 6, 6), frames=f, ctx_before=start, ctx_after=lens[c] - start - f)
    T1 = max(first)
    o1 = [np.full((T1, 8), np.nan, np.float32) for _ in range(ncell)]
    h1 = [np.full((1, hidden), np.nan, np.float32) for _ in range(ncell)]
    ws = np.zeros(lib.pass_workspace_bytes(enc.d, ncell, T1) // 4, np.float32)
    lib.gru_rnn_forward_stacked_carry(enc.d, ptr(enc.prepared), [pin(c, 0, first[c]) for c in range(ncell)], [ptr(y_e)] * ncell, [None] * ncell,
                                      1, T1, 4, [ptr(o) for o in o1], [ptr(h) for h in h1], ptr(ws), ws.nbytes, fl)
    assert lib.workspace_status(ptr(ws))[0] == 0
    cont = [c for c in range(ncell) if first[c] < lens[c]]
    T2 = max(lens[c] - first[c] for c in cont)
    o2 = [np.full((T2, 8), np.nan, np.float32) for _ in cont]
    ws = np.zeros(lib.pass_workspace_bytes(enc.d, len(cont), T2) // 4, np.float32)
    lib.gru_rnn_forward_stacked_carry(enc.d, ptr(enc.prepared), [pin(c, first[c], lens[c] - first[c]) for c in cont], [None] * len(cont),
                                      [ptr(h1[c]) for c in cont], 1, T2, 4, [ptr(o) for o in o2], [None] * len(cont), ptr(ws), ws.nbytes, fl)
    assert lib.workspace_status(ptr(ws))[0] == 0
    for c in range(ncell):
        assert np.array_equal(o1[c][:first[c]], whole[c][:first[c]]), ("first window", c)
    for k, c in enumerate(cont):
        n = lens[c] - first[c]
        assert np.array_equal(o2[k][:n], whole[c][first[c]:lens[c]]), ("continued", c, float(np.abs(o2[k][:n] - whole[c][first[c]:lens[c]]).max()))


def limb_selftest_values():
    x = (synth.normal("limbs/x", (4096,)) * np.exp2(synth.uniform01("limbs/e", (4096,)) * 16.0 - 12.0)).astype(np.float32)
    x[:8] = [0.0, 1.0, -1.0, 0.5, 3.14159274, -2.71828175, 1.0 + 2.0 ** -23, 0.99999994]
    # every binade from 2^-30 up to 2^11, both signs: the exact range and the range below it
    e = np.repeat(np.arange(-30, 12), 16).astype(np.float64)
    m = 1.0 + synth.uniform01("limbs/m", (e.size,))
    sgn = np.where(np.arange(e.size) % 2 == 0, 1.0, -1.0)
    return np.concatenate([x, (sgn * m * np.exp2(e)).astype(np.float32)])


def check_limb_transport(x, y):
    """The contract of the limb transport (two fp16 halves + a bf8 byte per value): BIT-EXACT for |x| >= 2^-16 (and 0), absolute
    error <= 2^-40 below (the third limb's bf8 form leaves its normal range there)."""
    big = (np.abs(x) >= 2.0 ** -16) | (x == 0)
    assert np.array_equal(x[big], y[big]), int((x[big] != y[big]).sum())
    err = np.abs(y[~big].astype(np.float64) - x[~big].astype(np.float64))
    assert err.size > 100 and float(err.max()) <= 2.0 ** -40, float(err.max())


def test_limb_transport_selftest(lib):
    """cvae_selftest_limbs on the host build: the producer's split (two halves + a bf8 byte) and the consumer's packed decode rebuild
    every value of at least 2^-16 bit for bit -- all eight positions of a group, not only the first four."""
    x = limb_selftest_values()
    y = np.zeros_like(x)
    lib.selftest_limbs(ptr(x), ptr(y), x.size)
    check_limb_transport(x, y)
    pos = np.arange(x.size) % 8
    big = np.abs(x) >= 2.0 ** -16
    for q in range(8):
        assert big[pos == q].sum() > 100 and np.array_equal(x[big & (pos == q)], y[big & (pos == q)]), q
