"""Pin the CPU oracle (oracle/cyclevae_oracle.py) against vectors recorded from the reference itself.

The goldens in tests/golden/*.npz were produced by tests/golden/make_golden.py running the imported
reference (torch 2.10 CPU, fp32).  Tolerances: the oracle and the reference sum the same fp32 products in
different orders (numpy/OpenBLAS vs ATen/oneDNN), so float results agree to fp32 re-association noise:
  single op / single pass : max|d| <= 2e-5
  10-pass cyc2 chain      : max|d| <= 2e-4   and   MCD <= 1e-3 dB (a tenth of the 0.01 dB budget)
INT bookkeeping must be bit-exact.
"""
import numpy as np
import pytest

import synth
from oracle import cyclevae_oracle as orc


def maxabs(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))


def mcd_db(a, b):
    return float(np.mean(orc.mcd_frames(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]), L2=True)))


def tiny_problem():
    return synth.CycleVAEProblem(B=2, T=12, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="tiny")


def test_weights_are_reproducible(golden):
    g = golden("tiny_ops")
    P = tiny_problem()
    assert synth.sha256_state(P.enc) == str(g["sha_enc"])
    assert synth.sha256_state(P.dec) == str(g["sha_dec"])


def test_tiny_ops(golden):
    g = golden("tiny_ops")
    P = tiny_problem()
    xc = orc.front_end(P.enc, P.x)
    assert maxabs(xc, g["xconv"]) <= 2e-5
    h0 = orc.gru_cell(P.enc, np.concatenate([xc[:, 0], P.y_in_enc[:, 0]], 1), np.zeros((2, 32), np.float32))
    assert maxabs(h0[None], g["h_step0"]) <= 2e-5
    y0 = h0 @ P.enc["out_1.weight"][:, :, 0].T + P.enc["out_1.bias"]
    assert maxabs(y0[:, None], g["y_step0"]) <= 2e-5
    lat, ly, lh = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, clamp_vae=True, lat_dim=4)
    assert maxabs(lat, g["lat"]) <= 2e-5 and maxabs(ly, g["lat_y"]) <= 2e-5 and maxabs(lh, g["lat_h"]) <= 2e-5
    assert maxabs(orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, lat_dim=4)[0], g["lat_noclamp"]) <= 2e-5
    z = orc.sampling_vae_batch(g["lat"], P.eps[0, 0], 4)
    assert maxabs(z, g["z"]) <= 2e-6
    rec, ry, rh = orc.gru_rnn_forward(P.dec, np.concatenate([P.code_src, g["z"]], 2), P.y_in_dec)
    assert maxabs(rec, g["rec"]) <= 2e-5 and maxabs(ry, g["rec_y"]) <= 2e-5 and maxabs(rh, g["rec_h"]) <= 2e-5
    tgt = P.x[0, :, P.stdim:]
    np.testing.assert_allclose(np.array(orc.twfse_loss(g["rec"][0], tgt, L2=False)), g["twfse_l1"], rtol=2e-6)
    np.testing.assert_allclose(np.array(orc.twfse_loss(g["rec"][0], tgt, L2=True)), g["twfse_l2"], rtol=2e-6)
    np.testing.assert_allclose(orc.loss_vae(g["lat"][0], 4), g["kl"], rtol=2e-6)


def test_clamp_is_active_somewhere(golden):
    g = golden("tiny_ops")
    # the clamp must be exercised by at least the equality lat >= floor; tiny nets rarely reach -13.8, so
    # force it: an oracle pass on shifted log-variances
    P = tiny_problem()
    sd = dict(P.enc)
    sd["out_1.bias"] = sd["out_1.bias"] - np.float32(20.0)
    lat = orc.gru_rnn_forward(sd, P.x, P.y_in_enc, clamp_vae=True, lat_dim=4)[0]
    assert np.all(lat[:, :, 4:] >= orc.LOG_VAR_FLOOR) and np.any(lat[:, :, 4:] == orc.LOG_VAR_FLOOR)
    assert np.any(lat[:, :, :4] < orc.LOG_VAR_FLOOR)   # means are never clamped
    assert g["lat"].shape == lat.shape


def test_tiny_chain(golden):
    g = golden("tiny_chain")
    P = tiny_problem()
    o = orc.cycle_chain(P.enc, P.dec, P.x, P.cvx, P.code_src, P.code_trg, P.y_in_enc, P.y_in_dec, P.eps, 2, 4)
    for k in ("lat", "rec", "cv", "latcv", "reccyc"):
        assert maxabs(np.stack(o[k]), g[k]) <= 2e-4, k


def test_full_pass_and_carry(golden):
    g = golden("full_pass")
    P = synth.CycleVAEProblem(B=2, T=80, bias_scale=0.05, tag="full")
    assert synth.sha256_state(P.enc) == str(g["sha_enc"]) and synth.sha256_state(P.dec) == str(g["sha_dec"])
    lat, ly, lh = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, clamp_vae=True, lat_dim=32)
    assert maxabs(lat, g["lat"]) <= 2e-5 and maxabs(ly, g["lat_y"]) <= 2e-5 and maxabs(lh, g["lat_h"]) <= 2e-5
    z = orc.sampling_vae_batch(g["lat"], P.eps[0, 0], 32)
    rec, ry, rh = orc.gru_rnn_forward(P.dec, np.concatenate([P.code_src, z], 2), P.y_in_dec)
    assert maxabs(rec, g["rec"]) <= 2e-5 and maxabs(ry, g["rec_y"]) <= 2e-5 and maxabs(rh, g["rec_h"]) <= 2e-5
    assert mcd_db(rec, g["rec"]) <= 1e-3
    lat2d = orc.gru_rnn_forward(P.enc, P.x[0], P.y_in_enc[:1], clamp_vae=True, lat_dim=32)[0]
    assert lat2d.shape == (80, 64) and maxabs(lat2d, g["lat2d"]) <= 2e-5
    a, ay, ah = orc.gru_rnn_forward(P.enc, P.x[:, :40], P.y_in_enc, clamp_vae=True, lat_dim=32)
    b, by, bh = orc.gru_rnn_forward(P.enc, P.x[:, 40:], ay, h_in=ah, clamp_vae=True, lat_dim=32)
    assert maxabs(a, g["carry_a"]) <= 2e-5 and maxabs(b, g["carry_b"]) <= 2e-5
    assert maxabs(by, g["carry_by"]) <= 2e-5 and maxabs(bh, g["carry_bh"]) <= 2e-5
    # windowed != whole-utterance from frame 80-4 on (conv re-padded per window)
    assert maxabs(a[:, :36], g["lat"][:, :36]) <= 2e-5 and maxabs(a[:, 36:], g["lat"][:, 36:40]) > 1e-3


def test_full_chain(golden):
    g = golden("full_chain")
    P = synth.CycleVAEProblem(B=2, T=80, bias_scale=0.05, tag="full")
    o = orc.cycle_chain(P.enc, P.dec, P.x, P.cvx, P.code_src, P.code_trg, P.y_in_enc, P.y_in_dec, P.eps, 2, 32)
    for k in ("lat", "rec", "cv", "latcv", "reccyc"):
        assert maxabs(np.stack(o[k]), g[k]) <= 2e-4, k
    for k in ("rec", "cv", "reccyc"):
        assert mcd_db(np.stack(o[k]), g[k]) <= 1e-3, k


def test_stress_pass(golden):
    g = golden("stress_pass")
    P = synth.CycleVAEProblem(B=1, T=16, in_dim=54, out_dim=50, lat_dim=64, hidden=2048, n_cyc=1, bias_scale=0.05, tag="stress")
    lat, ly, lh = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, clamp_vae=True, lat_dim=64)
    assert maxabs(lat, g["lat"]) <= 2e-5 and maxabs(lh, g["lat_h"]) <= 2e-5
    z = orc.sampling_vae_batch(g["lat"], P.eps[0, 0], 64)
    rec = orc.gru_rnn_forward(P.dec, np.concatenate([P.code_src, z], 2), P.y_in_dec)[0]
    assert maxabs(rec, g["rec"]) <= 2e-5


def test_stage6(golden):
    g = golden("stage6")
    P = synth.CycleVAEProblem(B=1, T=203, bias_scale=0.05, tag="st6")
    eps = synth.normal("st6/eps_dec", (5, 203, 32))
    lat, zbar, cv = orc.stage6_convert(P.enc, P.dec, P.x[0], P.y_in_enc, P.y_in_dec, eps, 32)
    assert cv.dtype == np.float64 and cv.shape == (203, 50)
    assert maxabs(lat, g["lat"]) <= 2e-5 and maxabs(zbar, g["lat_feat"]) <= 2e-5
    assert maxabs(cv, g["cvmcep"]) <= 5e-5 and mcd_db(cv, g["cvmcep"]) <= 1e-3


def _int_cases():
    r = lambda a, b: list(range(a, b + 1))
    return [
        ([205, 170, 90], [r(10, 69) + r(85, 159) + r(161, 199), r(5, 59) + r(100, 164), r(82, 87)]),
        ([80, 79, 81, 1], [r(0, 79), r(3, 70), r(79, 80), [0]]),
        ([637, 400, 12], [r(30, 600), r(0, 79) + r(81, 159) + r(320, 399), r(2, 9)]),
    ]


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_int_window_bookkeeping_bit_exact(golden, ci):
    g = golden("int_windows")
    flens, spcs = _int_cases()[ci]
    U = len(flens)
    spc = np.zeros((U, 2200), np.int64)
    for j, s in enumerate(spcs):
        spc[j, :len(s)] = s
    wins = orc.window_bookkeeping(np.array(flens), spc, np.array([len(s) for s in spcs]), 80)
    assert len(wins) == g["c%d_se" % ci].shape[0]
    for w, se, si, ei, fa, sel in zip(wins, g["c%d_se" % ci], g["c%d_s_idx" % ci], g["c%d_e_idx" % ci],
                                      g["c%d_flen_acc" % ci], g["c%d_select" % ci]):
        assert (w["s"], w["e"]) == tuple(se)
        assert np.array_equal(w["s_idx"], si) and np.array_equal(w["e_idx"], ei)
        assert np.array_equal(w["flen_acc"], fa)
        assert [j for j in range(U) if sel[j]] == w["select_utt_idx"]


def test_int_worked_example_matches_survey(golden):
    """SURVEY App. B.1 table."""
    g = golden("int_windows")
    assert g["c0_se"].tolist() == [[0, 79], [80, 159], [160, 204]]
    assert g["c0_s_idx"].tolist() == [[0, 0, -1], [60, 55, 0], [135, 115, 0]]
    assert g["c0_e_idx"].tolist() == [[59, 54, -1], [134, 114, 5], [173, 119, 5]]
    assert g["c0_flen_acc"].tolist() == [[80, 80, 80], [80, 80, 10], [80, 10, 10]]
    assert g["c0_select"].tolist() == [[1, 1, 1], [1, 1, 1], [1, 1, 0]]


def test_torch_stock_baseline_matches_golden(golden):
    """The stock-torch composition timed as bench.py's cpu_baseline reproduces the reference's chain."""
    import torch
    from oracle import torch_stock as ts
    g = golden("full_chain")
    P = synth.CycleVAEProblem(B=2, T=80, bias_scale=0.05, tag="full")
    enc, dec = ts.StockGRURNN(P.enc, 54, 64, 1024), ts.StockGRURNN(P.dec, 34, 50, 1024)
    t = torch.from_numpy
    o = ts.cycle_chain(enc, dec, t(P.x), t(P.cvx), t(P.code_src), t(P.code_trg), t(P.y_in_enc), t(P.y_in_dec), t(P.eps), 2, 32)
    for k in ("lat", "rec", "cv", "latcv", "reccyc"):
        assert maxabs(np.stack([v.numpy() for v in o[k]]), g[k]) <= 2e-5, k


@pytest.mark.parametrize("tag,hid,B,T", [("train_h32", 32, 3, 10), ("train_h64", 64, 18, 7)])
def test_torch_stock_train_pass_matches_reference_gradients(golden, tag, hid, B, T):
    """The differentiable checker used for the HIP backward reproduces the reference's train-mode outputs and gradients
    when fed the dropout masks the reference drew."""
    import torch
    from oracle import torch_stock as ts
    g = golden(tag)
    P = synth.CycleVAEProblem(B=B, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=hid, n_cyc=1, bias_scale=0.1, tag=tag)
    x_dec = np.concatenate([P.code_src, synth.normal(tag + "/z", (B, T, 4))], 2)
    x2 = synth.features(tag + "/x2", B, T, P.mu, P.sigma)
    cases = [("enc", P.enc, P.x, P.y_in_enc, None, 4), ("dec", P.dec, x_dec, P.y_in_dec, None, -1),
             ("enc2", P.enc, x2, g["enc_y_last"], g["enc_h_last"], 4)]
    for name, sd, x, y_in, h_in, clamp in cases:
        o = 8 if name.startswith("enc") else 4
        cot = synth.normal(tag + "/cot_" + name.replace("2", ""), (B, T, o))
        out, y, h, Pm, xt = ts.train_forward(sd, x, y_in, h_in, g[name + "_cmask"], g[name + "_gmask"], clamp)
        (out * torch.from_numpy(cot)).sum().backward()
        assert maxabs(out.detach().numpy(), g[name + "_out"]) <= 2e-5
        assert maxabs(h.detach().numpy()[None], g[name + "_h_last"]) <= 2e-5
        assert maxabs(xt.grad.numpy(), g[name + "_dx"]) <= 5e-5 * max(1.0, float(np.abs(g[name + "_dx"]).max()))
        for k in ("conv.conv.0.weight", "conv.conv.1.weight", "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0",
                  "gru.bias_hh_l0", "out_1.weight", "out_1.bias", "conv.conv.0.bias", "conv.conv.1.bias"):
            ref = g[name + "_g_" + k]
            assert maxabs(Pm[k].grad.numpy(), ref) <= 5e-5 * max(1.0, float(np.abs(ref).max())), (name, k)


@pytest.mark.parametrize("fixture", ["stage4_step", "stage4_step_cyc4"])
@pytest.mark.parametrize("stack", [False, True], ids=["ten_passes", "rec_cv_stacked"])
def test_stage4_step_code_vs_reference_recorded_step(golden, stack, fixture):
    """stage4.chain_loss (the package's stage-4 step: flen_acc / select_utt_idx masking, the train...:1393 concat, windows
    continued from detached (y_last, h) carries) driven with the stock-torch checker must reproduce the two consecutive steps
    that the REFERENCE'S OWN statements executed (tests/golden/stage4_step.npz: loss, every gradient of the first step, gradient
    norms and post-Adam weight checksums of both); stage4_step_cyc4.npz: the same with n_cyc = 4, the cycle count of BASELINE
    configs[4] (8 encoder + 12 decoder passes per window)."""
    import torch
    import train_util
    from oracle import torch_stock as ts
    g = golden(fixture)
    P, x, cvx = train_util.golden_step_problem(g)
    leaf = {k: {n: torch.from_numpy(v.copy()).requires_grad_(n in train_util.TRAINABLE) for n, v in sd.items()}
            for k, sd in (("enc", P.enc), ("dec", P.dec))}
    opt = torch.optim.Adam([leaf[k][n] for k in ("enc", "dec") for n in train_util.TRAINABLE], lr=1e-4)

    def run_pass(kind, xin, y_in, clamp, mk, h_in=None):
        return ts.train_forward_t(leaf[kind.rstrip("2")], xin, y_in, torch.from_numpy(mk[0]), torch.from_numpy(mk[1]), clamp, h_in, True)

    names = {"lat": "batch_lat_src", "rec": "batch_trj_src_src", "cv": "batch_trj_src_trg", "latcv": "batch_lat_src_trg",
             "reccyc": "batch_trj_src_trg_src"}
    for w, loss, trajs in train_util.run_golden_windows(g, P, x, cvx, run_pass, opt, torch.device("cpu"), stack):
        for i in range(P.n_cyc):
            for k, gk in names.items():
                assert np.max(np.abs(trajs[i][k].detach().numpy() - g["w%d_%s" % (w, gk)][i])) <= 2e-5, (w, i, k)
        assert abs(loss.item() - float(g["w%d_loss" % w])) <= 2e-6 * abs(float(g["w%d_loss" % w])), (w, loss.item())
        for kind in ("enc", "dec"):
            for n in train_util.TRAINABLE:
                gr = leaf[kind][n].grad.numpy().astype(np.float64)
                ref_norm = float(g["w%d_%s_gnorm_%s" % (w, kind, n)])
                assert abs(np.sqrt((gr ** 2).sum()) - ref_norm) <= 1e-4 * ref_norm, (w, kind, n)
                ref = g["w%d_%s_g_%s" % (w, kind, n)]          # every gradient tensor of both windows
                assert np.max(np.abs(gr - ref)) <= 1e-4 * max(1e-6, np.max(np.abs(ref))), (w, kind, n)
    for kind in ("enc", "dec"):          # weights after both Adam steps
        for n in train_util.TRAINABLE:
            v = leaf[kind][n].detach().numpy().astype(np.float64)
            ref = g["w1_%s_after_%s" % (kind, n)]
            got = np.array([v.sum(), (v * v).sum(), v.ravel()[0], v.ravel()[-1]])
            assert np.allclose(got, ref, rtol=1e-5, atol=1e-7), (kind, n, got, ref)


def test_gv_postfilter_restatement_vs_reference_statements(golden):
    """oracle.gv_postfilter against the output of the reference's own lines decode...:419-422 (tests/golden/gv_postfilter.npz)."""
    g = golden("gv_postfilter")
    T, D = 211, 50
    c = (synth.normal("gvpin/c", (T, D)) * np.linspace(2.0, 0.1, D)).astype(np.float32)
    gv_t = (0.05 + synth.uniform01("gvpin/gv", (D - 1,))).astype(np.float64)
    cg = (0.02 + 0.5 * synth.uniform01("gvpin/cg", (D - 1,))).astype(np.float64)
    out, var = orc.gv_postfilter(c, gv_t, cg)
    assert np.array_equal(out, g["cvmcep_gv"]) and np.array_equal(var, g["cvgv"])


def test_stress_cyc4_chain_oracle_vs_reference(golden):
    """BASELINE configs[4] dims (hu2048 / ld64 / n_cyc = 4): the numpy restatement against the reference's own 20-pass chain."""
    g = golden("stress_chain")
    P = synth.CycleVAEProblem(B=2, T=16, in_dim=54, out_dim=50, lat_dim=64, hidden=2048, n_cyc=4, bias_scale=0.05, tag="stress4")
    out = orc.cycle_chain(P.enc, P.dec, P.x, P.cvx, P.code_src, P.code_trg, P.y_in_enc, P.y_in_dec, P.eps, 4, 64)
    for k in ("lat", "rec", "cv", "latcv", "reccyc"):
        assert np.max(np.abs(np.stack(out[k]) - g[k])) <= 2e-4, k


def test_mc2e_restatement_satisfies_its_definitions():
    """oracle.mc2e (SPTK mc2e of the reference's mod_pow, feature_extract_vc.py:131-138; pysptk is absent: parity unpinned) is
    held to the identities that define it: with alpha = 0 the impulse response is exp() of the cepstral power series, and a
    c0-only cepstrum has energy exp(2 c0) for any alpha."""
    c = np.array([0.1, 0.2, -0.05, 0.03])
    L = 32
    q = np.zeros(L)
    q[1:4] = c[1:]
    p, term = np.zeros(L), np.zeros(L)
    p[0] = term[0] = 1.0
    for n in range(1, 40):
        term = np.convolve(term, q)[:L] / n
        p += term
    assert abs(orc.mc2e(c[None], alpha=0.0, irlen=L)[0] - np.sum((np.exp(c[0]) * p) ** 2)) <= 1e-13
    for alpha in (0.0, 0.455):
        assert abs(orc.mc2e(np.array([[0.3] + [0.0] * 9]), alpha=alpha, irlen=64)[0] - np.exp(0.6)) <= 1e-12
    a = synth.normal("mc2e/a", (3, 50)).astype(np.float64) * np.linspace(1.0, 0.05, 50)
    d = orc.mod_pow_dpow(a, a * 1.0)
    assert np.max(np.abs(d)) == 0.0
