"""Parity of the HIP path (through gru_vae.py -> ctypes -> libcyclevae_hip.so) against the oracle and the
goldens recorded from the reference.  Run on the GPU box:  python -m pytest tests -m gpu

Tolerances (fp32 everywhere; differences are summation order plus the fp64-computed load-time folds).  Two levels, both asserted:
  BUDGET  the north star's acceptance line: MCD <= 0.01 dB against the CPU path (d=0..49 and d=1..49)
  TIGHT   regression grade, a few times what the kernels deliver on the MI355X (measured: max|d| 2e-7..2.3e-6 for a pass,
          <= 7.2e-7 over a cyc2 chain, MCD 3e-6..6.6e-6 dB, kernel vs kernel <= 8.3e-7): a 2^-22-level operand error -- the
          limb-decode bug of round 2 that every MCD-budget test passed -- moves max|d| by ~1e-5 and fails these
  frame indexing        exact: T_out == T_in, row b / frame t of the output belongs to row b / frame t of the input
A short report of every measured difference is appended to gpurun_out/gpu_parity_report.txt.
"""
import os

import numpy as np
import pytest

import synth
from oracle import cyclevae_oracle as orc

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "gpu_parity_report.txt")

BUDGET_MCD = 0.01       # dB, BASELINE.json north_star
TIGHT_PASS = 5e-6       # max|d| of one pass (any length up to 1500 frames) against the golden / the oracle
TIGHT_CHAIN = 5e-6      # max|d| of any trajectory of a cyc2 / cyc4 chain
TIGHT_MCD = 5e-5        # dB
TIGHT_KERNELS = 3e-6    # max|d| between two forms of the recurrent kernel on the same operands


def note(msg):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(msg + "\n")
    print(msg)


def maxabs(a, b, name=""):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert np.all(np.isfinite(a)), name + ": non-finite output"
    d = float(np.max(np.abs(a.astype(np.float64) - b.astype(np.float64))))
    note("%-40s max|d| = %.3e" % (name, d))
    return d


def mcd_db(a, b, lo=0):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    return float(np.mean(orc.mcd_frames(a.reshape(-1, a.shape[-1])[:, lo:], b.reshape(-1, b.shape[-1])[:, lo:], L2=True)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gv():
    import gru_vae
    return gru_vae


def module(gv, sd, i, o, h, enc, dev):
    m = gv.GRU_RNN(in_dim=i, out_dim=o, hidden_units=h, kernel_size=3, dilation_size=2, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval()


def T_(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_native_library_is_the_path(gv, dev):
    lib = gv._lib()
    assert lib.path.endswith("libcyclevae_hip.so") and os.path.exists(lib.path)
    with pytest.raises(RuntimeError):
        m = gv.GRU_RNN(in_dim=6, out_dim=8, hidden_units=32, scale_out_flag=False)
        with torch.no_grad():
            m(torch.zeros(2, 5, 6), torch.zeros(2, 1, 8))     # CPU tensors: no fallback


def test_tiny_ops_vs_golden(gv, dev, golden, monkeypatch):
    g = golden("tiny_ops")
    P = synth.CycleVAEProblem(B=2, T=12, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="tiny")
    enc, dec = module(gv, P.enc, 6, 8, 32, True, dev), module(gv, P.dec, 6, 4, 32, False, dev)
    for persist in (True, False):
        if not persist:
            monkeypatch.setattr(gv, "_persistent", False)
        with torch.no_grad():
            lat, ly, lh = enc(T_(P.x, dev), T_(P.y_in_enc, dev), clamp_vae=True, lat_dim=4)
            z = gv.sampling_with_eps(lat, T_(P.eps[0, 0], dev), 4)
            rec, ry, rh = dec(torch.cat((T_(P.code_src, dev), z), 2), T_(P.y_in_dec, dev))
        tag = "tiny[persist=%d] " % persist
        assert lat.shape == (2, 12, 8) and ly.shape == (2, 1, 8) and lh.shape == (1, 2, 32)
        assert maxabs(lat, g["lat"], tag + "lat") <= TIGHT_PASS and maxabs(ly, g["lat_y"], tag + "lat_y") <= TIGHT_PASS
        assert maxabs(lh, g["lat_h"], tag + "lat_h") <= TIGHT_PASS and maxabs(z, g["z"], tag + "z") <= TIGHT_PASS
        assert maxabs(rec, g["rec"], tag + "rec") <= TIGHT_PASS and maxabs(ry, g["rec_y"], tag + "rec_y") <= TIGHT_PASS
        assert maxabs(rh, g["rec_h"], tag + "rec_h") <= TIGHT_PASS
    with torch.no_grad():
        a, b = T_(g["rec"][0], dev), T_(P.x[0, :, P.stdim:], dev)
        crit = gv.TWFSEloss()
        l1 = np.array([v.item() for v in crit(a, b, L2=False, GV=False)])
        l2 = np.array([v.item() for v in crit(a, b, L2=True, GV=False)])
        kl = gv.loss_vae(T_(g["lat"][0], dev), lat_dim=4).item()
    np.testing.assert_allclose(l1, g["twfse_l1"], rtol=1e-5)
    np.testing.assert_allclose(l2, g["twfse_l2"], rtol=1e-5)
    np.testing.assert_allclose(kl, g["kl"], rtol=1e-5)


def test_full_pass_2d_and_carry_vs_golden(gv, dev, golden):
    g = golden("full_pass")
    P = synth.CycleVAEProblem(B=2, T=80, bias_scale=0.05, tag="full")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    with torch.no_grad():
        lat, ly, lh = enc(T_(P.x, dev), T_(P.y_in_enc, dev), clamp_vae=True, lat_dim=32)
        z = gv.sampling_with_eps(T_(g["lat"], dev), T_(P.eps[0, 0], dev), 32)
        rec, ry, rh = dec(torch.cat((T_(P.code_src, dev), z), 2), T_(P.y_in_dec, dev))
        lat2d = enc(T_(P.x[0], dev), T_(P.y_in_enc[:1], dev), clamp_vae=True, lat_dim=32)[0]
        a, ay, ah = enc(T_(P.x[:, :40], dev), T_(P.y_in_enc, dev), clamp_vae=True, lat_dim=32)
        b, by, bh = enc(T_(P.x[:, 40:], dev), ay, h_in=ah, clamp_vae=True, lat_dim=32)
    assert maxabs(lat, g["lat"], "full lat") <= TIGHT_PASS and maxabs(ly, g["lat_y"], "full lat_y") <= TIGHT_PASS
    assert maxabs(lh, g["lat_h"], "full lat_h") <= TIGHT_PASS
    assert maxabs(rec, g["rec"], "full rec") <= TIGHT_PASS and maxabs(ry, g["rec_y"], "full rec_y") <= TIGHT_PASS
    assert maxabs(rh, g["rec_h"], "full rec_h") <= TIGHT_PASS
    assert lat2d.shape == (80, 64) and maxabs(lat2d, g["lat2d"], "full lat2d") <= TIGHT_PASS
    assert maxabs(a, g["carry_a"], "carry a") <= TIGHT_PASS and maxabs(b, g["carry_b"], "carry b") <= TIGHT_PASS
    assert maxabs(by, g["carry_by"], "carry by") <= TIGHT_PASS and maxabs(bh, g["carry_bh"], "carry bh") <= TIGHT_PASS
    m = mcd_db(rec, g["rec"])
    note("full rec MCD vs reference = %.3e dB" % m)
    assert m <= BUDGET_MCD
    assert m <= TIGHT_MCD


def test_full_chain_vs_golden(gv, dev, golden, monkeypatch):
    g = golden("full_chain")
    P = synth.CycleVAEProblem(B=2, T=80, bias_scale=0.05, tag="full")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    args = [T_(v, dev) for v in (P.x, P.cvx, P.code_src, P.code_trg, P.y_in_enc, P.y_in_dec)]
    res = {}
    for persist in (True, False):
        if not persist:
            monkeypatch.setattr(gv, "_persistent", False)
        with torch.no_grad():
            out = chain(*args, eps=T_(P.eps, dev))
        torch.cuda.synchronize()
        assert chain.status()[0] == 0, "grid barrier timed out"
        for k in ("lat", "rec", "cv", "latcv", "reccyc"):
            assert maxabs(out[k], g[k], "chain[persist=%d] %s" % (persist, k)) <= TIGHT_CHAIN
        for k in ("rec", "cv", "reccyc"):
            m0, m1 = mcd_db(out[k], g[k], 0), mcd_db(out[k], g[k], 1)
            note("chain[persist=%d] %-7s MCD = %.3e dB (0..49)  %.3e dB (1..49)" % (persist, k, m0, m1))
            assert m0 <= BUDGET_MCD and m1 <= BUDGET_MCD
            assert m0 <= TIGHT_MCD and m1 <= TIGHT_MCD
        res[persist] = {k: v.cpu().numpy() for k, v in out.items()}
    # the cooperative one-launch recurrence (hardware-exp gates) and the per-step launches (libm gates) agree to rounding
    for k in res[True]:
        assert float(np.max(np.abs(res[True][k] - res[False][k]))) <= 2e-5, k


def test_stress_dims_vs_golden(gv, dev, golden):
    g = golden("stress_pass")
    P = synth.CycleVAEProblem(B=1, T=16, in_dim=54, out_dim=50, lat_dim=64, hidden=2048, n_cyc=1, bias_scale=0.05, tag="stress")
    enc, dec = module(gv, P.enc, 54, 128, 2048, True, dev), module(gv, P.dec, 66, 50, 2048, False, dev)
    with torch.no_grad():
        lat, ly, lh = enc(T_(P.x, dev), T_(P.y_in_enc, dev), clamp_vae=True, lat_dim=64)
        z = gv.sampling_with_eps(T_(g["lat"], dev), T_(P.eps[0, 0], dev), 64)
        rec = dec(torch.cat((T_(P.code_src, dev), z), 2), T_(P.y_in_dec, dev))[0]
    assert maxabs(lat, g["lat"], "stress lat") <= TIGHT_PASS and maxabs(lh, g["lat_h"], "stress lat_h") <= TIGHT_PASS
    assert maxabs(rec, g["rec"], "stress rec") <= TIGHT_PASS


def test_stage6_path_vs_golden(gv, dev, golden):
    """decode_gru-cyclevae_gauss.py:302-319 on the drop-in modules: 2-D encoder input, n-draw mean, decoder, float64."""
    g = golden("stage6")
    T, nd = 203, 5
    P = synth.CycleVAEProblem(B=1, T=T, bias_scale=0.05, tag="st6")
    eps = synth.normal("st6/eps_dec", (nd, T, 32))
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    with torch.no_grad():
        lat = enc(T_(P.x[0], dev), T_(P.y_in_enc, dev), clamp_vae=True, lat_dim=32)[0]
        lf = torch.mean(gv.sampling_with_eps(lat.unsqueeze(0).repeat(nd, 1, 1), T_(eps, dev), 32), 0)
        code = torch.zeros(T, 2, device=dev)
        code[:, 1] = 1
        cv = dec(torch.cat((code, lf), 1), T_(P.y_in_dec, dev))[0]
        cv64 = np.array(cv.cpu().data.numpy(), dtype=np.float64)
    assert cv64.shape == (T, 50)
    assert maxabs(lat, g["lat"], "stage6 lat") <= TIGHT_PASS and maxabs(lf, g["lat_feat"], "stage6 lat_feat") <= TIGHT_PASS
    assert maxabs(cv64, g["cvmcep"], "stage6 cvmcep") <= TIGHT_PASS
    m = mcd_db(cv64, g["cvmcep"])
    note("stage6 cvmcep MCD = %.3e dB" % m)
    assert m <= BUDGET_MCD
    assert m <= TIGHT_MCD


def test_headline_size_against_oracle_and_row_independence(gv, dev):
    """B=64, T=80, hu1024/ld32/cyc2 (BASELINE config 2): whole-chain parity with the oracle, plus properties that
    do not need an oracle -- every utterance row is independent of its batch mates, frame order is preserved."""
    P = synth.CycleVAEProblem(B=64, T=80, bias_scale=0.0, tag="bench")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    names = ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")
    full = [T_(getattr(P, n), dev) for n in names]
    with torch.no_grad():
        out = chain(*full, eps=T_(P.eps, dev))
        torch.cuda.synchronize()
        assert chain.status()[0] == 0
        rows = list(range(5, 24)) + [41, 63]    # 21 rows: more than 16, so the sub-batch runs the same (32-row tile) kernel
        sub = chain(*[v[rows] for v in full], eps=T_(P.eps[:, :, rows], dev))
    for k in out:
        assert out[k].shape[1:3] == (64, 80)
        assert torch.equal(out[k][:, rows], sub[k]), "row independence broken for " + k
    ref = orc.cycle_chain(P.enc, P.dec, P.x[:8], P.cvx[:8], P.code_src[:8], P.code_trg[:8], P.y_in_enc[:8],
                          P.y_in_dec[:8], P.eps[:, :, :8], 2, 32)
    for k in ref:
        assert maxabs(out[k][:, :8], np.stack(ref[k]), "headline %s (rows 0..7)" % k) <= TIGHT_CHAIN
    for k in ("rec", "cv", "reccyc"):
        m = mcd_db(out[k][:, :8], np.stack(ref[k]))
        note("headline %-7s MCD vs oracle = %.3e dB" % (k, m))
        assert m <= BUDGET_MCD
        assert m <= TIGHT_MCD
    # frame order: reversing nothing but reading frame t must equal a run truncated to t+5 frames up to frame t-... (conv sees +-4)
    with torch.no_grad():
        short = enc(full[0][:4, :40], full[4][:4], clamp_vae=True, lat_dim=32)[0]
        long_ = enc(full[0][:4], full[4][:4], clamp_vae=True, lat_dim=32)[0]
    assert torch.equal(short[:, :36], long_[:, :36])     # frames < 40-4 cannot see the truncation
    assert not torch.equal(short[:, 36:40], long_[:, 36:40])


def test_long_utterance_single_row(gv, dev):
    """Stage-5/6 shape: ONE utterance of 1500 frames through the hu1024 encoder and decoder (2-D input path), against the
    oracle; 1500 dependent steps also exercise the flag counters far beyond the 80-frame windows of training."""
    P = synth.CycleVAEProblem(B=1, T=1500, bias_scale=0.0, tag="longutt")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    with torch.no_grad():
        lat, ylast, h = enc(T_(P.x[0], dev), T_(P.y_in_enc, dev), clamp_vae=True, lat_dim=32)
        dec_in = torch.cat((T_(P.code_src[0], dev), lat[:, :32]), 1)
        rec, _, _ = dec(dec_in, T_(P.y_in_dec, dev))
        torch.cuda.synchronize()
    assert lat.shape == (1500, 64) and rec.shape == (1500, 50)
    r_lat, r_y, r_h = orc.gru_rnn_forward(P.enc, P.x[0], P.y_in_enc, clamp_vae=True, lat_dim=32)
    assert maxabs(lat, r_lat, "long utterance enc trj (T=1500)") <= TIGHT_PASS
    assert maxabs(ylast, r_y, "long utterance enc y_last") <= TIGHT_PASS
    assert maxabs(h, r_h, "long utterance enc h_last") <= TIGHT_PASS
    r_rec, _, _ = orc.gru_rnn_forward(P.dec, np.concatenate((P.code_src[0], r_lat[:, :32]), 1), P.y_in_dec)
    assert maxabs(rec, r_rec, "long utterance dec trj (T=1500)") <= TIGHT_PASS
    m = mcd_db(rec, r_rec)
    note("long utterance MCD vs oracle = %.3e dB" % m)
    assert m <= BUDGET_MCD
    assert m <= TIGHT_MCD


def test_stage6_postprocessing_on_device(gv, dev):
    """GV post-filter and aligned frame-wise MCD on the decoder output without leaving the device, f64,
    against the numpy restatement of decode_gru-cyclevae_gauss.py:377-378 / :417-421."""
    import stage6
    T, D = 1501, 50
    c = (synth.normal("s6/c", (T, D)) * np.linspace(2.0, 0.1, D)).astype(np.float32)
    ref_c = (c + 0.03 * synth.normal("s6/n", (T, D))).astype(np.float32)
    gv_t = (0.05 + synth.uniform01("s6/gv", (D - 1,))).astype(np.float64)
    cg = (0.02 + 0.5 * synth.uniform01("s6/cg", (D - 1,))).astype(np.float64)
    dp = (0.1 * synth.normal("s6/dp", (T,))).astype(np.float64)
    out, var = stage6.gv_postfilter(T_(c, dev), gv_t, cg, dpow=dp)
    r_out, r_var = orc.gv_postfilter(c, gv_t, cg, dp)
    assert out.dtype == torch.float64 and out.is_cuda
    d1 = float(np.max(np.abs(out.cpu().numpy() - r_out) / np.maximum(1.0, np.abs(r_out))))
    d2 = float(np.max(np.abs(var.cpu().numpy() - r_var) / np.abs(r_var)))
    note("stage6 gv_postfilter  rel|d| = %.3e (out), %.3e (var)" % (d1, d2))
    assert d1 <= 1e-12 and d2 <= 1e-11
    for d0 in (0, 1):
        frames, stats = stage6.mcd_aligned(T_(c, dev), T_(ref_c, dev), d0=d0, L2=True)
        r_frames, r_mean, r_std = orc.mcd_aligned(c, ref_c, d0, True)
        d3 = float(np.max(np.abs(frames.cpu().numpy() - r_frames) / r_frames))
        st = stats.cpu().numpy()
        note("stage6 mcd_aligned d0=%d rel|d| = %.3e (frames); mean %.6f dB vs %.6f" % (d0, d3, st[1], r_mean))
        assert d3 <= 1e-13 and abs(st[1] - r_mean) <= 1e-12 * r_mean and abs(st[2] - r_std) <= 1e-10 * r_std
    with pytest.raises(RuntimeError):
        stage6.mcd_aligned(torch.zeros(3, 5), torch.zeros(3, 5))      # CPU tensors: no fallback


@pytest.mark.parametrize("B", [64, 8])
def test_three_recurrent_kernels_agree(gv, dev, monkeypatch, B):
    """The three forms of the persistent recurrent kernel on the headline shape: exact3 (default: fp32 operands carried exactly
    as three fp16 limbs, six f16 MFMAs per product, k_gru_steps_v6), split2 (22-bit fp16 pairs, k_gru_steps_v5) and fp32
    (v_mfma_f32_16x16x4_f32, k_gru_steps_v4).  All must sit at the same distance from the oracle; exact3 and fp32 multiply the
    same fp32 operands exactly, so they differ by summation order only and must be closer to each other than split2 is.
    B = 8 (the recipe's batch_size_utt, run.sh:173): passes of 4..16 rows take the SAME exact-operand kernel (one half-empty
    32-row tile), not a narrower one."""
    P = synth.CycleVAEProblem(B=B, T=80, bias_scale=0.0, tag="bench")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    full = [T_(getattr(P, n), dev) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    ref = orc.cycle_chain(P.enc, P.dec, P.x[:4], P.cvx[:4], P.code_src[:4], P.code_trg[:4], P.y_in_enc[:4], P.y_in_dec[:4],
                          P.eps[:, :, :4], 2, 32)
    outs = {}
    for name in ("exact3", "split2", "fp32"):
        monkeypatch.setattr(gv, "_force_kernel", name)
        with torch.no_grad():
            outs[name] = chain(*full, eps=T_(P.eps, dev))
            torch.cuda.synchronize()
        assert chain.status()[0] == 0
        worst = max(mcd_db(outs[name][k][:, :4], np.stack(ref[k])) for k in ("rec", "cv", "reccyc"))
        dmax = max(maxabs(outs[name][k][:, :4], np.stack(ref[k]), "%s %s" % (name, k)) for k in ref)
        note("recurrence %-7s: MCD vs oracle %.3e dB, max|d| %.3e" % (name, worst, dmax))
        assert worst <= BUDGET_MCD and worst <= TIGHT_MCD and dmax <= TIGHT_CHAIN
    dist = {}
    for a, b in (("exact3", "fp32"), ("split2", "fp32"), ("exact3", "split2")):
        assert not torch.equal(outs[a]["reccyc"], outs[b]["reccyc"])       # they really are different kernels
        dist[a, b] = max(float((outs[a][k] - outs[b][k]).abs().max()) for k in outs[a])
        note("%s vs %s: max|d| over all 10 trajectories = %.3e" % (a, b, dist[a, b]))
        assert dist[a, b] <= TIGHT_KERNELS
    assert dist["exact3", "fp32"] <= 1.5 * dist["split2", "fp32"] + 1e-7


@pytest.mark.parametrize("B", [200, 512])
def test_many_row_tiles_per_block(gv, dev, B):
    """B=200 rows = 7 row tiles of 32 on 2 block rows (up to 4 tiles per block), B=512 = the whole batch of BASELINE configs[3] in ONE
    launch (16 tiles, 8 per block): a thread's previous h comes back from the exchange buffer instead of a register; every row must
    still equal its small-batch result to rounding."""
    P = synth.CycleVAEProblem(B=B, T=12, bias_scale=0.0, tag="manytiles%d" % B)
    enc = module(gv, P.enc, 54, 64, 1024, True, dev)
    with torch.no_grad():
        big = enc(T_(P.x, dev), T_(P.y_in_enc, dev), clamp_vae=True, lat_dim=32)[0]
        rows = [0, 17, 101, B - 1]
        small = enc(T_(P.x[rows], dev), T_(P.y_in_enc[rows], dev), clamp_vae=True, lat_dim=32)[0]
        torch.cuda.synchronize()
    ref = orc.gru_rnn_forward(P.enc, P.x[rows], P.y_in_enc[rows], clamp_vae=True, lat_dim=32)[0]
    assert maxabs(big[rows], ref, "B=%d rows vs oracle" % B) <= TIGHT_PASS
    d = float((big[rows] - small).abs().max())
    note("B=%d vs 4-row batch: max|d| = %.3e" % (B, d))
    assert d <= 2e-6      # (different kernels: 32-row tiles with the own h re-read in fp32 vs the 16-row-tile pair kernel)


def test_philox_sampling_on_device(gv, dev):
    torch.manual_seed(1)
    p = torch.zeros(300, 64, 64, device=dev)
    z1 = gv.sampling_vae_batch(p, lat_dim=32)
    torch.manual_seed(1)
    z2 = gv.sampling_vae_batch(p, lat_dim=32)
    z3 = gv.sampling_vae_batch(p, lat_dim=32)
    assert z1.shape == (300, 64, 32) and torch.equal(z1, z2) and not torch.equal(z1, z3)
    assert abs(z1.mean().item()) < 5e-3 and abs(z1.std().item() - 1.0) < 5e-3
    # mean of many draws tends to mu (decode_gru-cyclevae_gauss.py:304-305)
    mu = torch.randn(50, 32, device=dev)
    par = torch.cat((mu, torch.full((50, 32), -2.0, device=dev)), 1)
    m = torch.mean(gv.sampling_vae_batch(par.unsqueeze(0).repeat(300, 1, 1), lat_dim=32), 0)
    assert (m - mu).abs().max().item() < 0.15


def test_draws_keyed_by_global_row(gv, dev):
    """with cvae_set_draw_origin a batch row draws the same latent eps on whichever rank it lands, so a chain on
    rows 20..39 of a 40-row job equals rows 20..39 of the 40-row chain bit for bit (same kernel, on-device Philox, same seed),
    and differs from what the same rows draw when they are numbered from 0."""
    P = synth.CycleVAEProblem(B=40, T=24, bias_scale=0.0, tag="origin")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    full = [T_(getattr(P, n), dev) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    try:
        with torch.no_grad():
            gv.set_draw_origin(0, 40, 24)
            whole = chain(*full, seed=77)
            lo = chain(*[v[:20] for v in full], seed=77)
            gv.set_draw_origin(20, 40, 24)
            hi = chain(*[v[20:] for v in full], seed=77)
            gv.set_draw_origin(0, 0, 0)
            hi_local = chain(*[v[20:] for v in full], seed=77)
            z_whole = gv.sampling_vae_batch(whole["lat"][0])          # stand-alone sampling follows the same numbering
        torch.cuda.synchronize()
        assert chain.status()[0] == 0
        for k in whole:
            assert torch.equal(whole[k][:, :20], lo[k]) and torch.equal(whole[k][:, 20:], hi[k]), k
        assert not torch.equal(hi["rec"], hi_local["rec"])
        assert z_whole.shape == (40, 24, 32)
    finally:
        gv.set_draw_origin(0, 0, 0)


def test_cycle_chain_carry_form_on_device(gv, dev):
    """CycleChain with state (cvae_cycle_forward_carry): a second 40-frame window continued from the first one's per-pass
    (y_last, h) equals the ten module calls of the reference's windowed loop (train...:1299-1311) with the same carries."""
    P = synth.CycleVAEProblem(B=20, T=80, bias_scale=0.0, tag="carrydev")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    names = ("x", "cvx", "code_src", "code_trg")
    state, carries = None, {}
    with torch.no_grad():
        for w in range(2):
            sl = slice(40 * w, 40 * w + 40)
            a = [T_(getattr(P, n)[:, sl], dev) for n in names]
            eps = T_(P.eps[:, :, :, sl], dev)
            out, state = chain(*a, T_(P.y_in_enc, dev), T_(P.y_in_dec, dev), eps=eps, state=state, return_state=True)
            prev = None
            for i in range(2):
                def run(m, xin, slot, y0, **kw):
                    y, h = carries.get((i, slot), (T_(y0, dev), None))
                    o, yl, hl = m(xin, y, h_in=h, **kw)
                    carries[(i, slot)] = (yl, hl)
                    return o
                e_in = a[0] if i == 0 else torch.cat((a[0][:, :, :4], prev), 2)
                lat = run(enc, e_in, "lat", P.y_in_enc, clamp_vae=True, lat_dim=32)
                rec = run(dec, torch.cat((a[2], gv.sampling_with_eps(lat, eps[i, 0], 32)), 2), "rec", P.y_in_dec)
                cv = run(dec, torch.cat((a[3], gv.sampling_with_eps(lat, eps[i, 1], 32)), 2), "cv", P.y_in_dec)
                latcv = run(enc, torch.cat((a[1], cv), 2), "latcv", P.y_in_enc, clamp_vae=True, lat_dim=32)
                prev = run(dec, torch.cat((a[2], gv.sampling_with_eps(latcv, eps[i, 2], 32)), 2), "reccyc", P.y_in_dec)
                for k, v in (("lat", lat), ("rec", rec), ("cv", cv), ("latcv", latcv), ("reccyc", prev)):
                    assert maxabs(out[k][i], v.cpu().numpy(), "carry chain w%d c%d %s" % (w, i, k)) <= TIGHT_KERNELS
            assert maxabs(state["h_dec"][1, 2], carries[(1, "reccyc")][1][0].cpu().numpy(), "carry chain h state") <= TIGHT_KERNELS
    torch.cuda.synchronize()
    assert chain.status()[0] == 0


@pytest.mark.gpu
def test_stage6_pair_as_a_wavefront_of_windows(gv, dev):
    """stage6.convert_pair(window=...): the utterance pair cut into windows, the decoder launch of window w beside the encoder
    launch of window w+1 on two streams (passes with carried state whose conv front-end sees the neighbouring frames, ABI 5).
    Must equal the unbroken two-launch form BIT FOR BIT -- explicit eps and Philox draws, window sizes that do and do not divide
    the lengths, a target shorter and longer than the source."""
    import stage6
    n = 5
    # (94, 150) at 32 and (446, 660) at 224: the source rows end inside (stop - reach, stop] of a window while the target row is
    # still unfinished -- before the schedule deferred them, the unfinished row got fewer frames than the pass ran
    for (Ts, Tt), windows in (((203, 180), (64, 50, 199)), ((97, 150), (32,)), ((94, 150), (32,)), ((446, 660), (224,)), ((660, 446), (224,))):
        Ps = synth.CycleVAEProblem(B=1, T=Ts, bias_scale=0.0, tag="s6win/src%d" % Ts)
        Pt = synth.CycleVAEProblem(B=1, T=Tt, bias_scale=0.0, tag="s6win/trg%d" % Tt)
        enc, dec = module(gv, Ps.enc, 54, 64, 1024, True, dev), module(gv, Ps.dec, 34, 50, 1024, False, dev)
        es, et = T_(synth.normal("s6win/eps_s", (n, Ts, 32)), dev), T_(synth.normal("s6win/eps_t", (n, Tt, 32)), dev)
        y_pp, y_d = T_(Ps.y_in_enc, dev), T_(Ps.y_in_dec, dev)
        xs, xt = T_(Ps.x[0], dev), T_(Pt.x[0], dev)
        with torch.no_grad():
            ref_e = stage6.convert_pair(enc, dec, xs, xt, y_pp, y_d, y_d, 32, n_smpl_dec=n, eps_src=es, eps_trg=et)
            ref_p = stage6.convert_pair(enc, dec, xs, xt, y_pp, y_d, y_d, 32, n_smpl_dec=300, seed=21)
            torch.cuda.synchronize()
            for W in windows:
                for rep in range(2):
                    got_e = stage6.convert_pair(enc, dec, xs, xt, y_pp, y_d, y_d, 32, n_smpl_dec=n, eps_src=es, eps_trg=et, window=W)
                    got_p = stage6.convert_pair(enc, dec, xs, xt, y_pp, y_d, y_d, 32, n_smpl_dec=300, seed=21, window=W)
                    torch.cuda.synchronize()
                    for name, a, b, c, d in zip(("cvmcep", "cvmcep_src", "cvmcep_trg", "lat_src", "lat_trg"), ref_e, got_e, ref_p, got_p):
                        assert a.shape == b.shape and torch.equal(a, b), (Ts, Tt, W, rep, "eps", name, float((a - b).abs().max()))
                        assert torch.equal(c, d), (Ts, Tt, W, rep, "philox", name, float((c - d).abs().max()))
    gv.check_status()


def test_stage6_list_pipelined_over_two_streams(gv, dev):
    """stage6.convert_list: the encoder pass of utterance pair g+1 side by side with the decoder pass of pair g (two streams, two
    word-exchange recurrences co-resident on every CU) must give exactly what one convert_pairs call per pair gives, for
    pairs of different lengths, and twice in a row (buffers recycled by the allocator across streams)."""
    import stage6
    lens = [(203, 180), (150, 203), (180, 150), (64, 97)]
    P = {T: synth.CycleVAEProblem(B=1, T=T, bias_scale=0.0, tag="s6list/%d" % T) for T in (203, 180, 150, 64, 97)}
    enc, dec = module(gv, P[203].enc, 54, 64, 1024, True, dev), module(gv, P[203].dec, 34, 50, 1024, False, dev)
    y_pp, y_d = T_(P[203].y_in_enc, dev), T_(P[203].y_in_dec, dev)
    groups = [[(T_(P[a].x[0], dev), T_(P[b].x[0], dev))] for a, b in lens]
    seeds = [11, 12, 13, 14]
    with torch.no_grad():
        ref = [stage6.convert_pairs(enc, dec, g, y_pp, y_d, y_d, 32, n_smpl_dec=7, seed=sd) for g, sd in zip(groups, seeds)]
        torch.cuda.synchronize()
        for rep in range(2):
            got = stage6.convert_list(enc, dec, groups, y_pp, y_d, y_d, 32, n_smpl_dec=7, seeds=seeds)
            torch.cuda.synchronize()
            assert len(got) == len(ref)
            for gi, (rg, gg) in enumerate(zip(ref, got)):
                for name, a, b in zip(("cvmcep", "cvmcep_src", "cvmcep_trg", "lat_src", "lat_trg"), rg[0], gg[0]):
                    assert a.shape == b.shape and torch.equal(a, b), (rep, gi, name)
        # calls with SEVERAL pairs run the tile kernel, whose single 32-row tile occupies 128 of the 256 CUs: the encoder launch of
        # one call and the decoder launch of the previous one are resident together on disjoint halves of the chip
        big = [[groups[0][0], groups[1][0], groups[2][0], groups[3][0]], [groups[2][0], groups[0][0], groups[3][0]],
               [groups[1][0]] * 5, [groups[3][0], groups[2][0]]]
        bseeds = [31, 32, 33, 34]
        bref = [stage6.convert_pairs(enc, dec, g, y_pp, y_d, y_d, 32, n_smpl_dec=7, seed=sd) for g, sd in zip(big, bseeds)]
        torch.cuda.synchronize()
        bgot = stage6.convert_list(enc, dec, big, y_pp, y_d, y_d, 32, n_smpl_dec=7, seeds=bseeds)
        torch.cuda.synchronize()
        for gi, (rg, gg) in enumerate(zip(bref, bgot)):
            assert len(rg) == len(gg)
            for q, (rp, gp) in enumerate(zip(rg, gg)):
                for a, b in zip(rp, gp):
                    assert torch.equal(a, b), ("several pairs per call", gi, q)
        # mixed: one-pair calls (word-exchange kernels, 256 small blocks) next to several-pair calls (tile kernel, 128 whole-CU blocks)
        mixed = [groups[0], big[0], groups[1], big[3], big[2], groups[3]]
        mseeds = [41, 42, 43, 44, 45, 46]
        mref = [stage6.convert_pairs(enc, dec, g, y_pp, y_d, y_d, 32, n_smpl_dec=7, seed=sd) for g, sd in zip(mixed, mseeds)]
        torch.cuda.synchronize()
        mgot = stage6.convert_list(enc, dec, mixed, y_pp, y_d, y_d, 32, n_smpl_dec=7, seeds=mseeds)
        torch.cuda.synchronize()
        for gi, (rg, gg) in enumerate(zip(mref, mgot)):
            for q, (rp, gp) in enumerate(zip(rg, gg)):
                for a, b in zip(rp, gp):
                    assert torch.equal(a, b), ("mixed list", gi, q)
        gv.check_status()
        # the flat-list helper: calls of per_call pairs sorted by length, results in the caller's order
        flat = [g[0] for g in groups] * 2 + [groups[1][0]]
        many = stage6.convert_many(enc, dec, flat, y_pp, y_d, y_d, 32, n_smpl_dec=7, per_call=4, seed=50)
        torch.cuda.synchronize()
        assert len(many) == len(flat)
        for (fs_, ft_), r in zip(flat, many):
            assert r[0].shape == (fs_.shape[0], 50) and r[2].shape == (ft_.shape[0], 50) and r[3].shape == (fs_.shape[0], 64)
            assert all(torch.isfinite(o).all() for o in r)
        # (latents do not depend on the seed or on a call's other rows; the one-pair reference ran the word-exchange kernel)
        assert float((many[0][3] - ref[0][0][3]).abs().max()) <= TIGHT_KERNELS and float((many[4][3] - ref[0][0][3]).abs().max()) <= TIGHT_KERNELS
    gv.check_status()


def test_stage6_file_list_fan_out_one_process_per_device(gv, dev, tmp_path):
    """stage6.convert_files (the reference's multi-GPU stage 5 / 6: np.array_split of the file list, one process per GPU,
    decode...:190-195, 591-602) with the PRODUCT worker: a spawned process per device rebuilds the networks from their state dicts,
    reads the HDF5 feature files, runs convert_many on its GPU and returns the results in caller order.  The one-GPU box gives one
    device; the result must be what convert_many gives in this process with the same (seed, list position) draw keys -- and must not
    depend on per_call (calls of 2 pairs: tile kernel; the in-process reference below runs one pair per call: word-exchange kernel,
    so the two differ by kernel rounding only while the draws are identical)."""
    import hdf5io
    import stage6
    lens = [(60, 45), (33, 70), (52, 52)]
    P = synth.CycleVAEProblem(B=1, T=8, bias_scale=0.0, tag="s6files")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    files, pairs = [], []
    for i, (a, b) in enumerate(lens):
        fa, fb = synth.features("s6files/s%d" % i, 1, a, P.mu, P.sigma)[0], synth.features("s6files/t%d" % i, 1, b, P.mu, P.sigma)[0]
        pa, pb = str(tmp_path / ("s%d.h5" % i)), str(tmp_path / ("t%d.h5" % i))
        hdf5io.write_hdf5(pa, "/feat_org_lf0", fa)
        hdf5io.write_hdf5(pb, "/feat_org_lf0", fb)
        files.append((pa, pb))
        pairs.append((T_(fa, dev), T_(fb, dev)))
    y_pp, y_d = T_(P.y_in_enc, dev), T_(P.y_in_dec, dev)
    got = stage6.convert_files(enc, dec, files, [0], y_pp, y_d, y_d, 32, n_smpl_dec=7, per_call=2, seed=5, timeout=900)
    with torch.no_grad():
        same = stage6.convert_many(enc, dec, pairs, y_pp, y_d, y_d, 32, n_smpl_dec=7, per_call=2, seed=5, first_pair_id=0)
        single = stage6.convert_many(enc, dec, pairs, y_pp, y_d, y_d, 32, n_smpl_dec=7, per_call=1, seed=5, first_pair_id=0)
    torch.cuda.synchronize()
    gv.check_status()
    for q, (a, b) in enumerate(lens):
        assert [x.shape for x in got[q]] == [(a, 50), (a, 50), (b, 50), (a, 64), (b, 64)]
        for name, x, y_, z in zip(("cvmcep", "cvmcep_src", "cvmcep_trg", "lat_src", "lat_trg"), got[q], same[q], single[q]):
            assert np.array_equal(x, y_.cpu().numpy()), (q, name)                       # the worker process == this process, bit for bit
            assert maxabs(x, z.cpu().numpy(), "file fan-out, calls of 2 vs 1 pair: %s" % name) <= TIGHT_KERNELS


def test_stage6_pair_stacked_passes(gv, dev):
    """stage6.convert_pair: E(src) || E(trg) and the three decoder passes of decode...:303-323 as two stacked launches, with the
    5-draw latent mean taken in the prologue, against the same five passes through the module API one by one (utterances of
    203 and 180 frames: the shorter one's padding must look like the conv padding it would see alone)."""
    import stage6
    Ps = synth.CycleVAEProblem(B=1, T=203, bias_scale=0.0, tag="s6pair/src")
    Pt = synth.CycleVAEProblem(B=1, T=180, bias_scale=0.0, tag="s6pair/trg")
    enc, dec = module(gv, Ps.enc, 54, 64, 1024, True, dev), module(gv, Ps.dec, 34, 50, 1024, False, dev)
    n = 5
    es = synth.normal("s6pair/eps_s", (n, 203, 32))
    et = synth.normal("s6pair/eps_t", (n, 180, 32))
    y_pp, y_d = T_(Ps.y_in_enc, dev), T_(Ps.y_in_dec, dev)
    with torch.no_grad():
        got = stage6.convert_pair(enc, dec, T_(Ps.x[0], dev), T_(Pt.x[0], dev), y_pp, y_d, y_d, 32, n_smpl_dec=n,
                                  eps_src=T_(es, dev), eps_trg=T_(et, dev))
        ref = []
        lats = []
        for feat, e in ((Ps.x[0], es), (Pt.x[0], et)):
            lat = enc(T_(feat, dev), y_pp, clamp_vae=True, lat_dim=32)[0]
            z = torch.mean(gv.sampling_with_eps(lat.unsqueeze(0).repeat(n, 1, 1), T_(e, dev), 32), 0)
            lats.append((lat, z))
        code = lambda i, T: T_(np.tile(np.eye(2, dtype=np.float32)[i], (T, 1)), dev)
        ref.append(dec(torch.cat((code(1, 203), lats[0][1]), 1), y_d)[0])
        ref.append(dec(torch.cat((code(0, 203), lats[0][1]), 1), y_d)[0])
        ref.append(dec(torch.cat((code(1, 180), lats[1][1]), 1), y_d)[0])
    torch.cuda.synchronize()
    for name, a, b in (("cvmcep", got[0], ref[0]), ("cvmcep_src", got[1], ref[1]), ("cvmcep_trg", got[2], ref[2]),
                       ("lat_src", got[3], lats[0][0]), ("lat_trg", got[4], lats[1][0])):
        assert a.shape == b.shape
        assert maxabs(a, b.cpu().numpy(), "stage6 pair " + name) <= TIGHT_PASS
    # Philox path: runs, deterministic in the seed
    with torch.no_grad():
        a = stage6.convert_pair(enc, dec, T_(Ps.x[0], dev), T_(Pt.x[0], dev), y_pp, y_d, y_d, 32, n_smpl_dec=300, seed=3)
        b = stage6.convert_pair(enc, dec, T_(Ps.x[0], dev), T_(Pt.x[0], dev), y_pp, y_d, y_d, 32, n_smpl_dec=300, seed=3)
    assert torch.equal(a[0], b[0]) and torch.isfinite(a[2]).all()
    # several pairs in one call (15 stacked decoder rows, the 16-row dataflow kernel) against the single-pair call (2 / 3 rows: the
    # word-exchange kernel k_gru_steps_ll, plain fp32 FMAs): same values up to the two kernels' rounding
    Pu = synth.CycleVAEProblem(B=1, T=150, bias_scale=0.0, tag="s6pair/u")
    with torch.no_grad():
        many = stage6.convert_pairs(enc, dec, [(T_(Ps.x[0], dev), T_(Pt.x[0], dev)), (T_(Pu.x[0], dev), T_(Ps.x[0], dev)),
                                               (T_(Pt.x[0], dev), T_(Pu.x[0], dev)), (T_(Pu.x[0], dev), T_(Pu.x[0], dev)),
                                               (T_(Ps.x[0], dev), T_(Ps.x[0], dev))], y_pp, y_d, y_d, 32, n_smpl_dec=n,
                                    eps=[(T_(es, dev), T_(et, dev)), (T_(es[:, :150], dev), T_(es, dev)), (T_(et, dev), T_(es[:, :150], dev)),
                                         (T_(et[:, :150], dev), T_(et[:, :150], dev)), (T_(es, dev), T_(es, dev))])
    torch.cuda.synchronize()
    for name, a_, b_ in zip(("cvmcep", "cvmcep_src", "cvmcep_trg", "lat_src", "lat_trg"), many[0], got):
        assert maxabs(a_, b_.cpu().numpy(), "stage6 five pairs vs one pair " + name) <= TIGHT_KERNELS, name
    assert many[1][0].shape == (150, 50) and many[2][2].shape == (150, 50) and all(torch.isfinite(o).all() for q in many for o in q)
    # ten pairs = 20 encoder / 30 decoder rows: still ONE 32-row tile of the dataflow kernel (the most a call takes)
    ten_pairs = [(T_(Ps.x[0], dev), T_(Pt.x[0], dev)), (T_(Pu.x[0], dev), T_(Ps.x[0], dev))] * 5
    ten_eps = [(T_(es, dev), T_(et, dev)), (T_(es[:, :150], dev), T_(es, dev))] * 5
    with torch.no_grad():
        ten = stage6.convert_pairs(enc, dec, ten_pairs, y_pp, y_d, y_d, 32, n_smpl_dec=n, eps=ten_eps)
        with pytest.raises(ValueError):
            stage6.convert_pairs(enc, dec, ten_pairs + ten_pairs[:1], y_pp, y_d, y_d, 32, n_smpl_dec=n, eps=ten_eps + ten_eps[:1])
    torch.cuda.synchronize()
    for q in (0, 8):
        for name, a_, b_ in zip(("cvmcep", "cvmcep_src", "cvmcep_trg", "lat_src", "lat_trg"), ten[q], got):
            assert maxabs(a_, b_.cpu().numpy(), "stage6 ten pairs vs one pair " + name) <= TIGHT_KERNELS, (q, name)
    for q in (1, 9):
        for a_, b_ in zip(ten[q], many[1]):
            assert torch.equal(a_, b_), q          # (same kernel, same rows' arithmetic: a row does not see its neighbours)


def test_stress_config_cyc4_chain(gv, dev, golden):
    """BASELINE configs[4] dims, hu2048 / ld64 / n_cyc = 4 (8 encoder + 12 decoder passes): (a) B=2, T=16 against the chain the
    REFERENCE ran (tests/golden/stress_chain.npz; any-H kernels at this batch size); (b) B=40, T=24 through the persistent
    H = 2048 kernel (k_gru_steps_v6<32, ., 3, W2S>: 8-unit x 32-row blocks on all 256 CUs, exact operands with the third weight limb
    streamed, stacked decoder passes
    = 3 row tiles per block) against the oracle on three of its rows, MCD within the 0.01 dB budget."""
    g = golden("stress_chain")
    P = synth.CycleVAEProblem(B=2, T=16, in_dim=54, out_dim=50, lat_dim=64, hidden=2048, n_cyc=4, bias_scale=0.05, tag="stress4")
    enc, dec = module(gv, P.enc, 54, 128, 2048, True, dev), module(gv, P.dec, 66, 50, 2048, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=64, n_cyc=4)
    names = ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")
    with torch.no_grad():
        out = chain(*[T_(getattr(P, n), dev) for n in names], eps=T_(P.eps, dev))
    torch.cuda.synchronize()
    assert chain.status()[0] == 0
    for k in g.files:
        assert maxabs(out[k], g[k], "stress cyc4 (reference golden) " + k) <= TIGHT_CHAIN
    Q = synth.CycleVAEProblem(B=40, T=24, in_dim=54, out_dim=50, lat_dim=64, hidden=2048, n_cyc=4, bias_scale=0.0, tag="stress4b")
    enc2, dec2 = module(gv, Q.enc, 54, 128, 2048, True, dev), module(gv, Q.dec, 66, 50, 2048, False, dev)
    chain2 = gv.CycleChain(enc2, dec2, lat_dim=64, n_cyc=4)
    with torch.no_grad():
        big = chain2(*[T_(getattr(Q, n), dev) for n in names], eps=T_(Q.eps, dev))
    torch.cuda.synchronize()
    assert chain2.status()[0] == 0
    rows = [0, 17, 39]
    ref = orc.cycle_chain(Q.enc, Q.dec, Q.x[rows], Q.cvx[rows], Q.code_src[rows], Q.code_trg[rows], Q.y_in_enc[rows],
                          Q.y_in_dec[rows], Q.eps[:, :, rows], 4, 64)
    for k in ref:
        assert maxabs(big[k][:, rows], np.stack(ref[k]), "stress cyc4 B=40 persistent " + k) <= TIGHT_CHAIN
    m = max(mcd_db(big[k][:, rows], np.stack(ref[k])) for k in ("rec", "cv", "reccyc"))
    note("stress cyc4 B=40 persistent kernel: MCD vs oracle %.3e dB" % m)
    assert m <= BUDGET_MCD
    assert m <= TIGHT_MCD


def test_mc2e_and_mod_pow_on_device(gv, dev):
    """SURVEY 8(f) row 1, second half: the power correction of mod_pow (feature_extract_vc.py:131-138) on the device --
    cvae_mc2e (SPTK freqt + c2ir + energy, f64) against the numpy restatement, fp32 and fp64 inputs, recipe parameters
    alpha = 0.455, irlen = 1024; then through the GV post-filter as `dpow`."""
    import stage6
    T, D = 37, 50
    cv = (synth.normal("mc2e/cv", (T, D)) * np.linspace(1.5, 0.05, D)).astype(np.float32)
    rf = (cv + 0.1 * synth.normal("mc2e/rf", (T, D)) * np.linspace(1.0, 0.05, D)).astype(np.float64)
    e32 = stage6.mc2e(T_(cv, dev)).cpu().numpy()
    e64 = stage6.mc2e(torch.from_numpy(rf).to(dev)).cpu().numpy()
    r32, r64 = orc.mc2e(cv), orc.mc2e(rf)
    d1, d2 = float(np.max(np.abs(e32 / r32 - 1))), float(np.max(np.abs(e64 / r64 - 1)))
    note("mc2e on device: rel|d| = %.3e (fp32 input), %.3e (fp64 input)" % (d1, d2))
    assert d1 <= 1e-11 and d2 <= 1e-11
    dp = stage6.mod_pow_dpow(T_(cv, dev), torch.from_numpy(rf).to(dev))
    assert float(np.max(np.abs(dp.cpu().numpy() - orc.mod_pow_dpow(cv, rf)))) <= 1e-11
    gv_t = (0.05 + synth.uniform01("mc2e/gv", (D - 1,))).astype(np.float64)
    cg = (0.02 + 0.5 * synth.uniform01("mc2e/cg", (D - 1,))).astype(np.float64)
    out, _ = stage6.gv_postfilter(T_(cv, dev), gv_t, cg, dpow=dp)
    ref, _ = orc.gv_postfilter(cv, gv_t, cg, orc.mod_pow_dpow(cv, rf))
    assert float(np.max(np.abs(out.cpu().numpy() - ref))) <= 1e-10


def test_hand_off_under_memory_traffic(gv, dev):
    """The dataflow hand-off of the persistent kernels (write-through publishes, flag polls, plain first-touch operand loads in
    k_gru_steps_v6) under uneven load: the same chain repeated while a second stream keeps the fabric busy with large device
    copies must reproduce the quiet run bit for bit, every word of all ten trajectories, and report no timed-out spin."""
    P = synth.CycleVAEProblem(B=64, T=40, bias_scale=0.0, tag="traffic")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    full = [T_(getattr(P, n), dev) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    eps = T_(P.eps, dev)
    with torch.no_grad():
        quiet = {k: v.clone() for k, v in chain(*full, eps=eps).items()}
    torch.cuda.synchronize()
    a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    side = torch.cuda.Stream()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(4):
                b.copy_(a, non_blocking=True)
                a.copy_(b, non_blocking=True)
        with torch.no_grad():
            out = chain(*full, eps=eps)
        torch.cuda.synchronize()
        assert chain.status()[0] == 0
        for k in quiet:
            assert torch.equal(out[k], quiet[k]), (rep, k)


@pytest.mark.gpu
def test_hand_off_under_cu_contention(gv, dev):
    """The all-resident recurrent kernels are launched PLAINLY after a one-time occupancy check (cvae_launch_coop): nothing at launch
    time guarantees that the whole grid is co-resident.  Here a second stream fills every CU with LDS-hungry workgroups (96 KB each,
    two waves of them, ~1.5 ms per workgroup: cvae_selftest_occupy) right before the chain is enqueued, so the recurrent kernels'
    blocks reach their CUs late and unevenly while the early ones spin on flags.  Every repetition must either reproduce the quiet
    run BIT FOR BIT or be refused cleanly (a time-out status raised by the module, never silently wrong values); at least one must
    go through.  Same for a train-mode pass + backward (k_train_fwd_steps_x3h, k_train_bwd_steps_x3)."""
    import _cabi
    lib = gv._lib()
    P = synth.CycleVAEProblem(B=64, T=24, bias_scale=0.0, tag="contention")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    full = [T_(getattr(P, n), dev) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    eps = T_(P.eps, dev)

    def train_pass():
        m = gv.GRU_RNN(in_dim=54, out_dim=64, hidden_units=1024, kernel_size=3, dilation_size=2, do_prob=0.5, scale_in_flag=True,
                       scale_out_flag=False)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in P.enc.items()})
        m = m.to(dev).train()
        torch.manual_seed(11)                       # the same Philox masks every time
        xt = full[0].clone().requires_grad_(True)
        out = m(xt, full[4], do=True, clamp_vae=True, lat_dim=32)[0]
        out.square().sum().backward()
        return [out.detach(), xt.grad, m.gru.weight_hh_l0.grad, m.conv.conv[0].weight.grad]

    with torch.no_grad():
        quiet = {k: v.clone() for k, v in chain(*full, eps=eps).items()}
    quiet_t = [v.clone() for v in train_pass()]
    torch.cuda.synchronize()
    gv.check_status(True)
    side = torch.cuda.Stream()
    through = refused = 0
    for rep in range(5):
        lib.selftest_occupy(512, 96 * 1024, 3000000, side.cuda_stream)
        try:
            with torch.no_grad():
                out = chain(*full, eps=eps)
            lib.selftest_occupy(512, 96 * 1024, 3000000, side.cuda_stream)
            tr = train_pass()
            torch.cuda.synchronize()
            gv.check_status(True)
        except _cabi.CvaeError:
            refused += 1
            torch.cuda.synchronize()
            continue
        through += 1
        for k in quiet:
            assert torch.equal(out[k], quiet[k]), (rep, k)
        for a, b in zip(tr, quiet_t):
            assert torch.equal(a, b), rep
    note("CU contention (512 x 96 KB LDS workgroups on a second stream): %d repetitions bit-identical, %d refused" % (through, refused))
    assert through >= 1


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 3])
def test_word_exchange_under_memory_traffic(gv, dev, rows):
    """k_gru_steps_ll (passes of at most three rows: tagged 16-byte words polled by the consumers) under uneven load: a 400-frame
    pass with h_in / y_in carries repeated while a second stream keeps the fabric busy reproduces the quiet run bit for bit
    (a word accepted with the wrong step's values, or a torn word, would change h from that step on) and reports no time-out."""
    P = synth.CycleVAEProblem(B=rows, T=400, bias_scale=0.05, tag="traffic_ll%d" % rows)
    enc = module(gv, P.enc, 54, 64, 1024, True, dev)
    x, y0 = T_(P.x, dev), T_(P.y_in_enc, dev)
    h0 = T_((0.3 * synth.normal("traffic_ll/h%d" % rows, (1, rows, 1024))).astype(np.float32), dev)
    with torch.no_grad():
        quiet = [t.clone() for t in enc(x, y0, h_in=h0, clamp_vae=True, lat_dim=32)]
        o = orc.gru_rnn_forward(P.enc, P.x, P.y_in_enc, h_in=h0.cpu().numpy(), clamp_vae=True, lat_dim=32)
    torch.cuda.synchronize()
    assert maxabs(quiet[0], o[0], "word exchange %d rows T=400 vs oracle" % rows) <= TIGHT_PASS
    a = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    side = torch.cuda.Stream()
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(4):
                b.copy_(a, non_blocking=True)
                a.copy_(b, non_blocking=True)
        with torch.no_grad():
            out = enc(x, y0, h_in=h0, clamp_vae=True, lat_dim=32)
        torch.cuda.synchronize()
        gv.check_status(True)
        for u, v in zip(out, quiet):
            assert torch.equal(u, v), rep


@pytest.mark.gpu
def test_limb_transport_selftest_device_equals_host_build(gv, dev):
    """cvae_selftest_limbs: producer split + consumer packed decode on the device, bit for bit against the host build of the same
    code (tests/emu); bit-exact for |x| >= 2^-16, absolute error <= 2^-40 below.  (Round 2: the packed decode used the low four bytes for both halves of a
    group of eight -- results stayed inside the MCD budget, so only this direct check sees it.)"""
    import test_emu_library as tel
    from emu_util import emu_lib, ptr
    x = tel.limb_selftest_values()
    y_host = np.zeros_like(x)
    emu_lib().selftest_limbs(ptr(x), ptr(y_host), x.size)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.empty_like(xd)
    gv._lib().selftest_limbs(xd.data_ptr(), yd.data_ptr(), x.size, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    y = yd.cpu().numpy()
    assert np.array_equal(y, y_host), float(np.abs(y - y_host).max())
    tel.check_limb_transport(x, y)


def test_dtw_on_device_vs_restated_algorithm(gv, dev):
    """stage6.dtw_org_to_trg (cvae_dtw_org_to_trg) at utterance size against oracle.dtw_org_to_trg -- PARITY UNPINNED (dtw_c's source
    is not in the reference tree): the oracle documents the algorithm; same path, same warp, costs to 1e-10; plus the defining
    property: a sequence with every frame held twice warps back onto the original at zero cost."""
    import stage6
    a = synth.normal("dtwg/a", (203, 50)).astype(np.float64)
    b = (0.9 * synth.normal("dtwg/b", (187, 50)) + 0.05).astype(np.float64)
    for mcd in (-1, 0):
        al, twf, mean, fr = stage6.dtw_org_to_trg(T_(a, dev), T_(b, dev), mcd=mcd)
        torch.cuda.synchronize()
        ra, rt, rm, rf = orc.dtw_org_to_trg(a, b, mcd=mcd)
        assert np.array_equal(twf.cpu().numpy(), rt) and np.array_equal(al.cpu().numpy(), ra)
        assert np.abs(fr.cpu().numpy() - rf).max() <= 1e-10 and abs(float(mean) - rm) <= 1e-10
    big = synth.normal("dtwg/big", (700, 50)).astype(np.float32)
    al, twf, mean, fr = stage6.dtw_org_to_trg(T_(np.repeat(big, 2, axis=0), dev), T_(big, dev))
    torch.cuda.synchronize()
    assert float(mean) == 0.0 and np.array_equal(twf.cpu().numpy() // 2, np.arange(700)) and np.array_equal(al.cpu().numpy(), big.astype(np.float64))


@pytest.mark.gpu
def test_plain_and_cooperative_launch_give_the_same_bits(gv, dev):
    """The all-resident recurrent kernels are launched plainly after a one-time occupancy check (default) or through
    hipLaunchCooperativeKernel (option coop_launch=1): the same kernels, the same results to the bit -- dataflow kernel (B=20),
    three-row low-latency kernel (B=3) and a stacked chain."""
    lib = gv._lib()
    P = synth.CycleVAEProblem(B=20, T=24, bias_scale=0.0, tag="launchmode")
    enc, dec = module(gv, P.enc, 54, 64, 1024, True, dev), module(gv, P.dec, 34, 50, 1024, False, dev)
    chain = gv.CycleChain(enc, dec, lat_dim=32, n_cyc=2)
    full = [T_(getattr(P, n), dev) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")]
    outs = []
    try:
        for mode in (0, 1, 0):
            lib.set_option("coop_launch", mode)
            with torch.no_grad():
                o = chain(*full, eps=T_(P.eps, dev))
                small = enc(T_(P.x[:3], dev), T_(P.y_in_enc[:3], dev), clamp_vae=True, lat_dim=32)[0]
                torch.cuda.synchronize()
            assert chain.status()[0] == 0
            outs.append((o, small))
    finally:
        lib.set_option("coop_launch", 0)
    for o, small in outs[1:]:
        assert torch.equal(small, outs[0][1])
        for k in outs[0][0]:
            assert torch.equal(o[k], outs[0][0][k]), k
    ref = orc.gru_rnn_forward(P.enc, P.x[:3], P.y_in_enc[:3], clamp_vae=True, lat_dim=32)[0]
    assert maxabs(outs[0][1], ref, "3-row pass vs oracle") <= TIGHT_PASS
