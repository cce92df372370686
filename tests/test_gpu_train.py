"""GPU tests of the training path through the drop-in module: train-mode passes and their autograd backward against
gradients recorded from the reference, and a whole stage-4 step (cyc2 chain + loss + backward + Adam) against the
stock-torch CPU checker with identical dropout masks and eps.

Tolerances: gradients relative to each tensor's largest entry, 2e-4 for single passes, 1e-3 for the 10-pass step
(fp32, sums over thousands of terms, different summation orders); loss 1e-5 relative.
"""
import os

import numpy as np
import pytest

import synth
from train_util import TRAINABLE, chain_loss, cpu_step, make_masks

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "gpu_parity_report.txt")


def note(msg):
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    with open(REPORT, "a") as f:
        f.write(msg + "\n")
    print(msg)


def rel_err(a, ref, name):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    assert a.shape == ref.shape, (name, a.shape, ref.shape)
    assert np.all(np.isfinite(a)), name
    e = float(np.max(np.abs(a.astype(np.float64) - ref))) / max(1.0, float(np.max(np.abs(ref))))
    note("%-48s rel err = %.3e" % (name, e))
    return e


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def gv():
    import gru_vae
    return gru_vae


def module(gv, sd, i, o, h, enc, dev, do_prob=0.5):
    m = gv.GRU_RNN(in_dim=i, out_dim=o, hidden_units=h, kernel_size=3, dilation_size=2, do_prob=do_prob, scale_in_flag=enc,
                   scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    for n, p in m.named_parameters():
        p.requires_grad_(n in TRAINABLE)     # scale_in / scale_out frozen (train...:369-372)
    return m


@pytest.mark.parametrize("tag,hid,B,T", [("train_h32", 32, 3, 10), ("train_h64", 64, 18, 7)])
def test_train_pass_autograd_vs_reference(gv, dev, golden, tag, hid, B, T):
    g = golden(tag)
    P = synth.CycleVAEProblem(B=B, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=hid, n_cyc=1, bias_scale=0.1, tag=tag)
    x_dec = np.concatenate([P.code_src, synth.normal(tag + "/z", (B, T, 4))], 2)
    x2 = synth.features(tag + "/x2", B, T, P.mu, P.sigma)
    enc, dec = module(gv, P.enc, 6, 8, hid, True, dev), module(gv, P.dec, 6, 4, hid, False, dev)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cases = [("enc", enc, P.x, P.y_in_enc, None, True), ("dec", dec, x_dec, P.y_in_dec, None, False),
             ("enc2", enc, x2, g["enc_y_last"], g["enc_h_last"], True)]
    for name, m, x, y_in, h_in, clamp in cases:
        for p in m.parameters():
            p.grad = None
        cot = synth.normal(tag + "/cot_" + name.replace("2", ""), (B, T, m.out_dim))
        xt = t(x).requires_grad_(True)
        m._debug_masks = (t(g[name + "_cmask"]), t(g[name + "_gmask"]))
        out, yl, hl = m(xt, t(y_in), h_in=None if h_in is None else t(h_in), do=True, clamp_vae=clamp, lat_dim=4)
        assert not yl.requires_grad and not hl.requires_grad
        (out * t(cot)).sum().backward()
        assert rel_err(out, g[name + "_out"], tag + " " + name + " out") <= 1e-4
        assert rel_err(hl, g[name + "_h_last"], tag + " " + name + " h_last") <= 1e-4
        assert rel_err(xt.grad, g[name + "_dx"], tag + " " + name + " dx") <= 2e-4
        for k in TRAINABLE:
            assert rel_err(dict(m.named_parameters())[k].grad, g[name + "_g_" + k], tag + " " + name + " d" + k) <= 2e-4
        assert m.scale_in.weight.grad is None if name.startswith("enc") else m.scale_out.weight.grad is None


@pytest.mark.parametrize("hid,B,T", [(64, 4, 12), (1024, 2, 16), (64, 50, 6), (2048, 2, 5)])   # 50 rows: 4 row tiles, split GEMMs; 2048: stress config
def test_stage4_step_vs_cpu_checker(gv, dev, hid, B, T):
    """cyc2 chain in train mode (dropout 0.5) + loss + backward + Adam through the drop-in modules vs stock torch on CPU."""
    big = hid >= 1024
    kw = dict(B=B, T=T, hidden=hid, n_cyc=2, bias_scale=0.05, tag="step%d" % hid)
    if hid == 2048:      # BASELINE configs[4] dims (hu2048 / ld64): the any-H kernels (per-step launches) carry this size
        P = synth.CycleVAEProblem(lat_dim=64, **kw)
    else:
        P = synth.CycleVAEProblem(**kw) if big else synth.CycleVAEProblem(in_dim=10, out_dim=6, lat_dim=4, **kw)
    masks = make_masks(P, 4, 6)
    ref_loss, ref_grads = cpu_step(P, masks)
    ed, eo, dd, do_ = (54, 128, 66, 50) if hid == 2048 else ((54, 64, 34, 50) if big else (10, 8, 6, 6))
    enc, dec = module(gv, P.enc, ed, eo, hid, True, dev), module(gv, P.dec, dd, do_, hid, False, dev)
    mods = {"enc": enc, "dec": dec}

    def run_pass(kind, x, y_in, clamp, mk):
        m = mods[kind]
        m._debug_masks = (torch.from_numpy(mk[0]).to(dev), torch.from_numpy(mk[1]).to(dev))
        return m(x, y_in, do=True, clamp_vae=clamp >= 0, lat_dim=P.lat_dim)[0]

    opt = torch.optim.Adam([p for m in mods.values() for p in m.parameters() if p.requires_grad], lr=1e-4)
    opt.zero_grad()
    loss = chain_loss(run_pass, P, dev, masks)
    loss.backward()
    note("stage-4 step hu%d: loss gpu %.6f cpu %.6f" % (hid, loss.item(), ref_loss))
    assert abs(loss.item() - ref_loss) <= 1e-5 * abs(ref_loss)
    for kind, m in mods.items():
        for k in TRAINABLE:
            gr = dict(m.named_parameters())[k].grad
            assert rel_err(gr, ref_grads[kind][k], "step hu%d %s d%s" % (hid, kind, k)) <= 1e-3
    before = {k: v.detach().clone() for k, v in enc.named_parameters()}
    opt.step()
    # Adam's first step moves every trainable entry with a non-zero gradient by ~lr; frozen layers stay put
    assert torch.equal(before["scale_in.weight"], enc.scale_in.weight)
    d = (enc.gru.weight_hh_l0.detach() - before["gru.weight_hh_l0"]).abs()
    assert 0.5e-4 < d.max().item() <= 1.01e-4
    # the next forward sees the updated weights (train image is rebuilt)
    loss2 = chain_loss(run_pass, P, dev, masks)
    assert loss2.item() < loss.item()
