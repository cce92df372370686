"""SURVEY 8(f) row 4, first variant: the Laplace posterior (reference gru_vae.py:101-114 sampling_vae_laplace, :130-145 loss_vae_laplace,
:415-417 the clamp_vae_laplace branch of GRU_RNN.forward).  tests/golden/laplace.npz was recorded by RUNNING the reference
(tests/golden/make_golden.py laplace: the reference's own uniform draw under torch.manual_seed, recorded next to its output).
CPU: the oracle restatement, the drop-in module's torch-op functions, and the library's kernels on the host build; -m gpu: the
HIP kernels through the drop-in module."""
import numpy as np
import pytest

import _cabi
import synth
from oracle import cyclevae_oracle as orc

torch = pytest.importorskip("torch")

FLOOR = np.float32(-7.2543288692621097)


def problem():
    P = synth.CycleVAEProblem(B=2, T=12, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="laplace")
    sd = dict(P.enc)
    sd["out_1.bias"] = sd["out_1.bias"] - np.float32(8.0) * (np.arange(8) >= 4).astype(np.float32)
    return P, sd


def test_oracle_restatement_vs_the_reference(golden):
    g = golden("laplace")
    P, sd = problem()
    assert synth.sha256_state(sd) == str(g["sha_enc"])
    lat = orc.gru_rnn_forward(sd, P.x, P.y_in_enc, clamp_vae_laplace=True, lat_dim=4)[0]
    assert np.abs(lat - g["lat"]).max() <= 2e-5 and np.array_equal(lat[:, :, 4:] == FLOOR, g["lat"][:, :, 4:] == FLOOR)
    assert np.any(g["lat"][:, :, 4:] == FLOOR) and np.any(g["raw"][:, :, 4:] < FLOOR) and np.all(g["lat"][:, :, :4] == g["raw"][:, :, :4])
    lat2d = orc.gru_rnn_forward(sd, P.x[1], P.y_in_enc[1:], clamp_vae_laplace=True, lat_dim=4)[0]
    assert lat2d.shape == (12, 8) and np.abs(lat2d - g["lat2d"]).max() <= 2e-5
    z = orc.sampling_vae_laplace(g["lat"][0], g["eps"], 4)
    assert np.abs(z - g["z"]).max() <= 2e-6
    assert abs(float(orc.loss_vae_laplace(g["lat"][0], 4)) - float(g["kl"])) <= 2e-6 * abs(float(g["kl"]))
    # clamp_vae wins when both flags are given (gru_vae.py:410 comes before :415)
    both = orc.gru_rnn_forward(sd, P.x, P.y_in_enc, clamp_vae=True, clamp_vae_laplace=True, lat_dim=4)[0]
    assert np.all(both[:, :, 4:] >= orc.LOG_VAR_FLOOR) and np.any(both[:, :, 4:] < FLOOR)


def test_module_functions_on_cpu_tensors(golden):
    """loss_vae_laplace is torch ops (any device, like the reference); its value and gradient are the reference's."""
    import gru_vae
    g = golden("laplace")
    p = torch.from_numpy(g["lat"][0].copy()).requires_grad_(True)
    kl = gru_vae.loss_vae_laplace(p, lat_dim=4)
    assert abs(kl.item() - float(g["kl"])) <= 1e-6 * abs(float(g["kl"]))
    kl.backward()
    assert np.abs(p.grad.numpy() - g["d_kl"]).max() <= 1e-6
    assert abs(gru_vae.loss_vae_laplace(p.detach()).item() - float(g["kl"])) <= 1e-6 * abs(float(g["kl"]))     # lat_dim defaults to half
    with pytest.raises(RuntimeError):
        gru_vae.sampling_vae_laplace(p.detach(), lat_dim=4)             # the draw is a HIP kernel: no CPU fallback


def test_two_sided_dil_conv_is_callable_on_its_own(golden):
    """the reference's TwoSidedDilConv1d.forward (gru_vae.py:53-66) works stand-alone; the drop-in's did not."""
    import gru_vae
    g = golden("tiny_ops")
    P = synth.CycleVAEProblem(B=2, T=12, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="tiny")
    m = gru_vae.GRU_RNN(in_dim=6, out_dim=8, hidden_units=32, kernel_size=3, dilation_size=2, scale_out_flag=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.enc.items()})
    with torch.no_grad():
        xconv = m.conv(m.scale_in(torch.from_numpy(P.x).transpose(1, 2))).transpose(1, 2)
    assert tuple(xconv.shape) == (2, 12, 54) and np.abs(xconv.numpy() - g["xconv"]).max() <= 1e-5
    alone = gru_vae.TwoSidedDilConv1d(in_dim=3, kernel_size=3, layers=2)
    assert tuple(alone(torch.zeros(1, 3, 7)).shape) == (1, 27, 7)


def test_library_kernels_on_the_host_build(golden):
    from emu_util import NpNet, emu_lib, ptr
    lib = emu_lib()
    g = golden("laplace")
    P, sd = problem()
    net = NpNet(lib, sd, 6, 8, 32)
    for flags in (0, _cabi.FLAG_PERSISTENT):
        lat = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4 | _cabi.CLAMP_LAPLACE, flags=flags)[0]
        assert np.abs(lat - g["lat"]).max() <= 5e-5 and np.any(lat[:, :, 4:] == FLOOR) and np.all(lat[:, :, 4:] >= FLOOR)
    gauss = net.forward(P.x, P.y_in_enc, clamp_lat_dim=4)[0]
    assert np.any(gauss[:, :, 4:] < FLOOR)                               # the Gaussian floor is lower: another clamp
    lat0 = np.ascontiguousarray(g["lat"][0])
    eps = np.ascontiguousarray(g["eps"])
    z, eo = np.zeros((12, 4), np.float32), np.zeros((12, 4), np.float32)
    lib.sample_laplace(ptr(lat0), 12, 4, ptr(eps), 0, 0, ptr(z), ptr(eo))
    assert np.abs(z - g["z"]).max() <= 2e-6 and np.array_equal(eo, eps)
    cot = np.ascontiguousarray(synth.normal("laplace/cot", (12, 4)).astype(np.float32))
    dlat = np.zeros((12, 8), np.float32)
    lib.sample_laplace_backward(ptr(cot), ptr(lat0), ptr(z), 12, 4, ptr(dlat))
    assert np.abs(dlat - g["d_sample"]).max() <= 2e-6
    # Philox draw: deterministic in (seed, draw), uniform on the reference's interval, z follows Laplace(0, 1) for mu = 0, s = 0
    big = np.zeros((8192, 8), np.float32)
    z1, z2, z3, e1 = (np.zeros((8192, 4), np.float32) for _ in range(4))
    lib.sample_laplace(ptr(big), 8192, 4, None, 5, 0, ptr(z1), ptr(e1))
    lib.sample_laplace(ptr(big), 8192, 4, None, 5, 0, ptr(z2), None)
    lib.sample_laplace(ptr(big), 8192, 4, None, 5, 1, ptr(z3), None)
    assert np.array_equal(z1, z2) and not np.array_equal(z1, z3)
    assert e1.min() >= -0.4999 and e1.max() < 0.5 and abs(e1.mean()) < 0.01 and abs(e1.std() - 0.9999 / np.sqrt(12.0)) < 0.01
    assert np.all(np.isfinite(z1)) and abs(z1.mean()) < 0.05 and abs(z1.var() - 2.0) < 0.15       # Var Laplace(0, 1) = 2
    assert np.array_equal(z1, orc.sampling_vae_laplace(big, e1, 4)) or np.abs(z1 - orc.sampling_vae_laplace(big, e1, 4)).max() <= 2e-6


@pytest.mark.gpu
def test_laplace_variant_on_the_device(golden):
    import gru_vae
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    g = golden("laplace")
    P, sd = problem()
    m = gru_vae.GRU_RNN(in_dim=6, out_dim=8, hidden_units=32, kernel_size=3, dilation_size=2, scale_out_flag=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).eval()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    with torch.no_grad():
        lat = m(t(P.x), t(P.y_in_enc), clamp_vae_laplace=True, lat_dim=4)[0].cpu().numpy()
        lat2d = m(t(P.x[1]), t(P.y_in_enc[1:]), clamp_vae_laplace=True, lat_dim=4)[0].cpu().numpy()
    assert np.abs(lat - g["lat"]).max() <= 5e-6 and np.any(lat[:, :, 4:] == FLOOR) and np.all(lat[:, :, 4:] >= FLOOR)
    assert lat2d.shape == (12, 8) and np.abs(lat2d - g["lat2d"]).max() <= 5e-6
    # train-mode pass (autograd) with the same clamp: gradient is zero where the floor is active
    m.train()
    for n, p in m.named_parameters():
        p.requires_grad_(not n.startswith("scale"))
    x = t(P.x).requires_grad_(True)
    out = m(x, t(P.y_in_enc), clamp_vae_laplace=True, lat_dim=4)[0]
    assert np.abs(out.detach().cpu().numpy() - g["lat"]).max() <= 5e-6
    out[:, :, 4:].sum().backward()
    assert torch.isfinite(x.grad).all() and float(x.grad.abs().max()) > 0
    # sampling: injected eps reproduces the reference's z and gradient; Philox draws are reproducible from torch's seed
    p = t(g["lat"][0]).requires_grad_(True)
    z = gru_vae.sampling_vae_laplace(p, lat_dim=4, training=True, eps=t(g["eps"]))
    assert np.abs(z.detach().cpu().numpy() - g["z"]).max() <= 2e-6
    (z * t(synth.normal("laplace/cot", (12, 4)).astype(np.float32))).sum().backward()
    assert np.abs(p.grad.cpu().numpy() - g["d_sample"]).max() <= 2e-6
    torch.manual_seed(3)
    a = gru_vae.sampling_vae_laplace(p.detach(), lat_dim=4)
    torch.manual_seed(3)
    b = gru_vae.sampling_vae_laplace(p.detach())
    assert torch.equal(a, b) and torch.isfinite(a).all()
    kl = gru_vae.loss_vae_laplace(p.detach(), lat_dim=4)
    assert abs(kl.item() - float(g["kl"])) <= 2e-6 * abs(float(g["kl"]))
    gru_vae.check_status()


def test_vq_helpers_vs_the_reference(golden):
    """nn_search / nn_search_batch / weighted_ctr (reference gru_vae.py:148-195), torch ops in the drop-in as in the
    reference; tests/golden/vq.npz was recorded by running the reference's own functions."""
    import gru_vae
    g = golden("vq")
    enc = torch.from_numpy(synth.normal("vq/enc", (14, 5)).astype(np.float32))
    encb = torch.from_numpy(synth.normal("vq/encb", (2, 7, 5)).astype(np.float32))
    ctr = torch.from_numpy((1.5 * synth.normal("vq/ctr", (6, 5))).astype(np.float32))
    assert np.array_equal(gru_vae.nn_search(enc, ctr).numpy(), g["ids"]) and np.array_equal(gru_vae.nn_search_batch(encb, ctr).numpy(), g["idsb"])
    wc, wd = gru_vae.weighted_ctr(enc, ctr)
    assert np.abs(wc.numpy() - g["wc"]).max() <= 1e-6 and abs(wd.item() - float(g["wd"])) <= 1e-6
    p = torch.from_numpy(golden("laplace")["lat"][0].copy())
    with pytest.raises(RuntimeError):
        gru_vae.sampling_vae(p, lat_dim=4)            # the 2-D form of sampling_vae_batch: a HIP kernel, no CPU fallback


@pytest.mark.gpu
def test_vq_helpers_and_2d_sampling_on_the_device(golden):
    """The pieces of SURVEY 8(f) row 4 that have a device path, on device tensors: nn_search / nn_search_batch (INT: the indices must
    be bit-identical to the reference-recorded ones), weighted_ctr, and sampling_vae -- the 2-D form of the Gaussian draw (reference
    gru_vae.py:69-82), a HIP kernel: with injected eps it reproduces the reference-recorded z of tests/golden/tiny_ops.npz utterance
    by utterance, and its own Philox draw is the 3-D form's draw of the same seed (mu + exp(logvar / 2) * eps, eps ~ N(0, 1))."""
    import gru_vae
    assert torch.cuda.is_available()
    dev = torch.device("cuda:0")
    g = golden("vq")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    enc, encb = t(synth.normal("vq/enc", (14, 5)).astype(np.float32)), t(synth.normal("vq/encb", (2, 7, 5)).astype(np.float32))
    ctr = t((1.5 * synth.normal("vq/ctr", (6, 5))).astype(np.float32))
    ids, idsb = gru_vae.nn_search(enc, ctr), gru_vae.nn_search_batch(encb, ctr)
    assert ids.is_cuda and ids.dtype == torch.int64 and np.array_equal(ids.cpu().numpy(), g["ids"])
    assert idsb.is_cuda and np.array_equal(idsb.cpu().numpy(), g["idsb"])
    wc, wd = gru_vae.weighted_ctr(enc, ctr)
    assert wc.is_cuda and np.abs(wc.cpu().numpy() - g["wc"]).max() <= 1e-6 and abs(wd.item() - float(g["wd"])) <= 1e-6
    # 2-D Gaussian draw: injected eps against the reference-recorded z (recorded through sampling_vae_batch on [2, 12, 8])
    tg = golden("tiny_ops")
    P = synth.CycleVAEProblem(B=2, T=12, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="tiny")
    for b in range(2):
        z2 = gru_vae.sampling_with_eps(t(tg["lat"][b]), t(P.eps[0, 0][b]), 4)
        assert z2.shape == (12, 4) and np.abs(z2.cpu().numpy() - tg["z"][b]).max() <= 2e-6
    # the module's own draw: deterministic in torch's seed, the same stream as the 3-D form, and a standard normal eps
    p2 = t(tg["lat"][0])
    torch.manual_seed(5)
    a = gru_vae.sampling_vae(p2, lat_dim=4)
    torch.manual_seed(5)
    b3 = gru_vae.sampling_vae_batch(p2.unsqueeze(0), lat_dim=4)
    assert a.shape == (12, 4) and torch.equal(a, b3[0])
    big = torch.cat((torch.zeros(4096, 16), torch.zeros(4096, 16)), 1).to(dev)      # mu = 0, logvar = 0: z IS eps
    torch.manual_seed(6)
    eps = gru_vae.sampling_vae(big).cpu().numpy()
    assert abs(eps.mean()) <= 0.02 and abs(eps.std() - 1.0) <= 0.02 and np.abs(eps).max() < 6.5
    gru_vae.check_status()
