"""The drop-in module keeps the reference's module surface: names, constructor arguments, submodule names,
state_dict keys and shapes, attributes the stage scripts read or assign -- checked without a GPU.  Also: no silent CPU path."""
import inspect

import pytest
import torch

import gru_vae

ENC_KEYS = {"scale_in.weight": (54, 54, 1), "scale_in.bias": (54,), "conv.conv.0.weight": (162, 54, 3), "conv.conv.0.bias": (162,),
            "conv.conv.1.weight": (486, 162, 3), "conv.conv.1.bias": (486,), "gru.weight_ih_l0": (3072, 550),
            "gru.weight_hh_l0": (3072, 1024), "gru.bias_ih_l0": (3072,), "gru.bias_hh_l0": (3072,),
            "out_1.weight": (64, 1024, 1), "out_1.bias": (64,)}
DEC_KEYS = {"conv.conv.0.weight": (102, 34, 3), "conv.conv.0.bias": (102,), "conv.conv.1.weight": (306, 102, 3),
            "conv.conv.1.bias": (306,), "gru.weight_ih_l0": (3072, 356), "gru.weight_hh_l0": (3072, 1024),
            "gru.bias_ih_l0": (3072,), "gru.bias_hh_l0": (3072,), "out_1.weight": (50, 1024, 1), "out_1.bias": (50,),
            "scale_out.weight": (50, 50, 1), "scale_out.bias": (50,)}


def test_exported_names_and_signatures():
    for name in ("GRU_RNN", "TwoSidedDilConv1d", "sampling_vae_batch", "loss_vae", "TWFSEloss", "initialize", "sampling_vae_laplace",
                 "loss_vae_laplace"):
        assert hasattr(gru_vae, name), name
    sig = inspect.signature(gru_vae.GRU_RNN.__init__)
    assert list(sig.parameters)[1:] == ["in_dim", "out_dim", "hidden_units", "hidden_layers", "kernel_size", "dilation_size",
                                        "do_prob", "scale_in_flag", "scale_out_flag", "scale_in_out_flag"]
    assert [sig.parameters[k].default for k in list(sig.parameters)[1:]] == [39, 35, 1024, 1, 3, 2, 0, True, True, False]
    fwd = inspect.signature(gru_vae.GRU_RNN.forward)
    assert list(fwd.parameters)[1:] == ["x", "y_in", "softmax", "sigmoid", "exp", "h_in", "noise", "res", "res_stdim", "res_endim",
                                        "do", "clamp_vae", "relu_vae", "lat_dim", "clamp_vae_laplace"]
    assert list(inspect.signature(gru_vae.sampling_vae_batch).parameters) == ["param", "lat_dim", "training", "relu_vae"]
    assert list(inspect.signature(gru_vae.loss_vae).parameters) == ["param", "lat_dim", "relu_vae"]
    assert list(inspect.signature(gru_vae.sampling_vae_laplace).parameters)[:4] == ["param", "lat_dim", "training", "relu_vae"]
    assert list(inspect.signature(gru_vae.loss_vae_laplace).parameters) == ["param", "lat_dim", "relu_vae"]


def test_state_dict_keys_and_shapes_match_the_reference():
    enc = gru_vae.GRU_RNN(in_dim=54, out_dim=64, hidden_units=1024, kernel_size=3, dilation_size=2, do_prob=0.5, scale_out_flag=False)
    dec = gru_vae.GRU_RNN(in_dim=34, out_dim=50, hidden_units=1024, kernel_size=3, dilation_size=2, do_prob=0.5, scale_in_flag=False)
    assert {k: tuple(v.shape) for k, v in enc.state_dict().items()} == ENC_KEYS
    assert {k: tuple(v.shape) for k, v in dec.state_dict().items()} == DEC_KEYS
    assert sum(p.numel() for p in enc.parameters()) == 5173130 and sum(p.numel() for p in dec.parameters()) == 4403752
    assert enc.out_dim == 64 and enc.tot_in_dim == 550 and enc.receptive_field == 9 and dec.tot_in_dim == 356
    # what the training script does to a freshly built net (train...:340-347, :365-377)
    enc.apply(gru_vae.initialize)
    assert float(enc.out_1.bias.abs().max()) == 0.0 and float(enc.gru.weight_hh_l0.abs().max()) > 0.0
    enc.scale_in.weight = torch.nn.Parameter(torch.diag(torch.ones(54)).unsqueeze(2))
    enc.scale_in.bias = torch.nn.Parameter(torch.zeros(54))
    for p in enc.scale_in.parameters():
        p.requires_grad = False
    trainable = list(enc.conv.parameters()) + list(enc.gru.parameters()) + list(enc.out_1.parameters())
    assert sum(p.numel() for p in trainable) + sum(
        p.numel() for p in list(dec.conv.parameters()) + list(dec.gru.parameters()) + list(dec.out_1.parameters())) == 9571362
    torch.optim.Adam(trainable, lr=1e-4)
    assert enc.train() is enc and enc.eval() is enc


def test_losses_on_cpu_tensors():
    """loss_vae / TWFSEloss are plain torch ops (the reference calls them on whatever device the tensors are)."""
    p = torch.tensor([[0.5, -1.0, 0.2, 0.1]])
    kl = gru_vae.loss_vae(p, lat_dim=2)
    exp = 0.5 * ((torch.exp(p[:, 2:]) + p[:, :2] ** 2 - p[:, 2:] - 1.0).sum(1)).mean()
    assert torch.allclose(kl, exp)
    x, y = torch.zeros(3, 4), torch.ones(3, 4)
    s, m, sd = gru_vae.TWFSEloss()(x, y, GV=False, L2=True)
    k = 10.0 / 2.302585092994046
    assert abs(m.item() - k * (2 * 4) ** 0.5) < 1e-5 and abs(s.item() - 3 * m.item()) < 1e-4


def test_twfse_every_branch_matches_the_reference():
    """tests/golden/twfse_branches.npz: outputs of the reference's TWFSEloss for twf x rmse x L2 x GV (make_golden.py)."""
    import os
    import numpy as np
    import synth
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "twfse_branches.npz"))
    x = synth.normal("twfse/x", (12, 5)).astype(np.float32)
    y = (synth.normal("twfse/y", (9, 5)) * 0.5 + 0.1).astype(np.float32)
    twf = torch.from_numpy(g["twf"])
    crit = gru_vae.TWFSEloss()
    for use_twf in (0, 1):
        xx = torch.from_numpy(x) if use_twf else torch.from_numpy(x[:9])
        yy = torch.from_numpy(y if use_twf else x[:9] * 0.8 + y * 0.2)
        for rmse in (0, 1):
            for l2 in (0, 1):
                for gv in (0, 1):
                    out = crit(xx, yy, twf=twf if use_twf else None, GV=bool(gv), rmse=bool(rmse), L2=bool(l2))
                    ref = g["out_twf%d_rmse%d_l2%d_gv%d" % (use_twf, rmse, l2, gv)]
                    assert len(out) == len(ref)
                    np.testing.assert_allclose(np.array([v.item() for v in out]), ref, rtol=2e-6, atol=1e-7)


def test_no_cpu_fallback_and_dead_flags():
    m = gru_vae.GRU_RNN(in_dim=6, out_dim=8, hidden_units=32, scale_out_flag=False)
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP device only"):
        m(torch.zeros(2, 5, 6), torch.zeros(2, 1, 8))
    with pytest.raises(RuntimeError, match="HIP device only"):
        gru_vae.sampling_vae_batch(torch.zeros(2, 5, 8))
    for kw in ({"softmax": True}, {"res": True}, {"noise": 0.1}, {"relu_vae": True}):
        with pytest.raises(NotImplementedError):
            m(torch.zeros(2, 5, 6), torch.zeros(2, 1, 8), **kw)
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP device only"):      # (a live flag since round 5: tests/test_laplace.py)
        m(torch.zeros(2, 5, 6), torch.zeros(2, 1, 8), clamp_vae_laplace=True)
    with pytest.raises(NotImplementedError):
        gru_vae.GRU_RNN(in_dim=6, out_dim=8, hidden_units=32, scale_in_out_flag=True)


@pytest.mark.parametrize("sel,flen", [(None, None), ([0, 2, 3], [7, 3, 5, 7]), ([1], [2, 4, 6, 1])])
def test_script_loss_loop_equals_the_vectorised_loss(sel, flen):
    """stage4.script_loss_loop (the per-utterance loop of train...:1363-1410 as the unchanged script runs it, the quirk of :1393
    included) and stage4.loss_terms (what Stage4Step uses) are the same number, for ragged windows and utterance selections."""
    import stage4
    torch.manual_seed(3)
    B, T, L, D, st = 4, 7, 3, 5, 2
    trajs = [{k: torch.randn(B, T, 2 * L if "lat" in k else D) for k in ("lat", "rec", "cv", "latcv", "reccyc")} for _ in range(2)]
    x = torch.randn(B, T, st + D)
    for half in (False, True):
        a = float(stage4.loss_terms(trajs, x, st, L, flen, sel, half))
        log = []
        b = float(stage4.script_loss_loop(trajs, x, st, L, flen, sel, half, log=log))
        assert abs(a - b) <= 2e-6 * abs(a), (a, b)
        assert len(log) == 2 * (B if sel is None else len(sel)) and len(log[0]) == 5


def test_direct_gradient_accumulation_is_vetoed_where_a_hook_could_read_p_grad_early(monkeypatch):
    """gru_vae._auto_sink_ok (plain `loss.backward()` flows, set_backward_overlap): adding a pass's parameter gradients straight into
    p.grad with the weight-gradient GEMMs on a side stream is only safe when nothing reads p.grad before backward() returns.  A
    post-accumulate-grad hook on a parameter (optimizer-in-backward) or an initialised process group of more than one rank (a DDP /
    FSDP reducer hooks AccumulateGrad from C++, invisible from Python) must send the pass down the autograd-returned-gradients path
    (ADVICE r5); with_process_group=True lifts the second condition for loops that all-reduce after backward()."""
    import torch
    import gru_vae

    class Ctx(object):
        needs_input_grad = (False,) * 7 + (True,) * 10

    m = gru_vae.GRU_RNN(in_dim=6, out_dim=8, hidden_units=32, kernel_size=3, dilation_size=2, scale_out_flag=False)
    seen = []
    monkeypatch.setattr(torch._C, "_will_engine_execute_node", lambda n: seen.append(n) or True)
    prev = gru_vae.set_backward_overlap(True)
    try:
        assert gru_vae._auto_sink_ok(Ctx(), m) and len(seen) == 10                 # plain leaves, no hooks, no process group
        h = m.gru.weight_hh_l0.register_post_accumulate_grad_hook(lambda p: None)
        assert not gru_vae._auto_sink_ok(Ctx(), m)
        h.remove()
        assert gru_vae._auto_sink_ok(Ctx(), m)
        monkeypatch.setattr(torch.distributed, "is_initialized", lambda: True)
        monkeypatch.setattr(torch.distributed, "get_world_size", lambda *a, **k: 2)
        assert not gru_vae._auto_sink_ok(Ctx(), m)
        gru_vae.set_backward_overlap(True, with_process_group=True)
        assert gru_vae._auto_sink_ok(Ctx(), m)
        gru_vae.set_backward_overlap(False)
        assert not gru_vae._auto_sink_ok(Ctx(), m)
    finally:
        gru_vae.set_backward_overlap(prev)
