"""SURVEY 8(f) row 3, the part around the loader: joint statistics -> scale_in / scale_out and the initial feedback vectors, and
the checkpoint dictionary, against tests/golden/recipe.npz (the reference's own statements at train...:344-347, :357-359 and its
save_checkpoint :152-167 executed through ast by make_golden.py::case_recipe; statistics by sklearn's StandardScaler as
calc_stats_vc_joint.py uses it)."""
import os

import numpy as np
import pytest
from conftest import have_hdf5
import torch

import recipe
import synth


class _Net(torch.nn.Module):
    """CPU stand-in with the reference module's parameter names (the drop-in GRU_RNN is the same container; its forward needs the GPU)."""

    def __init__(self, in_dim, out_dim, scale_in, scale_out):
        super().__init__()
        import gru_vae
        m = gru_vae.GRU_RNN(in_dim=in_dim, out_dim=out_dim, hidden_units=32, kernel_size=3, dilation_size=2, do_prob=0.5,
                            scale_in_flag=scale_in, scale_out_flag=scale_out)
        self.m = m


def nets():
    import gru_vae
    enc = gru_vae.GRU_RNN(in_dim=10, out_dim=8, hidden_units=32, kernel_size=3, dilation_size=2, do_prob=0.5, scale_out_flag=False)
    dec = gru_vae.GRU_RNN(in_dim=6, out_dim=6, hidden_units=32, kernel_size=3, dilation_size=2, do_prob=0.5, scale_in_flag=False)
    return enc, dec


def test_joint_stats_equal_sklearn_standard_scaler(golden):
    g = golden("recipe")
    mean, scale = recipe.joint_stats([g["feat0"], g["feat1"], g["feat2"]])
    assert np.allclose(mean, g["mean"], rtol=1e-12, atol=1e-12) and np.allclose(scale, g["scale"], rtol=1e-12, atol=1e-12)
    const = np.concatenate([g["feat0"], np.full((17, 1), 2.5, np.float32)], 1)          # a constant column: scale 1, like sklearn
    assert recipe.joint_stats([const])[1][-1] == 1.0


@pytest.mark.skipif(not have_hdf5(), reason="no HDF5 C library on this machine")
def test_statistics_file_round_trip_feeds_the_scalers(golden, tmp_path):
    """calc_stats_vc_joint.py:83-127 -> train...:296-299, :344-347 through real HDF5 files (hdf5io = the HDF5 C library)."""
    import hdf5io
    g = golden("recipe")
    files = []
    for i in range(3):
        files.append(str(tmp_path / "hdf5" / ("utt%d.h5" % i)))
        hdf5io.write_hdf5(files[-1], "/feat_org_lf0", g["feat%d" % i])
    stats = str(tmp_path / "stats" / "stats_jnt.h5")
    recipe.write_joint_stats(stats, files)
    assert hdf5io.read_hdf5(stats, "/mean_feat_org_lf0_jnt").dtype == np.float64
    assert np.allclose(hdf5io.read_hdf5(stats, "/scale_feat_org_lf0_jnt"), g["scale"], rtol=1e-12, atol=1e-12)
    mean, std, mean_trg, std_trg = recipe.read_joint_stats(stats, stdim=4)
    assert mean.dtype == torch.float32 and torch.equal(mean_trg, mean[4:]) and torch.equal(std_trg, std[4:])
    enc, dec = nets()
    recipe.set_scalers(enc, dec, mean, std, stdim=4)
    assert np.allclose(enc.scale_in.weight.detach().numpy(), g["scale_in_w"], rtol=1e-6, atol=0)
    assert np.allclose(dec.scale_out.bias.detach().numpy(), g["scale_out_b"], rtol=1e-6, atol=1e-7)


def test_scalers_and_initial_feedback_equal_the_reference_statements(golden):
    g = golden("recipe")
    enc, dec = nets()
    mean_trg, std_trg = recipe.set_scalers(enc, dec, g["mean"], g["scale"], stdim=4)
    assert np.array_equal(enc.scale_in.weight.detach().numpy(), g["scale_in_w"])
    assert np.array_equal(enc.scale_in.bias.detach().numpy(), g["scale_in_b"])
    assert np.array_equal(dec.scale_out.weight.detach().numpy(), g["scale_out_w"])
    assert np.array_equal(dec.scale_out.bias.detach().numpy(), g["scale_out_b"])
    y_pp, y_in = recipe.initial_feedback(mean_trg, std_trg, 3, 4)
    assert np.array_equal(y_pp.numpy(), g["y_in_pp"]) and np.array_equal(y_in.numpy(), g["y_in_src"])


def test_checkpoint_has_the_reference_layout_and_round_trips(golden, tmp_path):
    g = golden("recipe")
    enc, dec = nets()
    recipe.set_scalers(enc, dec, g["mean"], g["scale"], stdim=4)
    params = [p for m in (enc, dec) for n, p in m.named_parameters() if not n.startswith("scale")]
    opt = torch.optim.Adam(params, lr=1e-4)
    for p in params:
        p.grad = torch.ones_like(p)
    opt.step()
    path = recipe.save_checkpoint(str(tmp_path), enc, dec, opt, np.random.get_state(), torch.get_rng_state(), 7)
    assert [os.path.basename(path)] == list(g["ck_file"])
    ck = torch.load(path, weights_only=False)
    assert sorted(ck.keys()) == list(g["ck_keys"]) and list(ck["model_encoder"].keys()) == list(g["ck_enc_keys"])
    assert sorted(ck["optimizer"]["param_groups"][0].keys()) == list(g["ck_opt_group_keys"])
    assert sorted(ck["optimizer"]["state"][0].keys()) == list(g["ck_opt_state_keys"])
    assert len(ck["optimizer"]["param_groups"][0]["params"]) == int(g["ck_opt_nparams"][1]) and int(g["ck_iterations"][0]) == ck["iterations"]
    enc2, dec2 = nets()
    opt2 = torch.optim.Adam([p for m in (enc2, dec2) for n, p in m.named_parameters() if not n.startswith("scale")], lr=1e-4)
    assert recipe.resume(path, enc2, dec2, opt2) == 7
    for a, b in zip(list(enc.state_dict().values()) + list(dec.state_dict().values()),
                    list(enc2.state_dict().values()) + list(dec2.state_dict().values())):
        assert torch.equal(a, b)
    assert torch.equal(opt.state_dict()["state"][3]["exp_avg"], opt2.state_dict()["state"][3]["exp_avg"])


class _FlatStep(object):
    """The fields of stage4.Stage4Step's fused optimiser (flat buffers), without a GPU."""

    def __init__(self, params):
        self.opt, self.params, self.grads = None, params, None
        n = sum(p.numel() for p in params)
        self.exp_avg, self.exp_avg_sq = torch.arange(n, dtype=torch.float32) * 1e-3, torch.arange(n, dtype=torch.float32) * 1e-6
        self.step_no, self.lr, self.betas, self.eps = 5, 1e-4, (0.9, 0.999), 1e-8


def test_flat_adam_state_converts_to_torch_layout_and_back(golden):
    enc, dec = nets()
    params = [p for m in (enc, dec) for n, p in m.named_parameters() if not n.startswith("scale")]
    st = _FlatStep(params)
    sd = recipe.adam_state_dict(st)
    opt = torch.optim.Adam(params, lr=1.0)
    opt.load_state_dict(sd)                                  # torch accepts it: the reference's resume path (:378-379) would too
    assert opt.state_dict()["param_groups"][0]["lr"] == 1e-4 and float(opt.state_dict()["state"][2]["step"]) == 5.0
    o = sum(p.numel() for p in params[:2])
    assert torch.equal(opt.state_dict()["state"][2]["exp_avg"].reshape(-1), st.exp_avg[o:o + params[2].numel()])
    st2 = _FlatStep(params)
    st2.exp_avg.zero_(); st2.exp_avg_sq.zero_(); st2.step_no = 0
    recipe.load_adam_state_dict(st2, opt.state_dict())
    assert st2.step_no == 5 and torch.equal(st2.exp_avg, st.exp_avg) and torch.equal(st2.exp_avg_sq, st.exp_avg_sq)
