"""CPU BASELINE (test/bench infrastructure, NOT product code): the hot path composed from stock torch.nn
modules the way the reference composes them -- Conv1d front-end, nn.GRU called once per frame with seq_len=1,
1x1 Conv1d projection, torch.cat to grow the trajectory (reference src/nets/gru_vae.py:322-455) and the cycle
loop of src/bin/train_gru_cyclevae_gauss_batch.py:1326-1338.  The reference's own Python never travels to the
GPU box; this restatement is what bench.py times on the host cores as `cpu_baseline` (kind "port").  It is
checked against the goldens recorded from the reference in tests/test_oracle_golden.py::test_torch_stock_*.
"""
import torch
import torch.nn.functional as F
from torch import nn


class StockGRURNN(nn.Module):
    def __init__(self, sd, in_dim, out_dim, hidden):
        super(StockGRURNN, self).__init__()
        t = lambda k: torch.from_numpy(sd[k].copy())
        self.sin = (t("scale_in.weight"), t("scale_in.bias")) if "scale_in.weight" in sd else None
        self.c0 = (t("conv.conv.0.weight"), t("conv.conv.0.bias"))
        self.c1 = (t("conv.conv.1.weight"), t("conv.conv.1.bias"))
        self.gru = nn.GRU(9 * in_dim + out_dim, hidden, 1, batch_first=True)
        with torch.no_grad():
            for name in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                getattr(self.gru, name).copy_(t("gru." + name))
        self.out = (t("out_1.weight"), t("out_1.bias"))
        self.sout = (t("scale_out.weight"), t("scale_out.bias")) if "scale_out.weight" in sd else None

    @torch.no_grad()
    def forward(self, x, y, h=None, clamp_lat_dim=-1):
        xt = x.transpose(1, 2)                                   # gru_vae.py:336-338
        if self.sin is not None:
            xt = F.conv1d(xt, *self.sin)
        xc = F.conv1d(F.conv1d(xt, *self.c0, padding=4), *self.c1, dilation=3).transpose(1, 2)   # :357
        trj = None
        for t in range(x.shape[1]):                              # :365 / :391-394
            o, h = self.gru(torch.cat((xc[:, t:t + 1], y), 2), h)
            y = F.conv1d(o.transpose(1, 2), *self.out).transpose(1, 2)
            trj = y if trj is None else torch.cat((trj, y), 1)
        if self.sout is not None:                                # :402-404
            out = F.conv1d(trj.transpose(1, 2), *self.sout).transpose(1, 2)
        else:
            out = trj
            if clamp_lat_dim >= 0:                               # :410-412
                out = torch.cat((out[:, :, :clamp_lat_dim],
                                 torch.clamp(out[:, :, clamp_lat_dim:], min=-13.815510557964274)), 2)
        return out, y, h


@torch.no_grad()
def cycle_chain(enc, dec, x, cvx, cs, ct, ye, yd, eps, n_cyc, L):
    """train_gru_cyclevae_gauss_batch.py:1326-1338, eval form, eps supplied [n_cyc,3,B,T,L]."""
    stdim = cvx.shape[2]
    smp = lambda p, e: p[:, :, :L] + torch.exp(p[:, :, L:] / 2) * e   # gru_vae.py:96
    out = {k: [] for k in ("lat", "rec", "cv", "latcv", "reccyc")}
    for i in range(n_cyc):
        e_in = x if i == 0 else torch.cat((x[:, :, :stdim], out["reccyc"][i - 1]), 2)
        lat = enc(e_in, ye, clamp_lat_dim=L)[0]
        rec = dec(torch.cat((cs, smp(lat, eps[i, 0])), 2), yd)[0]
        cv = dec(torch.cat((ct, smp(lat, eps[i, 1])), 2), yd)[0]
        latcv = enc(torch.cat((cvx, cv), 2), ye, clamp_lat_dim=L)[0]
        reccyc = dec(torch.cat((cs, smp(latcv, eps[i, 2])), 2), yd)[0]
        for k, v in zip(("lat", "rec", "cv", "latcv", "reccyc"), (lat, rec, cv, latcv, reccyc)):
            out[k].append(v)
    return out


def train_forward(sd, x, y_in, h_in, cmask, gmask, clamp_lat_dim=-1, requires=("conv.conv.0.weight", "conv.conv.0.bias",
                  "conv.conv.1.weight", "conv.conv.1.bias", "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0",
                  "gru.bias_hh_l0", "out_1.weight", "out_1.bias")):
    """Differentiable train-mode pass with SUPPLIED dropout masks (already scaled by 1/(1-p)): the checker for the HIP
    backward.  Same math as reference gru_vae.py:353-355 (conv_drop on the conv output), :378-382 (gru_drop on the GRU
    output feeding out_1, the carried h stays un-dropped).  Returns (out, y_last, h_last, params dict, x tensor)."""
    P = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    for k in requires:
        P[k].requires_grad_(True)
    xt = torch.from_numpy(x.copy()).requires_grad_(True)
    H = P["gru.weight_hh_l0"].shape[1]
    v = xt.transpose(1, 2)
    if "scale_in.weight" in P:
        v = F.conv1d(v, P["scale_in.weight"], P["scale_in.bias"])
    xc = F.conv1d(F.conv1d(v, P["conv.conv.0.weight"], P["conv.conv.0.bias"], padding=4), P["conv.conv.1.weight"],
                  P["conv.conv.1.bias"], dilation=3).transpose(1, 2) * torch.from_numpy(cmask)
    y = torch.from_numpy(y_in.copy())[:, 0]
    h = torch.zeros(x.shape[0], H) if h_in is None else torch.from_numpy(h_in.copy())[0]
    Wih, Whh, bih, bhh = P["gru.weight_ih_l0"], P["gru.weight_hh_l0"], P["gru.bias_ih_l0"], P["gru.bias_hh_l0"]
    Wo, bo = P["out_1.weight"][:, :, 0], P["out_1.bias"]
    gm = torch.from_numpy(gmask)
    ys = []
    for t in range(x.shape[1]):
        gi = torch.cat((xc[:, t], y), 1) @ Wih.t() + bih
        gh = h @ Whh.t() + bhh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = n + z * (h - n)
        y = (h * gm[t]) @ Wo.t() + bo
        ys.append(y)
    trj = torch.stack(ys, 1)
    if "scale_out.weight" in P:
        out = trj @ P["scale_out.weight"][:, :, 0].t() + P["scale_out.bias"]
    else:
        out = trj
        if clamp_lat_dim >= 0:
            out = torch.cat((out[:, :, :clamp_lat_dim], torch.clamp(out[:, :, clamp_lat_dim:], min=-13.815510557964274)), 2)
    return out, y, h, P, xt


def train_forward_t(P, x, y_in, cmask, gmask, clamp_lat_dim=-1, h_in=None, with_state=False):
    """train_forward on tensors that are already part of an autograd graph (P: dict of leaf tensors, x may carry grad):
    used to chain several passes (the stage-4 step) on the CPU checker.  Returns trj_out, or (trj_out, y_last [B,1,Co],
    h_last [1,B,H]) with with_state=True; h_in [1,B,H] continues a window (reference train...:1301)."""
    H = P["gru.weight_hh_l0"].shape[1]
    v = x.transpose(1, 2)
    if "scale_in.weight" in P:
        v = F.conv1d(v, P["scale_in.weight"], P["scale_in.bias"])
    xc = F.conv1d(F.conv1d(v, P["conv.conv.0.weight"], P["conv.conv.0.bias"], padding=4), P["conv.conv.1.weight"],
                  P["conv.conv.1.bias"], dilation=3).transpose(1, 2) * cmask
    y = y_in[:, 0]
    h = torch.zeros(x.shape[0], H) if h_in is None else h_in[0]
    Wih, Whh, bih, bhh = P["gru.weight_ih_l0"], P["gru.weight_hh_l0"], P["gru.bias_ih_l0"], P["gru.bias_hh_l0"]
    Wo, bo = P["out_1.weight"][:, :, 0], P["out_1.bias"]
    ys = []
    for t in range(x.shape[1]):
        gi = torch.cat((xc[:, t], y), 1) @ Wih.t() + bih
        gh = h @ Whh.t() + bhh
        r = torch.sigmoid(gi[:, :H] + gh[:, :H])
        z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
        h = n + z * (h - n)
        y = (h * gmask[t]) @ Wo.t() + bo
        ys.append(y)
    trj = torch.stack(ys, 1)
    if "scale_out.weight" in P:
        out = trj @ P["scale_out.weight"][:, :, 0].t() + P["scale_out.bias"]
    elif clamp_lat_dim >= 0:
        out = torch.cat((trj[:, :, :clamp_lat_dim], torch.clamp(trj[:, :, clamp_lat_dim:], min=-13.815510557964274)), 2)
    else:
        out = trj
    return (out, y.unsqueeze(1), h.unsqueeze(0)) if with_state else out
