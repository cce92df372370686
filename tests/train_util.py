"""Shared by the training tests: the stage-4 step (cyc2 chain in train mode + loss, reference
train_gru_cyclevae_gauss_batch.py:1326-1410) written once over an abstract `run_pass`, so the same code drives the HIP
modules on the GPU and the stock-torch checker on the CPU with identical dropout masks and eps."""
import numpy as np
import torch

import synth

K_MCD = (10.0 / 2.3025850929940456840179914546844) * 1.4142135623730950488016887242097


def make_masks(P, n_pass_enc, n_pass_dec, p=0.5, tag="masks"):
    """Deterministic inverted-dropout masks for every pass of a chain: list of (cmask [B,T,9C], gmask [T,B,H])."""
    out = {"enc": [], "dec": []}
    for kind, n, cin in (("enc", n_pass_enc, P.in_dim), ("dec", n_pass_dec, P.lat_dim + 2)):
        for i in range(n):
            cm = (synth.uniform01("%s/%s%d/c" % (tag, kind, i), (P.B, P.T, 9 * cin)) >= p).astype(np.float32) / (1 - p)
            gm = (synth.uniform01("%s/%s%d/g" % (tag, kind, i), (P.T, P.B, P.hidden)) >= p).astype(np.float32) / (1 - p)
            out[kind].append((cm, gm))
    return out


def chain_loss(run_pass, P, dev, masks, n_cyc=2):
    """run_pass(kind, x[B,T,C] tensor, y_in tensor, clamp_lat_dim, (cmask, gmask)) -> trj_out.  Returns the batch loss."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    x, cvx, cs, ct = t(P.x), t(P.cvx), t(P.code_src), t(P.code_trg)
    ye, yd, eps = t(P.y_in_enc), t(P.y_in_dec), t(P.eps)
    L, stdim = P.lat_dim, P.stdim
    smp = lambda par, e: par[:, :, :L] + torch.exp(par[:, :, L:] / 2) * e
    ie = idc = 0
    loss = 0.0
    prev = None
    tgt = x[:, :, stdim:]
    for i in range(n_cyc):
        e_in = x if i == 0 else torch.cat((x[:, :, :stdim], prev), 2)
        lat = run_pass("enc", e_in, ye, L, masks["enc"][ie]); ie += 1
        rec = run_pass("dec", torch.cat((cs, smp(lat, eps[i, 0])), 2), yd, -1, masks["dec"][idc]); idc += 1
        cv = run_pass("dec", torch.cat((ct, smp(lat, eps[i, 1])), 2), yd, -1, masks["dec"][idc]); idc += 1
        latcv = run_pass("enc", torch.cat((cvx, cv), 2), ye, L, masks["enc"][ie]); ie += 1
        reccyc = run_pass("dec", torch.cat((cs, smp(latcv, eps[i, 2])), 2), yd, -1, masks["dec"][idc]); idc += 1
        prev = reccyc
        # loss per utterance = mean over its frames, summed over utterances (train...:1363-1410); every utterance of
        # this synthetic batch has T frames, so the per-utterance loop collapses to one mean over frames per term
        loss = loss + (K_MCD * (rec - tgt).abs().sum(2)).mean(1).sum() + (K_MCD * (reccyc - tgt).abs().sum(2)).mean(1).sum()
        for par in (lat, latcv):
            mu, s = par[:, :, :L], par[:, :, L:]
            loss = loss + (0.5 * (s.exp() + mu * mu - s - 1.0).sum(2)).mean(1).sum()
    return loss


TRAINABLE = ("conv.conv.0.weight", "conv.conv.0.bias", "conv.conv.1.weight", "conv.conv.1.bias", "gru.weight_ih_l0",
             "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0", "out_1.weight", "out_1.bias")


def cpu_step(P, masks, n_cyc=2):
    """Loss and gradients from the stock-torch checker (oracle/torch_stock.py) on the CPU."""
    from oracle import torch_stock as ts
    params = {}
    sds = {"enc": P.enc, "dec": P.dec}
    leaf = {k: {n: torch.from_numpy(v.copy()).requires_grad_(n in TRAINABLE) for n, v in sd.items()} for k, sd in sds.items()}

    def run_pass(kind, x, y_in, clamp, mk):
        out = ts.train_forward_t(leaf[kind], x, y_in, torch.from_numpy(mk[0]), torch.from_numpy(mk[1]), clamp)
        return out

    loss = chain_loss(run_pass, P, torch.device("cpu"), masks, n_cyc)
    loss.backward()
    grads = {k: {n: leaf[k][n].grad.numpy() for n in TRAINABLE} for k in leaf}
    return float(loss.item()), grads
