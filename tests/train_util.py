"""Shared by the training tests: the stage-4 step (cyc2 chain in train mode + loss, reference
train_gru_cyclevae_gauss_batch.py:1326-1410) written once over an abstract `run_pass`, so the same code drives the HIP
modules on the GPU and the stock-torch checker on the CPU with identical dropout masks and eps."""
import numpy as np
import torch

import synth

import stage4
from stage4 import TRAINABLE  # noqa: F401


def make_masks(P, n_pass_enc, n_pass_dec, p=0.5, tag="masks"):
    """Deterministic inverted-dropout masks for every pass of a chain: list of (cmask [B,T,9C], gmask [T,B,H])."""
    out = {"enc": [], "dec": []}
    for kind, n, cin in (("enc", n_pass_enc, P.in_dim), ("dec", n_pass_dec, P.lat_dim + 2)):
        for i in range(n):
            cm = (synth.uniform01("%s/%s%d/c" % (tag, kind, i), (P.B, P.T, 9 * cin)) >= p).astype(np.float32) / (1 - p)
            gm = (synth.uniform01("%s/%s%d/g" % (tag, kind, i), (P.T, P.B, P.hidden)) >= p).astype(np.float32) / (1 - p)
            out[kind].append((cm, gm))
    return out


def chain_loss(run_pass, P, dev, masks, n_cyc=2, stack_rec_cv=False):
    """stage4.chain_loss on a synthetic problem P; masks: dict from make_masks (numpy) or {"enc": [None]*k, "dec": ...}."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return stage4.chain_loss(run_pass, t(P.x), t(P.cvx), t(P.code_src), t(P.code_trg), t(P.y_in_enc), t(P.y_in_dec), t(P.eps),
                             P.lat_dim, n_cyc, masks, stack_rec_cv=stack_rec_cv)


def cpu_step(P, masks, n_cyc=2, stack_rec_cv=False):
    """Loss and gradients from the stock-torch checker (oracle/torch_stock.py) on the CPU."""
    from oracle import torch_stock as ts
    params = {}
    sds = {"enc": P.enc, "dec": P.dec}
    leaf = {k: {n: torch.from_numpy(v.copy()).requires_grad_(n in TRAINABLE) for n, v in sd.items()} for k, sd in sds.items()}

    def run_pass(kind, x, y_in, clamp, mk):
        out = ts.train_forward_t(leaf[kind.rstrip("2")], x, y_in, torch.from_numpy(mk[0]), torch.from_numpy(mk[1]), clamp)
        return out

    loss = chain_loss(run_pass, P, torch.device("cpu"), masks, n_cyc, stack_rec_cv)
    loss.backward()
    grads = {k: {n: leaf[k][n].grad.numpy() for n in TRAINABLE} for k in leaf}
    return float(loss.item()), grads


def golden_step_problem(g):
    """Inputs of tests/golden/stage4_step.npz / stage4_step_cyc4.npz (make_golden.py::case_step / case_step4): three utterances of
    20 / 15 / 9 frames, raw features zero-padded behind an utterance's end like the reference's dataset does, 12-frame windows."""
    ncyc = int(g["n_cyc"][0]) if "n_cyc" in g.files else 2
    P = synth.CycleVAEProblem(B=3, T=20, in_dim=10, out_dim=6, lat_dim=4, hidden=32, n_cyc=ncyc, bias_scale=0.1,
                              tag="step" if ncyc == 2 else "step%d" % ncyc)
    x, cvx = P.x.copy(), P.cvx.copy()
    for j, n in enumerate(g["flens"]):
        x[j, int(n):] = 0.0
        cvx[j, int(n):] = 0.0
    return P, x, cvx


def run_golden_windows(g, P, x, cvx, run_pass, optimizer, dev, stack_rec_cv=False):
    """Both windows of the golden step through stage4.chain_loss + backward + optimizer.step; run_pass(kind, x, y_in, clamp,
    (cmask, gmask) numpy pair, h_in=None) -> (trj, y_last, h_last).  Yields (window, loss tensor) after each backward and
    before the optimizer step, so that the caller can look at the gradients."""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    carry = None
    ncyc = P.n_cyc
    for w in range(2):
        s0, e0 = (int(v) for v in g["w%d_se" % w])
        masks = {k: [((g["w%d_%s%d_cmask" % (w, k, i)] * 2.0).astype(np.float32), (g["w%d_%s%d_gmask" % (w, k, i)] * 2.0).astype(np.float32))
                     for i in range(n)] for k, n in (("enc", 2 * ncyc), ("dec", 3 * ncyc))}
        optimizer.zero_grad()
        loss, carry, trajs = stage4.chain_loss(
            run_pass, t(x[:, s0:e0 + 1]), t(cvx[:, s0:e0 + 1]), t(P.code_src[:, s0:e0 + 1]), t(P.code_trg[:, s0:e0 + 1]),
            t(P.y_in_enc), t(P.y_in_dec), t(P.eps[:, :, :, s0:e0 + 1]), P.lat_dim, ncyc, masks,
            flen_acc=[int(v) for v in g["w%d_flen_acc" % w]], select_utt_idx=[int(v) for v in g["w%d_select" % w]],
            carry=carry, return_state=True, stack_rec_cv=stack_rec_cv)
        loss.backward()
        yield w, loss, trajs
        optimizer.step()
