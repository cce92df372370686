#!/usr/bin/env python
"""GPU-only: ms per stage-4 step for the forms of stage4.Stage4Step (fused glue or torch ops, host sync per step or not), and the
gradient / weight differences between the fused and the torch-op form after one and two steps.
    python tools/train_step_variants.py [B] [T] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import gru_vae
import stage4
import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 80
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="variants")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def mods():
    out = []
    for sd, i, o, enc in ((P.enc, 54, 64, True), (P.dec, 34, 50, False)):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        out.append(m.to(dev).train())
    return out


data = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec", "eps")]
for name, kw in (("fused, sync per step", dict(fused=True)), ("fused, no sync", dict(fused=True, sync=False)),
                 ("torch glue + torch Adam", dict(fused=False))):
    enc, dec = mods()
    step = stage4.Stage4Step(enc, dec, lat_dim=32, n_cyc=2, lr=1e-4, **kw)
    for _ in range(2):
        step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(*data)
    torch.cuda.synchronize()
    print("%-28s %.2f ms per step" % (name, 1e3 * (time.perf_counter() - t0) / steps))

# fused vs torch form: same masks (seeded Philox: torch.manual_seed before every step), same eps
res = {}
for fused in (False, True):
    enc, dec = mods()
    step = stage4.Stage4Step(enc, dec, lat_dim=32, n_cyc=2, lr=1e-4, fused=fused)
    snaps = []
    for k in range(2):
        torch.manual_seed(100 + k)
        loss = step(*data)
        torch.cuda.synchronize()
        snaps.append((float(loss), step.grads.flat.clone(), torch.cat([p.detach().reshape(-1) for p in step.params]).clone()))
    res[fused] = snaps
for k in range(2):
    (l0, g0, w0), (l1, g1, w1) = res[False][k], res[True][k]
    print("step %d: loss %.6f / %.6f, grad max|d| %.3e (max|g| %.3e), weights max|d| %.3e" %
          (k + 1, l0, l1, float((g0 - g1).abs().max()), float(g0.abs().max()), float((w0 - w1).abs().max())))
    o = 0
    for m, kind in ((step.mods["enc"], "enc"), (step.mods["dec"], "dec")):
        for n, p in m.named_parameters():
            if p.requires_grad:
                d = (g0[o:o + p.numel()] - g1[o:o + p.numel()]).abs().max()
                print("   %s %-22s grad max|d| %.3e of %.3e   |g| min %.3e" % (kind, n, float(d), float(g0[o:o + p.numel()].abs().max()),
                                                                           float(g0[o:o + p.numel()].abs().min())))
                o += p.numel()
