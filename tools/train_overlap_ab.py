#!/usr/bin/env python
"""GPU-only: same-run A/B of the stage-4 step with the weight-gradient GEMMs on the side stream (default) or on the launch stream.
    python tools/train_overlap_ab.py [B] [T] [steps] [NAME=VALUE library options ...]"""
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

pos = [a for a in sys.argv[1:] if "=" not in a]
B = int(pos[0]) if len(pos) > 0 else 64
T = int(pos[1]) if len(pos) > 1 else 80
steps = int(pos[2]) if len(pos) > 2 else 8
for kv in sys.argv[1:]:
    if "=" in kv:
        gru_vae._lib().set_option(kv.split("=")[0], int(kv.split("=")[1]))
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
for rep in range(2):
    for name, kw in (("side stream", dict(overlap_wgrad=True)), ("launch stream", dict(overlap_wgrad=False))):
        enc, dec = mods()
        step = stage4.Stage4Step(enc, dec, lat_dim=32, n_cyc=2, lr=1e-4, **kw)
        for _ in range(2):
            step(*data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(*data)
        torch.cuda.synchronize()
        print("weight-gradient GEMMs on the %-14s %.2f ms per step" % (name, 1e3 * (time.perf_counter() - t0) / steps), flush=True)
