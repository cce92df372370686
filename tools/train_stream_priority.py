#!/usr/bin/env python
"""GPU-only experiment: does a HIGH-priority main stream (recurrences, dx chain) against the default-priority side stream
(weight-gradient GEMMs) shorten the stage-4 step?    python tools/train_stream_priority.py [B] [T] [steps]"""
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
P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="prio")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
print("priority range", torch.cuda.Stream.priority_range())


def mods():
    out = []
    for sd, i, o, enc in ((P.enc, 54, 64, True), (P.dec, 34, 50, False)):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        out.append(m.to(dev).train())
    return out


data = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")] + [None]
for name, main_prio in (("default main stream", None), ("high-priority main stream", -1), ("default main stream", None), ("high-priority main stream", -1)):
    enc, dec = mods()
    step = stage4.Stage4Step(enc, dec, lat_dim=32, n_cyc=2, lr=1e-4)
    s = torch.cuda.Stream(priority=main_prio) if main_prio is not None else torch.cuda.current_stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for _ in range(2):
            step(*data)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(*data)
        torch.cuda.synchronize()
    print("%-28s %.2f ms per step" % (name, 1e3 * (time.perf_counter() - t0) / steps))
