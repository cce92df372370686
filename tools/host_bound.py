#!/usr/bin/env python
"""GPU-only: is the stage-4 step bound by the host?  Enqueues N unsynchronised steps (Stage4Step(sync=False)) and reports the host
time to enqueue them next to the time until the device has finished them.    python tools/host_bound.py [B] [T] [steps]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 80
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="hostbound")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
mods = []
for sd, i, o, enc in ((P.enc, 54, 64, True), (P.dec, 34, 50, False)):
    m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mods.append(m.to(dev).train())
data = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")] + [None]
step = stage4.Stage4Step(mods[0], mods[1], lat_dim=32, n_cyc=2, lr=1e-4, sync=False)
step.MAX_IN_FLIGHT = 10 ** 6
for _ in range(3):
    step(*data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step(*data)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("B=%d T=%d: host enqueues a step in %.2f ms; %d steps done on the device after %.2f ms per step" %
      (B, T, 1e3 * (t1 - t0) / steps, steps, 1e3 * (t2 - t0) / steps))
