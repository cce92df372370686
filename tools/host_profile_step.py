#!/usr/bin/env python
"""GPU-only: where does the HOST spend its time in the fused stage-4 step (stage4.Stage4Step)?  Enqueue-only timing of the forward
and the backward half (sync=False: no host wait inside the step) next to the device time, then a cProfile of ten steps.
    python tools/host_profile_step.py [B]"""
import cProfile
import os
import pstats
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
dev = torch.device("cuda:0")
P = synth.CycleVAEProblem(B=B, T=80, bias_scale=0.0, tag="hostprof")
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
# host time of the halves of ONE step with an idle device in front (what a sync=True step sees after its status read)
orig_fb = step._fused_loss_backward
marks = {}


def timed_fb(*a, **k):
    marks["fwd_done"] = time.perf_counter()
    r = orig_fb(*a, **k)
    marks["bwd_done"] = time.perf_counter()
    return r


step._fused_loss_backward = timed_fb
hf, hb, dv = [], [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step(*data)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    hf.append(marks["fwd_done"] - t0); hb.append(marks["bwd_done"] - marks["fwd_done"]); dv.append(t2 - t0)
print("B=%d: host enqueues the forward half in %.2f ms, loss + backward in %.2f ms (rest of the call %.2f ms); device done %.2f ms after the call began"
      % (B, 1e3 * np.median(hf), 1e3 * np.median(hb), 1e3 * np.median([d for d in dv]) - 1e3 * np.median(hf) - 1e3 * np.median(hb), 1e3 * np.median(dv)))
step._fused_loss_backward = orig_fb
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(*data)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
