#!/usr/bin/env python
"""GPU-only: where does the HOST spend its time in the unchanged-script flow (ten GRU_RNN passes with per-pass autograd, the script's
per-utterance loss loop, torch.optim.Adam)?  cProfile over a few steps.   python tools/host_profile_dropin.py [B] [script_loss 0/1]"""
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
script = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
dev = torch.device("cuda:0")
P = synth.CycleVAEProblem(B=B, T=80, bias_scale=0.0, tag="hostprof")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
mods = []
for sd, i, o, enc in ((P.enc, 54, 64, True), (P.dec, 34, 50, False)):
    m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    mods.append(m.to(dev).train())
data = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")] + [None]
step = stage4.Stage4Step(mods[0], mods[1], lat_dim=32, n_cyc=2, lr=1e-4, fused=False, stack_rec_cv=False, overlap_wgrad=False,
                         script_loss=script)
for _ in range(3):
    step(*data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step(*data)
torch.cuda.synchronize()
print("B=%d script_loss=%d: %.2f ms per step" % (B, script, 1e2 * (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    step(*data)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
