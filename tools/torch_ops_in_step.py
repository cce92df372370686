#!/usr/bin/env python
"""GPU-only: which torch operators (and how many device kernels each) run inside one fused stage-4 step beside the library's own
launches -- torch.profiler over three steps.
    python tools/torch_ops_in_step.py [B]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

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
step = stage4.Stage4Step(mods[0], mods[1], lat_dim=32, n_cyc=2, lr=1e-4)
for _ in range(3):
    step(*data)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step(*data)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = [s for s in e.stack if "cyclevae-vc_amd" in s or "stage4" in s][:2]
    print("%-28s x%-4.1f dev %6.1f us/step  %s" % (e.key, e.count / N, e.device_time_total / N, " <- ".join(s.split("/")[-1] for s in st)))
