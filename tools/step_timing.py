#!/usr/bin/env python
"""GPU-only debugging tool: per-phase cycle breakdown of the tuned persistent recurrence (one encoder pass).

    python tools/step_timing.py [B] [T]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd")]
import numpy as np
import torch

import _cabi
import gru_vae
import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda:0")
for kv in sys.argv[3:]:          # NAME=VALUE library options
    gru_vae._lib().set_option(kv.split("=")[0], int(kv.split("=")[1]))
P = synth.CycleVAEProblem(B=B, T=T, tag="timing")
enc = gru_vae.GRU_RNN(in_dim=54, out_dim=64, hidden_units=1024, scale_out_flag=False)
enc.load_state_dict({k: torch.from_numpy(v) for k, v in P.enc.items()})
enc = enc.to(dev).eval()
x, y = torch.from_numpy(P.x).to(dev), torch.from_numpy(P.y_in_enc).to(dev)
lib = gru_vae._lib()
with torch.no_grad():
    for flags in (0, _cabi.FLAG_STEP_TIMING):
        gru_vae._flags_extra = flags
        for _ in range(3):
            enc(x, y, clamp_vae=True, lat_dim=32)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            enc(x, y, clamp_vae=True, lat_dim=32)
        torch.cuda.synchronize()
        print("flags=%d: %.3f ms per encoder pass (whole pass, host-timed)" % (flags, 1e3 * (time.perf_counter() - t0) / n))
d, _ = enc.prepared(dev)
ws = enc._prep.workspace(B, T, dev)
out = lib.step_timing(d, B, T, ws.data_ptr(), torch.cuda.current_stream().cuda_stream)
names = ("front-end mfma", "flag wait", "loads+mfma", "reduce+gates+publish")
tot = sum(out[:4])
print("cycle sums over %d steps (mean over blocks | max over blocks | mean per step | share)" % T)
for i, nme in enumerate(names):
    print("  %-22s %12.0f %12.0f %10.1f %6.1f%%" % (nme, out[i], out[4 + i], out[i] / T, 100 * out[i] / tot))
print("  total per step: %.1f cycles" % (tot / T))
print("  status words (0: timeouts, 1: early-request hits, 2: probes):", lib.workspace_status(ws.data_ptr(), torch.cuda.current_stream().cuda_stream))
