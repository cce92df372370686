#!/usr/bin/env python
"""GPU-only: phase cycle sums of the persistent training forward recurrence (one encoder pass).
    python tools/train_phase_timing.py [B] [T] [option=value ...]      (library options, e.g. train_bwd_geom=1)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")]
import torch

import gru_vae
import synth
from train_util import TRAINABLE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 80
dev = torch.device("cuda:0")
gru_vae._lib().set_option("train_prof", 1)
for kv in sys.argv[3:]:
    gru_vae._lib().set_option(kv.split("=")[0], int(kv.split("=")[1]))
P = synth.CycleVAEProblem(B=B, T=T, tag="tphase")
m = gru_vae.GRU_RNN(in_dim=54, out_dim=64, hidden_units=1024, do_prob=0.5, scale_out_flag=False)
m.load_state_dict({k: torch.from_numpy(v) for k, v in P.enc.items()})
m = m.to(dev).train()
for n, p in m.named_parameters():
    p.requires_grad_(n in TRAINABLE)
x, y = torch.from_numpy(P.x).to(dev), torch.from_numpy(P.y_in_enc).to(dev)
for _ in range(3):
    for p in m.parameters():
        p.grad = None
    out = m(x, y, do=True, clamp_vae=True, lat_dim=32)[0]
    out.sum().backward()
torch.cuda.synchronize()
d, _ = m._prep_train.get(m, dev)
scr = m._prep_train.scratch_for(B, T, dev)
c = gru_vae._lib().train_debug_counters(d, B, T, scr.data_ptr(), torch.cuda.current_stream().cuda_stream)
names = ("poll", "loads+mfma", "reduce+cell", "publish")
for tag, v in (("forward", c[:4]), ("reverse (exact kernel only)", c[4:])):
    tot = sum(v)
    print("%s: per step %s  total %.0f cycles" % (tag, "  ".join("%s %.0f" % (n, q / T) for n, q in zip(names, v)), tot / T))
