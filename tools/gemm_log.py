#!/usr/bin/env python
"""GPU-only: shapes, tiles and rates of every GEMM of one train-mode encoder and decoder pass (forward + backward) at B x T.
    python tools/gemm_log.py [B] [T] [gemm_force]      (lines go to stderr; gemm_force = TM*10000 + TN*100 + ks)"""
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
gru_vae._lib().set_option("gemm_log", 1)
if len(sys.argv) > 3:
    gru_vae._lib().set_option("gemm_force", int(sys.argv[3]))
P = synth.CycleVAEProblem(B=B, T=T, tag="gemmlog")
for kind, sd, i, o, enc in (("enc", P.enc, 54, 64, True), ("dec", P.dec, 34, 50, False)):
    m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    for n, p in m.named_parameters():
        p.requires_grad_(n in TRAINABLE)
    x = torch.randn(B, T, i, device=dev, requires_grad=True)
    y = torch.zeros(B, 1, o, device=dev)
    for it in range(2):
        if it == 1:
            sys.stderr.write("---- %s pass B=%d T=%d: forward\n" % (kind, B, T))
        out = m(x, y, do=True, clamp_vae=enc, lat_dim=32)[0]
        torch.cuda.synchronize()
        if it == 1:
            sys.stderr.write("---- backward\n")
        out.sum().backward()
        torch.cuda.synchronize()
