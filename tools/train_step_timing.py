#!/usr/bin/env python
"""GPU-only: wall time of one stage-4 step (cyc2 chain in train mode, dropout 0.5, loss, backward, Adam) through the drop-in
modules (BASELINE configs[2]).   python tools/train_step_timing.py [B ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import gru_vae
import synth
from train_util import TRAINABLE, chain_loss

dev = torch.device("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 64]:
    T = 80
    P = synth.CycleVAEProblem(B=B, T=T, tag="trainstep")
    mods = {}
    for kind, sd, i, o, enc in (("enc", P.enc, 54, 64, True), ("dec", P.dec, 34, 50, False)):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        m = m.to(dev).train()
        for n, p in m.named_parameters():
            p.requires_grad_(n in TRAINABLE)
        mods[kind] = m
    opt = torch.optim.Adam([p for m in mods.values() for p in m.parameters() if p.requires_grad], lr=1e-4)
    run = lambda kind, x, y_in, clamp, mk: mods[kind](x, y_in, do=True, clamp_vae=clamp >= 0, lat_dim=32)[0]
    none_masks = {"enc": [None] * 4, "dec": [None] * 6}

    def step():
        opt.zero_grad()
        loss = chain_loss(run, P, dev, none_masks)
        loss.backward()
        opt.step()
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("B=%d T=%d: %.1f ms per step, %.0f frames/s, loss %.3f, peak mem %.2f GB" % (
        B, T, 1e3 * dt, B * T / dt, loss.item(), torch.cuda.max_memory_allocated() / 2 ** 30))
