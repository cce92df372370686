#!/usr/bin/env python
"""GPU-only: the stage-6 forms of tools/b1_timing.py a few times each, for a rocprofv3 kernel trace (tools/prof_stage6.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd")]
import numpy as np, torch
import gru_vae, synth, stage6
dev = torch.device("cuda:0")
L = 32
W_ = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="bench/rank0")
def mod(sd, i, o, enc):
    m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, kernel_size=3, dilation_size=2, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval()
enc, dec = mod(W_.enc, 54, 64, True), mod(W_.dec, 34, 50, False)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
PU = synth.CycleVAEProblem(B=1, T=637, bias_scale=0.0, tag="bench/utt")
PT = synth.CycleVAEProblem(B=1, T=660, bias_scale=0.0, tag="bench/utt_trg")
xu, yu, ydu, xt_ = tt(PU.x[0]), tt(PU.y_in_enc), tt(PU.y_in_dec), tt(PT.x[0])
with torch.no_grad():
    for _ in range(3):
        stage6.convert_pair(enc, dec, xu, xt_, yu, ydu, ydu, L, n_smpl_dec=300)
        stage6.convert_pair(enc, dec, xu, xt_, yu, ydu, ydu, L, n_smpl_dec=300, window=224)
        stage6.convert_list(enc, dec, [[(xu, xt_)]] * 4, yu, ydu, ydu, L, n_smpl_dec=300)
        stage6.convert_pairs(enc, dec, [(xu, xt_)] * 10, yu, ydu, ydu, L, n_smpl_dec=300)
        stage6.convert_list(enc, dec, [[(xu, xt_)] * 10] * 3, yu, ydu, ydu, L, n_smpl_dec=300)
    torch.cuda.synchronize()
