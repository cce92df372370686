"""Round 6 diagnostic: the step forms of test_stage4step_forms_agree (hu64) under both reverse-recurrence geometries,
with the step-2 flat gradient compared as well (is a W_hh difference after two steps Adam noise or a race?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gru_vae as gv, synth, stage4
from train_util import TRAINABLE, make_masks
dev = torch.device("cuda:0")
def module(sd, i, o, h, enc):
    m = gv.GRU_RNN(in_dim=i, out_dim=o, hidden_units=h, kernel_size=3, dilation_size=2, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    for n, p in m.named_parameters():
        p.requires_grad_(n in TRAINABLE)
    return m
hid, B, T = 64, 20, 9
P = synth.CycleVAEProblem(in_dim=10, out_dim=6, lat_dim=4, B=B, T=T, hidden=hid, n_cyc=2, bias_scale=0.05, tag="forms%d" % hid)
masks_np = make_masks(P, 4, 6)
masks = {k: [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b in v] for k, v in masks_np.items()}
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
args = [t(P.x), t(P.cvx), t(P.code_src), t(P.code_trg), t(P.y_in_enc), t(P.y_in_dec), t(P.eps)]
for geom in (0, 1, 1):
    gv._lib().set_option("train_bwd_geom", geom)
    res = []
    for stack, overlap, fused in ((False, False, False), (True, False, False), (True, True, False), (True, True, False)):
        enc, dec = module(P.enc, 10, 8, hid, True), module(P.dec, 6, 6, hid, False)
        step = stage4.Stage4Step(enc, dec, lat_dim=P.lat_dim, n_cyc=2, lr=1e-4, stack_rec_cv=stack, overlap_wgrad=overlap, fused=fused)
        gs = []
        for k in range(2):
            step(*args, masks=masks)
            torch.cuda.synchronize()
            gs.append(step.grads.flat.detach().cpu().numpy().copy())
        res.append((gs, enc.gru.weight_hh_l0.detach().cpu().numpy().copy()))
    for i in range(1, len(res)):
        g1 = np.abs(res[i][0][0] - res[0][0][0]).max() / np.abs(res[0][0][0]).max()
        g2 = np.abs(res[i][0][1] - res[0][0][1]).max() / np.abs(res[0][0][1]).max()
        d = np.abs(res[i][1] - res[0][1])
        print("geom", geom, "form", i, "grad1 rel %.2e grad2 rel %.2e  W_hh max|d| %.2e frac>1e-6 %.2e" % (g1, g2, d.max(), (d > 1e-6).mean()), flush=True)
