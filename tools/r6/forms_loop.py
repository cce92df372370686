"""Round 6 diagnostic: repeat the hu64 'stacked + side stream' step form N times per reverse-recurrence geometry and count how often
its weights after two steps differ from the one-stream form's (a rare mismatch was seen once in test_stage4step_forms_agree)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gru_vae as gv, synth, stage4
from train_util import TRAINABLE, make_masks
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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
def run(stack, overlap, fused, junk):
    enc, dec = module(P.enc, 10, 8, hid, True), module(P.dec, 6, 6, hid, False)
    step = stage4.Stage4Step(enc, dec, lat_dim=P.lat_dim, n_cyc=2, lr=1e-4, stack_rec_cv=stack, overlap_wgrad=overlap, fused=fused)
    gs = []
    for k in range(2):
        step(*args, masks=masks)
        if junk:      # churn the caching allocator between steps, like a test session does
            tmp = [torch.randn(np.random.randint(1, 1 << 18), device=dev) for _ in range(8)]
            del tmp
        torch.cuda.synchronize()
        gs.append(step.grads.flat.detach().cpu().numpy().copy())
    return gs, {n: p.detach().cpu().numpy().copy() for n, p in enc.named_parameters()}
for geom in (1, 0):
    gv._lib().set_option("train_bwd_geom", geom)
    base = run(False, False, False, False)
    bad = 0
    for it in range(N):
        for form in ((True, True, False), (True, True, True)):
            gs, w = run(*form, junk=(it % 2 == 1))
            g2 = np.abs(gs[1] - base[0][1]).max() / np.abs(base[0][1]).max()
            d = np.abs(w["gru.weight_hh_l0"] - base[1]["gru.weight_hh_l0"])
            if g2 > 2e-6 or (d > 1e-6).mean() > 2e-3:
                bad += 1
                worst = {n: float(np.abs(w[n] - base[1][n]).max()) for n in w}
                idx = np.argwhere(np.abs(gs[1] - base[0][1]) > 1e-6 * np.abs(base[0][1]).max())
                print("geom", geom, "iter", it, form, "grad2 rel %.2e frac %.2e" % (g2, (d > 1e-6).mean()), "bad grad entries", len(idx),
                      "first/last", idx[:1].ravel(), idx[-1:].ravel(), flush=True)
    print("geom", geom, "mismatches", bad, "of", 2 * N, flush=True)
