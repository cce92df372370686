"""Round 6 measurement: how close does training come to the range limit of the gate-gradient exchange (status 5:
|gate gradient| >= ~234 in the persistent reverse recurrences), and what does a step cost when it crosses it?
B = 64 and B = 1 stage-4 steps (hu1024 / ld32 / cyc2, T = 80) with the input features scaled x1 / x4 / x16 and lr = 1e-3, 50 steps:
incidents (steps repeated on the fp32 per-step reverse path), time of a clean and of a repeated step, and the largest gate gradient
seen, bracketed by lowering the library's overflow threshold (option bwd_overflow_at) on short runs from the same initial weights.
    python tools/r6/status5_cliff.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gru_vae as gv, synth, stage4
from train_util import TRAINABLE
dev = torch.device("cuda:0")
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, L, NC, H = 80, 32, 2, 1024
W = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="bench/rank0")

def module(sd, i, o, enc):
    m = gv.GRU_RNN(in_dim=i, out_dim=o, hidden_units=H, kernel_size=3, dilation_size=2, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.to(dev).train()
    for n, p in m.named_parameters():
        p.requires_grad_(n in TRAINABLE)
    return m

def run(B, scale, lr, steps, ovf=None):
    lib = gv._lib()
    lib.reset_options()
    if ovf is not None:
        lib.set_option("bwd_overflow_at", ovf)
    P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="cliff%d" % B)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    args = [t(P.x * scale), t(P.cvx * scale), t(P.code_src), t(P.code_trg), t(P.y_in_enc), t(P.y_in_dec), None]
    enc, dec = module(W.enc, 54, 2 * L, True), module(W.dec, 2 + L, 50, False)
    step = stage4.Stage4Step(enc, dec, lat_dim=L, n_cyc=NC, lr=lr)
    torch.manual_seed(1)
    clean, rep, losses = [], [], []
    for k in range(steps):
        f0 = step.fallbacks
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = step(*args)
        torch.cuda.synchronize()
        dt = 1e3 * (time.perf_counter() - t0)
        (rep if step.fallbacks > f0 else clean).append(dt)
        losses.append(float(loss.item()))
    lib.reset_options()
    med = lambda v: float(np.median(v)) if v else float("nan")
    return step.fallbacks, med(clean[2:] if len(clean) > 4 else clean), med(rep), losses

print("| rows | features x | lr | steps | steps repeated on the fp32 reverse path | ms per clean step | ms per repeated step | loss first -> last | largest |gate gradient| seen |")
print("|---|---|---|---|---|---|---|---|---|")
for B in (64, 1):
    for scale in (1.0, 4.0, 16.0):
        fb, tc, tr, losses = run(B, scale, 1e-3, STEPS)
        # bracket the largest gate gradient of the first 10 steps: status 5 is raised from |g| * 256 >= threshold
        seen = "< 0.06"
        for thr in (15, 60, 250, 1000, 4000, 15000, 60000):
            if run(B, scale, 1e-3, 10, ovf=thr)[0] > 0:
                seen = ">= %.3g" % (thr / 256.0)
            else:
                break
        print("| %d | %g | 1e-3 | %d | %d | %.2f | %s | %.1f -> %.1f | %s |" % (B, scale, STEPS, fb, tc, ("%.2f" % tr) if fb else "-", losses[0], losses[-1], seen), flush=True)
