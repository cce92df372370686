"""Round 6: per-step wall times of the fused stage-4 step (64 x 80) over many steps and several fresh Stage4Step objects, per library
option set -- is there a slow mode (the 16-unit reverse recurrence on half the chip next to side-stream GEMMs)?
    python tools/r6/step_jitter.py "opt=v opt=v" "opt=v" ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gru_vae as gv, synth, stage4
dev = torch.device("cuda:0")
B, T = 64, 80
P = synth.CycleVAEProblem(B=B, T=T, bias_scale=0.0, tag="jitter")
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
data = [tt(getattr(P, n)) for n in ("x", "cvx", "code_src", "code_trg", "y_in_enc", "y_in_dec")] + [None]
def mods():
    out = []
    for sd, i, o, enc in ((P.enc, 54, 64, True), (P.dec, 34, 50, False)):
        m = gv.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, do_prob=0.5, scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        out.append(m.to(dev).train())
    return out
for spec in sys.argv[1:] or [""]:
    for rep in range(4):
        gv._lib().reset_options()
        for kv in spec.split():
            gv._lib().set_option(kv.split("=")[0], int(kv.split("=")[1]))
        enc, dec = mods()
        step = stage4.Stage4Step(enc, dec, lat_dim=32, n_cyc=2, lr=1e-4)
        for _ in range(2):
            step(*data)
        torch.cuda.synchronize()
        ts = []
        for _ in range(24):
            t0 = time.perf_counter()
            step(*data)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        ts = np.array(ts)
        print("[%s] object %d: median %.2f  min %.2f  max %.2f  steps > 1.1 x median: %d  coop_fallback %s fallbacks %d" %
              (spec, rep, np.median(ts), ts.min(), ts.max(), int((ts > 1.1 * np.median(ts)).sum()), step.coop_fallback, step.fallbacks), flush=True)
        del step, enc, dec
