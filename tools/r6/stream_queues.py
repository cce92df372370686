"""Round 6: do two HIP streams always run concurrently?  A one-block spin kernel (cvae_selftest_occupy, ~1 ms) on the launch stream and
on the k-th stream torch hands out: wall time ~1 ms = concurrent, ~2 ms = the two streams share a hardware queue and serialise.
Then the fused stage-4 step with that stream as its side stream."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import gru_vae as gv, synth, stage4
dev = torch.device("cuda:0")
lib = gv._lib()
main = torch.cuda.current_stream()
streams = [torch.cuda.Stream() for _ in range(10)]
def overlap_ms(s):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lib.selftest_occupy(1, 1024, 2000000, main.cuda_stream)
    lib.selftest_occupy(1, 1024, 2000000, s.cuda_stream)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)
overlap_ms(streams[0])
print("current stream", main.cuda_stream, "spin kernel alone: %.2f ms" % (overlap_ms(main) / 2))
for k, s in enumerate(streams):
    print("stream %d (handle %#x): main + this = %.2f ms" % (k, s.cuda_stream, overlap_ms(s)), flush=True)
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
enc, dec = mods()
for k in range(8):
    step = stage4.Stage4Step(enc, dec, lat_dim=32, n_cyc=2, lr=1e-4)
    step.side = streams[k]
    for _ in range(2):
        step(*data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        step(*data)
    torch.cuda.synchronize()
    print("side stream %d: %.2f ms per step" % (k, 1e3 * (time.perf_counter() - t0) / 8), flush=True)
    del step
