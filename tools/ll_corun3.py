#!/usr/bin/env python
"""GPU-only: how do THREE word-exchange recurrences of three rows each (what an 8- or 9-row pass would split into) share the chip?
Three encoder passes of 3 rows x T frames on three streams against one after the other, and against ONE pass of 9 rows on the
row-tile kernel (what such a pass runs today).  Modules with separate workspaces, so the launches do not share buffers."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd")]
import numpy as np, torch
import gru_vae, synth
dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 80
W = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="bench/rank0")
def mod():
    m = gru_vae.GRU_RNN(in_dim=54, out_dim=64, hidden_units=1024, kernel_size=3, dilation_size=2, scale_in_flag=True, scale_out_flag=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.enc.items()})
    return m.to(dev).eval()
encs = [mod() for _ in range(3)]
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
P = synth.CycleVAEProblem(B=9, T=T, bias_scale=0.0, tag="corun3")
x, y = tt(P.x), tt(P.y_in_enc)
xs, ys = [x[3 * i:3 * i + 3].contiguous() for i in range(3)], [y[3 * i:3 * i + 3].contiguous() for i in range(3)]
with torch.no_grad():
    def timed(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
    print("one 3-row pass (word exchange)            %.3f ms" % timed(lambda: encs[0](xs[0], ys[0], clamp_vae=True, lat_dim=32)))
    print("one 9-row pass (row-tile kernel, today)   %.3f ms" % timed(lambda: encs[0](x, y, clamp_vae=True, lat_dim=32)))
    def serial():
        for i in range(3): encs[i](xs[i], ys[i], clamp_vae=True, lat_dim=32)
    print("three 3-row passes one after the other    %.3f ms" % timed(serial))
    ss = [torch.cuda.Stream() for _ in range(3)]
    outs = [None] * 3
    def three(n=3):
        cur = torch.cuda.current_stream()
        for i in range(n):
            ss[i].wait_stream(cur)
            with torch.cuda.stream(ss[i]):
                outs[i] = encs[i](xs[i], ys[i], clamp_vae=True, lat_dim=32)[0]
        for i in range(n): cur.wait_stream(ss[i])
    print("two 3-row passes on two streams           %.3f ms" % timed(lambda: three(2)))
    print("three 3-row passes on three streams       %.3f ms" % timed(three))
    ref = encs[0](x, y, clamp_vae=True, lat_dim=32)[0]
    torch.cuda.synchronize()
    print("max |d| vs the 9-row pass: %.2e" % max(float((outs[i] - ref[3 * i:3 * i + 3]).abs().max()) for i in range(3)))
gru_vae.check_status()
