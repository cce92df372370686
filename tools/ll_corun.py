#!/usr/bin/env python
"""GPU-only: do two word-exchange recurrences (an encoder pass and a decoder pass of one utterance each) run side by side on two
streams as fast as one alone?  The premise of overlapping the encoder of one utterance with the decoder of the previous one."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd")]
import numpy as np, torch
import gru_vae, synth
dev = torch.device("cuda:0")
for kv in sys.argv[1:]:
    gru_vae._lib().set_option(kv.split("=")[0], int(kv.split("=")[1]))
L = 32
W = synth.CycleVAEProblem(B=1, T=1, bias_scale=0.0, tag="bench/rank0")
def mod(sd, i, o, enc):
    m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=1024, kernel_size=3, dilation_size=2, scale_in_flag=enc, scale_out_flag=not enc)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.to(dev).eval()
enc, dec = mod(W.enc, 54, 64, True), mod(W.dec, 34, 50, False)
tt = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
PU = synth.CycleVAEProblem(B=1, T=637, bias_scale=0.0, tag="bench/utt")
xu, yu, cu, ydu = tt(PU.x[0]), tt(PU.y_in_enc), tt(PU.code_trg[0]), tt(PU.y_in_dec)
with torch.no_grad():
    lat = enc(xu, yu, clamp_vae=True, lat_dim=L)[0]
    z = torch.mean(gru_vae.sampling_vae_batch(lat.unsqueeze(0).repeat(300, 1, 1), lat_dim=L), 0)
    din = torch.cat((cu, z), 1)
    ref_e, ref_d = enc(xu, yu, clamp_vae=True, lat_dim=L)[0].clone(), dec(din, ydu)[0].clone()
    torch.cuda.synchronize()
    def timed(fn, n=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
    print("encoder pass alone          %.3f ms" % timed(lambda: enc(xu, yu, clamp_vae=True, lat_dim=L)))
    print("decoder pass alone          %.3f ms" % timed(lambda: dec(din, ydu)))
    def serial():
        enc(xu, yu, clamp_vae=True, lat_dim=L); dec(din, ydu)
    print("one after the other         %.3f ms" % timed(serial))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = {}
    def both():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            outs["e"] = enc(xu, yu, clamp_vae=True, lat_dim=L)[0]
        with torch.cuda.stream(s2):
            outs["d"] = dec(din, ydu)[0]
        cur.wait_stream(s1); cur.wait_stream(s2)
    print("side by side on two streams %.3f ms" % timed(both))
    torch.cuda.synchronize()
    print("results equal to the serial ones:", bool(torch.equal(outs["e"], ref_e)), bool(torch.equal(outs["d"], ref_d)))
