import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "cyclevae-vc_amd")]
import numpy as np, torch
import gru_vae, synth
dev = torch.device("cuda:0")
for kv in sys.argv[1:]:          # NAME=VALUE library options (cvae_set_option), e.g. ll_backoff=14
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
def one():
    lat = enc(xu, yu, clamp_vae=True, lat_dim=L)[0]
    z = torch.mean(gru_vae.sampling_vae_batch(lat.unsqueeze(0).repeat(300, 1, 1), lat_dim=L), 0)
    return dec(torch.cat((cu, z), 1), ydu)[0]
with torch.no_grad():
    for _ in range(3): one()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): one()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print("single utterance T=637 (enc + 300-draw mean + dec): %.3f ms = %.0f frames/s, %.2f us per dependent step" % (1e3*dt, 637/dt, 1e6*dt/1274))

# the whole stage-6 network path of one utterance pair (decode...:302-323): five passes one by one vs two stacked launches
import stage6
PT = synth.CycleVAEProblem(B=1, T=660, bias_scale=0.0, tag="bench/utt_trg")
xt_ = tt(PT.x[0])
code = lambda i, T: tt(np.tile(np.eye(2, dtype=np.float32)[i], (T, 1)))
def five():
    ls = enc(xu, yu, clamp_vae=True, lat_dim=L)[0]
    zs = torch.mean(gru_vae.sampling_vae_batch(ls.unsqueeze(0).repeat(300, 1, 1), lat_dim=L), 0)
    lt = enc(xt_, yu, clamp_vae=True, lat_dim=L)[0]
    zt = torch.mean(gru_vae.sampling_vae_batch(lt.unsqueeze(0).repeat(300, 1, 1), lat_dim=L), 0)
    a = dec(torch.cat((code(1, 637), zs), 1), ydu)[0]
    b = dec(torch.cat((code(0, 637), zs), 1), ydu)[0]
    c = dec(torch.cat((code(1, 660), zt), 1), ydu)[0]
    return a, b, c
def two():
    return stage6.convert_pair(enc, dec, xu, xt_, yu, ydu, ydu, L, n_smpl_dec=300)
for name, fn in (("five passes one by one", five), ("stage6.convert_pair (2 stacked launches)", two)):
    with torch.no_grad():
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("stage-6 network path, T=637/660 pair, %-42s %.3f ms = %.0f converted frames/s" % (name + ":", 1e3*dt, 637/dt))
def five_pairs():
    return stage6.convert_pairs(enc, dec, [(xu, xt_)] * 5, yu, ydu, ydu, L, n_smpl_dec=300)
with torch.no_grad():
    for _ in range(2): five_pairs()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): five_pairs()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("stage-6 network path, FIVE T=637/660 pairs per call (10 / 15 stacked rows):                 %.3f ms = %.0f converted frames/s" % (1e3*dt, 5*637/dt))
def ten_pairs():
    return stage6.convert_pairs(enc, dec, [(xu, xt_)] * 10, yu, ydu, ydu, L, n_smpl_dec=300)
with torch.no_grad():
    for _ in range(2): ten_pairs()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): ten_pairs()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("stage-6 network path, TEN T=637/660 pairs per call (20 / 30 stacked rows):                  %.3f ms = %.0f converted frames/s" % (1e3*dt, 10*637/dt))
def listed():
    return stage6.convert_list(enc, dec, [[(xu, xt_)]] * 8, yu, ydu, ydu, L, n_smpl_dec=300)
with torch.no_grad():
    for _ in range(2): listed()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): listed()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 24
print("stage-6 network path, a LIST of 8 pairs, encoder of pair g+1 beside the decoder of pair g (convert_list):  %.3f ms per pair = %.0f converted frames/s" % (1e3*dt, 637/dt))
for W in (96, 160, 224, 330, [40, 224]):
    with torch.no_grad():
        fn = lambda: stage6.convert_pair(enc, dec, xu, xt_, yu, ydu, ydu, L, n_smpl_dec=300, window=W)
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print("stage-6 network path, T=637/660 pair as a wavefront of windows %s (decoder of window w beside encoder of w+1): %.3f ms = %.0f converted frames/s" % (W, 1e3*dt, 637/dt))
