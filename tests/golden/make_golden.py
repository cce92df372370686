"""Generate golden vectors by RUNNING THE REFERENCE (build container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Imports /root/reference/src/nets/gru_vae.py (needs only torch + numpy), feeds it the repo's deterministic
synthetic weights / features (cyclevae-vc_amd/synth.py) and records the reference's outputs.  Only DATA is
written: inputs are regenerated from (seed, tag) on any box, weights are never stored (their SHA-256 is).
The reference hard-codes `.cuda()` in sampling_vae_batch (gru_vae.py:91,94); in this GPU-less container
`torch.Tensor.cuda` is shimmed to identity and `torch.randn` inside the reference module is replaced by a
feeder of the supplied eps, so the *reference's own code* runs on the eps the parity tests inject.

The INT fixtures run the reference's `train_generator` (train_gru_cyclevae_gauss_batch.py:45-149): only that
function definition is exec'ed (via ast) because the script imports h5py / torchvision / dtw_c, absent here.
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "cyclevae-vc_amd"))
sys.path.insert(0, "/root/reference/src/nets")

import synth  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self
import gru_vae as ref  # noqa: E402  (the reference)

torch.set_num_threads(8)


def to_t(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def build(sd, in_dim, out_dim, hidden, enc):
    m = ref.GRU_RNN(in_dim=in_dim, out_dim=out_dim, hidden_units=hidden, kernel_size=3, dilation_size=2,
                    scale_out_flag=not enc, scale_in_flag=enc)
    m.load_state_dict(to_t(sd))
    m.eval()
    return m


class EpsFeeder(object):
    """Stands in for torch.randn inside the reference module: returns the queued eps tensors in order."""

    def __init__(self):
        self.q = []

    def __call__(self, *shape):
        e = self.q.pop(0)
        assert tuple(e.shape) == tuple(shape), (e.shape, shape)
        return torch.from_numpy(e.copy())


class _TorchProxy(object):
    def __init__(self, feeder):
        self._f = feeder

    def __getattr__(self, k):
        return self._f if k == "randn" else getattr(torch, k)


FEED = EpsFeeder()
ref.torch = _TorchProxy(FEED)


def ref_sample(param, eps, lat_dim):
    FEED.q.append(eps)
    return ref.sampling_vae_batch(param, lat_dim=lat_dim)


def run_pass(m, x, y_in, h_in=None, clamp=False, lat_dim=16):
    with torch.no_grad():
        o, y, h = m(torch.from_numpy(x), torch.from_numpy(y_in),
                    h_in=None if h_in is None else torch.from_numpy(h_in), clamp_vae=clamp, lat_dim=lat_dim)
    return o.numpy(), y.numpy(), h.numpy()


def chain(encm, decm, P):
    """Reference modules composed as train_gru_cyclevae_gauss_batch.py:1326-1338 (eval, do=False)."""
    L = P.lat_dim
    tt = torch.from_numpy
    x, cvx, cs, ct = tt(P.x), tt(P.cvx), tt(P.code_src), tt(P.code_trg)
    ye, yd = tt(P.y_in_enc), tt(P.y_in_dec)
    out = {k: [] for k in ("lat", "rec", "cv", "latcv", "reccyc")}
    with torch.no_grad():
        for i in range(P.n_cyc):
            e_in = x if i == 0 else torch.cat((x[:, :, :P.stdim], out["reccyc"][i - 1]), 2)
            lat = encm(e_in, ye, clamp_vae=True, lat_dim=L)[0]
            rec = decm(torch.cat((cs, ref_sample(lat, P.eps[i, 0], L)), 2), yd)[0]
            cv = decm(torch.cat((ct, ref_sample(lat, P.eps[i, 1], L)), 2), yd)[0]
            latcv = encm(torch.cat((cvx, cv), 2), ye, clamp_vae=True, lat_dim=L)[0]
            reccyc = decm(torch.cat((cs, ref_sample(latcv, P.eps[i, 2], L)), 2), yd)[0]
            for k, v in zip(("lat", "rec", "cv", "latcv", "reccyc"), (lat, rec, cv, latcv, reccyc)):
                out[k].append(v)
    return {k: np.stack([v.numpy() for v in vs]) for k, vs in out.items()}


def save(name, **arrs):
    p = os.path.join(HERE, name + ".npz")
    np.savez_compressed(p, **arrs)
    print("wrote %s (%.1f KB)" % (p, os.path.getsize(p) / 1024.0))


def case_tiny():
    """Per-op goldens at tiny dims (H=32, Cin=6, L=4, B=2, T=12), non-zero biases."""
    P = synth.CycleVAEProblem(B=2, T=12, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="tiny")
    encm, decm = build(P.enc, 6, 8, 32, True), build(P.dec, 6, 4, 32, False)
    with torch.no_grad():
        xin = encm.scale_in(torch.from_numpy(P.x).transpose(1, 2))
        xconv = encm.conv(xin).transpose(1, 2)
        u0 = torch.cat((xconv[:, :1], torch.from_numpy(P.y_in_enc)), 2)
        out0, h0 = encm.gru(u0)
        y0 = encm.out_1(out0.transpose(1, 2)).transpose(1, 2)
    lat, ylast, hlast = run_pass(encm, P.x, P.y_in_enc, clamp=True, lat_dim=4)
    lat_noclamp = run_pass(encm, P.x, P.y_in_enc, clamp=False, lat_dim=4)[0]
    z = ref_sample(torch.from_numpy(lat), P.eps[0, 0], 4).numpy()
    dec_in = np.concatenate([P.code_src, z], 2)
    rec, rec_y, rec_h = run_pass(decm, dec_in, P.y_in_dec)
    crit = ref.TWFSEloss()
    a, b = torch.from_numpy(rec[0]), torch.from_numpy(P.x[0, :, P.stdim:])
    l1 = [v.item() for v in crit(a, b, L2=False, GV=False)]
    l2 = [v.item() for v in crit(a, b, L2=True, GV=False)]
    kl = ref.loss_vae(torch.from_numpy(lat[0]), lat_dim=4).item()
    save("tiny_ops", sha_enc=synth.sha256_state(P.enc), sha_dec=synth.sha256_state(P.dec),
         xconv=xconv.numpy(), h_step0=h0.numpy(), y_step0=y0.numpy(), lat=lat, lat_y=ylast, lat_h=hlast,
         lat_noclamp=lat_noclamp, z=z, rec=rec, rec_y=rec_y, rec_h=rec_h,
         twfse_l1=np.array(l1, np.float32), twfse_l2=np.array(l2, np.float32), kl=np.float32(kl))
    save("tiny_chain", **chain(encm, decm, P))


def case_full():
    """hu1024/ld32 single passes (B=2,T=80), 2-D path, state carry, cyc2 chain."""
    P = synth.CycleVAEProblem(B=2, T=80, bias_scale=0.05, tag="full")
    encm, decm = build(P.enc, 54, 64, 1024, True), build(P.dec, 34, 50, 1024, False)
    lat, lat_y, lat_h = run_pass(encm, P.x, P.y_in_enc, clamp=True, lat_dim=32)
    z = ref_sample(torch.from_numpy(lat), P.eps[0, 0], 32).numpy()
    rec, rec_y, rec_h = run_pass(decm, np.concatenate([P.code_src, z], 2), P.y_in_dec)
    lat2d = run_pass(encm, P.x[0], P.y_in_enc[:1], clamp=True, lat_dim=32)[0]
    # carry: two 40-frame windows, second one fed the first one's (y_last, h)  (train...:1301-1311)
    a, ay, ah = run_pass(encm, P.x[:, :40], P.y_in_enc, clamp=True, lat_dim=32)
    b, by, bh = run_pass(encm, P.x[:, 40:], ay, h_in=ah, clamp=True, lat_dim=32)
    save("full_pass", sha_enc=synth.sha256_state(P.enc), sha_dec=synth.sha256_state(P.dec),
         lat=lat, lat_y=lat_y, lat_h=lat_h, rec=rec, rec_y=rec_y, rec_h=rec_h, lat2d=lat2d,
         carry_a=a, carry_b=b, carry_by=by, carry_bh=bh)
    save("full_chain", **chain(encm, decm, P))


def case_stress():
    """hu2048/ld64 single encoder+decoder pass, B=1, T=16."""
    P = synth.CycleVAEProblem(B=1, T=16, in_dim=54, out_dim=50, lat_dim=64, hidden=2048, n_cyc=1, bias_scale=0.05, tag="stress")
    encm, decm = build(P.enc, 54, 128, 2048, True), build(P.dec, 66, 50, 2048, False)
    lat, lat_y, lat_h = run_pass(encm, P.x, P.y_in_enc, clamp=True, lat_dim=64)
    z = ref_sample(torch.from_numpy(lat), P.eps[0, 0], 64).numpy()
    rec = run_pass(decm, np.concatenate([P.code_src, z], 2), P.y_in_dec)[0]
    save("stress_pass", lat=lat, lat_y=lat_y, lat_h=lat_h, rec=rec)


def case_stage6():
    """Stage-6 network path (decode_gru-cyclevae_gauss.py:302-319) on one utterance, T=203, 5 draws."""
    T, nd = 203, 5
    P = synth.CycleVAEProblem(B=1, T=T, bias_scale=0.05, tag="st6")
    eps = synth.normal("st6/eps_dec", (nd, T, 32))
    encm, decm = build(P.enc, 54, 64, 1024, True), build(P.dec, 34, 50, 1024, False)
    tt = torch.from_numpy
    with torch.no_grad():
        lat = encm(tt(P.x[0]), tt(P.y_in_enc), clamp_vae=True, lat_dim=32)[0]
        FEED.q.append(eps)
        lf = torch.mean(ref.sampling_vae_batch(lat.unsqueeze(0).repeat(nd, 1, 1), lat_dim=32), 0)
        code = torch.zeros(T, 2)
        code[:, 1] = 1
        cv = decm(torch.cat((code, lf), 1), tt(P.y_in_dec))[0]
    save("stage6", lat=lat.numpy(), lat_feat=lf.numpy(), cvmcep=np.array(cv.numpy(), dtype=np.float64))


def train_pass(m, x, y_in, h_in, cot, clamp, lat_dim, p_drop):
    """One train-mode pass of a reference module (do=True): returns outputs, the dropout masks it drew (captured by
    forward hooks on conv_drop / gru_drop, scaled by 1/(1-p)), and gradients of sum(out*cot) w.r.t. x and every
    parameter."""
    masks = {"conv": [], "gru": []}
    h1 = m.conv_drop.register_forward_hook(lambda mod, i, o: masks["conv"].append((o.detach() != 0).float() / (1 - p_drop)))
    h2 = m.gru_drop.register_forward_hook(lambda mod, i, o: masks["gru"].append((o.detach() != 0).float() / (1 - p_drop)))
    m.train()
    for q in m.parameters():
        q.requires_grad_(True)
        q.grad = None
    xt = torch.from_numpy(x).requires_grad_(True)
    o, y, h = m(xt, torch.from_numpy(y_in), h_in=None if h_in is None else torch.from_numpy(h_in), do=True,
                clamp_vae=clamp, lat_dim=lat_dim)
    (o * torch.from_numpy(cot)).sum().backward()
    h1.remove()
    h2.remove()
    # a mask entry is ambiguous only where the masked value is exactly 0; conv/GRU outputs never are
    out = {"out": o.detach().numpy(), "y_last": y.detach().numpy(), "h_last": h.detach().numpy(), "dx": xt.grad.numpy(),
           "cmask": masks["conv"][0].numpy(), "gmask": torch.cat(masks["gru"], 1).transpose(0, 1).contiguous().numpy()}
    for k, q in m.named_parameters():
        out["g_" + k] = q.grad.numpy() if q.grad is not None else np.zeros(tuple(q.shape), np.float32)
    return out


def case_train():
    """Train-mode (dropout 0.5) forward + backward of single passes at hidden 32 and 64, with state carry."""
    for tag, hid, B, T in (("train_h32", 32, 3, 10), ("train_h64", 64, 18, 7)):
        P = synth.CycleVAEProblem(B=B, T=T, in_dim=6, out_dim=4, lat_dim=4, hidden=hid, n_cyc=1, bias_scale=0.1, tag=tag)
        arrs = {}
        for name, sd, i, o, enc in (("enc", P.enc, 6, 8, True), ("dec", P.dec, 6, 4, False)):
            m = ref.GRU_RNN(in_dim=i, out_dim=o, hidden_units=hid, kernel_size=3, dilation_size=2, do_prob=0.5,
                            scale_out_flag=not enc, scale_in_flag=enc)
            m.load_state_dict(to_t(sd))
            torch.manual_seed(7 + hid)
            x = P.x if enc else np.concatenate([P.code_src, synth.normal(tag + "/z", (B, T, 4))], 2)
            y_in = P.y_in_enc if enc else P.y_in_dec
            cot = synth.normal(tag + "/cot_" + name, (B, T, o))
            r = train_pass(m, x, y_in, None, cot, enc, 4, 0.5)
            for k, v in r.items():
                arrs[name + "_" + k] = v
            if enc:   # second window fed the first one's detached (y_last, h)  (train...:1301)
                x2 = synth.features(tag + "/x2", B, T, P.mu, P.sigma)
                r2 = train_pass(m, x2, r["y_last"], r["h_last"], cot, True, 4, 0.5)
                for k, v in r2.items():
                    arrs["enc2_" + k] = v
        save(tag, **arrs)


def _load_train_generator():
    src = open("/root/reference/src/bin/train_gru_cyclevae_gauss_batch.py").read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "train_generator"][0]
    ns = {"np": np, "torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "<reference train_generator>", "exec"), ns)
    return ns["train_generator"]


def int_cases():
    """(flens, spcidx lists) of the INT fixtures; case 0 is SURVEY App. B.1's worked example."""
    r = lambda a, b: list(range(a, b + 1))
    return [
        ([205, 170, 90], [r(10, 69) + r(85, 159) + r(161, 199), r(5, 59) + r(100, 164), r(82, 87)]),
        ([80, 79, 81, 1], [r(0, 79), r(3, 70), r(79, 80), [0]]),
        ([637, 400, 12], [r(30, 600), r(0, 79) + r(81, 159) + r(320, 399), r(2, 9)]),
    ]


def case_int():
    gen_fn = _load_train_generator()
    pad_len = 2200
    arrs = {}
    for ci, (flens, spcs) in enumerate(int_cases()):
        U = len(flens)
        spc = np.zeros((U, pad_len), np.int64)
        for j, s in enumerate(spcs):
            spc[j, :len(s)] = s
        fl_spc = [len(s) for s in spcs]
        batch = {"flen_src": torch.tensor(flens), "flen_spc_src": torch.tensor(fl_spc),
                 "flen_src_trg": torch.tensor(flens), "flen_spc_src_trg": torch.tensor(fl_spc),
                 "h_src": torch.zeros(U, pad_len, 1), "src_code": torch.zeros(U, pad_len, 2),
                 "trg_code": torch.zeros(U, pad_len, 2), "cv_src": torch.zeros(U, pad_len, 1),
                 "spcidx_src": torch.from_numpy(spc), "h_src_trg": torch.zeros(U, pad_len, 1),
                 "spcidx_src_trg": torch.from_numpy(spc), "featfile_src": ["a"] * U, "featfile_src_trg": ["b"] * U}
        g = gen_fn([batch], torch.device("cpu"), batch_size=80)
        rows = []
        while True:
            y = next(g)
            if y[9] < 0:  # c_idx sentinel
                break
            s, e, s_idx, e_idx, sel, facc = y[5], y[6], y[7], y[8], y[19], y[20]
            assert y[1].shape[1] == e - s + 1
            rows.append((s, e, np.array(s_idx).copy(), np.array(e_idx).copy(), np.array(facc).copy(),
                         np.array([1 if j in sel else 0 for j in range(U)])))
        arrs["c%d_se" % ci] = np.array([[r[0], r[1]] for r in rows], np.int64)
        arrs["c%d_s_idx" % ci] = np.stack([r[2] for r in rows]).astype(np.int64)
        arrs["c%d_e_idx" % ci] = np.stack([r[3] for r in rows]).astype(np.int64)
        arrs["c%d_flen_acc" % ci] = np.stack([r[4] for r in rows]).astype(np.int64)
        arrs["c%d_select" % ci] = np.stack([r[5] for r in rows]).astype(np.int64)
    save("int_windows", **arrs)


def case_twfse():
    """Every branch of TWFSEloss.forward (twf / rmse / L2 / GV) on small deterministic inputs."""
    x = synth.normal("twfse/x", (12, 5)).astype(np.float32)
    y = (synth.normal("twfse/y", (9, 5)) * 0.5 + 0.1).astype(np.float32)
    twf = np.array([0, 1, 1, 3, 4, 6, 8, 10, 11], np.int64)
    crit = ref.TWFSEloss()
    arrs = {"twf": twf}
    for use_twf in (0, 1):
        xx = torch.from_numpy(x)
        yy = torch.from_numpy(y if use_twf else x[:9] * 0.8 + y * 0.2)
        if not use_twf:
            xx = xx[:9]
        for rmse in (0, 1):
            for l2 in (0, 1):
                for gv in (0, 1):
                    out = crit(xx, yy, twf=torch.from_numpy(twf) if use_twf else None, GV=bool(gv), rmse=bool(rmse), L2=bool(l2))
                    arrs["out_twf%d_rmse%d_l2%d_gv%d" % (use_twf, rmse, l2, gv)] = np.array([v.item() for v in out], np.float64)
    save("twfse_branches", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "full", "stress", "stage6", "int", "train", "twfse"]
    for w in which:
        {"tiny": case_tiny, "full": case_full, "stress": case_stress, "stage6": case_stage6, "int": case_int,
         "train": case_train, "twfse": case_twfse}[w]()
