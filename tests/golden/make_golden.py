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


def case_stress_chain():
    """BASELINE configs[4] dims: the n_cyc = 4 reconversion chain at hu2048 / ld64 (8 encoder + 12 decoder passes), B=2, T=16,
    through the reference modules (eval form of train...:1326-1338)."""
    P = synth.CycleVAEProblem(B=2, T=16, in_dim=54, out_dim=50, lat_dim=64, hidden=2048, n_cyc=4, bias_scale=0.05, tag="stress4")
    encm, decm = build(P.enc, 54, 128, 2048, True), build(P.dec, 66, 50, 2048, False)
    out = chain(encm, decm, P)
    save("stress_chain", **out)


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


REF_TRAIN = "/root/reference/src/bin/train_gru_cyclevae_gauss_batch.py"


def _ref_stmt_at(tree, lineno, kind):
    """The statement of type `kind` that starts at `lineno` of the reference training script."""
    for n in ast.walk(tree):
        if isinstance(n, kind) and getattr(n, "lineno", -1) == lineno:
            return n
    raise KeyError((lineno, kind))


def _exec_stmts(stmts, ns, what):
    exec(compile(ast.Module(body=list(stmts), type_ignores=[]), "<reference %s>" % what, "exec"), ns)


def case_step(NC=2, tag="step", out="stage4_step"):
    """Two consecutive stage-4 steps EXECUTED BY THE REFERENCE'S OWN STATEMENTS (ast-extracted from
    train_gru_cyclevae_gauss_batch.py, which cannot be imported here): the window plan comes from its train_generator (:45-149),
    the forward from the `if src_idx_s > 0 and prev_featfile_src == featfile_src ...` statement at :1298 (fresh-window branch
    :1326-1338 for the first window, carry branch :1299-1311 for the second), the loss from the two loops under
    `if len(select_utt_idx) > 0:` (:1356-1410, flen_acc / select_utt_idx masking and the :1393 concat included) and the update
    from its `optimizer.zero_grad(); batch_loss.backward(); optimizer.step()` (:1418-1420) with the optimizer built like :373-377.
    Tiny dims (hidden 32), dropout 0.5 in train mode; the masks the reference drew are captured by forward hooks and eps is fed.
    Three utterances of 20 / 15 / 9 frames, 12-frame windows: ragged flen_acc, the short utterance drops out of the second window."""
    torch.manual_seed(20190721)      # the dropout masks of the two steps come from torch's global generator: pinned, so that the
                                     # fixture regenerates bit for bit
    src = open(REF_TRAIN).read()
    tree = ast.parse(src)
    fwd_if = _ref_stmt_at(tree, 1298, ast.If)
    loss_if = _ref_stmt_at(tree, 1356, ast.If)
    assert ast.get_source_segment(src, fwd_if.test).startswith("src_idx_s > 0 and prev_featfile_src == featfile_src")
    assert ast.get_source_segment(src, loss_if.test) == "len(select_utt_idx) > 0"
    loss_stmts = loss_if.body[:5]                       # init loop, loss loop, zero_grad, backward, step
    assert [ast.get_source_segment(src, n) for n in loss_stmts[2:]] == ["optimizer.zero_grad()", "batch_loss.backward()", "optimizer.step()"]

    U, W, L, HID, STD = 3, 12, 4, 32, 4
    flens = [20, 15, 9]
    P = synth.CycleVAEProblem(B=U, T=20, in_dim=10, out_dim=6, lat_dim=L, hidden=HID, n_cyc=NC, bias_scale=0.1, tag=tag)
    pad_len = 20
    x = P.x.copy()
    cvx = P.cvx.copy()
    for j, n in enumerate(flens):                      # the dataset zero-pads raw features behind an utterance's end (dataset.py:23-31)
        x[j, n:] = 0.0
        cvx[j, n:] = 0.0
    spc = [list(range(2, 18)), list(range(0, 14)), list(range(1, 8))]   # speech-frame indices (only steer select_utt_idx)
    spcidx = np.zeros((U, pad_len), np.int64)
    for j, q in enumerate(spc):
        spcidx[j, :len(q)] = q
    fl_spc = [len(q) for q in spc]
    tt = torch.from_numpy
    batch = {"flen_src": torch.tensor(flens), "flen_spc_src": torch.tensor(fl_spc), "flen_src_trg": torch.tensor(flens),
             "flen_spc_src_trg": torch.tensor(fl_spc), "h_src": tt(x), "src_code": tt(P.code_src.copy()),
             "trg_code": tt(P.code_trg.copy()), "cv_src": tt(cvx), "spcidx_src": tt(spcidx), "h_src_trg": tt(x),
             "spcidx_src_trg": tt(spcidx), "featfile_src": ["SRC/u%d.h5" % j for j in range(U)],
             "featfile_src_trg": ["TRG/u%d.h5" % j for j in range(U)]}
    gen = _load_train_generator()([batch], torch.device("cpu"), batch_size=W)

    def mk(sd, i, o, enc):
        m = ref.GRU_RNN(in_dim=i, out_dim=o, hidden_units=HID, kernel_size=3, dilation_size=2, do_prob=0.5,
                        scale_out_flag=not enc, scale_in_flag=enc)
        m.load_state_dict(to_t(sd))
        m.train()
        return m

    enc, dec = mk(P.enc, 10, 8, True), mk(P.dec, 6, 6, False)
    for q in enc.scale_in.parameters():                # train...:369-372
        q.requires_grad = False
    for q in dec.scale_out.parameters():
        q.requires_grad = False
    module_list = list(enc.conv.parameters()) + list(enc.gru.parameters()) + list(enc.out_1.parameters())
    module_list += list(dec.conv.parameters()) + list(dec.gru.parameters()) + list(dec.out_1.parameters())
    masks = {"enc": [], "dec": []}
    cur = {}

    def hook(kind, which):
        def f(mod, i, o):
            m_ = (o.detach() != 0).float() / 0.5
            if which == "conv":
                cur[kind] = {"c": m_.numpy(), "g": []}
                masks[kind].append(cur[kind])
            else:
                cur[kind]["g"].append(m_)
        return f

    for kind, m in (("enc", enc), ("dec", dec)):
        m.conv_drop.register_forward_hook(hook(kind, "conv"))
        m.gru_drop.register_forward_hook(hook(kind, "gru"))

    class A(object):
        pass

    args = A()
    args.n_cyc, args.lat_dim, args.batch_size_utt, args.spk_src, args.batch_size = NC, L, U, "SRC", W
    ns = {"torch": torch, "np": np, "os": os, "Variable": torch.autograd.Variable, "args": args, "model_encoder": enc,
          "model_decoder": dec, "sampling_vae_batch": ref.sampling_vae_batch, "loss_vae": ref.loss_vae,
          "criterion_mcd": ref.TWFSEloss(), "optimizer": torch.optim.Adam(module_list, lr=1e-4), "stdim": STD, "half_cyc": False,
          "y_in_pp": tt(P.y_in_enc.copy()), "y_in_src": tt(P.y_in_dec.copy()), "y_in_trg": tt(P.y_in_dec.copy()),
          "iter_count": 0, "prev_featfile_src": None}
    for name in ("batch_lat_src", "y_in_pp_src", "h_in_pp_src", "batch_trj_src_src", "y_in_src_src", "h_in_src_src",
                 "batch_trj_src_trg", "y_in_src_trg", "h_in_src_trg", "batch_lat_src_trg", "y_in_pp_src_trg", "h_in_pp_src_trg",
                 "batch_trj_src_trg_src", "y_in_src_trg_src", "h_in_src_trg_src", "batch_mcdpow_src_src", "batch_mcd_src_src",
                 "batch_mcdpow_src_trg_src", "batch_mcd_src_trg_src", "batch_loss_mcd_src_src", "batch_loss_mcd_src_trg_src",
                 "batch_loss_mcd_src_trg", "batch_loss_lat_src", "batch_loss_lat_src_cv"):
        ns[name] = [None] * NC
    for name in ("loss_mcd_src_src", "loss_mcd_src_trg_src", "loss_mcd_src_trg", "loss_lat_src_cv", "loss_lat_src",
                 "loss_mcd_trg_trg", "loss_mcd_trg_src_trg", "loss_mcd_trg_src", "loss_lat_trg_cv", "loss_lat_trg"):
        ns[name] = [[] for _ in range(NC)]
    arrs = {"flens": np.array(flens, np.int64), "spcidx": spcidx, "flens_spc": np.array(fl_spc, np.int64)}
    names = ("batch_src batch_src_src_code batch_src_trg_code batch_src_trg batch_cv_src src_idx_s src_idx_e spcidx_src_s_idx "
             "spcidx_src_e_idx c_idx utt_idx spcidx_src spcidx_src_trg featfile_src featfile_src_trg flens_src flens_src_trg "
             "flens_spc_src flens_spc_src_trg select_utt_idx flen_acc n_batch_utt").split()
    for w in range(2):
        y = next(gen)
        assert y[9] >= 0
        ns.update(dict(zip(names, y)))
        s0, e0 = ns["src_idx_s"], ns["src_idx_e"]
        for i in range(NC):
            for q in range(3):
                FEED.q.append(P.eps[i, q][:, s0:e0 + 1].copy())
        n0 = {k: len(v) for k, v in masks.items()}
        _exec_stmts([fwd_if], ns, "forward :1298-1354")
        ns["prev_featfile_src"] = ns["featfile_src"]    # :1354
        assert not FEED.q
        _exec_stmts(loss_stmts, ns, "loss and update :1356-1420")
        ns["iter_count"] += 1
        arrs["w%d_se" % w] = np.array([s0, e0], np.int64)
        arrs["w%d_flen_acc" % w] = np.array(ns["flen_acc"], np.int64)
        arrs["w%d_select" % w] = np.array(ns["select_utt_idx"], np.int64)
        arrs["w%d_loss" % w] = np.array(ns["batch_loss"].item(), np.float64)
        for k in ("batch_lat_src", "batch_trj_src_src", "batch_trj_src_trg", "batch_lat_src_trg", "batch_trj_src_trg_src"):
            arrs["w%d_%s" % (w, k)] = np.stack([v.detach().numpy() for v in ns[k]])
        for kind in ("enc", "dec"):
            for ci, mm in enumerate(masks[kind][n0[kind]:]):
                arrs["w%d_%s%d_cmask" % (w, kind, ci)] = (mm["c"] != 0)
                arrs["w%d_%s%d_gmask" % (w, kind, ci)] = (torch.cat(mm["g"], 1).transpose(0, 1).contiguous().numpy() != 0)
        for kind, m in (("enc", enc), ("dec", dec)):
            for k, q in m.named_parameters():
                if not q.requires_grad:
                    continue
                g = q.grad.numpy()
                arrs["w%d_%s_g_%s" % (w, kind, k)] = g.copy()       # (both windows: the carry branch's gradients are held to the full-tensor bound too)
                arrs["w%d_%s_gnorm_%s" % (w, kind, k)] = np.array(np.sqrt((g.astype(np.float64) ** 2).sum()))
                v = q.detach().numpy().astype(np.float64)
                arrs["w%d_%s_after_%s" % (w, kind, k)] = np.array([v.sum(), (v * v).sum(), v.ravel()[0], v.ravel()[-1]])
    arrs["n_cyc"] = np.array([NC], np.int64)
    save(out, **arrs)


def case_step4():
    """The same two reference-executed steps with n_cyc = 4 (the cycle count of BASELINE configs[4]): 8 encoder + 12 decoder passes per
    window, the cycle loop of :1326-1338 / :1299-1311 four times, loss and update as in case_step."""
    case_step(NC=4, tag="step4", out="stage4_step_cyc4")


def case_gv():
    """GV post-filter: the reference's own statements decode_gru-cyclevae_gauss.py:419-420,422 (ast-extracted; the script
    imports h5py / pysptk / pyworld and cannot be imported) on a synthetic converted mcep sequence."""
    path = "/root/reference/src/bin/decode_gru-cyclevae_gauss.py"
    src = open(path).read()
    tree = ast.parse(src)
    stmts = [_ref_stmt_at(tree, 419, ast.Assign), _ref_stmt_at(tree, 420, ast.Assign), _ref_stmt_at(tree, 422, ast.Expr)]
    assert ast.get_source_segment(src, stmts[0]).startswith("datamean = np.mean(cvmcep[:,1:]")
    assert ast.get_source_segment(src, stmts[2]).startswith("cvgvlist.append(np.var(cvmcep_gv[:,1:]")
    T, D = 211, 50
    c = (synth.normal("gvpin/c", (T, D)) * np.linspace(2.0, 0.1, D)).astype(np.float32)
    gv_t = (0.05 + synth.uniform01("gvpin/gv", (D - 1,))).astype(np.float64)
    cg = (0.02 + 0.5 * synth.uniform01("gvpin/cg", (D - 1,))).astype(np.float64)
    ns = {"np": np, "cvmcep": np.array(c, dtype=np.float64), "gv_mean_trg": gv_t, "cvgv_mean": cg, "cvgvlist": []}   # :319 float64
    _exec_stmts(stmts, ns, "GV post-filter :419-422")
    save("gv_postfilter", cvmcep_gv=ns["cvmcep_gv"], cvgv=ns["cvgvlist"][0])


def loader_store():
    """Synthetic on-disk content of the loader fixture: {(file, dataset): array}; two speakers, three utterance pairs of
    23 / 17 / 31 (source) and 20 / 19 / 31 (paired speaker) frames, 5 feature dims, 3 converted-F0 dims."""
    store = {}
    for u, (n_src, n_trg) in enumerate(((23, 20), (17, 19), (31, 31))):
        for spk, n in (("spkA", n_src), ("spkB", n_trg)):
            f = "/data/%s/utt%d.h5" % (spk, u)
            store[(f, "/feat_org_lf0")] = synth.normal("loader/%s/%d/feat" % (spk, u), (n, 5)).astype(np.float32)
            store[(f, "/cvuvlogf0fil_ap")] = synth.normal("loader/%s/%d/cv" % (spk, u), (n, 3)).astype(np.float32)
            keep = np.nonzero(synth.uniform01("loader/%s/%d/spc" % (spk, u), (n,)) > 0.3)[0]
            store[(f, "/spcidx_range")] = keep[None, :].astype(np.int64)
    return store


def loader_lists():
    src = ["/data/spkA/utt0.h5", "/data/spkB/utt1.h5", "/data/spkA/utt2.h5"]
    trg = ["/data/spkB/utt0.h5", "/data/spkA/utt1.h5", "/data/spkB/utt2.h5"]
    return src, trg


def case_recipe():
    """The reference's own statements around the hot path, executed through ast: scale_in / scale_out from the joint statistics
    (train...:344-347), the initial feedback vectors (:357-359), save_checkpoint (:152-167) on a torch.optim.Adam that has taken one
    step, and sklearn's StandardScaler over three ragged utterances like calc_stats_vc_joint.py:83-127 (partial_fit per file)."""
    import types
    from sklearn.preprocessing import StandardScaler
    src = open(REF_TRAIN).read()
    tree = ast.parse(src)
    feats = [(synth.normal("recipe/u%d" % i, (n, 10)) * (0.5 + 0.1 * np.arange(10)) + np.arange(10) * 0.3).astype(np.float32)
             for i, n in enumerate((17, 23, 9))]
    sc = StandardScaler()
    for f in feats:
        sc.partial_fit(f[:, :])
    mean, scale = sc.mean_, sc.scale_
    stdim, L, B = 4, 4, 3
    enc = ref.GRU_RNN(in_dim=10, out_dim=2 * L, hidden_units=32, kernel_size=3, dilation_size=2, do_prob=0.5, scale_out_flag=False)
    dec = ref.GRU_RNN(in_dim=L + 2, out_dim=6, hidden_units=32, kernel_size=3, dilation_size=2, do_prob=0.5, scale_in_flag=False)
    torch.manual_seed(11)
    enc.apply(ref.initialize)
    dec.apply(ref.initialize)
    ns = {"torch": torch, "np": np, "model_encoder": enc, "model_decoder": dec, "mean_jnt": torch.FloatTensor(mean),
          "std_jnt": torch.FloatTensor(scale), "mean_jnt_trg": torch.FloatTensor(mean[stdim:]), "std_jnt_trg": torch.FloatTensor(scale[stdim:]),
          "args": types.SimpleNamespace(batch_size_utt=B, lat_dim=L, batch_size_utt_eval=B)}
    stmts = [_ref_stmt_at(tree, ln, ast.Assign) for ln in (344, 345, 346, 347, 357, 358, 359)]
    assert ast.get_source_segment(src, stmts[0]).startswith("model_encoder.scale_in.weight = torch.nn.Parameter(torch.diag(1.0/std_jnt.data)")
    _exec_stmts(stmts, ns, "scalers")
    arrs = {"feat%d" % i: f for i, f in enumerate(feats)}
    arrs.update(mean=mean, scale=scale, scale_in_w=enc.scale_in.weight.detach().numpy(), scale_in_b=enc.scale_in.bias.detach().numpy(),
                scale_out_w=dec.scale_out.weight.detach().numpy(), scale_out_b=dec.scale_out.bias.detach().numpy(),
                y_in_pp=ns["y_in_pp"].numpy(), y_in_src=ns["y_in_src"].numpy())
    # checkpoint: the reference's save_checkpoint on CPU modules (its .cpu() / .cuda() round trip is the identity here)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "save_checkpoint"][0]
    import logging
    import tempfile
    ns2 = {"torch": torch, "os": os, "logging": logging}
    _exec_stmts([fn], ns2, "save_checkpoint")
    params = [p_ for m in (enc, dec) for n_, p_ in m.named_parameters() if not n_.startswith("scale")]
    opt = torch.optim.Adam(params, lr=1e-4)
    x = torch.from_numpy(np.stack([f[:9] for f in feats]))
    y, _, _ = enc(x, ns["y_in_pp"], do=True, clamp_vae=True, lat_dim=L)
    y.sum().backward()
    opt.step()
    with tempfile.TemporaryDirectory() as td:
        ns2["save_checkpoint"](td, enc, dec, opt, np.random.get_state(), torch.get_rng_state(), 7)
        names = os.listdir(td)
        ck = torch.load(os.path.join(td, names[0]), weights_only=False)
    arrs["ck_file"] = np.array(names)
    arrs["ck_keys"] = np.array(sorted(ck.keys()))
    arrs["ck_enc_keys"] = np.array(list(ck["model_encoder"].keys()))
    arrs["ck_opt_group_keys"] = np.array(sorted(ck["optimizer"]["param_groups"][0].keys()))
    arrs["ck_opt_state_keys"] = np.array(sorted(ck["optimizer"]["state"][0].keys()))
    arrs["ck_opt_nparams"] = np.array([len(ck["optimizer"]["state"]), len(ck["optimizer"]["param_groups"][0]["params"])], np.int64)
    arrs["ck_iterations"] = np.array([ck["iterations"]], np.int64)
    save("recipe", **arrs)


def case_loader():
    """The reference's own `padding` + `FeatureDatasetSingleVAE` (src/utils/dataset.py:23-98, ast-extracted: the module imports
    soundfile and h5py-backed utils) over a dict-backed `read_hdf5`, torch's DataLoader default collate, and the reference's
    `train_generator` (train...:45-149) with 12-frame windows: what cyclevae-vc_amd/loader.py must reproduce."""
    from torch.utils.data import DataLoader, Dataset
    path = "/root/reference/src/utils/dataset.py"
    tree = ast.parse(open(path).read())
    body = [n for n in tree.body if (isinstance(n, ast.FunctionDef) and n.name == "padding") or
            (isinstance(n, ast.ClassDef) and n.name == "FeatureDatasetSingleVAE")]
    assert len(body) == 2
    store = loader_store()
    ns = {"np": np, "torch": torch, "os": os, "Dataset": Dataset, "read_hdf5": lambda f, k: store[(f, k)]}
    exec(compile(ast.Module(body=body, type_ignores=[]), "<reference dataset.py>", "exec"), ns)
    src, trg = loader_lists()
    pad = lambda x: ns["padding"](x, 40, value=0.0)
    ds = ns["FeatureDatasetSingleVAE"](src, trg, pad, "spkA")
    batch = next(iter(DataLoader(ds, batch_size=3, shuffle=False)))
    arrs = {"item_" + k: v.numpy() for k, v in batch.items() if torch.is_tensor(v)}
    g = _load_train_generator()([batch], torch.device("cpu"), batch_size=12)
    w = 0
    while True:
        y = next(g)
        if y[9] < 0:
            break
        for i, name in ((0, "hs_src"), (1, "src_codes"), (2, "trg_codes"), (3, "hs_src_trg"), (4, "cvs_src"), (11, "spcidcs_src"),
                        (12, "spcidcs_src_trg")):
            arrs["w%d_%s" % (w, name)] = y[i].numpy()
        arrs["w%d_ints" % w] = np.array([y[5], y[6], y[9], y[10], y[21]], np.int64)
        arrs["w%d_s_idx" % w], arrs["w%d_e_idx" % w] = np.array(y[7], np.int64), np.array(y[8], np.int64)
        arrs["w%d_select" % w], arrs["w%d_flen_acc" % w] = np.array(y[19], np.int64), np.array(y[20], np.int64)
        arrs["w%d_lens" % w] = np.stack([np.asarray(y[i], np.int64) for i in (15, 16, 17, 18)])
        w += 1
    arrs["n_windows"] = np.array([w], np.int64)
    y = next(_load_train_generator()([batch], torch.device("cpu"), batch_size=0))
    arrs["utt_src_codes"] = y[1].numpy()
    arrs["utt_ints"] = np.array([y[5], y[6], y[15]], np.int64)
    save("loader", **arrs)


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


def case_laplace():
    """SURVEY 8(f) row 4, first variant: the Laplace posterior of the sibling recipes (dead code in egs/one-to-one, live in the module):
    sampling_vae_laplace (gru_vae.py:101-114), loss_vae_laplace (:130-145) and the clamp_vae_laplace branch of GRU_RNN.forward
    (:415-417, log-scale floor -7.2543...).  The encoder's out_1 bias is shifted by -8 so that the floor is active on about half of the
    log-scales; the uniform draw is torch's own (`torch.empty(..).uniform_(-0.4999, 0.5)` under manual_seed), recorded next to the
    reference's output so that the parity tests inject it."""
    P = synth.CycleVAEProblem(B=2, T=12, in_dim=6, out_dim=4, lat_dim=4, hidden=32, n_cyc=2, bias_scale=0.1, tag="laplace")
    sd = dict(P.enc)
    sd["out_1.bias"] = sd["out_1.bias"] - np.float32(8.0) * (np.arange(8) >= 4).astype(np.float32)
    encm = build(sd, 6, 8, 32, True)
    with torch.no_grad():
        lat = encm(torch.from_numpy(P.x), torch.from_numpy(P.y_in_enc), clamp_vae_laplace=True, lat_dim=4)[0].numpy()
        lat2d = encm(torch.from_numpy(P.x[1]), torch.from_numpy(P.y_in_enc[1:]), clamp_vae_laplace=True, lat_dim=4)[0].numpy()
        raw = encm(torch.from_numpy(P.x), torch.from_numpy(P.y_in_enc), lat_dim=4)[0].numpy()
    floor = np.float32(-7.2543288692621097067625904247823)
    assert np.any(lat[:, :, 4:] == floor) and np.any(lat[:, :, 4:] > floor) and np.any(raw[:, :, 4:] < floor)
    param = torch.from_numpy(lat[0])                       # [T, 2L]: the 2-D form both functions take
    torch.manual_seed(1234)
    eps = torch.empty(param.shape[0], 4).uniform_(-0.4999, 0.5).numpy()
    torch.manual_seed(1234)
    z = ref.sampling_vae_laplace(param, lat_dim=4).numpy()
    zt = torch.from_numpy(eps)
    assert np.array_equal(z, (param[:, :4] - torch.exp(param[:, 4:]) * zt.sign() * torch.log1p(-2 * zt.abs())).numpy())
    kl = ref.loss_vae_laplace(param, lat_dim=4).item()
    # gradients of both (the reference differentiates through them in the sibling recipes' training)
    pg = param.clone().requires_grad_(True)
    cot = torch.from_numpy(synth.normal("laplace/cot", (12, 4)).astype(np.float32))
    torch.manual_seed(1234)
    (ref.sampling_vae_laplace(pg, lat_dim=4, training=True) * cot).sum().backward()
    dz_dparam = pg.grad.numpy().copy()
    pg.grad = None
    ref.loss_vae_laplace(pg, lat_dim=4).backward()
    save("laplace", sha_enc=synth.sha256_state(sd), lat=lat, lat2d=lat2d, raw=raw, eps=eps, z=z, kl=np.float32(kl),
         d_sample=dz_dparam, d_kl=pg.grad.numpy())


def case_vq():
    """the VQ helpers of the sibling recipes (gru_vae.py:148-195), run as the reference wrote them."""
    enc = torch.from_numpy(synth.normal("vq/enc", (14, 5)).astype(np.float32))
    encb = torch.from_numpy(synth.normal("vq/encb", (2, 7, 5)).astype(np.float32))
    ctr = torch.from_numpy((1.5 * synth.normal("vq/ctr", (6, 5))).astype(np.float32))
    ids, idsb = ref.nn_search(enc, ctr).numpy(), ref.nn_search_batch(encb, ctr).numpy()
    wc, wd = ref.weighted_ctr(enc, ctr)
    save("vq", ids=ids, idsb=idsb, wc=wc.numpy(), wd=np.float32(wd.item()))


def m2m_store():
    """Synthetic on-disk content of the many-to-many fixture: three source and two target speakers, two utterances each (different
    lengths per speaker), 5 feature dims, one converted-F0 stream (3 dims) per speaker of the OTHER group."""
    src_spk, trg_spk = ["sA", "sB", "sC"], ["tA", "tB"]
    store = {}
    for si, spk in enumerate(src_spk + trg_spk):
        for u in range(2):
            n = 9 + 3 * si + 5 * u
            f = "/data/%s/utt%d.h5" % (spk, u)
            store[(f, "/feat_org_lf0")] = synth.normal("m2m/%s/%d/feat" % (spk, u), (n, 5)).astype(np.float32)
            for other in (trg_spk if spk in src_spk else src_spk):
                store[(f, "/cvuvlogf0fil_ap_" + other)] = synth.normal("m2m/%s/%d/cv_%s" % (spk, u, other), (n, 3)).astype(np.float32)
            keep = np.nonzero(synth.uniform01("m2m/%s/%d/spc" % (spk, u), (n,)) > 0.3)[0]
            store[(f, "/spcidx_range")] = keep[None, :].astype(np.int64)
    return store, src_spk, trg_spk


def _flatten_item(prefix, item, arrs):
    for k, v in item.items():
        if torch.is_tensor(v):
            arrs["%s_%s" % (prefix, k)] = v.numpy()
        elif isinstance(v, list) and v and torch.is_tensor(v[0]):
            for i, t in enumerate(v):
                arrs["%s_%s_%d" % (prefix, k, i)] = t.numpy()
        elif isinstance(v, list):
            arrs["%s_%s" % (prefix, k)] = np.array(v)
        elif isinstance(v, str):
            arrs["%s_%s" % (prefix, k)] = np.array([v])
        else:
            arrs["%s_%s" % (prefix, k)] = np.array([v], np.int64)


def case_m2m():
    """the many-to-many datasets of src/utils/dataset.py:101-492, ast-extracted and run over a dict-backed read_hdf5
    (np.random seeded: the random conversion pairs are the reference's own draws)."""
    from torch.utils.data import Dataset
    path = "/root/reference/src/utils/dataset.py"
    tree = ast.parse(open(path).read())
    want = {"padding", "proc_multspk_data_random", "FeatureDatasetMultTrainVAE", "FeatureDatasetMultEvalVAE", "proc_multspk_data_random_cls",
            "FeatureDatasetMultTrainVAECls", "FeatureDatasetMultEvalVAECls"}
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert len(body) == len(want)
    store, src_spk, trg_spk = m2m_store()
    ns = {"np": np, "torch": torch, "os": os, "Dataset": Dataset, "read_hdf5": lambda f, k: store[(f, k)]}
    exec(compile(ast.Module(body=body, type_ignores=[]), "<reference dataset.py>", "exec"), ns)
    pad = lambda x: ns["padding"](x, 30, value=0.0)
    files = ["/data/%s/utt%d.h5" % (s, u) for s in src_spk + trg_spk for u in range(2)]
    arrs = {}
    for cls in ("FeatureDatasetMultTrainVAE", "FeatureDatasetMultTrainVAECls"):
        ds = ns[cls](files, pad, src_spk, trg_spk, 2)
        np.random.seed(1234)
        for i in range(len(ds)):
            _flatten_item("%s_%d" % (cls, i), ds[i], arrs)
    src_lists = [["/data/%s/utt%d.h5" % (s, u) for u in range(2)] for s in src_spk]
    trg_lists = [["/data/%s/utt%d.h5" % (s, u) for u in range(2)] for s in trg_spk]
    for cls in ("FeatureDatasetMultEvalVAE", "FeatureDatasetMultEvalVAECls"):
        ds = ns[cls](src_lists, trg_lists, pad, src_spk, trg_spk)
        arrs["%s_len" % cls] = np.array([len(ds)], np.int64)
        arrs["%s_pairs" % cls] = np.array([[ds.count_spk_pair_cv[s][t] for t in trg_spk] for s in src_spk], np.int64)
        arrs["%s_file_list_src_trg" % cls] = np.array(ds.file_list_src_trg)
        for i in range(len(ds)):
            _flatten_item("%s_%d" % (cls, i), ds[i], arrs)
    # one target speaker only: the pairing's other branch
    ds1 = ns["FeatureDatasetMultEvalVAE"](src_lists, trg_lists[:1], pad, src_spk, trg_spk[:1])
    arrs["one_trg_file_list_src_trg"] = np.array(ds1.file_list_src_trg)
    save("m2m", **arrs)


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "full", "stress", "stage6", "int", "train", "twfse", "step", "step4", "gv", "stress_chain", "loader", "recipe", "laplace", "vq", "m2m"]
    for w in which:
        {"tiny": case_tiny, "full": case_full, "stress": case_stress, "stage6": case_stage6, "int": case_int,
         "train": case_train, "twfse": case_twfse, "step": case_step, "gv": case_gv, "stress_chain": case_stress_chain, "loader": case_loader, "recipe": case_recipe, "step4": case_step4, "laplace": case_laplace, "vq": case_vq, "m2m": case_m2m}[w]()
