"""CPU ORACLE (test infrastructure, NOT product code) for the CycleVAE-VC hot path.

A numpy float32 restatement of the reference's encoder -> latent -> decoder forward.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(cyclevae-vc_amd/) never does and fails loudly without its HIP library.

Parity pinning: the reference ships no tests, golden vectors or fixtures for this path (SURVEY.md
section 4), and its arithmetic lives in PyTorch.  This oracle is therefore pinned against outputs of the
reference itself, imported in the build container from /root/reference/src/nets/gru_vae.py by
tests/golden/make_golden.py; the resulting vectors are committed under tests/golden/*.npz and
tests/test_oracle_golden.py checks this file against them (tolerances stated there).

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
import numpy as np

F32 = np.float32
LOG_VAR_FLOOR = F32(-13.815510557964274)  # ln(1e-6), src/nets/gru_vae.py:412
LOG_SCALE_FLOOR_LAPLACE = F32(-7.2543288692621097)  # clamp_vae_laplace, src/nets/gru_vae.py:417
MCD_K = 10.0 / 2.3025850929940456840179914546844  # src/nets/gru_vae.py:523


def _sigmoid(a):
    return (1.0 / (1.0 + np.exp(-a, dtype=F32))).astype(F32)


def conv1d(x, w, b, dilation=1, padding=0):
    """torch.nn.Conv1d on [B,C,T] (stride 1).  src/nets/gru_vae.py:49-51 build the two instances."""
    B, C, T = x.shape
    O, _, K = w.shape
    if padding:
        x = np.pad(x, ((0, 0), (0, 0), (padding, padding)))
        T = T + 2 * padding
    To = T - dilation * (K - 1)
    y = np.broadcast_to(b[None, :, None], (B, O, To)).astype(F32).copy()
    for k in range(K):
        y += np.einsum("oc,bct->bot", w[:, :, k], x[:, :, k * dilation:k * dilation + To], dtype=F32)
    return y


def front_end(sd, x):
    """scale_in + TwoSidedDilConv1d.  src/nets/gru_vae.py:331-346 (layout, scale_in) and :53-66, :357 (convs).

    x [B,T,Cin] -> x_conv [B,T,9*Cin] for kernel_size=3, dilation_size(layers)=2.
    """
    xt = np.transpose(x, (0, 2, 1)).astype(F32)
    if "scale_in.weight" in sd:
        xt = conv1d(xt, sd["scale_in.weight"], sd["scale_in.bias"])
    ks = sd["conv.conv.0.weight"].shape[2]
    layers = 2
    pad = (ks ** layers - 1) // 2  # src/nets/gru_vae.py:44-45
    c = conv1d(xt, sd["conv.conv.0.weight"], sd["conv.conv.0.bias"], dilation=1, padding=pad)
    c = conv1d(c, sd["conv.conv.1.weight"], sd["conv.conv.1.bias"], dilation=ks, padding=0)
    return np.transpose(c, (0, 2, 1)).copy()


def gru_cell(sd, u, h):
    """One torch.nn.GRU step (gate rows r,z,n).  Called per frame at src/nets/gru_vae.py:365,392."""
    H = h.shape[1]
    gi = u @ sd["gru.weight_ih_l0"].T + sd["gru.bias_ih_l0"]
    gh = h @ sd["gru.weight_hh_l0"].T + sd["gru.bias_hh_l0"]
    r = _sigmoid(gi[:, :H] + gh[:, :H])
    z = _sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
    n = np.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:], dtype=F32)
    return (n + z * (h - n)).astype(F32)


def gru_rnn_forward(sd, x, y_in, h_in=None, clamp_vae=False, lat_dim=16, conv_mask=None, gru_masks=None, clamp_vae_laplace=False):
    """GRU_RNN.forward, live branch only.  src/nets/gru_vae.py:322-455.

    x [B,T,Cin] or [T,Cin]; y_in [B,1,Cout]; h_in [1,B,H] or None.
    conv_mask [B,T,9Cin] / gru_masks [T,B,H]: dropout masks already scaled by 1/(1-p) (train mode,
    src/nets/gru_vae.py:355,380); None = eval.
    Returns (trj_out, y_last [B,1,Cout] raw, h [1,B,H]); for 2-D x trj_out is [T,Cout] (:406,:423).
    """
    two_d = x.ndim == 2
    if two_d:
        x = x[None]
    x = x.astype(F32)
    B, T, _ = x.shape
    xc = front_end(sd, x)
    if conv_mask is not None:
        xc = xc * conv_mask
    H = sd["gru.weight_hh_l0"].shape[1]
    Wo, bo = sd["out_1.weight"][:, :, 0], sd["out_1.bias"]
    h = np.zeros((B, H), F32) if h_in is None else h_in[0].astype(F32)
    y = y_in[:, 0].astype(F32)
    trj = np.empty((B, T, Wo.shape[0]), F32)
    for t in range(T):
        h = gru_cell(sd, np.concatenate([xc[:, t], y], 1), h)  # :365 / :392
        o = h if gru_masks is None else h * gru_masks[t]       # :369 / :380 (carried h stays un-dropped)
        y = (o @ Wo.T + bo).astype(F32)                          # :371 / :393
        trj[:, t] = y
    if "scale_out.weight" in sd:                                 # :402-406
        out = (trj @ sd["scale_out.weight"][:, :, 0].T + sd["scale_out.bias"]).astype(F32)
    else:
        out = trj.copy()
        if clamp_vae:                                            # :408-412
            out[:, :, lat_dim:] = np.maximum(out[:, :, lat_dim:], LOG_VAR_FLOOR)
        elif clamp_vae_laplace:                                  # :415-417 (log-scale floor of the Laplace variant)
            out[:, :, lat_dim:] = np.maximum(out[:, :, lat_dim:], LOG_SCALE_FLOOR_LAPLACE)
    if two_d:
        out = out[0]
    return out, y[:, None, :].copy(), h[None].copy()


def sampling_vae_batch(param, eps, lat_dim=None):
    """z = mu + exp(log_var/2) * eps.  src/nets/gru_vae.py:85-98 with eps supplied instead of torch.randn."""
    if lat_dim is None:
        lat_dim = param.shape[-1] // 2
    return (param[..., :lat_dim] + np.exp(param[..., lat_dim:] / F32(2), dtype=F32) * eps).astype(F32)


def loss_vae(param, lat_dim=None):
    """KL to N(0,I), mean over frames.  src/nets/gru_vae.py:117-123.  param [T,2L]."""
    if lat_dim is None:
        lat_dim = param.shape[1] // 2
    mu, s = param[:, :lat_dim], param[:, lat_dim:]
    return F32(np.mean(0.5 * np.sum(np.exp(s, dtype=F32) + mu * mu - s - F32(1.0), 1), dtype=F32))


def sampling_vae_laplace(param, eps, lat_dim=None):
    """z = mu - exp(log_scale) * sign(eps) * log1p(-2|eps|), eps ~ U(-0.4999, 0.5) supplied.  src/nets/gru_vae.py:101-112 (the
    log-scale branch)."""
    if lat_dim is None:
        lat_dim = param.shape[-1] // 2
    eps = eps.astype(F32)
    return (param[..., :lat_dim] - np.exp(param[..., lat_dim:], dtype=F32) * np.sign(eps) * np.log1p(F32(-2) * np.abs(eps), dtype=F32)).astype(F32)


def loss_vae_laplace(param, lat_dim=None):
    """KL of Laplace(mu, exp(s)) to Laplace(0, 1), mean over frames.  src/nets/gru_vae.py:130-139.  param [T,2L]."""
    if lat_dim is None:
        lat_dim = param.shape[1] // 2
    mu_abs, s = np.abs(param[:, :lat_dim]), param[:, lat_dim:]
    scale = np.exp(s, dtype=F32)
    return F32(np.mean(np.sum(-s + scale * np.exp(-mu_abs / scale, dtype=F32) + mu_abs - F32(1.0), 1), dtype=F32))


def mcd_frames(x, y, L2=True):
    """Per-frame mel-cepstral distortion.  src/nets/gru_vae.py:523 (L2) / :525 (L1)."""
    d = x.astype(F32) - y.astype(F32)
    if L2:
        return (F32(MCD_K) * np.sqrt(F32(2.0) * np.sum(d * d, 1, dtype=F32))).astype(F32)
    return (F32(MCD_K) * F32(1.4142135623730950488) * np.sum(np.abs(d), 1, dtype=F32)).astype(F32)


def twfse_loss(x, y, L2=True):
    """TWFSEloss.forward(twf=None, GV=False, rmse=False) -> (sum, mean, std).  src/nets/gru_vae.py:521-534."""
    m = mcd_frames(x, y, L2)
    return F32(m.sum(dtype=F32)), F32(m.mean(dtype=F32)), F32(m.std(ddof=1, dtype=F32)) if m.size > 1 else F32(np.nan)


def cycle_chain(enc, dec, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, n_cyc=2, lat_dim=32):
    """n_cyc reconversion loop, eval form.  src/bin/train_gru_cyclevae_gauss_batch.py:1326-1338 (fresh window).

    eps [n_cyc,3,B,T,L] in draw order (rec, cv, rec_cyc).  Returns dict of lists lat, rec, cv, latcv, reccyc.
    """
    stdim = cvx.shape[2]
    out = {k: [] for k in ("lat", "rec", "cv", "latcv", "reccyc")}
    for i in range(n_cyc):
        e_in = x if i == 0 else np.concatenate([x[:, :, :stdim], out["reccyc"][i - 1]], 2)
        lat = gru_rnn_forward(enc, e_in, y_in_enc, clamp_vae=True, lat_dim=lat_dim)[0]
        rec = gru_rnn_forward(dec, np.concatenate([code_src, sampling_vae_batch(lat, eps[i, 0], lat_dim)], 2), y_in_dec)[0]
        cv = gru_rnn_forward(dec, np.concatenate([code_trg, sampling_vae_batch(lat, eps[i, 1], lat_dim)], 2), y_in_dec)[0]
        latcv = gru_rnn_forward(enc, np.concatenate([cvx, cv], 2), y_in_enc, clamp_vae=True, lat_dim=lat_dim)[0]
        reccyc = gru_rnn_forward(dec, np.concatenate([code_src, sampling_vae_batch(latcv, eps[i, 2], lat_dim)], 2), y_in_dec)[0]
        for k, v in zip(("lat", "rec", "cv", "latcv", "reccyc"), (lat, rec, cv, latcv, reccyc)):
            out[k].append(v)
    return out


def stage6_convert(enc, dec, feat, y_in_enc, y_in_dec, eps, lat_dim=32, trg_is_second=True):
    """Stage-6 network path for one utterance.  src/bin/decode_gru-cyclevae_gauss.py:302-319.

    feat [T,Cin]; eps [n_smpl,T,L].  Returns (lat_src [T,2L], lat_feat [T,L], cvmcep [T,Cout] float64).
    """
    lat = gru_rnn_forward(enc, feat, y_in_enc, clamp_vae=True, lat_dim=lat_dim)[0]
    z = sampling_vae_batch(np.broadcast_to(lat[None], (eps.shape[0],) + lat.shape), eps, lat_dim)
    zbar = z.mean(0, dtype=F32)
    code = np.zeros((feat.shape[0], 2), F32)
    code[:, 1 if trg_is_second else 0] = 1
    cv = gru_rnn_forward(dec, np.concatenate([code, zbar], 1), y_in_dec)[0]
    return lat, zbar, cv.astype(np.float64)


def cycle_loss(outs, x, flen_acc, select_utt_idx, lat_dim=32, half_cyc=False):
    """Stage-4 batch loss of one fresh window starting at frame 0.  src/bin/train_gru_cyclevae_gauss_batch.py:1363-1410.

    Mirrors the reference exactly, including the `batch_loss_lat_src_cv` quirk at :1393.
    """
    stdim = x.shape[2] - outs["rec"][0].shape[2]
    total = F32(0)
    for i in range(len(outs["lat"])):
        m_ss, m_sts, l_s, l_cv = [], [], [], []
        for k, j in enumerate(select_utt_idx):
            n = int(flen_acc[j])
            tgt = x[j, :n, stdim:]
            a = twfse_loss(outs["rec"][i][j, :n], tgt, L2=False)[1]
            b = twfse_loss(outs["reccyc"][i][j, :n], tgt, L2=False)[1]
            c = loss_vae(outs["lat"][i][j, :n], lat_dim)
            d = loss_vae(outs["latcv"][i][j, :n], lat_dim)
            m_ss.append(a)
            m_sts.append(b)
            l_s.append(c)
            l_cv = list(l_s) + [d] if k > 0 else [d]   # :1393 concatenates onto the *lat_src* list
        total = total + F32(np.sum(m_ss, dtype=F32)) + F32(np.sum(l_s, dtype=F32))
        if not half_cyc:
            total = total + F32(np.sum(m_sts, dtype=F32)) + F32(np.sum(l_cv, dtype=F32))
    return F32(total)


# --------------------------------------------------------------------------------------------------------
# INT path: frame-window bookkeeping of the stage-4 generator (must be bit-exact)
# --------------------------------------------------------------------------------------------------------
def _advance(spc, flen_spc, s, e, st):
    """Shared scan body of src/bin/train_gru_cyclevae_gauss_batch.py:79-99 and :112-132 for one utterance."""
    for i in range(st["e_idx"] + 1, flen_spc):
        v = int(spc[i])
        if not st["s_flag"] and v >= s:
            if v > e:
                st["s_idx"] = -1
                break
            st["s_idx"] = i
            st["s_flag"], st["e_flag"] = True, False
            if i == flen_spc - 1:
                st["e_idx"] = i
                st["s_flag"], st["e_flag"] = False, True
                break
        elif not st["e_flag"] and (v >= e or i == flen_spc - 1):
            st["e_idx"] = i - 1 if v > e else i
            st["s_flag"], st["e_flag"] = False, True
            break


def window_bookkeeping(flens, spcidcs, flens_spc, batch_size=80):
    """All windows of ONE dataloader batch as yielded by train_generator (batch_size>0 branch).

    src/bin/train_gru_cyclevae_gauss_batch.py:70-134.  flens [U]; spcidcs [U,>=max(flens_spc)] int64 (zero
    padded, src/utils/dataset.py:23-31,93); flens_spc [U].
    Returns a list of dicts {s, e, s_idx[U], e_idx[U], flen_acc[U], select_utt_idx[list]} (copies per window).
    """
    U = len(flens)
    max_flen = int(np.max(flens))
    st = [dict(s_idx=-1, e_idx=-1, s_flag=False, e_flag=True) for _ in range(U)]
    flen_acc = np.repeat(batch_size, U).astype(np.int64)
    s, e = 0, batch_size - 1
    for j in range(U):
        _advance(spcidcs[j], int(flens_spc[j]), s, e, st[j])
    wins = []

    def snap(sel):
        wins.append(dict(s=s, e=e, s_idx=np.array([q["s_idx"] for q in st], np.int64),
                         e_idx=np.array([q["e_idx"] for q in st], np.int64),
                         flen_acc=flen_acc.copy(), select_utt_idx=list(sel)))

    snap(range(U))
    while e < max_flen - 1:
        s = e + 1
        e = s + batch_size - 1
        if e >= max_flen:
            e = max_flen - 1
        sel = []
        for j in range(U):
            if st[j]["e_idx"] < int(flens_spc[j]) - 1:
                if e >= flens[j]:
                    flen_acc[j] = flens[j] - s
                _advance(spcidcs[j], int(flens_spc[j]), s, e, st[j])
                sel.append(j)
        snap(sel)
    return wins


# ---- stage-6 post-processing -------------------------------------------------------------------
# The reference does these inline in float64 numpy (decode_gru-cyclevae_gauss.py); the aligned MCD comes from the
# third-party `dtw_c` extension (unpinned in tools/requirements.txt, source not in the tree): its per-frame value for aligned
# inputs is restated from the in-tree formula gru_vae.py:523, which IS pinned (tests/golden/tiny_ops.npz, twfse_branches.npz).
# The GV post-filter is pinned by tests/golden/gv_postfilter.npz: the output of the reference's own statements
# (decode_gru-cyclevae_gauss.py:419-422, ast-extracted by tests/golden/make_golden.py::case_gv), reproduced bit for bit.

def gv_postfilter(cvmcep, gv_mean_trg, cvgv_mean, dpow=None):
    """decode_gru-cyclevae_gauss.py:417-421: sqrt(gv_trg/cvgv) * (c - mean_t c) + mean_t c on coefficients 1.., coefficient 0
    kept (plus the power correction `dpow` of mod_pow, feature_extract_vc.py:131-138, when given).  Returns (out [T,D] f64,
    np.var(out[:,1:], 0))."""
    c = np.array(cvmcep, dtype=np.float64)
    if dpow is not None:
        c[:, 0] += np.asarray(dpow, np.float64)
    datamean = np.mean(c[:, 1:], axis=0)
    out = np.c_[c[:, 0], np.sqrt(np.asarray(gv_mean_trg, np.float64) / np.asarray(cvgv_mean, np.float64)) * (c[:, 1:] - datamean) + datamean]
    return out, np.var(out[:, 1:], axis=0)


def mc2e(mc, alpha=0.455, irlen=1024):
    """SPTK mc2e as pysptk.mc2e applies it per frame (reference mod_pow, feature_extract_vc.py:131-138).  pysptk (unpinned in
    tools/requirements.txt) is absent here and no reference test pins it: PARITY UNPINNED for this function; it restates SPTK's
    published freqt (frequency transformation, recursion over the input coefficients from the last to the first) and c2ir
    (h[0] = exp(c[0]), h[n] = (1/n) sum_{k=1..n} k c[k] h[n-k])."""
    mc = np.atleast_2d(np.asarray(mc, np.float64))
    m2 = irlen - 1
    out = np.empty(mc.shape[0], np.float64)
    a, b = -alpha, 1.0 - alpha * alpha
    kk = np.arange(irlen, dtype=np.float64)
    for f in range(mc.shape[0]):
        g = np.zeros(irlen, np.float64)
        for ci in mc[f][::-1]:                       # i = -m1 .. 0 reads c1[-i]
            d = g.copy()
            g[0] = ci + a * d[0]
            g[1] = b * d[0] + a * d[1]
            # g[j] = d[j-1] + a (d[j] - g[j-1]), j >= 2: first-order recurrence along j
            r = d[1:-1] + a * d[2:]
            for j in range(2, irlen):
                g[j] = r[j - 2] - a * g[j - 1]
        kc = kk * g
        h = np.zeros(irlen, np.float64)
        h[0] = np.exp(g[0])
        for n in range(1, irlen):
            h[n] = np.dot(kc[1:n + 1], h[n - 1::-1][:n]) / n
        out[f] = np.sum(h * h)
    return out


def mod_pow_dpow(cvmcep, mcep, alpha=0.455, irlen=1024):
    """feature_extract_vc.py:133-135: dpow = log(r_e / cv_e) / 2."""
    return np.log(mc2e(mcep, alpha, irlen) / mc2e(cvmcep, alpha, irlen)) / 2.0


def mcd_aligned(a, b, d0=1, L2=True):
    """Per-frame MCD of aligned sequences in float64 over coefficients d0.. (gru_vae.py:523 / :525; decode...:377-378 calls
    dtw_c.calc_mcd on float64 copies with d0 = 0 and d0 = 1).  Returns (frames, np.mean, np.std)."""
    d = np.asarray(a, np.float64)[:, d0:] - np.asarray(b, np.float64)[:, d0:]
    if L2:
        m = MCD_K * np.sqrt(2.0 * np.sum(d * d, 1))
    else:
        m = MCD_K * 1.4142135623730950488016887242097 * np.sum(np.abs(d), 1)
    return m, float(np.mean(m)), float(np.std(m))


def dtw_org_to_trg(org, trg, mcd=-1):
    """Dynamic time warping of `org` [T1,D] onto the time axis of `trg` [T2,D], the role dtw_c.dtw_org_to_trg plays at
    decode_gru-cyclevae_gauss.py:334-364, :424 and train...:679-688, :897-917.  dtw_c is a compiled third-party module whose source is
    NOT in the reference tree (tools/Makefile installs it from elsewhere, unpinned) and no reference test holds its outputs:
    PARITY UNPINNED.  What is restated is the textbook algorithm those call sites presuppose, with every choice written down:
      local cost  mcd != 0 (default): mel-cepstral distortion of the frame pair, 10/ln10 * sqrt(2 * sum_d (a_d - b_d)^2) [dB];
                  mcd == 0: cosine distance 1 - <a,b> / (|a| |b|)   (the call sites use it for latent-space "cosine similarity");
      steps       (i-1, j-1), (i-1, j), (i, j-1), all with weight 1 (symmetric type-1, no slope limit, no window);
      boundary    path from (0, 0) to (T1-1, T2-1); ties in the backtrack prefer the diagonal, then (i-1, j), then (i, j-1);
      warp        target frame j takes the org frame i with the smallest local cost among the path points (., j) (first on ties).
    Returns (aligned_org [T2,D], twf [T2] int64 = the chosen i per j, mean over j of the chosen local costs, those costs [T2]).
    float64 throughout."""
    a, b = np.asarray(org, np.float64), np.asarray(trg, np.float64)
    T1, T2 = a.shape[0], b.shape[0]
    if mcd != 0:
        d = a[:, None, :] - b[None, :, :]
        cost = MCD_K * np.sqrt(2.0 * np.sum(d * d, 2))
    else:
        na, nb = np.sqrt(np.sum(a * a, 1)), np.sqrt(np.sum(b * b, 1))
        cost = 1.0 - (a @ b.T) / (na[:, None] * nb[None, :])
    acc = np.full((T1, T2), np.inf)
    acc[0, 0] = cost[0, 0]
    for i in range(T1):
        for j in range(T2):
            if i == 0 and j == 0:
                continue
            best = np.inf
            if i > 0 and j > 0:
                best = acc[i - 1, j - 1]
            if i > 0 and acc[i - 1, j] < best:
                best = acc[i - 1, j]
            if j > 0 and acc[i, j - 1] < best:
                best = acc[i, j - 1]
            acc[i, j] = cost[i, j] + best
    twf = np.full(T2, -1, np.int64)
    fr = np.full(T2, np.inf)
    i, j = T1 - 1, T2 - 1
    while True:
        if cost[i, j] <= fr[j]:          # walking backwards: "<=" keeps the FIRST (smallest i) of equal costs
            fr[j], twf[j] = cost[i, j], i
        if i == 0 and j == 0:
            break
        cands = []
        if i > 0 and j > 0:
            cands.append((acc[i - 1, j - 1], 0, i - 1, j - 1))
        if i > 0:
            cands.append((acc[i - 1, j], 1, i - 1, j))
        if j > 0:
            cands.append((acc[i, j - 1], 2, i, j - 1))
        _, _, i, j = min(cands)
    return a[twf], twf, float(np.mean(fr)), fr
