"""Stage-6 post-processing on the device, next to the decoder output.

The reference copies every converted utterance to the host as float64 numpy (decode_gru-cyclevae_gauss.py:319) and then
applies the GV post-filter (:417-421) and computes frame-wise MCD through the `dtw_c` extension (:377-378).  These two
functions keep the trajectory in HBM: fp32 decoder output in, float64 results out (torch tensors on the same device),
through `cvae_gv_postfilter` / `cvae_mcd_aligned` of libcyclevae_hip.so.  HIP device tensors only; no fallback.
"""
import torch

import gru_vae


def _dev_f32(t, name):
    gru_vae._need_cuda(t, name)
    return t.to(torch.float32).contiguous()


def _f64(v, dev):
    return torch.as_tensor(v, dtype=torch.float64, device=dev).contiguous()


def gv_postfilter(cvmcep, gv_mean_trg, cvgv_mean, dpow=None):
    """cvmcep [T,D] (decoder output, device); gv_mean_trg, cvgv_mean [D-1] (statistics of the recipe: GV of the target speaker,
    mean GV of the converted training set); dpow [T] or None.  Returns (cvmcep_gv [T,D] float64, var [D-1] float64)."""
    c = _dev_f32(cvmcep, "gv_postfilter(cvmcep)")
    T, D = c.shape
    dev = c.device
    gv, cg = _f64(gv_mean_trg, dev), _f64(cvgv_mean, dev)
    if gv.numel() != D - 1 or cg.numel() != D - 1:
        raise ValueError("gv_mean_trg / cvgv_mean must have D-1 = %d entries" % (D - 1))
    dp = None if dpow is None else _f64(dpow, dev)
    if dp is not None and dp.numel() != T:
        raise ValueError("dpow must have T = %d entries" % T)
    out = torch.empty(T, D, dtype=torch.float64, device=dev)
    var = torch.empty(D - 1, dtype=torch.float64, device=dev)
    work = torch.empty(2 * D, dtype=torch.float64, device=dev)
    gru_vae._lib().gv_postfilter(c.data_ptr(), T, D, 0 if dp is None else dp.data_ptr(), gv.data_ptr(), cg.data_ptr(),
                                 out.data_ptr(), var.data_ptr(), work.data_ptr(), torch.cuda.current_stream().cuda_stream)
    return out, var


def mc2e(mc, alpha=0.455, irlen=1024):
    """Impulse-response energy per frame (SPTK mc2e, pysptk.mc2e in the reference's mod_pow): mc [T,D] device tensor (float32 or
    float64) -> [T] float64 on the device."""
    gru_vae._need_cuda(mc, "mc2e(mc)")
    m = mc.contiguous() if mc.dtype in (torch.float32, torch.float64) else mc.to(torch.float32).contiguous()
    T, D = m.shape
    e = torch.empty(T, dtype=torch.float64, device=m.device)
    gru_vae._lib().mc2e(m.data_ptr(), m.dtype == torch.float64, D, T, D, float(alpha), int(irlen), e.data_ptr(),
                        torch.cuda.current_stream().cuda_stream)
    return e


def mod_pow_dpow(cvmcep, mcep, alpha=0.455, irlen=1024):
    """The power correction of mod_pow (feature_extract_vc.py:131-138): dpow[t] = log(mc2e(mcep[t]) / mc2e(cvmcep[t])) / 2, [T]
    float64 on the device; pass it as `dpow` to gv_postfilter (mod_pow adds it to coefficient 0, decode...:406, before the GV
    post-filter of :419-420).  PARITY UNPINNED: mc2e is pysptk's (SPTK freqt + c2ir), a third-party binary that is neither in the
    reference tree nor in this image; cvae_mc2e restates the published algorithm and is held to the identities that define it
    (tests/test_emu_stage6.py, test_gpu_parity.py::test_mc2e_and_mod_pow_on_device), not to SPTK's output."""
    return torch.log(mc2e(mcep, alpha, irlen) / mc2e(cvmcep, alpha, irlen)) / 2.0


def mcd_aligned(a, b, d0=1, L2=True):
    """Frame-wise MCD [dB] of two aligned [rows,D] trajectories over coefficients d0.. (d0=0: "mcdpow", d0=1: "mcd").
    Returns (frames [rows] float64, stats [4] float64 = sum, mean, np.std, torch.std), all on the device."""
    a, b = _dev_f32(a, "mcd_aligned(a)"), _dev_f32(b, "mcd_aligned(b)")
    if a.shape != b.shape or a.dim() != 2:
        raise ValueError("mcd_aligned needs two [rows, D] tensors of the same shape")
    rows, D = a.shape
    frames = torch.empty(rows, dtype=torch.float64, device=a.device)
    stats = torch.empty(4, dtype=torch.float64, device=a.device)
    gru_vae._lib().mcd_aligned(a.data_ptr(), D, b.data_ptr(), D, rows, D, d0, L2, frames.data_ptr(), stats.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
    return frames, stats


def dtw_org_to_trg(org, trg, mcd=-1):
    """Device counterpart of dtw_c.dtw_org_to_trg as decode_gru-cyclevae_gauss.py:334-364 / :424 calls it: org [T1,D], trg [T2,D]
    device tensors (any float dtype; computed in f64) -> (aligned_org [T2,D] f64, twf [T2] int64, mean local cost (0-dim f64 tensor),
    per-frame local costs [T2] f64).  mcd != 0: mel-cd per frame pair in dB (the arrays the reference averages into "mcdpow" /
    "mcd"), mcd == 0: cosine distance.  dtw_c's source is not in the reference tree: PARITY UNPINNED, algorithm documented at
    oracle/cyclevae_oracle.py::dtw_org_to_trg."""
    lib = gru_vae._lib()
    a, b = org.to(torch.float64).contiguous(), trg.to(torch.float64).contiguous()
    T1, T2, D = a.shape[0], b.shape[0], a.shape[1]
    dev = a.device
    aligned = torch.empty(T2, D, dtype=torch.float64, device=dev)
    twf = torch.empty(T2, dtype=torch.int64, device=dev)
    frames = torch.empty(T2, dtype=torch.float64, device=dev)
    mean = torch.empty(1, dtype=torch.float64, device=dev)
    nb = lib.dtw_work_bytes(T1, T2)
    work = torch.empty(nb, dtype=torch.uint8, device=dev)
    lib.dtw_org_to_trg(a.data_ptr(), b.data_ptr(), T1, T2, D, int(mcd), aligned.data_ptr(), twf.data_ptr(), frames.data_ptr(),
                       mean.data_ptr(), work.data_ptr(), nb, gru_vae._stream())
    return aligned, twf, mean[0], frames


def _encode_pairs(model_encoder, pairs, y_in_pp, lat_dim):
    """First half of convert_pairs on the current stream: all 2N encoder passes as one pass over 2N stacked rows.  Returns what the
    decoder half needs."""
    N = len(pairs)
    if N < 1 or 3 * N > 32:
        raise ValueError("1..10 utterance pairs per call, got %d" % N)
    gru_vae._need_cuda(pairs[0][0], "convert_pairs(feat_src)")
    lib = gru_vae._lib()
    dev = pairs[0][0].device
    f = lambda t: t.to(torch.float32).contiguous()
    feats = [(f(a), f(b)) for a, b in pairs]
    lens = [(a.shape[0], b.shape[0]) for a, b in feats]
    T = max(max(l) for l in lens)
    L, Cin = lat_dim, model_encoder.in_dim
    st = torch.cuda.current_stream().cuda_stream
    de, ie = model_encoder.prepared(dev)
    ypp = f(y_in_pp.reshape(1, -1))
    lat = torch.empty(2 * N, T, 2 * L, dtype=torch.float32, device=dev)
    ws = torch.empty(lib.pass_workspace_bytes(de, 2 * N, T), dtype=torch.uint8, device=dev)
    pins = []
    for (a, b), (ta, tb) in zip(feats, lens):
        pins += [lib.pass_input((a.data_ptr(), Cin, Cin), frames=ta), lib.pass_input((b.data_ptr(), Cin, Cin), frames=tb)]
    lib.gru_rnn_forward_stacked(de, ie.data_ptr(), pins, [ypp.data_ptr()] * (2 * N), 1, T, L,
                                [lat[r].data_ptr() for r in range(2 * N)], ws.data_ptr(), ws.numel(), gru_vae._flags(), st)
    return {"N": N, "lens": lens, "T": T, "lat": lat, "dev": dev, "keep": (feats, ypp, ws)}


def _decode_pairs(model_decoder, enc, y_in_src, y_in_trg, lat_dim, n_smpl_dec, eps, seed, pair_ids=None):
    """Second half of convert_pairs on the current stream: the n_smpl_dec-draw latent means and all 3N decoder passes as one pass
    over 3N stacked rows."""
    lib = gru_vae._lib()
    N, lens, T, lat, dev = enc["N"], enc["lens"], enc["T"], enc["lat"], enc["dev"]
    L, Co = lat_dim, model_decoder.out_dim
    f = lambda t: t.to(torch.float32).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    dd, idd = model_decoder.prepared(dev)
    ws = torch.empty(lib.pass_workspace_bytes(dd, 3 * N, T), dtype=torch.uint8, device=dev)
    codes = torch.tensor([[1.0, 0.0], [0.0, 1.0]], dtype=torch.float32, device=dev)     # src_code, trg_code (decode...:309-314)

    def pad_eps(e):
        if e is None:
            return None
        e = f(e)
        if e.shape[1] == T:
            return e
        out_ = torch.zeros(e.shape[0], T, L, dtype=torch.float32, device=dev)
        out_[:, :e.shape[1]] = e
        return out_

    sd = gru_vae._draw_seed() if seed is None else seed
    n = int(n_smpl_dec)
    out = torch.empty(3 * N, T, Co, dtype=torch.float32, device=dev)
    ys, yt = f(y_in_src.reshape(1, -1)), f(y_in_trg.reshape(1, -1))
    keep, pins, yins = [], [], []
    for q, (ta, tb) in enumerate(lens):
        es, et = (None, None) if eps is None else (pad_eps(eps[q][0]), pad_eps(eps[q][1]))
        keep += [es, et]

        def cell(code_row, lat_row, e, frames, draw0):
            return lib.pass_input((codes[code_row].data_ptr(), 2, 0), lat=lat[lat_row].data_ptr(), lat_dim=L,
                                  eps=None if e is None else e.data_ptr(), seed=sd, draw_id=draw0, frames=frames, n_draws=n)

        # cvmcep and cvmcep_src share ONE sampling of lat_src (decode...:304-305), cvmcep_trg has its own (:307-308)
        # Philox draws are keyed (seed, draw id, frame, dim): pair q of this call takes the ids 2nq .. 2nq + 2n - 1 -- or, with
        # pair_ids, those of its position in the caller's whole list, so that its draws do not depend on how the list was grouped
        # into calls or split over devices (convert_many(first_pair_id=...), convert_files)
        d0 = 2 * n * (q if pair_ids is None else int(pair_ids[q]))
        pins += [cell(1, 2 * q, es, ta, d0), cell(0, 2 * q, es, ta, d0), cell(1, 2 * q + 1, et, tb, d0 + n)]
        yins += [yt.data_ptr(), ys.data_ptr(), yt.data_ptr()]
    lib.gru_rnn_forward_stacked(dd, idd.data_ptr(), pins, yins, 1, T, -1, [out[r].data_ptr() for r in range(3 * N)],
                                ws.data_ptr(), ws.numel(), gru_vae._flags(), st)
    return out, [(out[3 * q, :ta], out[3 * q + 1, :ta], out[3 * q + 2, :tb], lat[2 * q, :ta], lat[2 * q + 1, :tb])
                 for q, (ta, tb) in enumerate(lens)]


def convert_pairs(model_encoder, model_decoder, pairs, y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec=300, eps=None, seed=None,
                  window=None):
    """The network part of stage 6 (reference decode_gru-cyclevae_gauss.py:302-323) for SEVERAL (source, target) utterance pairs
    at once.  For every pair:

        lat_src = E(feat_src), lat_trg = E(feat_trg);  z = mean over n_smpl_dec draws of sampling_vae_batch(lat)
        cvmcep = D([trg_code; z_src]),  cvmcep_src = D([src_code; z_src]),  cvmcep_trg = D([trg_code; z_trg])

    All 2N encoder passes run as ONE pass over 2N stacked rows, all 3N decoder passes as one over 3N rows
    (cvae_gru_rnn_forward_stacked): rows are independent recurrences, a dependent step costs the same chip-wide hand-off for one
    row and for thirty-two (one row tile), utterances of different length are padded with zeros AFTER normalisation exactly like the conv padding
    they would see alone, and the n_smpl_dec-draw latent mean is taken inside the pass prologue (no [n_smpl_dec, T, L] tensor).
    pairs: list of (feat_src [Ts,Cin], feat_trg [Tt,Cin]) device tensors, at most 10 pairs (32 stacked rows per pass = one row tile);
    y_in_* as the reference passes them ([1,1,C]); eps None (Philox) or a list of (eps_src [n,Ts,L], eps_trg [n,Tt,L]).
    window (frames, or a list of window lengths; ONE pair only): run the pair as a pass-level wavefront over windows of that many
    frames, the decoder of window w beside the encoder of window w+1 (_convert_pair_windowed): same values bit for bit, lower
    latency (3.15 -> 2.8-2.9 ms for a 637 / 660-frame pair at window=224).
    Returns a list of (cvmcep [Ts,Co], cvmcep_src [Ts,Co], cvmcep_trg [Tt,Co], lat_src [Ts,2L], lat_trg [Tt,2L]) (fp32, device).
    """
    gru_vae.check_status()
    if window and len(pairs) != 1:
        raise ValueError("convert_pairs(window=...) runs ONE pair as a wavefront of windows, got %d pairs" % len(pairs))
    if window:
        gru_vae._need_cuda(pairs[0][0], "convert_pairs(feat_src)")
        return _convert_pair_windowed(model_encoder, model_decoder, pairs[0], y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec,
                                      None if eps is None else eps[0], seed, window)
    enc = _encode_pairs(model_encoder, pairs, y_in_pp, lat_dim)
    return _decode_pairs(model_decoder, enc, y_in_src, y_in_trg, lat_dim, n_smpl_dec, eps, seed)[1]


def _window_edges(Tmax, window, reach):
    """Window edges of the wavefront: `window` frames each, or the caller's own list of window lengths (the last one repeats).  (A
    short first window, so that the decoder starts early, measured no better: every window costs ~0.15 ms of launches on the
    decoder's chain.)  Every window must exceed twice the front-end's reach; a sliver at the end joins the window before it."""
    sizes = [int(v) for v in window] if isinstance(window, (list, tuple)) else [int(window)]
    if min(sizes) <= 2 * reach:
        raise ValueError("windows of %s frames: each must exceed twice the front-end's reach (%d)" % (sizes, reach))
    edges = [0]
    while edges[-1] < Tmax:
        edges.append(min(Tmax, edges[-1] + sizes[min(len(edges) - 1, len(sizes) - 1)]))
    if len(edges) > 2 and edges[-1] - edges[-2] <= 2 * reach:
        del edges[-2]
    return edges


def _decoder_window_schedule(nfr, edges, reach):
    """Which decoder rows advance in which window of the wavefront, and by how many frames.  nfr[i]: frames of decoder row i;
    edges: window edges of the ENCODER passes (frame numbers); reach: frames the conv front-end looks ahead.  Returns one
    (rows, spans) per window.  After encoder window w (frames < stop) a row can advance to stop - reach, or FINISH (advance to its
    last frame) once the encoder has passed its end.  The stacked pass runs T = max(spans) steps for EVERY row, so while another
    row is still unfinished a row may only finish if that takes no more steps than the unfinished rows take: an unfinished row
    with frames < T would step on over latent frames the encoder has not written yet, and before ABI 6 its carried state was read
    behind those extra steps (lens (446, 660) at window 224 gave spans (226, 226, 224)).  Such a row finishes one window
    later, with a span of at most `reach` frames.  Invariant (asserted): a span shorter than T always FINISHES its row."""
    done, out = [0] * len(nfr), []
    for w in range(len(edges) - 1):
        stop = edges[w + 1]
        unfinished = any(n > stop for n in nfr)
        target = [n if stop >= n and not (unfinished and n > stop - reach) else stop - reach for n in nfr]
        rows = [i for i in range(len(nfr)) if target[i] > done[i]]
        spans = [target[i] - done[i] for i in rows]
        T = max(spans) if spans else 0
        assert all(k == T or nfr[i] == done[i] + k for i, k in zip(rows, spans)), (nfr, edges, w, spans, done)
        out.append((rows, spans))
        for i, k in zip(rows, spans):
            done[i] += k
    assert done == list(nfr), (nfr, edges, done)
    return out


def _convert_pair_windowed(model_encoder, model_decoder, pair, y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec, eps, seed, window):
    """convert_pairs for ONE utterance pair as a pass-level wavefront: the utterances are cut into windows of `window` frames, the
    encoder launch of window w+1 runs on a second stream beside the decoder launch of window w (the decoder lags by the conv
    front-end's reach, 4 frames: its window w needs latent frames up to the end of encoder window w).  A window is a pass with
    carried state whose front-end sees the neighbouring frames of the utterance (cvae_gru_rnn_forward_stacked_carry, ABI 5), so the
    result is the unbroken pass bit for bit.  Both recurrences are the word-exchange kernels (<= 3 rows), co-resident on every CU."""
    lib = gru_vae._lib()
    f = lambda t: t.to(torch.float32).contiguous()
    fs, ft = f(pair[0]), f(pair[1])
    dev = fs.device
    lens = (fs.shape[0], ft.shape[0])
    Tmax, L, Cin, Co, H = max(lens), lat_dim, model_encoder.in_dim, model_decoder.out_dim, model_encoder.hidden_units
    reach = (model_decoder.kernel_size ** 2 - 1) // 2
    if (model_encoder.kernel_size ** 2 - 1) // 2 != reach:
        raise ValueError("encoder and decoder front-ends reach %d / %d frames: the window schedule assumes one reach"
                         % ((model_encoder.kernel_size ** 2 - 1) // 2, reach))
    if min(fs.shape[0], ft.shape[0]) < 1:
        raise ValueError("convert_pairs(window=...): empty utterance")
    edges = _window_edges(Tmax, window, reach)
    if dev not in _pipe_streams:
        with torch.cuda.device(dev):
            s0 = gru_vae.concurrent_stream(dev, slot=1)
            _pipe_streams[dev] = (s0, gru_vae.concurrent_stream(dev, slot=2, beside=[s0]))     # (probed: really concurrent with each other)
    s_enc, s_dec = _pipe_streams[dev]
    cur = torch.cuda.current_stream(dev)
    de, ie = model_encoder.prepared(dev)
    dd, idd = model_decoder.prepared(dev)
    flags = gru_vae._flags()
    ypp, ys, yt = f(y_in_pp.reshape(1, -1)), f(y_in_src.reshape(1, -1)), f(y_in_trg.reshape(1, -1))
    lat = torch.empty(2, Tmax, 2 * L, dtype=torch.float32, device=dev)
    out = torch.empty(3, Tmax, Co, dtype=torch.float32, device=dev)
    h_enc = torch.zeros(2, H, dtype=torch.float32, device=dev)
    h_dec = torch.zeros(3, model_decoder.hidden_units, dtype=torch.float32, device=dev)
    codes = torch.tensor([[1.0, 0.0], [0.0, 1.0]], dtype=torch.float32, device=dev)     # src_code, trg_code (decode...:309-314)
    n = int(n_smpl_dec)
    sd = gru_vae._draw_seed() if seed is None else seed
    e_src, e_trg = (None, None) if eps is None else (f(eps[0]), f(eps[1]))
    feats = (fs, ft)
    # decoder rows: (code row, latent row, eps, frames, first draw id, y_in) -- cvmcep and cvmcep_src share ONE sampling of lat_src
    drows = ((1, 0, e_src, lens[0], 0, yt), (0, 0, e_src, lens[0], 0, ys), (1, 1, e_trg, lens[1], n, yt))
    nwin = len(edges) - 1
    keep = [fs, ft, ypp, ys, yt, lat, out, h_enc, h_dec, codes, e_src, e_trg]
    # the schedule: per window, the encoder rows still running and the decoder rows that can advance (the decoder lags by `reach`:
    # its frames < d1 need latent frames < d1 + reach, which exist once encoder window w is through)
    enc_calls, dec_calls, done = [], [], [0, 0, 0]
    dec_sched = _decoder_window_schedule([row[3] for row in drows], edges, reach)
    for w in range(nwin):
        start, stop = edges[w], edges[w + 1]
        alive = [r for r in range(2) if lens[r] > start]
        fr = [min(stop, lens[r]) - start for r in alive]
        T = max(fr)
        ws = torch.empty(lib.pass_workspace_bytes(de, len(alive), T), dtype=torch.uint8, device=dev)
        keep.append(ws)
        enc_calls.append((de, ie.data_ptr(),
                          [lib.pass_input((feats[r].data_ptr() + start * Cin * 4, Cin, Cin), frames=k, ctx_before=start,
                                          ctx_after=lens[r] - start - k) for r, k in zip(alive, fr)],
                          [ypp.data_ptr() if w == 0 else None] * len(alive), [None if w == 0 else h_enc[r].data_ptr() for r in alive], 1, T, L,
                          [lat[r].data_ptr() + start * 2 * L * 4 for r in alive], [h_enc[r].data_ptr() for r in alive], ws.data_ptr(), ws.numel()))
        rows, spans = dec_sched[w]
        if not rows:
            dec_calls.append(None)
            continue
        T = max(spans)
        pins = []
        for i, k in zip(rows, spans):
            crow, lr, e, nfr, draw0, _ = drows[i]
            d0 = done[i]
            pins.append(lib.pass_input((codes[crow].data_ptr(), 2, 0), lat=lat[lr].data_ptr() + d0 * 2 * L * 4, lat_dim=L,
                                       eps=None if e is None else e.data_ptr() + d0 * L * 4, seed=sd, draw_id=draw0, frames=k, n_draws=n,
                                       ctx_before=d0, ctx_after=nfr - d0 - k, draw_frame0=d0, eps_draw_stride=0 if e is None else e.shape[1] * L))
        ws = torch.empty(lib.pass_workspace_bytes(dd, len(rows), T), dtype=torch.uint8, device=dev)
        keep.append(ws)
        dec_calls.append((dd, idd.data_ptr(), pins, [drows[i][5].data_ptr() if done[i] == 0 else None for i in rows],
                          [None if done[i] == 0 else h_dec[i].data_ptr() for i in rows], 1, T, -1,
                          [out[i].data_ptr() + done[i] * Co * 4 for i in rows], [h_dec[i].data_ptr() for i in rows], ws.data_ptr(), ws.numel()))
        for i, k in zip(rows, spans):
            done[i] += k
    # Two streams, one per net: the encoder windows form one dependent chain, the decoder windows another (each decoder window
    # behind the encoder window that produced its latent frames), and the two chains run side by side.  (Measured and dropped:
    # issuing every pass as two calls -- input-side front-end on a third / fourth stream ahead of the state-dependent rest -- made
    # the pair SLOWER, 3.06 vs 2.88 ms: front-end kernels that share the CUs with two polling recurrences slow both.)
    s_enc.wait_stream(cur)
    s_dec.wait_stream(cur)
    for w in range(nwin):
        lib.gru_rnn_forward_stacked_carry(*enc_calls[w], flags, s_enc.cuda_stream)
        if dec_calls[w] is None:
            continue
        ev = torch.cuda.Event()
        ev.record(s_enc)
        s_dec.wait_event(ev)
        lib.gru_rnn_forward_stacked_carry(*dec_calls[w], flags, s_dec.cuda_stream)
    cur.wait_stream(s_enc)
    cur.wait_stream(s_dec)
    for t in keep:
        if t is not None:
            t.record_stream(s_enc)
            t.record_stream(s_dec)
    return [(out[0, :lens[0]], out[1, :lens[0]], out[2, :lens[1]], lat[0, :lens[0]], lat[1, :lens[1]])]


_pipe_streams = {}


def convert_list(model_encoder, model_decoder, groups, y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec=300, eps=None, seeds=None,
                 pair_ids=None):
    """convert_pairs over a LIST of calls (the file list one GPU gets, decode...:190-195), software-pipelined over two streams: the
    encoder pass of group g+1 runs side by side with the decoder pass of group g.  Both are hand-off-bound recurrences that leave
    most of every CU idle: one utterance pair per group (the word-exchange kernels, <= 3 rows per pass) runs two such passes side
    by side in 1.66 ms where one after the other takes 2.52 ms (tools/ll_corun.py, MI355X).  Groups of several pairs (<= 32 rows
    per pass: ONE row tile of the dataflow kernel = 128 blocks of a whole CU each, half the chip) overlap too, on disjoint CUs:
    ten-pair calls 6.9 -> 4.6 ms per call = 1.38 M converted frames/s.  Results are bit-identical to convert_pairs group by group
    (tests/test_gpu_parity.py).  groups: list of lists of (feat_src, feat_trg); eps / seeds: None or
    one entry per group, as convert_pairs takes them; pair_ids: None or per group the list positions that key the pairs' draws
    (_decode_pairs).  Returns one convert_pairs result per group; everything is ordered behind the current stream on entry and
    ahead of it on return."""
    if not groups:
        return []
    gru_vae._need_cuda(groups[0][0][0], "convert_list(feat_src)")
    gru_vae.check_status()
    dev = groups[0][0][0].device
    if dev not in _pipe_streams:
        with torch.cuda.device(dev):
            s0 = gru_vae.concurrent_stream(dev, slot=1)
            _pipe_streams[dev] = (s0, gru_vae.concurrent_stream(dev, slot=2, beside=[s0]))     # (probed: really concurrent with each other)
    s_enc, s_dec = _pipe_streams[dev]
    cur = torch.cuda.current_stream(dev)
    s_enc.wait_stream(cur)
    s_dec.wait_stream(cur)
    results, pending = [], None
    for g in range(len(groups) + 1):
        nxt = None
        if g < len(groups):
            with torch.cuda.stream(s_enc):
                enc = _encode_pairs(model_encoder, groups[g], y_in_pp, lat_dim)
                ev = torch.cuda.Event()
                ev.record(s_enc)
            nxt = (g, enc, ev)
        if pending is not None:
            gi, enc_p, ev_p = pending
            with torch.cuda.stream(s_dec):
                s_dec.wait_event(ev_p)
                enc_p["lat"].record_stream(s_dec)
                for t in enc_p["keep"][0]:
                    for x in t:
                        x.record_stream(s_dec)
                out, res = _decode_pairs(model_decoder, enc_p, y_in_src, y_in_trg, lat_dim, n_smpl_dec,
                                         None if eps is None else eps[gi], None if seeds is None else seeds[gi],
                                         None if pair_ids is None else pair_ids[gi])
                out.record_stream(cur)
                enc_p["lat"].record_stream(cur)
            results.append(res)
        pending = nxt
    cur.wait_stream(s_enc)
    cur.wait_stream(s_dec)
    return results


def convert_many(model_encoder, model_decoder, pairs, y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec=300, per_call=10, seed=None,
                 first_pair_id=None):
    """A flat file list of (feat_src, feat_trg) pairs (decode...:190-195) through convert_list in calls of `per_call` pairs (<= 10:
    30 decoder rows are one row tile), sorted by length so that a call's rows are padded little; results in the order given.
    1.38 M converted frames/s at 637 / 660-frame pairs on one MI355X.  seed: None, or the base of the per-call Philox seeds.
    first_pair_id (with seed): pair i draws with (seed, first_pair_id + i) whatever call it lands in -- the results then do not depend
    on per_call, nor on how a longer list was split over devices (convert_files)."""
    if not 1 <= int(per_call) <= 10:
        raise ValueError("per_call must be 1..10, got %r" % (per_call,))
    if first_pair_id is not None and seed is None:
        raise ValueError("first_pair_id keys the draws of a pair by (seed, position): give a seed")
    order = sorted(range(len(pairs)), key=lambda i: -max(pairs[i][0].shape[0], pairs[i][1].shape[0]))
    groups = [order[k:k + int(per_call)] for k in range(0, len(order), int(per_call))]
    if first_pair_id is None:
        seeds, ids = None if seed is None else [int(seed) + k for k in range(len(groups))], None
    else:
        seeds, ids = [int(seed)] * len(groups), [[int(first_pair_id) + i for i in g] for g in groups]
    res = convert_list(model_encoder, model_decoder, [[pairs[i] for i in g] for g in groups], y_in_pp, y_in_src, y_in_trg, lat_dim,
                       n_smpl_dec, None, seeds, ids)
    out = [None] * len(pairs)
    for g, rg in zip(groups, res):
        for i, r in zip(g, rg):
            out[i] = r
    return out


def split_file_list(items, n_dev):
    """The reference's fan-out of a file list over its GPUs (decode_gru-cyclevae_gauss.py:190-195, calc_cvgv...:120-123):
    np.array_split -- contiguous chunks, the first len % n ones longer by one.  Returns [(first index, chunk)] per device."""
    import numpy as np
    idx = np.array_split(np.arange(len(items)), int(n_dev))
    return [(int(ix[0]) if len(ix) else len(items), [items[int(i)] for i in ix]) for ix in idx]


def _net_config(m):
    """What a worker process needs to rebuild a GRU_RNN: constructor arguments and the state dict on the host."""
    kw = dict(in_dim=m.in_dim, out_dim=m.out_dim, hidden_units=m.hidden_units, kernel_size=m.kernel_size,
              dilation_size=m.dilation_size, do_prob=m.do_prob, scale_in_flag=m.scale_in_flag, scale_out_flag=m.scale_out_flag)
    return kw, {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}      # (numpy: a worker need not import torch to unpickle it)


def _files_worker(rank, device, first, items, cfg, queue):
    """One process per GPU (decode...:591-602 starts one mp.Process per file-list chunk): rebuild both networks on `device`, read the
    chunk's feature files (read_hdf5(path, "/feat_org_lf0"), decode...:236,257), convert_many, results to the parent as numpy."""
    try:
        import numpy as np
        import hdf5io
        dev = torch.device("cuda", int(device))
        torch.cuda.set_device(dev)
        nets = []
        for kw, sd in (cfg["enc"], cfg["dec"]):
            m = gru_vae.GRU_RNN(**kw)
            m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
            nets.append(m.to(dev).eval())
        rd = cfg.get("reader") or (lambda path: hdf5io.read_hdf5(path, cfg["key"]))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        pairs = [(t(rd(a)), t(rd(b))) for a, b in items]
        y = [torch.from_numpy(v).to(dev) for v in cfg["y_in"]]
        with torch.no_grad():
            res = convert_many(nets[0], nets[1], pairs, y[0], y[1], y[2], cfg["lat_dim"], cfg["n_smpl_dec"], cfg["per_call"],
                               cfg["seed"], None if cfg["seed"] is None else first) if pairs else []
        torch.cuda.synchronize(dev)
        gru_vae.check_status()
        queue.put((rank, first, [tuple(x.cpu().numpy() for x in r) for r in res], None))
    except BaseException as e:      # the parent re-raises: a worker must never leave it waiting on the queue
        import traceback
        queue.put((rank, first, None, "%s\n%s" % (e, traceback.format_exc())))


def convert_files(model_encoder, model_decoder, file_pairs, devices, y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec=300, per_call=10,
                  seed=None, key="/feat_org_lf0", reader=None, worker=None, timeout=None):
    """The multi-GPU form of stage 5 / 6 (decode_gru-cyclevae_gauss.py:190-195, 591-602; calc_cvgv_gru-cyclevae_gauss.py:120-123,
    286-318): the file list is split into len(devices) contiguous chunks (np.array_split, as the reference does), ONE PROCESS PER
    DEVICE converts its chunk with convert_many, and the results come back in the order of `file_pairs`.  No collective: utterances
    are independent.
    file_pairs: list of (source feature file, target feature file), each an HDF5 file with `key` [T, Cin] (or whatever `reader(path)`
    turns into a [T, Cin] array); devices: list of CUDA device indices; model_*: GRU_RNN modules (any device: the workers get their
    state dicts); y_in_*: as convert_pairs takes them.  seed: None (every worker draws from its own generator) or the Philox seed --
    pair i of the list then draws with (seed, i), so the result does not depend on the number of devices or on per_call.
    worker: the per-device function (tests run the same fan-out on the host build of the library); default _files_worker.
    `reader` and `worker` cross a process boundary under the spawn context: they must be picklable top-level functions (no lambdas).
    Returns a list of (cvmcep [Ts,Co], cvmcep_src, cvmcep_trg [Tt,Co], lat_src [Ts,2L], lat_trg [Tt,2L]) float32 numpy arrays."""
    import torch.multiprocessing as mp
    devices = list(devices)
    if not devices:
        raise ValueError("convert_files needs at least one device")
    cfg = {"enc": _net_config(model_encoder), "dec": _net_config(model_decoder),
           "y_in": [v.detach().cpu().numpy() for v in (y_in_pp, y_in_src, y_in_trg)], "lat_dim": int(lat_dim), "n_smpl_dec": int(n_smpl_dec),
           "per_call": int(per_call), "seed": None if seed is None else int(seed), "key": key, "reader": reader}
    chunks = split_file_list(list(file_pairs), len(devices))
    fn = worker or _files_worker
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    procs = []
    for rank, (dev, (first, items)) in enumerate(zip(devices, chunks)):
        p = ctx.Process(target=fn, args=(rank, dev, first, items, cfg, queue))
        p.start()
        procs.append(p)
    out, errors = [None] * len(file_pairs), []
    try:
        import queue as _queue
        import time as _time
        deadline = None if timeout is None else _time.monotonic() + timeout
        reported = set()
        for _ in procs:
            # a worker that dies without reaching its except clause (segfault, OOM killer, hipErrorLaunchFailure abort, an import
            # failure under spawn) never posts: poll, and look at the processes between polls
            while True:
                try:
                    rank, first, res, err = queue.get(timeout=1.0)
                    break
                except _queue.Empty:
                    dead = [(r, q.exitcode) for r, q in enumerate(procs) if r not in reported and not q.is_alive() and q.exitcode not in (0, None)]
                    if dead:
                        raise RuntimeError("convert_files: " + "; ".join("the worker of device %s (chunk %d) died with exit code %s "
                                           "before it returned a result" % (devices[r], r, c) for r, c in dead))
                    if deadline is not None and _time.monotonic() > deadline:
                        raise RuntimeError("convert_files: no result within %s s" % timeout)
            reported.add(rank)
            if err is not None:
                errors.append("device %s (chunk %d): %s" % (devices[rank], rank, err))
                continue
            if len(res) != len(chunks[rank][1]):
                errors.append("device %s returned %d results for %d pairs" % (devices[rank], len(res), len(chunks[rank][1])))
                continue
            out[first:first + len(res)] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()
    if errors:
        raise RuntimeError("convert_files: " + "; ".join(errors))
    return out


def convert_pair(model_encoder, model_decoder, feat_src, feat_trg, y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec=300,
                 eps_src=None, eps_trg=None, seed=None, window=None):
    """convert_pairs for ONE utterance pair: two launches of dependent steps instead of the five passes of decode...:303-323."""
    e = None if eps_src is None and eps_trg is None else [(eps_src, eps_trg)]
    return convert_pairs(model_encoder, model_decoder, [(feat_src, feat_trg)], y_in_pp, y_in_src, y_in_trg, lat_dim, n_smpl_dec,
                         e, seed, window)[0]
