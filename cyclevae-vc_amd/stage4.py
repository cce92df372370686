"""The stage-4 step of the reference (src/bin/train_gru_cyclevae_gauss_batch.py:1326-1420) on the drop-in modules:
n_cyc reconversion chain in train mode, loss, backward, (gradient all-reduce,) Adam.

The reference open-codes this in its training script; here it is one function over an abstract `run_pass` so the same
code drives the HIP modules on the GPU and the stock-torch checker on the CPU (tests), and bench.py's train leg.
`chain_forward` + `loss_terms` (= `chain_loss`) are plain torch and run anywhere; `Stage4Step` is the GPU step: by default the
draw + concatenation that builds a decoder input, the loss with its gradients and Adam each run as ONE library launch
(cvae_sample_cat, cvae_stage4_loss, cvae_adam_step) instead of a few dozen element-wise torch kernels.
"""
import torch

K_MCD_L1 = (10.0 / 2.3025850929940456840179914546844) * 1.4142135623730950488016887242097   # gru_vae.py:525

TRAINABLE = ("conv.conv.0.weight", "conv.conv.0.bias", "conv.conv.1.weight", "conv.conv.1.bias", "gru.weight_ih_l0",
             "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0", "out_1.weight", "out_1.bias")   # train...:373-376


PASS_SLOTS = ("lat", "rec", "cv", "latcv", "reccyc")     # the five passes of a cycle, in the reference's order (train...:1328-1338)


def torch_dec_input(lat, codes, eps, lat_dim, cycle):
    """[code ; sampling_vae_batch(lat)] for one decoder pass, or for several stacked along the batch axis (gru_vae.py:96 + the
    torch.cat of train...:1335-1338); eps: one [B,T,L] tensor per part, or None to draw on the device."""
    L = lat_dim

    def draw(e):
        if e is None:           # no eps supplied: the module's own sampling_vae_batch, as the script calls it (on-device Philox)
            import gru_vae
            return gru_vae.sampling_vae_batch(lat, lat_dim=L)
        return lat[:, :, :L] + torch.exp(lat[:, :, L:] / 2) * e

    parts = [torch.cat((c, draw(e)), 2) for c, e in zip(codes, eps)]
    return parts[0] if len(parts) == 1 else torch.cat(parts, 0)


class _SplitRows(torch.autograd.Function):
    """out[:B], out[B:] of the stacked rec || cv decoder pass.  Plain slicing costs the backward two zero-filled [2B,T,C] tensors, two
    copies and an add (five launches per cycle); here the backward is one concatenation."""

    @staticmethod
    def forward(ctx, out, B):
        ctx.set_materialize_grads(False)
        ctx.shape, ctx.B = tuple(out.shape), B
        return out[:B], out[B:]           # (views: nothing is copied)

    @staticmethod
    def backward(ctx, da, db):
        if da is None and db is None:
            return None, None
        ref = da if da is not None else db
        if da is None:
            da = ref.new_zeros((ctx.B,) + ctx.shape[1:])
        if db is None:
            db = ref.new_zeros((ctx.shape[0] - ctx.B,) + ctx.shape[1:])
        return torch.cat((da, db), 0), None


def chain_forward(run_pass, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, lat_dim, n_cyc=2, masks=None, carry=None,
                  stack_rec_cv=False, dec_input=torch_dec_input):
    """The passes of one frame window (train...:1299-1338).  Returns (trajs, state): trajs[i] = {"lat", "rec", "cv", "latcv",
    "reccyc"} of cycle i, state = {(cycle, slot): (y_last, h_last)} when run_pass returns them.

    run_pass(kind, x[B,T,C], y_in, clamp_lat_dim, mask_pair_or_None[, h_in]) -> trj_out or (trj_out, y_last, h_last).
    eps [n_cyc,3,B,T,L] in draw order (rec, cv, rec_cyc), or None when dec_input draws on its own.
    carry: None for a fresh window (y_in_* are the initial feedbacks, h = 0), or the state of the previous window of the same
    utterances (:1299-1311: every pass continues from its own detached state).
    stack_rec_cv: the two decoder passes of :1335-1336 share weights and do not depend on each other, so they run as ONE call on
    2B rows (rows [0,B) = rec, [B,2B) = cv; run_pass is told kind "dec2")."""
    L, stdim = lat_dim, cvx.shape[2]
    B = x.shape[0]
    ie = idc = 0
    prev = y2 = None
    state, trajs = {}, []
    mk = lambda kind, i: None if masks is None else masks[kind][i]
    ep = lambda i, k: None if eps is None else eps[i, k]

    def one(kind, slot, i, xin, y0, clamp, mask):
        if carry is not None:
            y0, h0 = carry[(i, slot)]
            out = run_pass(kind, xin, y0.detach(), clamp, mask, h0.detach())
        else:
            out = run_pass(kind, xin, y0, clamp, mask)
        if isinstance(out, tuple):
            state[(i, slot)] = (out[1], out[2])
            return out[0]
        return out

    for i in range(n_cyc):
        e_in = x if i == 0 else torch.cat((x[:, :, :stdim], prev), 2)
        lat = one("enc", "lat", i, e_in, y_in_enc, L, mk("enc", ie)); ie += 1
        if stack_rec_cv:
            xin = dec_input(lat, (code_src, code_trg), (ep(i, 0), ep(i, 1)), L, (i, 0))
            ma, mb = mk("dec", idc), mk("dec", idc + 1)     # (conv mask [B,T,9C], gru mask [T,B,H]) per pass
            cat = lambda a, b, d: torch.cat((a, b), d) if torch.is_tensor(a) else __import__("numpy").concatenate((a, b), d)
            m2 = None if ma is None else (cat(ma[0], mb[0], 0), cat(ma[1], mb[1], 1))
            if carry is not None:
                (ya, ha), (yb, hb) = carry[(i, "rec")], carry[(i, "cv")]
                out = run_pass("dec2", xin, torch.cat((ya, yb), 0).detach(), -1, m2, torch.cat((ha, hb), 1).detach())
            else:
                y2 = torch.cat((y_in_dec, y_in_dec), 0) if y2 is None else y2      # (once per window, not per cycle)
                out = run_pass("dec2", xin, y2, -1, m2)
            if isinstance(out, tuple):
                state[(i, "rec")] = (out[1][:B], out[2][:, :B])
                state[(i, "cv")] = (out[1][B:], out[2][:, B:])
                out = out[0]
            rec, cv = _SplitRows.apply(out, B) if torch.is_tensor(out) and out.requires_grad else (out[:B], out[B:])
            idc += 2
        else:
            rec = one("dec", "rec", i, dec_input(lat, (code_src,), (ep(i, 0),), L, (i, 0)), y_in_dec, -1, mk("dec", idc)); idc += 1
            cv = one("dec", "cv", i, dec_input(lat, (code_trg,), (ep(i, 1),), L, (i, 1)), y_in_dec, -1, mk("dec", idc)); idc += 1
        latcv = one("enc", "latcv", i, torch.cat((cvx, cv), 2), y_in_enc, L, mk("enc", ie)); ie += 1
        reccyc = one("dec", "reccyc", i, dec_input(latcv, (code_src,), (ep(i, 2),), L, (i, 2)), y_in_dec, -1, mk("dec", idc)); idc += 1
        prev = reccyc
        trajs.append({"lat": lat, "rec": rec, "cv": cv, "latcv": latcv, "reccyc": reccyc})
    return trajs, state


def frame_weights(B, T, flen_acc=None, select_utt_idx=None):
    """What the reference's per-utterance slicing amounts to (train...:1363-1410), as three host lists: w[j][t] = 1/n_j for the
    first n_j = min(flen_acc[j], T) frames of a selected utterance, else 0 (mean over frames, summed over utterances);
    last[j] = 1 for the LAST selected utterance; n_sel.  Defaults = every utterance, whole window."""
    sel = list(range(B)) if select_utt_idx is None else [int(j) for j in select_utt_idx]
    nfr = [T] * B if flen_acc is None else [min(int(n), T) for n in flen_acc]
    w = [[0.0] * T for _ in range(B)]
    for j in sel:
        n = nfr[j]
        if n > 0:
            w[j][:n] = [1.0 / n] * n
    last = [0.0] * B
    if sel:
        last[sel[-1]] = 1.0
    return w, last, len(sel)


def loss_terms(trajs, x, stdim, lat_dim, flen_acc=None, select_utt_idx=None, half_cyc=False):
    """Batch loss of one frame window from the trajectories of chain_forward, as the reference computes it (:1363-1410).

    Utterance j of select_utt_idx contributes the first n = flen_acc[j] frames of the window (Python slice clipping: at most T); the
    others are computed and ignored.  Per utterance and cycle: mean-over-frames L1 mel-cd of rec and of rec_cyc against
    x[..., stdim:], KL of lat and of latcv (cv against the source is only logged).  The per-utterance terms are SUMMED over
    utterances and cycles -- with the reference's quirk at :1393 kept: for more than one selected utterance the `lat_src_cv` list is
    rebuilt from the `lat_src` list, so its sum is (sum of KL(lat) over the utterances) + KL(latcv of the LAST utterance) (SURVEY
    App. C.2; harmless at the recipe's batch_size_utt = 1).  half_cyc (n_cyc < 1 in the reference, :283-287) drops the rec_cyc /
    latcv terms."""
    L = lat_dim
    B, T = x.shape[0], x.shape[1]
    wl, lastl, n_sel = frame_weights(B, T, flen_acc, select_utt_idx)
    w = torch.tensor(wl, dtype=torch.float32, device=x.device)
    last = torch.tensor(lastl, dtype=torch.float32, device=x.device)
    tgt = x[:, :, stdim:]
    mcd = lambda trj: (K_MCD_L1 * (trj - tgt).abs().sum(2) * w).sum()                          # gru_vae.py:525-527
    kl_rows = lambda par: ((0.5 * (par[:, :, L:].exp() + par[:, :, :L] ** 2 - par[:, :, L:] - 1.0).sum(2)) * w).sum(1)   # :123
    loss = 0.0
    for tr in trajs:
        kl_lat = kl_rows(tr["lat"])
        loss = loss + mcd(tr["rec"]) + kl_lat.sum()
        if not half_cyc and n_sel:
            loss = loss + mcd(tr["reccyc"])
            if n_sel > 1:          # :1393: [KL(lat) of every utterance ..., KL(latcv) of the LAST selected one]
                loss = loss + kl_lat.sum()
            loss = loss + (kl_rows(tr["latcv"]) * last).sum()
    return loss


def script_loss_loop(trajs, x, stdim, lat_dim, flen_acc=None, select_utt_idx=None, half_cyc=False, log=None):
    """The same batch loss the way the UNCHANGED training script forms it (train...:1363-1410): a Python loop over the selected
    utterances that slices each one's frames, calls the module's TWFSEloss / loss_vae on them, reads five scalars back to the host
    for the log (`.item()`, a device synchronisation each) and concatenates the per-utterance terms.  loss_terms is the vectorised
    equivalent; this form exists so that bench.py can time the flow a user gets by changing nothing but path.sh:11.
    log: optional list that receives the scalars the script logs."""
    import gru_vae
    crit = gru_vae.TWFSEloss()
    B, T = x.shape[0], x.shape[1]
    sel = list(range(B)) if select_utt_idx is None else [int(j) for j in select_utt_idx]
    nfr = [T] * B if flen_acc is None else [int(n) for n in flen_acc]
    total = None
    for tr in trajs:
        terms = {"rec": [], "reccyc": [], "lat": [], "latcv": []}
        for k, j in enumerate(sel):
            n = nfr[j]
            tgt = x[j, :n, stdim:]
            m_rec = crit(tr["rec"][j, :n], tgt, L2=False, GV=False)[1]
            m_cyc = crit(tr["reccyc"][j, :n], tgt, L2=False, GV=False)[1]
            m_cv = crit(tr["cv"][j, :n], tgt, L2=False, GV=False)[1]
            kl, kl_cv = gru_vae.loss_vae(tr["lat"][j, :n], lat_dim=lat_dim), gru_vae.loss_vae(tr["latcv"][j, :n], lat_dim=lat_dim)
            scalars = [v.item() for v in (m_rec, m_cyc, m_cv, kl_cv, kl)]          # the script's per-utterance log entries
            if log is not None:
                log.append(scalars)
            terms["rec"].append(m_rec)
            terms["reccyc"].append(m_cyc)
            if k > 0:                     # :1393: the list is rebuilt from the `lat` list
                terms["latcv"] = list(terms["lat"]) + [kl, kl_cv]
            else:
                terms["latcv"] = [kl_cv]
            terms["lat"].append(kl)
        tot = lambda name: torch.stack(terms[name]).sum()
        cyc = tot("rec") + tot("lat")
        if not half_cyc:
            cyc = cyc + tot("reccyc") + tot("latcv")
        total = cyc if total is None else total + cyc
    return total


def chain_loss(run_pass, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, lat_dim, n_cyc=2, masks=None,
               flen_acc=None, select_utt_idx=None, half_cyc=False, carry=None, return_state=False, stack_rec_cv=False):
    """chain_forward + loss_terms: the batch loss of one frame window (train...:1299-1338 forward, :1363-1410 loss).
    x is the WINDOW of the padded utterance batch (batch_src[:, s:e+1]); flen_acc / select_utt_idx: the generator's bookkeeping
    (windows.plan_windows, reference train...:45-149).  return_state=True additionally returns the state dict for the next window
    and the five trajectories per cycle."""
    trajs, state = chain_forward(run_pass, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, lat_dim, n_cyc, masks, carry,
                                 stack_rec_cv)
    loss = loss_terms(trajs, x, cvx.shape[2], lat_dim, flen_acc, select_utt_idx, half_cyc)
    if return_state:
        return loss, state, trajs
    return loss


def freeze_scalers(*modules):
    """scale_in / scale_out are statistics, not trained (train...:365-372)."""
    for m in modules:
        for n, p in m.named_parameters():
            p.requires_grad_(n in TRAINABLE)


class _SampleCat(torch.autograd.Function):
    """[code ; mu + exp(s/2) eps] for one decoder pass or two stacked ones, one launch forward and one backward."""

    @staticmethod
    def forward(ctx, lat, codes, eps, seed, draws):
        import gru_vae
        lib = gru_vae._lib()
        B, T, L2 = lat.shape
        L, parts, ncode = L2 // 2, len(codes), codes[0].shape[2]
        latc = lat.contiguous()
        cs = [c.to(torch.float32).contiguous() for c in codes]
        es = [None if e is None else e.to(torch.float32).contiguous() for e in eps]
        out = torch.empty(parts * B, T, ncode + L, dtype=torch.float32, device=lat.device)
        eps_used = torch.empty(parts, B, T, L, dtype=torch.float32, device=lat.device)
        lib.sample_cat(latc.data_ptr(), [c.data_ptr() for c in cs], [None if e is None else e.data_ptr() for e in es], seed, draws,
                       B, T, L, ncode, out.data_ptr(), eps_used.data_ptr(), gru_vae._stream())
        ctx.save_for_backward(latc, eps_used)
        ctx.dims = (B, T, L, ncode, parts)
        return out

    @staticmethod
    def backward(ctx, dout):
        import gru_vae
        latc, eps_used = ctx.saved_tensors
        B, T, L, ncode, parts = ctx.dims
        dlat = torch.empty_like(latc)
        gru_vae._lib().sample_cat_backward(dout.contiguous().data_ptr(), latc.data_ptr(), eps_used.data_ptr(), B, T, L, ncode, parts,
                                           dlat.data_ptr(), gru_vae._stream())
        return dlat, None, None, None, None


class Stage4Step(object):
    """zero_grad -> chain (train mode) -> loss.backward() -> [all-reduce] -> optimizer.step()   (train...:1418-1420).

    fused=True (default on the GPU): decoder inputs through cvae_sample_cat, the loss and its gradients through cvae_stage4_loss
    (the backward starts from those gradients: no scalar-loss graph), Adam as cvae_adam_step_counted over ONE flat parameter buffer
    (the parameters become views of it) gated ON THE DEVICE by the step's status word, so a step whose kernels reported a failed
    hand-off or a range overflow never touches parameters, moments or the step counter (which lives on the device as well).
    fused=False keeps torch ops and torch.optim.Adam (the drop-in flow).

    The status word: the kernels report into the pinned host sink (cvae_set_status_sink); at the end of every step ONE
    stream-ordered launch moves it into the device word `status_dev` (cvae_status_latch: latch = max(latch, sink), sink = 0), which
    is MAX-reduced over the ranks when data-parallel, gates the update and is copied to a pinned slot behind an event.  The host
    never writes the sink while steps are in flight; the latch stays raised until the host has SEEN the code and cleared it with a
    stream-ordered memset.

    sync=True: after the optimiser step the call waits for that event and reads the slot (the reference synchronises every
    step as well: it logs `batch_loss.item()`), so an error is raised by the step that had it.  Status 5 -- a gate gradient of the
    reverse recurrence outside the range of its limb exchange -- is not an error: the step is repeated with the fp32 reverse
    recurrence (same draws: the generator state is rewound), and only that result is applied.  sync=False never waits: see
    _lagged_check (the host then enqueues step k+1 while the device runs step k; measured on one MI355X: 25.3 vs 25.2 ms at B = 64, 5.4 vs
    5.3 ms at one utterance -- the device is the bottleneck either way, the mode exists for loops that must not block)."""

    def __init__(self, enc, dec, lat_dim, n_cyc=2, lr=1e-4, dist=None, stack_rec_cv=True, overlap_wgrad=True, fused=None,
                 betas=(0.9, 0.999), eps=1e-8, sync=True, force_collectives=False, script_loss=False):
        """stack_rec_cv: rec || cv as one decoder launch (chain_forward).  overlap_wgrad: parameter gradients accumulate straight
        into the flat gradient buffer and the recurrent weight-gradient GEMMs of every backward pass run on a second stream, under
        the next pass's reverse recurrence (gru_vae.set_side_stream); joined before the all-reduce / optimizer step.
        force_collectives: issue the gradient all-reduce and the status MAX-reduce even in a process group of ONE rank (bench.py
        --force-dist: the RCCL path executes on a one-GPU box).  script_loss (fused=False only): form the loss with the training
        script's own per-utterance loop and its host read-backs (script_loss_loop) instead of the vectorised loss_terms."""
        import shard
        self.script_loss = bool(script_loss)
        self.mods = {"enc": enc, "dec": dec}
        on_gpu = next(enc.parameters()).is_cuda
        self.overlap_wgrad = overlap_wgrad and on_gpu
        self.fused = on_gpu if fused is None else (fused and on_gpu)
        self.side = None
        self.lat_dim, self.n_cyc, self.dist, self.stack_rec_cv = lat_dim, n_cyc, dist, stack_rec_cv
        self.force_collectives = bool(force_collectives)
        self.lr, self.betas, self.eps, self.sync = lr, betas, eps, sync
        freeze_scalers(enc, dec)
        self.params = [p for m in (enc, dec) for p in m.parameters() if p.requires_grad]
        self.grads = shard.FlatGradients(self.params)     # p.grad = views of one flat buffer: the all-reduce needs no copies
        self.opt = None
        self._step_host = 0
        if self.fused:
            n = self.grads.flat.numel()
            self.flat_p = torch.empty(n, dtype=torch.float32, device=self.grads.flat.device)
            o = 0
            for p in self.params:                         # parameters become views of one flat buffer: Adam is one launch
                self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_p[o:o + p.numel()].view_as(p)
                o += p.numel()
            self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat_p), torch.zeros_like(self.flat_p)
            # the latch the update kernel is gated by and Adam's step counter, both in DEVICE memory (class docstring)
            self.status_dev = torch.zeros(4, dtype=torch.int32, device=self.flat_p.device)
            # data-parallel: the MAX over the ranks' latches goes into a SEPARATE word, which gates the update and is what the host
            # reads; the local latch is only ever raised by this rank's kernels and cleared by this rank's host (reducing
            # the latch in place let a rank that had already cleared get its latch re-raised by a rank that had not)
            self.status_red = torch.zeros(4, dtype=torch.int32, device=self.flat_p.device)
            self.step_state = torch.zeros(4, dtype=torch.int32, device=self.flat_p.device)
            self._slots = [torch.zeros(4, dtype=torch.int32).pin_memory() for _ in range(self.MAX_IN_FLIGHT + 2)] if on_gpu else []
            self._slot_i = 0
            self._pending = []                            # sync=False: [(event, pinned slot)] of the steps in flight, oldest first
            self._last = None
            self._wcache = {}
        else:
            self.opt = torch.optim.Adam(self.params, lr=lr, betas=betas, eps=eps)
        self.allreduce_ms = []                            # per step, when time_allreduce is set (bench.py train leg)
        self.time_allreduce = False
        self.fallbacks = 0                                # steps repeated with the fp32 reverse recurrence (status 5)
        self.skipped = 0                                  # sync=False: steps whose update the device skipped
        self.coop_fallback = False                        # a hand-off time-out was seen: the all-resident kernels launch cooperatively
        self.prep_dec_on_side = True
        self._timing_skip_prep = False
        self._fp32_left = 0
        self._incident = 0                                # sync=False: steps of the current run of raised status words seen so far
        self._incident_left = 0                           # ... and how many more raised words may still belong to it (_drain)
        self.verbose_incidents = False                    # True: print every step skipped as the continuation of a handled incident
        self._owns_status = False
        self._saved_bwd_per_step = 0
        self.last_trajs = None

    MAX_IN_FLIGHT = 6          # sync=False: the host waits for the oldest step when this many are queued

    @property
    def step_no(self):
        """Number of updates applied so far (torch.optim.Adam's `step`).  Fused: read from the device counter (synchronises)."""
        if self.fused:
            return int(self.step_state[0].item())
        return self._step_host

    @step_no.setter
    def step_no(self, v):
        self._step_host = int(v)
        if self.fused:
            self.step_state[0] = int(v)

    def _collective(self):
        d = self.dist
        return d is not None and d.is_initialized() and (d.get_world_size() > 1 or self.force_collectives)

    # -- passes -------------------------------------------------------------------------------------------------------------------
    def _run(self, kind, x, y_in, clamp, masks, h_in=None):
        import gru_vae
        parts = 2 if kind == "dec2" else 1
        m = self.mods["dec" if parts == 2 else kind]
        if masks is not None:
            m._debug_masks = masks
        if parts > 1:
            gru_vae.set_draw_parts(parts)
        try:
            out = m(x, y_in, do=True, clamp_vae=clamp >= 0, lat_dim=self.lat_dim, h_in=h_in)
        finally:
            if parts > 1:
                gru_vae.set_draw_parts(1)
        return out if self._want_state else out[0]

    def _dec_input(self, lat, codes, eps, lat_dim, cycle):
        i, k = cycle
        return _SampleCat.apply(lat, codes, eps, self._seed, [i * 3 + k + q for q in range(len(codes))])

    def _weights(self, B, T, flen_acc, select_utt_idx, dev):
        key = (B, T, None if flen_acc is None else tuple(int(v) for v in flen_acc),
               None if select_utt_idx is None else tuple(int(v) for v in select_utt_idx))
        if key not in self._wcache:
            if len(self._wcache) > 64:
                self._wcache.clear()
            w, last, n_sel = frame_weights(B, T, flen_acc, select_utt_idx)
            self._wcache[key] = (torch.tensor(w, dtype=torch.float32, device=dev), torch.tensor(last, dtype=torch.float32, device=dev), n_sel)
        return self._wcache[key]

    def _fused_loss_backward(self, trajs, x, stdim, flen_acc, select_utt_idx, half_cyc):
        """cvae_stage4_loss per cycle, then ONE autograd sweep started from the trajectories' gradients."""
        import gru_vae
        lib = gru_vae._lib()
        B, T = x.shape[0], x.shape[1]
        L, D = self.lat_dim, trajs[0]["rec"].shape[2]
        dev = x.device
        w, last, n_sel = self._weights(B, T, flen_acc, select_utt_idx, dev)
        full = (not half_cyc) and n_sel > 0
        kl_scale = 2.0 if (full and n_sel > 1) else 1.0
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        frame_loss = torch.empty(B * T, dtype=torch.float32, device=dev)
        xc = x.contiguous()
        outs, grads = [], []
        for i, tr in enumerate(trajs):
            rec, lat = tr["rec"].contiguous(), tr["lat"].contiguous()
            d_rec, d_lat = torch.empty_like(rec), torch.empty_like(lat)
            reccyc = latcv = d_reccyc = d_latcv = None
            if full:
                reccyc, latcv = tr["reccyc"].contiguous(), tr["latcv"].contiguous()
                d_reccyc, d_latcv = torch.empty_like(reccyc), torch.empty_like(latcv)
            p = lambda t_: None if t_ is None else t_.data_ptr()
            lib.stage4_loss(p(rec), p(reccyc), p(lat), p(latcv), xc.data_ptr(), xc.shape[2], stdim, w.data_ptr(), last.data_ptr(), kl_scale,
                            B, T, D, L, p(d_rec), p(d_reccyc), p(d_lat), p(d_latcv), frame_loss.data_ptr(), loss.data_ptr(), i > 0,
                            gru_vae._stream())
            outs += [tr["rec"], tr["lat"]]
            grads += [d_rec, d_lat]
            if full:
                outs += [tr["reccyc"], tr["latcv"]]
                grads += [d_reccyc, d_latcv]
        torch.autograd.backward(outs, grads)
        return loss[0]

    # -- the step -----------------------------------------------------------------------------------------------------------------
    def _forward_backward(self, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, masks, flen_acc, select_utt_idx, carry,
                          half_cyc):
        import gru_vae
        self.grads.zero()
        self._seed = gru_vae._draw_seed() if eps is None else 0     # (one draw from torch's generator per step, only when it is used)
        # two or three utterances: rec and cv as separate passes stay on the three-row word-exchange recurrences (cvae_train_ll.h),
        # stacked they would be 4 / 6 rows on the tile kernels, at twice the time per step
        stack = self.stack_rec_cv and not (2 <= x.shape[0] <= 3)
        if self.overlap_wgrad:
            # the side stream serves the whole step: in the forward passes the library draws the recurrence's dropout mask on it,
            # beside the front-end GEMMs (option masks_on_side); in the backward passes the weight-gradient GEMMs run there
            if self.side is None:
                self.side = gru_vae.concurrent_stream()       # (probed: not every stream runs beside the launch stream)
            gru_vae.set_side_stream(self.side)
        try:
            loss, trajs, state = self._chain_forward_loss_backward(x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, masks, flen_acc,
                                                                   select_utt_idx, carry, half_cyc, stack)
        finally:
            if self.overlap_wgrad:
                gru_vae.join_side_stream()
                gru_vae.set_side_stream(None)
                for m in self.mods.values():
                    m._grad_sink = False
        return loss.detach(), state, trajs

    def _chain_forward_loss_backward(self, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, masks, flen_acc, select_utt_idx, carry,
                                     half_cyc, stack):
        import gru_vae
        trajs, state = chain_forward(self._run, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, self.lat_dim, self.n_cyc, masks,
                                     carry, stack, self._dec_input if self.fused else torch_dec_input)
        loss = None
        if not self.fused:
            loss = (script_loss_loop if self.script_loss else loss_terms)(trajs, x, cvx.shape[2], self.lat_dim, flen_acc,
                                                                          select_utt_idx, half_cyc)
        if self.overlap_wgrad:
            for m in self.mods.values():
                m._grad_sink = True
        if self.fused:
            loss = self._fused_loss_backward(trajs, x, cvx.shape[2], flen_acc, select_utt_idx, half_cyc)
        else:
            loss.backward()
        return loss, trajs, state

    def _reduce_and_update(self):
        import gru_vae
        force = self.force_collectives
        if self.time_allreduce and self._collective():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.grads.allreduce(self.dist, force)
            e1.record()
            self.allreduce_ms.append((e0, e1))
        else:
            self.grads.allreduce(self.dist, force)
        if not self.fused:
            gru_vae.check_status(sync=True)      # never step on gradients of a pass that reported a failed hand-off
            self.opt.step()
            self._step_host += 1
            return
        lib = gru_vae._lib()
        gate = None
        if gru_vae._sink() is not None:
            lib.status_latch(self.status_dev.data_ptr(), gru_vae._stream())     # stream-ordered: the sink after this step's kernels
            gate = self.status_dev
            if self._collective():
                self.status_red.copy_(self.status_dev)
                self.dist.all_reduce(self.status_red, op=self.dist.ReduceOp.MAX)
                gate = self.status_red
        if self.params[0].data_ptr() != self.flat_p.data_ptr() or \
                self.params[-1].data_ptr() != self.flat_p.data_ptr() + 4 * (self.flat_p.numel() - self.params[-1].numel()):
            raise RuntimeError("the modules' parameters are no longer views of Stage4Step's flat buffer (moved with .to() / .cpu() / "
                               "load_state_dict(assign=True) after the step was built?): build a new Stage4Step")
        lib.adam_step_counted(self.flat_p.data_ptr(), self.grads.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                              self.flat_p.numel(), self.lr, self.betas[0], self.betas[1], self.eps, self.step_state.data_ptr(),
                              gru_vae._stream(), gate=None if gate is None else gate.data_ptr())
        slot = self._slots[self._slot_i % len(self._slots)]
        self._slot_i += 1
        slot.copy_(self.status_dev if gate is None else gate, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()                          # _status waits for THIS, not for the preparation kernels queued below
        self._last = (ev, slot)
        cur = torch.cuda.current_stream()
        for name, m in self.mods.items():
            if self._timing_skip_prep:           # (measurement only: stale weight images, what the re-layout costs per step)
                continue
            m.weights_changed()                  # (the flat buffer was written behind torch's version counters)
            # the next step's weight images right behind the update: the ~30 preparation kernels per net then run while the host waits
            # for this step's status word and launches the next step's first kernels, instead of in front of its first recurrence.
            # The DECODER's image is not needed before the encoder's first pass is through (~1 ms at 64 rows): it is built on the
            # side stream, beside that pass, and the first decoder pass of the next step waits for its event (_run).
            if m.training and m.do_prob > 0:
                if name == "dec" and self.prep_dec_on_side and self.overlap_wgrad and self.side is not None:
                    self.side.wait_stream(cur)
                    with torch.cuda.stream(self.side):
                        _, image = m._prep_train.get(m, self.flat_p.device, float(m.do_prob))
                        ev = torch.cuda.Event()
                        ev.record(self.side)
                    m._prep_train.ready = ev       # (the image's next user waits for it, whoever that is: _PreparedTrain.get)
                    image.record_stream(cur)     # (allocated under the side stream, read by the passes on the launch stream)
                else:
                    m._prep_train.get(m, self.flat_p.device, float(m.do_prob))

    def _status(self):
        """sync=True: waits for the step's update; its status word (MAX over ranks when data-parallel).  A raised latch is cleared
        in stream order -- nothing else of this object is in flight."""
        if self._last is None:
            return 0
        ev, slot = self._last
        ev.synchronize()
        code = int(slot[0])
        if code:
            self.status_dev.zero_()
        return code

    FP32_STEPS_AFTER_OVERFLOW = 200

    def _fp32_reverse(self, on):
        """Switch the process-wide option train_bwd_per_step, keeping what the caller had set (bench.py --train-kernel fp32)."""
        import gru_vae
        lib = gru_vae._lib()
        if on:
            self._saved_bwd_per_step = lib.get_option("train_bwd_per_step")
            lib.set_option("train_bwd_per_step", 1)
        else:
            lib.set_option("train_bwd_per_step", self._saved_bwd_per_step)

    def _enable_coop_launch(self):
        import gru_vae
        self.coop_fallback = True
        gru_vae._lib().set_option("coop_launch", 1)

    def _lagged_check(self):
        """sync=False: no host wait per step.  The device skips the update of a step whose status word is raised -- and, because
        the latch stays raised until the host clears it in stream order, of every step enqueued before the host noticed; Adam's
        step counter lives on the device and does not count them.  The host looks at the pinned slots of the steps whose events
        have completed when the NEXT call comes in.  Status 5 (a gate gradient outside the range of the limb exchange): the skipped
        minibatches are lost -- the policy of a gradient scaler on overflow -- and the next FP32_STEPS_AFTER_OVERFLOW steps run the
        fp32 reverse recurrence.  Anything else raises.
        Data-parallel: every rank reads the MAX-reduced word of every step, i.e. all ranks see the SAME sequence of codes by step
        number, and every decision is taken from that sequence alone: an INCIDENT is a maximal run of consecutive steps with a
        non-zero word; it is handled once, at its first step (the same step on every rank).  Ranks notice it at different host
        times and clear their own latches a few steps apart; until the last one has, the reduced word stays raised and every rank
        keeps skipping -- those steps continue the incident, they are not a new one (a rank that had cleared earlier
        read the other rank's still-raised code as a second incident, raised alone and left the others in the next collective)."""
        import gru_vae
        if self._fp32_left > 0:
            self._fp32_left -= 1
            if self._fp32_left == 0:
                self._fp32_reverse(False)
        while len(self._pending) >= self.MAX_IN_FLIGHT:
            self._pending[0][0].synchronize()
            self._drain()
        self._drain()

    INCIDENT_MAX_STEPS = 64     # a run of raised steps longer than this is not "the other ranks have not cleared yet" any more

    def _drain(self):
        import gru_vae
        while self._pending and self._pending[0][0].query():
            _, slot = self._pending.pop(0)
            code = int(slot[0])
            if code == 0:
                self._incident = 0                       # the run of raised steps is over: the next raised word is a new incident
                continue
            if self._incident > 0 and self._incident_left > 0:
                # the incident this rank has already handled continues (another rank's latch is still raised, or was when this step
                # was reduced): the device skipped the step everywhere; clear again in case THIS rank's kernels raised meanwhile
                self._incident += 1
                self._incident_left -= 1
                self.skipped += 1
                self.status_dev.zero_()
                if self.verbose_incidents:
                    print("stage4.Stage4Step: step skipped, continuation %d of the incident (status %d)" % (self._incident - 1, code), flush=True)
                continue
            # a new incident.  The steps already enqueued behind this one stay in the list: each carries its own (reduced) word --
            # raised for as long as some rank's latch was, which on this rank means until the clear below takes effect
            self.skipped += 1
            self._incident = 1
            # how many FOLLOWING raised words still belong to this incident: on one rank only the steps that were already enqueued
            # when the host noticed (their words were latched before the clear below); data-parallel, other ranks notice up to
            # MAX_IN_FLIGHT steps later, hence the larger bound there.  A word raised beyond that is a NEW incident and is handled
            # (a persisting time-out must not eat 64 minibatches in silence)
            self._incident_left = self.INCIDENT_MAX_STEPS if self._collective() else len(self._pending)
            self.status_dev.zero_()                      # stream-ordered: steps enqueued from here on are applied again
            if code == 5:
                self.fallbacks += 1
                if self._fp32_left == 0:
                    self._fp32_reverse(True)
                self._fp32_left = self.FP32_STEPS_AFTER_OVERFLOW
                break
            if not self.coop_fallback:                   # first time-out: cooperative launches from here on, the minibatches are lost
                self._enable_coop_launch()
                break
            self._release_status()
            raise gru_vae._cabi.CvaeError("stage-4 step: a persistent kernel reported status %d (hand-off time-out) in one of the "
                                          "previous steps; their updates were skipped on the device" % code)
        if not self._pending:
            self._release_status()

    def _own_status(self):
        """While steps of a sync=False object are in flight the status word belongs to its device latch: gru_vae.check_status (every
        entry point of the drop-in module calls it: evaluation passes, stage6.convert_*) must not read-and-clear the pinned sink
        from the host between two steps -- that would take the word away from the latch, and the in-flight step's update would be
        applied on gradients of a failed hand-off.  Released when nothing is pending any more (_drain, finish)."""
        import gru_vae
        if not self._owns_status:
            gru_vae._status_owned += 1
            self._owns_status = True

    def _release_status(self):
        import gru_vae
        if self._owns_status:
            gru_vae._status_owned -= 1
            self._owns_status = False

    def finish(self):
        """sync=False: wait for every step in flight and take the decisions their status words call for (a last skipped step raises or
        switches the reverse recurrence exactly as the next call would have).  Returns the number of steps skipped so far."""
        while self.fused and self._pending:
            self._pending[0][0].synchronize()
            self._drain()
        self._release_status()
        return self.skipped

    def __del__(self):
        try:
            self._release_status()
        except Exception:
            pass

    def __call__(self, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, masks=None, flen_acc=None, select_utt_idx=None,
                 carry=None, return_state=False, half_cyc=False):
        """One step on the frame window x [B,T,Cin].  flen_acc / select_utt_idx / carry as in chain_loss (windows after the first
        of an utterance batch pass the previous call's state as `carry`, train...:1299-1311).  Returns the loss (0-dim tensor),
        with return_state=True (loss, state)."""
        import gru_vae
        self._want_state = return_state
        lagged = self.fused and not self.sync and x.is_cuda
        if lagged:
            self._lagged_check()
        args = (x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, masks, flen_acc, select_utt_idx, carry, half_cyc)
        rng = torch.get_rng_state()
        # sync=False: the per-pass host checks of the drop-in module must neither clear nor raise -- the status word is this
        # object's, read through the latch only
        if lagged:
            self._own_status()
        try:
            loss, state, trajs = self._forward_backward(*args)
            self._reduce_and_update()
        except BaseException:
            if lagged and not self._pending:
                self._release_status()
            raise
        if not x.is_cuda:
            return (loss, state) if return_state else loss
        if lagged:
            self._pending.append(self._last)
            self.last_trajs = trajs
            return (loss, state) if return_state else loss
        code = self._status() if self.fused else 0
        for _ in range(2):
            if code == 5:
                # a gate gradient left the range of the limb exchange of k_train_bwd_steps: the device skipped the update; repeat the
                # step with the fp32 reverse recurrence (per-step launches, no range limit) on the same draws
                self.fallbacks += 1
                torch.set_rng_state(rng)
                self._fp32_reverse(True)
                try:
                    loss, state, trajs = self._forward_backward(*args)
                    self._reduce_and_update()
                    code = self._status()
                finally:
                    self._fp32_reverse(False)
            elif code and not self.coop_fallback:
                # a hand-off spin timed out: the grid of an all-resident kernel was most likely not co-resident (kernels of another
                # stream or process held CUs).  From here on those kernels go through hipLaunchCooperativeKernel, which checks
                # residency at EVERY launch (~27 us of idle GPU per launch); the skipped step is repeated once on the same draws
                self._enable_coop_launch()
                torch.set_rng_state(rng)
                loss, state, trajs = self._forward_backward(*args)
                self._reduce_and_update()
                code = self._status()
            else:
                break
        if code:
            raise gru_vae._cabi.CvaeError("stage-4 step: a persistent kernel reported status %d (hand-off time-out); the update was "
                                          "skipped on the device, parameters and optimiser state are those of the previous step" % code)
        self.last_trajs = trajs
        return (loss, state) if return_state else loss
