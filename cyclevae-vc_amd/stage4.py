"""The stage-4 step of the reference (src/bin/train_gru_cyclevae_gauss_batch.py:1326-1420) on the drop-in modules:
n_cyc reconversion chain in train mode, loss, backward, (gradient all-reduce,) Adam.

The reference open-codes this in its training script; here it is one function over an abstract `run_pass` so the same
code drives the HIP modules on the GPU and the stock-torch checker on the CPU (tests), and bench.py --mode train.
"""
import torch

K_MCD_L1 = (10.0 / 2.3025850929940456840179914546844) * 1.4142135623730950488016887242097   # gru_vae.py:525

TRAINABLE = ("conv.conv.0.weight", "conv.conv.0.bias", "conv.conv.1.weight", "conv.conv.1.bias", "gru.weight_ih_l0",
             "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0", "out_1.weight", "out_1.bias")   # train...:373-376


PASS_SLOTS = ("lat", "rec", "cv", "latcv", "reccyc")     # the five passes of a cycle, in the reference's order (train...:1328-1338)


def chain_loss(run_pass, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, lat_dim, n_cyc=2, masks=None,
               flen_acc=None, select_utt_idx=None, half_cyc=False, carry=None, return_state=False, stack_rec_cv=False):
    """Batch loss of one frame window, as the reference computes it (train...:1299-1338 forward, :1363-1410 loss).

    run_pass(kind, x[B,T,C], y_in, clamp_lat_dim, mask_pair_or_None[, h_in]) -> trj_out or (trj_out, y_last, h_last).
    eps [n_cyc,3,B,T,L] in draw order (rec, cv, rec_cyc).  x is the WINDOW of the padded utterance batch (batch_src[:, s:e+1]).

    flen_acc / select_utt_idx: the generator's bookkeeping (windows.plan_windows, reference train...:45-149).  Utterance j of
    select_utt_idx contributes the first n = flen_acc[j] frames of the window (Python slice clipping: at most T); the others are
    computed and ignored.  Defaults = every utterance, whole window.
    Per utterance and cycle: mean-over-frames L1 mel-cd of rec and of rec_cyc against x[..., stdim:], KL of lat and of latcv
    (cv against the source is only logged).  The per-utterance terms are SUMMED over utterances and cycles -- with the reference's
    quirk at :1393 kept: for more than one selected utterance the `lat_src_cv` list is rebuilt from the `lat_src` list, so its sum
    is (sum of KL(lat) over the utterances) + KL(latcv of the LAST utterance) (harmless at the recipe's
    batch_size_utt = 1).  half_cyc (n_cyc < 1 in the reference, :283-287) drops the rec_cyc / latcv terms.

    carry: None for a fresh window (y_in_* are the initial feedbacks, h = 0), or {(cycle, slot): (y_last, h_last)} from the
    previous window of the same utterances (:1299-1311: every pass continues from its own detached state).
    return_state=True additionally returns that dict for the next window and the five trajectories per cycle.

    stack_rec_cv: the two decoder passes of :1335-1336 share weights and do not depend on each other, so they run as ONE call
    on 2B rows (rows [0,B) = rec, [B,2B) = cv; run_pass is told kind "dec2"): T dependent steps less per cycle in the forward
    and in the backward recurrence.  Same values; the weight gradients sum the same terms in one contraction instead of two.
    """
    L, stdim = lat_dim, cvx.shape[2]
    B, T = x.shape[0], x.shape[1]
    smp = lambda par, e: par[:, :, :L] + torch.exp(par[:, :, L:] / 2) * e    # gru_vae.py:96
    sel = list(range(B)) if select_utt_idx is None else [int(j) for j in select_utt_idx]
    nfr = [T] * B if flen_acc is None else [min(int(n), T) for n in flen_acc]
    ie = idc = 0
    loss = 0.0
    prev = None
    state, trajs = {}, []
    mk = lambda kind, i: None if masks is None else masks[kind][i]

    def one(kind, slot, i, xin, y0, clamp, mask):
        args = (kind, xin, y0, clamp, mask)
        if carry is not None:
            y0, h0 = carry[(i, slot)]
            out = run_pass(kind, xin, y0.detach(), clamp, mask, h0.detach())
        else:
            out = run_pass(*args)
        if isinstance(out, tuple):
            state[(i, slot)] = (out[1], out[2])
            return out[0]
        return out

    for i in range(n_cyc):
        e_in = x if i == 0 else torch.cat((x[:, :, :stdim], prev), 2)
        lat = one("enc", "lat", i, e_in, y_in_enc, L, mk("enc", ie)); ie += 1
        if stack_rec_cv:
            xin = torch.cat((torch.cat((code_src, smp(lat, eps[i, 0])), 2), torch.cat((code_trg, smp(lat, eps[i, 1])), 2)), 0)
            ma, mb = mk("dec", idc), mk("dec", idc + 1)     # (conv mask [B,T,9C], gru mask [T,B,H]) per pass
            cat = lambda a, b, d: torch.cat((a, b), d) if torch.is_tensor(a) else __import__("numpy").concatenate((a, b), d)
            m2 = None if ma is None else (cat(ma[0], mb[0], 0), cat(ma[1], mb[1], 1))
            if carry is not None:
                (ya, ha), (yb, hb) = carry[(i, "rec")], carry[(i, "cv")]
                out = run_pass("dec2", xin, torch.cat((ya, yb), 0).detach(), -1, m2, torch.cat((ha, hb), 1).detach())
            else:
                out = run_pass("dec2", xin, torch.cat((y_in_dec, y_in_dec), 0), -1, m2)
            if isinstance(out, tuple):
                state[(i, "rec")] = (out[1][:B], out[2][:, :B])
                state[(i, "cv")] = (out[1][B:], out[2][:, B:])
                out = out[0]
            rec, cv = out[:B], out[B:]
            idc += 2
        else:
            rec = one("dec", "rec", i, torch.cat((code_src, smp(lat, eps[i, 0])), 2), y_in_dec, -1, mk("dec", idc)); idc += 1
            cv = one("dec", "cv", i, torch.cat((code_trg, smp(lat, eps[i, 1])), 2), y_in_dec, -1, mk("dec", idc)); idc += 1
        latcv = one("enc", "latcv", i, torch.cat((cvx, cv), 2), y_in_enc, L, mk("enc", ie)); ie += 1
        reccyc = one("dec", "reccyc", i, torch.cat((code_src, smp(latcv, eps[i, 2])), 2), y_in_dec, -1, mk("dec", idc)); idc += 1
        prev = reccyc
        trajs.append({"lat": lat, "rec": rec, "cv": cv, "latcv": latcv, "reccyc": reccyc})
        # per-utterance means over the first n_j frames, vectorised over the batch (a Python loop over utterances costs a dozen
        # tiny launches per row): w[j,t] = 1/n_j for t < n_j of a selected utterance, else 0
        if i == 0:
            nf = torch.tensor(nfr, dtype=torch.float32, device=x.device)
            selm = torch.zeros(B, dtype=torch.float32, device=x.device)
            if sel:
                selm[torch.tensor(sel, dtype=torch.long, device=x.device)] = 1.0
            w = (torch.arange(T, device=x.device)[None, :] < nf[:, None]).to(torch.float32) * (selm / nf.clamp(min=1.0))[:, None]
            last = torch.zeros(B, dtype=torch.float32, device=x.device)
            if sel:
                last[sel[-1]] = 1.0
        tgt = x[:, :, stdim:]
        mcd = lambda trj: (K_MCD_L1 * (trj - tgt).abs().sum(2) * w).sum()                          # gru_vae.py:525-527
        kl_rows = lambda par: ((0.5 * (par[:, :, L:].exp() + par[:, :, :L] ** 2 - par[:, :, L:] - 1.0).sum(2)) * w).sum(1)   # :123
        kl_lat = kl_rows(lat)
        loss = loss + mcd(rec) + kl_lat.sum()
        if not half_cyc and sel:
            loss = loss + mcd(reccyc)
            if len(sel) > 1:          # :1393: [KL(lat) of every utterance ..., KL(latcv) of the LAST selected one]
                loss = loss + kl_lat.sum()
            loss = loss + (kl_rows(latcv) * last).sum()
    if return_state:
        return loss, state, trajs
    return loss


def freeze_scalers(*modules):
    """scale_in / scale_out are statistics, not trained (train...:365-372)."""
    for m in modules:
        for n, p in m.named_parameters():
            p.requires_grad_(n in TRAINABLE)


class Stage4Step(object):
    """zero_grad -> chain (train mode) -> loss.backward() -> [all-reduce] -> optimizer.step()   (train...:1418-1420)."""

    def __init__(self, enc, dec, lat_dim, n_cyc=2, lr=1e-4, dist=None, stack_rec_cv=True, overlap_wgrad=True):
        """stack_rec_cv: rec || cv as one decoder launch (chain_loss).  overlap_wgrad: parameter gradients accumulate straight into
        the flat gradient buffer and the recurrent weight-gradient GEMMs of every backward pass run on a second stream, under the
        next pass's reverse recurrence (gru_vae.set_side_stream); joined before the all-reduce / optimizer step."""
        self.mods = {"enc": enc, "dec": dec}
        self.overlap_wgrad = overlap_wgrad and next(enc.parameters()).is_cuda
        self.side = None
        self.lat_dim, self.n_cyc, self.dist, self.stack_rec_cv = lat_dim, n_cyc, dist, stack_rec_cv
        freeze_scalers(enc, dec)
        self.params = [p for m in (enc, dec) for p in m.parameters() if p.requires_grad]
        self.opt = torch.optim.Adam(self.params, lr=lr)
        import shard
        self.grads = shard.FlatGradients(self.params)     # p.grad = views of one flat buffer: the all-reduce needs no copies
        self.allreduce_ms = []                            # per step, when time_allreduce is set (bench.py --mode train)
        self.time_allreduce = False

    def _run(self, kind, x, y_in, clamp, masks):
        import gru_vae
        parts = 2 if kind == "dec2" else 1
        m = self.mods["dec" if parts == 2 else kind]
        if masks is not None:
            m._debug_masks = masks
        if parts > 1:
            gru_vae.set_draw_parts(parts)
        try:
            return m(x, y_in, do=True, clamp_vae=clamp >= 0, lat_dim=self.lat_dim)[0]
        finally:
            if parts > 1:
                gru_vae.set_draw_parts(1)

    def __call__(self, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, masks=None):
        import gru_vae
        self.grads.zero()
        loss = chain_loss(self._run, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, self.lat_dim, self.n_cyc, masks,
                          stack_rec_cv=self.stack_rec_cv)
        if self.overlap_wgrad:
            if self.side is None:
                self.side = torch.cuda.Stream()
            gru_vae.set_side_stream(self.side)
            for m in self.mods.values():
                m._grad_sink = True
            try:
                loss.backward()
            finally:
                gru_vae.join_side_stream()
                gru_vae.set_side_stream(None)
                for m in self.mods.values():
                    m._grad_sink = False
        else:
            loss.backward()
        if self.time_allreduce and self.dist is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self.grads.allreduce(self.dist)
            e1.record()
            self.allreduce_ms.append((e0, e1))
        else:
            self.grads.allreduce(self.dist)
        import gru_vae
        gru_vae.check_status()      # never step on gradients of a pass that reported a timed-out hand-off
        self.opt.step()
        return loss
