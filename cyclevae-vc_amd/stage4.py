"""The stage-4 step of the reference (src/bin/train_gru_cyclevae_gauss_batch.py:1326-1420) on the drop-in modules:
n_cyc reconversion chain in train mode, loss, backward, (gradient all-reduce,) Adam.

The reference open-codes this in its training script; here it is one function over an abstract `run_pass` so the same
code drives the HIP modules on the GPU and the stock-torch checker on the CPU (tests), and bench.py --mode train.
"""
import torch

K_MCD_L1 = (10.0 / 2.3025850929940456840179914546844) * 1.4142135623730950488016887242097   # gru_vae.py:525

TRAINABLE = ("conv.conv.0.weight", "conv.conv.0.bias", "conv.conv.1.weight", "conv.conv.1.bias", "gru.weight_ih_l0",
             "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0", "out_1.weight", "out_1.bias")   # train...:373-376


def chain_loss(run_pass, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, lat_dim, n_cyc=2, masks=None):
    """Batch loss of one fresh frame window whose utterances all span the window (train...:1326-1338, :1363-1410).

    run_pass(kind, x[B,T,C], y_in, clamp_lat_dim, mask_pair_or_None) -> trj_out;  eps [n_cyc,3,B,T,L] (draw order rec, cv,
    rec_cyc) or None to let `sample` draw them.  Per-utterance terms are means over frames, summed over utterances.
    """
    L, stdim = lat_dim, cvx.shape[2]
    smp = lambda par, e: par[:, :, :L] + torch.exp(par[:, :, L:] / 2) * e    # gru_vae.py:96
    ie = idc = 0
    loss = 0.0
    prev = None
    tgt = x[:, :, stdim:]
    mk = lambda kind, i: None if masks is None else masks[kind][i]
    for i in range(n_cyc):
        e_in = x if i == 0 else torch.cat((x[:, :, :stdim], prev), 2)
        lat = run_pass("enc", e_in, y_in_enc, L, mk("enc", ie)); ie += 1
        rec = run_pass("dec", torch.cat((code_src, smp(lat, eps[i, 0])), 2), y_in_dec, -1, mk("dec", idc)); idc += 1
        cv = run_pass("dec", torch.cat((code_trg, smp(lat, eps[i, 1])), 2), y_in_dec, -1, mk("dec", idc)); idc += 1
        latcv = run_pass("enc", torch.cat((cvx, cv), 2), y_in_enc, L, mk("enc", ie)); ie += 1
        reccyc = run_pass("dec", torch.cat((code_src, smp(latcv, eps[i, 2])), 2), y_in_dec, -1, mk("dec", idc)); idc += 1
        prev = reccyc
        loss = loss + (K_MCD_L1 * (rec - tgt).abs().sum(2)).mean(1).sum() + (K_MCD_L1 * (reccyc - tgt).abs().sum(2)).mean(1).sum()
        for par in (lat, latcv):
            mu, s = par[:, :, :L], par[:, :, L:]
            loss = loss + (0.5 * (s.exp() + mu * mu - s - 1.0).sum(2)).mean(1).sum()      # gru_vae.py:123
    return loss


def freeze_scalers(*modules):
    """scale_in / scale_out are statistics, not trained (train...:365-372)."""
    for m in modules:
        for n, p in m.named_parameters():
            p.requires_grad_(n in TRAINABLE)


class Stage4Step(object):
    """zero_grad -> chain (train mode) -> loss.backward() -> [all-reduce] -> optimizer.step()   (train...:1418-1420)."""

    def __init__(self, enc, dec, lat_dim, n_cyc=2, lr=1e-4, dist=None):
        self.mods = {"enc": enc, "dec": dec}
        self.lat_dim, self.n_cyc, self.dist = lat_dim, n_cyc, dist
        freeze_scalers(enc, dec)
        self.params = [p for m in (enc, dec) for p in m.parameters() if p.requires_grad]
        self.opt = torch.optim.Adam(self.params, lr=lr)

    def _run(self, kind, x, y_in, clamp, masks):
        m = self.mods[kind]
        if masks is not None:
            m._debug_masks = masks
        return m(x, y_in, do=True, clamp_vae=clamp >= 0, lat_dim=self.lat_dim)[0]

    def __call__(self, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, masks=None):
        import shard
        self.opt.zero_grad()
        loss = chain_loss(self._run, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps, self.lat_dim, self.n_cyc, masks)
        loss.backward()
        shard.allreduce_gradients(self.params, self.dist)
        self.opt.step()
        return loss
