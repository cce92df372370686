"""Frame-window bookkeeping of the stage-4 generator, vectorised (INT, bit-exact).

The reference walks every utterance's speech-frame index list with Python loops that compare elements of a *device*
tensor one at a time (src/bin/train_gru_cyclevae_gauss_batch.py:78-99, 108-133): one host<->device sync per comparison.
The same state machine has a closed form per utterance (spc = its ascending speech-frame indices, n of them):

    e_idx[w]   = #{ i < n : spc[i] <= e_w } - 1                                   (never stale: the count is monotone)
    first[w]   = e_idx[w-1] + 1   (0 for the first window)
    visited[w] = (w == 0) or first[w] < n          <=> `spcidcs_src_e_idx[j] < flens_spc_src[j]-1` at :109
    s_idx[w]   = first[w] if spc[first[w]] <= e_w else -1      when visited and first[w] < n, else the previous value
    flen_acc[w]= flens - s_w  when w > 0, visited and e_w >= flens, else the previous value (batch_size initially)
    select[w]  = every utterance for w == 0, the visited ones afterwards

so one `searchsorted` per batch plus two forward-fills replace the loops.  Windows are [s, e] inclusive, batch_size
frames, the last one clipped to max_flen-1 (:71-72, :102-106).  Works on CPU or device int64 tensors, no per-element sync.
"""
import torch


def window_bounds(max_flen, batch_size):
    """[n_win, 2] inclusive (s, e) like train_gru_cyclevae_gauss_batch.py:71-72 and :102-106."""
    s = [0]
    e = [batch_size - 1]
    while e[-1] < max_flen - 1:
        s.append(e[-1] + 1)
        e.append(min(s[-1] + batch_size - 1, max_flen - 1))
    return torch.tensor([s, e], dtype=torch.int64).t().contiguous()


def _ffill(values, valid):
    """Forward-fill along dim 0: out[w] = values[w'] for the last w' <= w with valid[w'] (valid[0] must be all True)."""
    n = values.shape[0]
    idx = torch.arange(n, device=values.device).unsqueeze(1).expand_as(values)
    last = torch.cummax(torch.where(valid, idx, torch.zeros_like(idx)), dim=0).values
    return torch.gather(values, 0, last)


def plan_windows(flens, spcidcs, flens_spc, batch_size=80):
    """All windows of one dataloader batch.

    flens [U] frame counts; spcidcs [U, >=max(flens_spc)] ascending speech-frame indices, zero padded
    (src/utils/dataset.py:23-31,93); flens_spc [U].  Returns a dict of int64 tensors:
      bounds [W,2], s_idx [W,U], e_idx [W,U], flen_acc [W,U], select [W,U] (0/1).
    """
    flens = torch.as_tensor(flens, dtype=torch.int64)
    spc = torch.as_tensor(spcidcs, dtype=torch.int64)
    n = torch.as_tensor(flens_spc, dtype=torch.int64).to(spc.device)
    flens = flens.to(spc.device)
    U = flens.shape[0]
    bounds = window_bounds(int(flens.max().item()), batch_size).to(spc.device)
    W = bounds.shape[0]
    s_w, e_w = bounds[:, 0:1], bounds[:, 1:2]                                  # [W,1]
    # padding must not be counted: push it past every window end
    col = torch.arange(spc.shape[1], device=spc.device).unsqueeze(0)
    big = torch.iinfo(torch.int64).max
    spc_m = torch.where(col < n.unsqueeze(1), spc, torch.full_like(spc, big))  # [U,S]
    cnt = torch.searchsorted(spc_m, e_w.t().expand(U, W).contiguous(), right=True).t()   # [W,U] #{spc <= e_w}
    e_idx = cnt - 1
    first = torch.cat([torch.zeros(1, U, dtype=torch.int64, device=spc.device), cnt[:-1]], 0)  # [W,U]
    has = first < n.unsqueeze(0)                                               # a not-yet-consumed speech frame exists
    w0 = torch.zeros(W, 1, dtype=torch.bool, device=spc.device)
    w0[0] = True
    visited = has | w0
    spc_first = torch.gather(spc_m.t().contiguous(), 0, torch.clamp(first, max=spc.shape[1] - 1))   # [W,U]
    cand = torch.where(spc_first <= e_w, first, torch.full_like(first, -1))
    # first window with no speech index at all keeps the initial -1
    cand = torch.where(has, cand, torch.full_like(cand, -1))
    s_idx = _ffill(cand, has | w0)
    upd = visited & ~w0 & (e_w >= flens.unsqueeze(0))
    fa = torch.where(upd, flens.unsqueeze(0) - s_w, torch.full((W, U), batch_size, dtype=torch.int64, device=spc.device))
    flen_acc = _ffill(fa, upd | w0)
    return {"bounds": bounds, "s_idx": s_idx, "e_idx": e_idx, "flen_acc": flen_acc, "select": visited.to(torch.int64)}


def iter_windows(flens, spcidcs, flens_spc, batch_size=80):
    """Yield (s, e, s_idx[U], e_idx[U], select_utt_idx list, flen_acc[U]) per window as numpy / python ints, i.e. the
    bookkeeping fields the reference's generator yields (train_gru_cyclevae_gauss_batch.py:101,134)."""
    p = {k: v.cpu().numpy() for k, v in plan_windows(flens, spcidcs, flens_spc, batch_size).items()}
    for w in range(p["bounds"].shape[0]):
        sel = [j for j in range(p["select"].shape[1]) if p["select"][w, j]]
        yield int(p["bounds"][w, 0]), int(p["bounds"][w, 1]), p["s_idx"][w], p["e_idx"][w], sel, p["flen_acc"][w]
