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
