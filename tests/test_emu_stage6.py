"""Stage-6 post-processing entry points (cvae_gv_postfilter, cvae_mcd_aligned) of the REAL library on the host-fiber emulator
against the float64 numpy restatement in oracle/ (SURVEY 8(f) rows 1-2).  Tolerance: f64 sums in another order, 1e-12 relative."""
import numpy as np
import pytest

import synth
from emu_util import emu_lib, ptr
from oracle import cyclevae_oracle as orc


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _case(T, D, tag):
    c = (synth.normal(tag + "/c", (T, D)) * np.linspace(2.0, 0.1, D)).astype(np.float32)
    gv = (0.05 + synth.uniform01(tag + "/gv", (D - 1,))).astype(np.float64)
    cg = (0.02 + 0.5 * synth.uniform01(tag + "/cg", (D - 1,))).astype(np.float64)
    dp = (0.1 * synth.normal(tag + "/dp", (T,))).astype(np.float64)
    return c, gv, cg, dp


@pytest.mark.parametrize("T,D,use_dpow", [(1, 2, False), (205, 50, True), (637, 50, False), (1501, 25, True)])
def test_gv_postfilter(lib, T, D, use_dpow):
    c, gv, cg, dp = _case(T, D, "gv%d" % T)
    out, var, work = np.full((T, D), np.nan), np.full(D - 1, np.nan), np.zeros(2 * D)
    lib.gv_postfilter(ptr(c), T, D, ptr(dp) if use_dpow else 0, ptr(gv), ptr(cg), ptr(out), ptr(var), ptr(work))
    r_out, r_var = orc.gv_postfilter(c, gv, cg, dp if use_dpow else None)
    np.testing.assert_allclose(out, r_out, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(var, r_var, rtol=1e-11, atol=1e-14)
    if T > 1:   # the filter sets the global variance to gv_trg / cvgv times what it was and keeps the mean
        np.testing.assert_allclose(var, np.var(c.astype(np.float64)[:, 1:], 0) * gv / cg, rtol=1e-10)
        np.testing.assert_allclose(out[:, 1:].mean(0), c.astype(np.float64)[:, 1:].mean(0), rtol=0, atol=1e-12)


@pytest.mark.parametrize("rows,D,d0,l2", [(1, 50, 0, True), (300, 50, 1, True), (1501, 50, 0, True), (77, 50, 1, False)])
def test_mcd_aligned(lib, rows, D, d0, l2):
    a = synth.normal("mcd%d/a" % rows, (rows, D)).astype(np.float32)
    b = (a + 0.05 * synth.normal("mcd%d/b" % rows, (rows, D))).astype(np.float32)
    frames, stats = np.full(rows, np.nan), np.full(4, np.nan)
    lib.mcd_aligned(ptr(a), D, ptr(b), D, rows, D, d0, l2, ptr(frames), ptr(stats))
    r_frames, r_mean, r_std = orc.mcd_aligned(a, b, d0, l2)
    np.testing.assert_allclose(frames, r_frames, rtol=1e-13)
    assert abs(stats[0] - r_frames.sum()) <= 1e-11 * max(1.0, abs(r_frames.sum()))
    assert abs(stats[1] - r_mean) <= 1e-12 * max(1.0, r_mean) and abs(stats[2] - r_std) <= 1e-10 * max(1.0, r_std)
    if rows > 1:
        assert abs(stats[3] - np.std(r_frames, ddof=1)) <= 1e-10
    # against the fp32 formula pinned to the reference's TWFSEloss (tests/golden/tiny_ops.npz): same values to fp32 rounding
    np.testing.assert_allclose(frames, orc.mcd_frames(a[:, d0:], b[:, d0:], L2=l2), rtol=2e-5)


def test_bad_arguments(lib):
    c = np.zeros((4, 3), np.float32)
    with pytest.raises(Exception):
        lib.gv_postfilter(ptr(c), 4, 1, 0, ptr(np.ones(1)), ptr(np.ones(1)), ptr(np.zeros((4, 1))), 0, ptr(np.zeros(4)))
    with pytest.raises(Exception):
        lib.mcd_aligned(ptr(c), 3, ptr(c), 3, 4, 3, 3, True, ptr(np.zeros(4)), 0)


def _run_dtw(lib, a, b, mcd):
    T1, T2, D = a.shape[0], b.shape[0], a.shape[1]
    aligned, twf = np.full((T2, D), np.nan), np.full(T2, -7, np.int64)
    frames, mean = np.full(T2, np.nan), np.full(1, np.nan)
    work = np.zeros(lib.dtw_work_bytes(T1, T2) // 8)
    lib.dtw_org_to_trg(ptr(a), ptr(b), T1, T2, D, mcd, ptr(aligned), ptr(twf), ptr(frames), ptr(mean), ptr(work), work.nbytes)
    return aligned, twf, float(mean[0]), frames


@pytest.mark.parametrize("T1,T2,mcd", [(9, 13, -1), (14, 8, -1), (11, 11, 0), (1, 5, -1), (6, 1, 0)])
def test_dtw_matches_the_restated_algorithm(lib, T1, T2, mcd):
    """cvae_dtw_org_to_trg against oracle.dtw_org_to_trg (PARITY UNPINNED: dtw_c's source is not in the reference tree; the oracle
    states the algorithm): same path, same warp, costs to 1e-12."""
    a = synth.normal("dtw/a%d_%d" % (T1, T2), (T1, 5)).astype(np.float64)
    b = (0.8 * synth.normal("dtw/b%d_%d" % (T1, T2), (T2, 5)) + 0.1).astype(np.float64)
    aligned, twf, mean, frames = _run_dtw(lib, a, b, mcd)
    ra, rt, rm, rf = orc.dtw_org_to_trg(a, b, mcd=mcd)
    assert np.array_equal(twf, rt) and np.array_equal(aligned, ra)
    assert np.abs(frames - rf).max() <= 1e-12 and abs(mean - rm) <= 1e-12


def test_dtw_recovers_a_known_time_stretch(lib):
    """Properties that define the function whatever dtw_c's details are: identical sequences align on the diagonal at zero cost; a
    sequence whose frames are each held twice warps back onto the original exactly; the warp is monotone and ends at the ends."""
    a = synth.normal("dtw/prop", (12, 6)).astype(np.float64)
    aligned, twf, mean, frames = _run_dtw(lib, a, a, -1)
    assert np.array_equal(twf, np.arange(12)) and mean == 0.0 and np.array_equal(aligned, a)
    held = np.repeat(a, 2, axis=0)                       # 24 frames
    aligned, twf, mean, frames = _run_dtw(lib, held, a, -1)
    assert mean == 0.0 and np.array_equal(aligned, a) and np.array_equal(twf // 2, np.arange(12))
    aligned, twf, mean, frames = _run_dtw(lib, a, held, -1)
    assert mean == 0.0 and np.array_equal(twf, np.repeat(np.arange(12), 2))
    b = a[::-1].copy() + 0.3
    _, twf, _, _ = _run_dtw(lib, a, b, 0)
    assert np.all(np.diff(twf) >= 0) and twf[0] >= 0 and twf[-1] <= 11
