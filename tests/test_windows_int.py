"""Bit-exact tests of the vectorised window bookkeeping (cyclevae-vc_amd/windows.py) against the goldens recorded from
the reference's train_generator and, on random ragged batches, against the oracle's restatement of the same loops."""
import numpy as np
import pytest
import torch

import windows
from oracle import cyclevae_oracle as orc
from test_oracle_golden import _int_cases


def pack(spcs, pad=2200):
    spc = np.zeros((len(spcs), pad), np.int64)
    for j, s in enumerate(spcs):
        spc[j, :len(s)] = s
    return spc, np.array([len(s) for s in spcs])


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_against_reference_goldens(golden, ci):
    g = golden("int_windows")
    flens, spcs = _int_cases()[ci]
    spc, n = pack(spcs)
    p = windows.plan_windows(np.array(flens), spc, n, 80)
    assert np.array_equal(p["bounds"].numpy(), g["c%d_se" % ci])
    assert np.array_equal(p["s_idx"].numpy(), g["c%d_s_idx" % ci])
    assert np.array_equal(p["e_idx"].numpy(), g["c%d_e_idx" % ci])
    assert np.array_equal(p["flen_acc"].numpy(), g["c%d_flen_acc" % ci])
    assert np.array_equal(p["select"].numpy(), g["c%d_select" % ci])
    rows = list(windows.iter_windows(np.array(flens), spc, n, 80))
    assert len(rows) == g["c%d_se" % ci].shape[0] and rows[0][4] == list(range(len(flens)))


def test_random_ragged_batches_match_the_loop_restatement():
    rng = np.random.RandomState(20190721)
    for trial in range(300):
        U = rng.randint(1, 9)
        bs = int(rng.choice([1, 7, 40, 80]))
        flens = rng.randint(1, 420, size=U)
        spcs = []
        for j in range(U):
            k = rng.randint(1, flens[j] + 1)
            mode = rng.randint(3)
            if mode == 0:       # dense prefix
                s = np.arange(k)
            elif mode == 1:     # random subset
                s = np.sort(rng.choice(flens[j], size=k, replace=False))
            else:               # a late burst (windows with no speech frame at all)
                lo = rng.randint(0, flens[j])
                s = np.arange(lo, min(flens[j], lo + k))
            spcs.append(list(map(int, s)))
        spc, n = pack(spcs, pad=512)
        ref = orc.window_bookkeeping(flens, spc, n, bs)
        got = windows.plan_windows(flens, spc, n, bs)
        assert got["bounds"].shape[0] == len(ref), (trial, bs, flens)
        for w, r in enumerate(ref):
            assert (int(got["bounds"][w, 0]), int(got["bounds"][w, 1])) == (r["s"], r["e"])
            assert np.array_equal(got["s_idx"][w].numpy(), r["s_idx"]), (trial, w, "s_idx")
            assert np.array_equal(got["e_idx"][w].numpy(), r["e_idx"]), (trial, w, "e_idx")
            assert np.array_equal(got["flen_acc"][w].numpy(), r["flen_acc"]), (trial, w, "flen_acc")
            assert [j for j in range(U) if got["select"][w, j]] == r["select_utt_idx"], (trial, w, "select")


def test_single_frame_and_exact_multiple_edges():
    # max_flen == batch_size (one window), max_flen == 2*batch_size (two full windows), 1-frame utterance
    for flens, bs in (([80], 80), ([160, 3], 80), ([1], 80), ([81], 80)):
        spcs = [list(range(f)) for f in flens]
        spc, n = pack(spcs, pad=256)
        ref = orc.window_bookkeeping(np.array(flens), spc, n, bs)
        got = windows.plan_windows(np.array(flens), spc, n, bs)
        assert got["bounds"].shape[0] == len(ref)
        for w, r in enumerate(ref):
            assert np.array_equal(got["s_idx"][w].numpy(), r["s_idx"]) and np.array_equal(got["e_idx"][w].numpy(), r["e_idx"])
            assert np.array_equal(got["flen_acc"][w].numpy(), r["flen_acc"])


@pytest.mark.gpu
def test_on_device_int_tensors_bit_exact(golden):
    g = golden("int_windows")
    flens, spcs = _int_cases()[0]
    spc, n = pack(spcs)
    dev = torch.device("cuda:0")
    p = windows.plan_windows(torch.tensor(flens, device=dev), torch.from_numpy(spc).to(dev), torch.from_numpy(n).to(dev), 80)
    for k, key in (("s_idx", "c0_s_idx"), ("e_idx", "c0_e_idx"), ("flen_acc", "c0_flen_acc"), ("select", "c0_select")):
        assert p[k].is_cuda and np.array_equal(p[k].cpu().numpy(), g[key])
