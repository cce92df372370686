"""CPU: stage6.convert_files -- the reference's multi-GPU form of stage 5 / 6 is a file-list fan-out, one process per GPU
(decode_gru-cyclevae_gauss.py:190-195, 591-602).  The fan-out itself (np.array_split chunks, one spawned process per "device",
results gathered in caller order, Philox draws keyed by list position so that the result does not depend on the device count,
a failing worker reported instead of a hang) runs here on the host build of the library: two and three "devices" against one."""
import os

import numpy as np
import pytest

import hdf5io
import stage6
import synth
from emu_files_worker import emu_files_worker

torch = pytest.importorskip("torch")
import gru_vae  # noqa: E402


def test_split_is_the_reference_array_split():
    for n_items in range(0, 12):
        items = ["f%d" % i for i in range(n_items)]
        for n_dev in (1, 2, 3, 8):
            ref = [c.tolist() for c in np.array_split(items, n_dev)]            # decode...:190-191
            got = stage6.split_file_list(items, n_dev)
            assert [c for _, c in got] == ref
            assert all(first == sum(len(c) for c in ref[:k]) for k, (first, _) in enumerate(got))


@pytest.fixture(scope="module")
def problem(tmp_path_factory):
    d = tmp_path_factory.mktemp("h5")
    lens = [(6, 5), (4, 7), (5, 4), (6, 3)]          # (short: every frame is a full kernel step on the host-fiber emulator)
    P = synth.CycleVAEProblem(B=1, T=8, in_dim=10, out_dim=6, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.05, tag="files")
    files = []
    for i, (a, b) in enumerate(lens):
        pa, pb = str(d / ("src%d.h5" % i)), str(d / ("trg%d.h5" % i))
        hdf5io.write_hdf5(pa, "/feat_org_lf0", synth.features("files/s%d" % i, 1, a, P.mu, P.sigma)[0])
        hdf5io.write_hdf5(pb, "/feat_org_lf0", synth.features("files/t%d" % i, 1, b, P.mu, P.sigma)[0])
        files.append((pa, pb))
    mods = []
    for sd, i, o, enc in ((P.enc, 10, 8, True), (P.dec, 6, 6, False)):
        m = gru_vae.GRU_RNN(in_dim=i, out_dim=o, hidden_units=64, kernel_size=3, dilation_size=2, scale_in_flag=enc, scale_out_flag=not enc)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        mods.append(m)
    y = [torch.from_numpy(P.y_in_enc), torch.from_numpy(P.y_in_dec), torch.from_numpy(P.y_in_dec)]
    return files, lens, mods, y


def run(problem, devices, files=None):
    all_files, lens, (enc, dec), y = problem
    return stage6.convert_files(enc, dec, all_files if files is None else files, devices, y[0], y[1], y[2], 4, n_smpl_dec=3, seed=11,
                                worker=emu_files_worker, timeout=600)


def test_two_and_three_devices_give_the_one_device_result_in_caller_order(problem):
    files, lens, _, _ = problem
    one = run(problem, [0])
    assert len(one) == len(files)
    for r, (a, b) in zip(one, lens):
        assert [x.shape for x in r] == [(a, 6), (a, 6), (b, 6), (a, 8), (b, 8)] and all(np.isfinite(x).all() for x in r)
    assert not np.array_equal(one[0][0], one[0][1])            # trg-code and src-code conversions differ
    for devices in ([0, 1],):
        got = run(problem, devices)
        for q, (r1, rn) in enumerate(zip(one, got)):
            for x1, xn in zip(r1, rn):
                assert np.array_equal(x1, xn), (devices, q)
    # more devices than files: empty chunks are fine
    few = run(problem, [0, 1, 2], files=files[:2])
    assert all(np.array_equal(a, b) for r1, rn in zip(one[:2], few) for a, b in zip(r1, rn))


def test_draws_are_keyed_by_list_position_not_by_chunk(problem):
    files, _, _, _ = problem
    # the same file at positions 0 and 2 draws differently; the file at position 2 converts the same whichever device gets it
    twice = run(problem, [0, 1], files=[files[0], files[1], files[0]])
    assert np.array_equal(twice[0][3], twice[2][3])              # latents: no draws involved
    assert not np.array_equal(twice[0][0], twice[2][0])          # decoder outputs: other draw ids
    again = run(problem, [0], files=[files[0], files[1], files[0]])
    assert np.array_equal(twice[2][0], again[2][0])


def test_a_failing_worker_is_reported(problem):
    files, _, _, _ = problem
    with pytest.raises(RuntimeError, match="cannot read"):
        run(problem, [0, 1], files=[files[0], (os.path.join("boom", "x.h5"), files[1][1])])
