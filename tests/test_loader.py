"""cyclevae-vc_amd/loader.py (dataset items, pinned collate, the stage-4 generator) against tests/golden/loader.npz,
which make_golden.py::case_loader recorded by running the REFERENCE'S OWN `padding`, `FeatureDatasetSingleVAE`
(src/utils/dataset.py:23-98), torch's default collate and `train_generator` (train_gru_cyclevae_gauss_batch.py:45-149) on a
dict-backed `read_hdf5`.  Everything here is copied bytes and integer bookkeeping: bit-exact."""
import os
import sys

import numpy as np
import pytest
from conftest import have_hdf5
import torch
from torch.utils.data import DataLoader

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, "golden")]

import loader
import synth  # noqa: F401  (make_golden imports it)


def fixture():
    import make_golden as mg
    store = mg.loader_store()
    src, trg = mg.loader_lists()
    pad = lambda x: loader.padding(x, 40, value=0.0)
    ds = loader.FeatureDatasetSingleVAE(src, trg, pad, "spkA", reader=lambda f, k: store[(f, k)])
    return ds


def test_items_and_pinned_collate_equal_the_reference(golden):
    g = golden("loader")
    ds = fixture()
    assert len(ds) == 3
    for collate in (None, loader.collate_pinned):
        kw = {} if collate is None else {"collate_fn": collate}
        batch = next(iter(DataLoader(ds, batch_size=3, shuffle=False, **kw)))
        for k, v in batch.items():
            if torch.is_tensor(v):
                ref = g["item_" + k]
                assert v.dtype == torch.from_numpy(ref).dtype and np.array_equal(v.numpy(), ref), k
        assert batch["featfile_src"] == ["/data/spkA/utt0.h5", "/data/spkB/utt1.h5", "/data/spkA/utt2.h5"]
    # speaker codes: one-hot of the file's own speaker / of the other one (dataset.py:77-82)
    it = ds[1]
    assert it["flen_src"] == 19 and it["src_code"][:19, 1].eq(1).all() and it["trg_code"][:19, 0].eq(1).all() and it["src_code"][19:].eq(0).all()


@pytest.mark.parametrize("collate", ["default", "pinned"])
def test_generator_yields_equal_the_reference(golden, collate):
    g = golden("loader")
    kw = {} if collate == "default" else {"collate_fn": loader.collate_pinned}
    dl = DataLoader(fixture(), batch_size=3, shuffle=False, **kw)
    gen = loader.train_generator(dl, torch.device("cpu"), batch_size=12)
    nw = int(g["n_windows"][0])
    for w in range(nw):
        y = next(gen)
        assert len(y) == 22
        for i, name in ((0, "hs_src"), (1, "src_codes"), (2, "trg_codes"), (3, "hs_src_trg"), (4, "cvs_src"), (11, "spcidcs_src"),
                        (12, "spcidcs_src_trg")):
            assert np.array_equal(y[i].numpy(), g["w%d_%s" % (w, name)]), (w, name)
        assert [y[5], y[6], y[9], y[10], y[21]] == list(g["w%d_ints" % w])
        assert np.array_equal(np.asarray(y[7]), g["w%d_s_idx" % w]) and np.array_equal(np.asarray(y[8]), g["w%d_e_idx" % w])
        assert list(y[19]) == list(g["w%d_select" % w]) and np.array_equal(np.asarray(y[20]), g["w%d_flen_acc" % w])
        assert np.array_equal(np.stack([np.asarray(y[i], np.int64) for i in (15, 16, 17, 18)]), g["w%d_lens" % w])
        assert y[13] == ["/data/spkA/utt0.h5", "/data/spkB/utt1.h5", "/data/spkA/utt2.h5"]
    end = next(gen)                      # end-of-pass sentinel, then the next pass starts over
    assert len(end) == 22 and end[9] == -1 and end[10] == -1 and end[0] == []
    again = next(gen)
    assert again[5] == 0 and np.array_equal(again[0].numpy(), g["w0_hs_src"])
    # whole-utterance mode (batch_size = 0, :137-138)
    y = next(loader.train_generator(dl, torch.device("cpu"), batch_size=0))
    assert len(y) == 16 and np.array_equal(y[1].numpy(), g["utt_src_codes"]) and [y[5], y[6], y[15]] == list(g["utt_ints"])


def hdf5_fixture(root):
    """The same three utterances as `fixture()`, but as the recipe's per-utterance HDF5 files under `root`, read by the DEFAULT reader."""
    import hdf5io
    import make_golden as mg
    for (f, key), arr in mg.loader_store().items():
        hdf5io.write_hdf5(str(root) + f, key, arr)
    src, trg = mg.loader_lists()
    pad = lambda x: loader.padding(x, 40, value=0.0)
    return loader.FeatureDatasetSingleVAE([str(root) + f for f in src], [str(root) + f for f in trg], pad, "spkA")


@pytest.mark.skipif(not have_hdf5(), reason="no HDF5 C library on this machine")
def test_hdf5_files_give_the_recorded_items_and_windows(golden, tmp_path):
    """The files on disk in the reference's format (one .h5 per utterance, /feat_org_lf0, /cvuvlogf0fil_ap, /spcidx_range), read
    through loader.read_hdf5 = the HDF5 C library: items, collate and generator yields equal the reference-recorded ones."""
    g = golden("loader")
    ds = hdf5_fixture(tmp_path)
    batch = next(iter(DataLoader(ds, batch_size=3, shuffle=False, collate_fn=loader.collate_pinned)))
    for k, v in batch.items():
        if torch.is_tensor(v):
            ref = g["item_" + k]
            assert v.dtype == torch.from_numpy(ref).dtype and np.array_equal(v.numpy(), ref), k
    assert batch["featfile_src"] == [str(tmp_path) + f for f in ("/data/spkA/utt0.h5", "/data/spkB/utt1.h5", "/data/spkA/utt2.h5")]
    gen = loader.train_generator(DataLoader(ds, batch_size=3, shuffle=False), torch.device("cpu"), batch_size=12)
    for w in range(int(g["n_windows"][0])):
        y = next(gen)
        for i, name in ((0, "hs_src"), (3, "hs_src_trg"), (4, "cvs_src"), (11, "spcidcs_src"), (12, "spcidcs_src_trg")):
            assert np.array_equal(y[i].numpy(), g["w%d_%s" % (w, name)]), (w, name)
        assert [y[5], y[6], y[9], y[10], y[21]] == list(g["w%d_ints" % w])


@pytest.mark.skipif(not have_hdf5(), reason="no HDF5 C library on this machine")
def test_read_hdf5_errors_are_loud(tmp_path):
    import hdf5io
    with pytest.raises(FileNotFoundError):
        loader.read_hdf5("/nonexistent.h5", "/feat_org_lf0")
    f = str(tmp_path / "a.h5")
    hdf5io.write_hdf5(f, "/feat_org_lf0", np.zeros((3, 54), np.float32))
    with pytest.raises(KeyError):
        loader.read_hdf5(f, "/spcidx_range")
    junk = tmp_path / "junk.h5"
    junk.write_bytes(b"not an hdf5 file")
    with pytest.raises(OSError):
        loader.read_hdf5(str(junk), "/feat_org_lf0")


@pytest.mark.gpu
def test_pinned_batches_reach_the_device_and_feed_the_windows():
    """On the GPU box: the collate pins, the generator's trimmed copies land on the device with the host values, and the window
    bookkeeping it yields is the device-side plan of windows.plan_windows."""
    import windows
    dev = torch.device("cuda:0")
    dl = DataLoader(fixture(), batch_size=3, shuffle=False, collate_fn=loader.collate_pinned)
    batch = next(iter(dl))
    assert batch["h_src"].is_pinned() and batch["spcidx_src"].is_pinned()
    gen = loader.train_generator(dl, dev, batch_size=12)
    y = next(gen)
    torch.cuda.synchronize()
    assert y[0].device.type == "cuda" and torch.equal(y[0].cpu(), batch["h_src"][:, :31])
    plan = windows.plan_windows(torch.as_tensor(y[15]), y[11], torch.as_tensor(y[17]), 12)
    assert np.array_equal(plan["s_idx"][0].cpu().numpy(), np.asarray(y[7]))


@pytest.mark.gpu
def test_device_yields_equal_the_reference_recorded_windows(golden):
    """On the GPU box: every tensor the generator puts on the device (pinned collate, trimmed asynchronous copies) and every piece
    of window bookkeeping equal tests/golden/loader.npz -- the yields of the reference's own train_generator on the same files."""
    g = golden("loader")
    dev = torch.device("cuda:0")
    dl = DataLoader(fixture(), batch_size=3, shuffle=False, collate_fn=loader.collate_pinned)
    gen = loader.train_generator(dl, dev, batch_size=12)
    for w in range(int(g["n_windows"][0])):
        y = next(gen)
        torch.cuda.synchronize()
        for i, name in ((0, "hs_src"), (1, "src_codes"), (2, "trg_codes"), (3, "hs_src_trg"), (4, "cvs_src"), (11, "spcidcs_src"),
                        (12, "spcidcs_src_trg")):
            assert y[i].device.type == "cuda" and np.array_equal(y[i].cpu().numpy(), g["w%d_%s" % (w, name)]), (w, name)
        assert [y[5], y[6], y[9], y[10], y[21]] == list(g["w%d_ints" % w])
        assert np.array_equal(np.asarray(y[7]), g["w%d_s_idx" % w]) and np.array_equal(np.asarray(y[8]), g["w%d_e_idx" % w])
        assert list(y[19]) == list(g["w%d_select" % w]) and np.array_equal(np.asarray(y[20]), g["w%d_flen_acc" % w])
