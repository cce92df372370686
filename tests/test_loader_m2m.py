"""the many-to-many datasets (reference src/utils/dataset.py:101-492) in cyclevae-vc_amd/loader_m2m.py against
tests/golden/m2m.npz, recorded by executing the reference's own class definitions (ast-extracted) over the same dict-backed reader
with the same np.random seed: every item of every class, key for key, bit for bit."""
import numpy as np
import pytest

import loader
import loader_m2m
import synth

torch = pytest.importorskip("torch")


def store():
    src_spk, trg_spk = ["sA", "sB", "sC"], ["tA", "tB"]
    st = {}
    for si, spk in enumerate(src_spk + trg_spk):
        for u in range(2):
            n = 9 + 3 * si + 5 * u
            f = "/data/%s/utt%d.h5" % (spk, u)
            st[(f, "/feat_org_lf0")] = synth.normal("m2m/%s/%d/feat" % (spk, u), (n, 5)).astype(np.float32)
            for other in (trg_spk if spk in src_spk else src_spk):
                st[(f, "/cvuvlogf0fil_ap_" + other)] = synth.normal("m2m/%s/%d/cv_%s" % (spk, u, other), (n, 3)).astype(np.float32)
            keep = np.nonzero(synth.uniform01("m2m/%s/%d/spc" % (spk, u), (n,)) > 0.3)[0]
            st[(f, "/spcidx_range")] = keep[None, :].astype(np.int64)
    return st, src_spk, trg_spk


def check_item(prefix, item, g):
    seen = 0
    for k, v in item.items():
        if torch.is_tensor(v):
            ref = g["%s_%s" % (prefix, k)]
            assert v.numpy().dtype == ref.dtype and np.array_equal(v.numpy(), ref), (prefix, k)
        elif isinstance(v, list) and v and torch.is_tensor(v[0]):
            for i, t in enumerate(v):
                ref = g["%s_%s_%d" % (prefix, k, i)]
                assert t.numpy().dtype == ref.dtype and np.array_equal(t.numpy(), ref), (prefix, k, i)
        elif isinstance(v, list):
            assert list(g["%s_%s" % (prefix, k)]) == v, (prefix, k)
        elif isinstance(v, str):
            assert str(g["%s_%s" % (prefix, k)][0]) == v, (prefix, k)
        else:
            assert int(g["%s_%s" % (prefix, k)][0]) == int(v), (prefix, k)
        seen += 1
    # and nothing the reference returns is missing
    want = {n[len(prefix) + 1:] for n in g.files if n.startswith(prefix + "_")}
    have = set()
    for k, v in item.items():
        have |= {"%s_%d" % (k, i) for i in range(len(v))} if isinstance(v, list) and v and torch.is_tensor(v[0]) else {k}
    assert want == have, (prefix, sorted(want ^ have))


def test_training_datasets_item_for_item(golden):
    g = golden("m2m")
    st, src_spk, trg_spk = store()
    read = lambda f, k: st[(f, k)]
    pad = lambda x: loader.padding(x, 30, value=0.0)
    files = ["/data/%s/utt%d.h5" % (s, u) for s in src_spk + trg_spk for u in range(2)]
    for name, cls in (("FeatureDatasetMultTrainVAE", loader_m2m.FeatureDatasetMultTrainVAE),
                      ("FeatureDatasetMultTrainVAECls", loader_m2m.FeatureDatasetMultTrainVAECls)):
        ds = cls(files, pad, src_spk, trg_spk, 2, reader=read)
        assert len(ds) == 10
        np.random.seed(1234)
        for i in range(len(ds)):
            check_item("%s_%d" % (name, i), ds[i], g)


def test_evaluation_datasets_and_their_deterministic_pairing(golden):
    g = golden("m2m")
    st, src_spk, trg_spk = store()
    read = lambda f, k: st[(f, k)]
    pad = lambda x: loader.padding(x, 30, value=0.0)
    src_lists = [["/data/%s/utt%d.h5" % (s, u) for u in range(2)] for s in src_spk]
    trg_lists = [["/data/%s/utt%d.h5" % (s, u) for u in range(2)] for s in trg_spk]
    for name, cls in (("FeatureDatasetMultEvalVAE", loader_m2m.FeatureDatasetMultEvalVAE),
                      ("FeatureDatasetMultEvalVAECls", loader_m2m.FeatureDatasetMultEvalVAECls)):
        ds = cls(src_lists, trg_lists, pad, src_spk, trg_spk, reader=read)
        assert len(ds) == int(g[name + "_len"][0]) == 6
        assert np.array_equal(np.array([[ds.count_spk_pair_cv[s][t] for t in trg_spk] for s in src_spk]), g[name + "_pairs"])
        assert list(g[name + "_file_list_src_trg"]) == ds.file_list_src_trg
        for i in range(len(ds)):
            check_item("%s_%d" % (name, i), ds[i], g)
    one = loader_m2m.FeatureDatasetMultEvalVAE(src_lists, trg_lists[:1], pad, src_spk, trg_spk[:1], reader=read)
    assert list(g["one_trg_file_list_src_trg"]) == one.file_list_src_trg
    # the reference's quirk is kept: 'src_trg_code' carries the code over the TARGET utterance's frames
    it = loader_m2m.FeatureDatasetMultEvalVAE(src_lists, trg_lists, pad, src_spk, trg_spk, reader=read)[0]
    assert torch.equal(it["src_trg_code"], it["trg_code"])


def test_proc_functions_fill_the_code_in_place():
    st, src_spk, trg_spk = store()
    read = lambda f, k: st[(f, k)]
    code = np.zeros((9, 5))
    np.random.seed(3)
    cv, tc, spk, f_trg, pairs = loader_m2m.proc_multspk_data_random("/data/sA/utt0.h5", src_spk, trg_spk, 3, code, reader=read)
    assert spk == "sA" and np.all(code[:, 0] == 1) and code.sum() == 9 and len(cv) == len(tc) == len(pairs) == 3
    assert all(p in trg_spk for p in pairs) and f_trg == "/data/%s/utt0.h5" % pairs[0]
    code = np.zeros((9 + 3 * 4, 5))
    out = loader_m2m.proc_multspk_data_random_cls("/data/tB/utt0.h5", src_spk, trg_spk, 2, code, reader=read)
    assert np.all(code[:, 4] == 1) and np.all(out[5] == 4) and all(p in src_spk for p in out[4]) and out[6][0].dtype == np.int64
