"""Data path of the sibling MANY-TO-MANY recipes (reference src/utils/dataset.py:101-492): the datasets that feed
the same GRU_RNN encoder / decoder when several source and target speakers share one model.  Dead code in egs/one-to-one, kept
here under the reference's names, arguments and item keys so that a many-to-many script finds what it imports:

  proc_multspk_data_random(_cls)              :101-135 / :290-329  speaker one-hot of the utterance, a RANDOM conversion pair per cycle
  FeatureDatasetMultTrainVAE(Cls)             :138-186 / :332-382  training items (n_cyc converted-F0 streams and target codes)
  FeatureDatasetMultEvalVAE(Cls)              :189-287 / :385-492  evaluation items with a deterministic speaker pairing

`reader(path, key)` (default: loader.read_hdf5) replaces the reference's module-level read_hdf5, as in loader.py.  The random pair of
a cycle is drawn with np.random.randint exactly where the reference draws it, so a seeded run selects the same speakers.  One quirk
of the reference is kept on purpose (never "fix" silently): the evaluation item's key 'src_trg_code' carries
`trg_code` (the code over the TARGET utterance's frames, :283), not the `src_trg_code` array built two lines above it.
"""
import itertools
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from loader import read_hdf5


def _proc(featfile, spk_src_list, spk_trg_list, n_cyc, src_code, reader, with_class):
    featfile_spk = os.path.basename(os.path.dirname(featfile))
    n_src = len(spk_src_list)
    flag_src = featfile_spk in spk_src_list
    src_class_code = None
    if flag_src:
        own = spk_src_list.index(featfile_spk)                               # (first match, as the reference's loop)
        src_code[:, own] = 1
        src_class_code = np.ones(src_code.shape[0], dtype=np.int64) * own
    elif featfile_spk in spk_trg_list:
        own = spk_trg_list.index(featfile_spk) + n_src
        src_code[:, own] = 1
        src_class_code = np.ones(src_code.shape[0], dtype=np.int64) * own
    cv_src_list, trg_code_list, pair_spk_list, trg_class_code_list = [None] * n_cyc, [None] * n_cyc, [None] * n_cyc, [None] * n_cyc
    pool, offset = (spk_trg_list, n_src) if flag_src else (spk_src_list, 0)  # a source utterance converts to a target speaker and v.v.
    for i in range(n_cyc):
        trg_code_list[i] = np.zeros((src_code.shape[0], src_code.shape[1]))
        pair_idx = np.random.randint(0, len(pool))
        trg_code_list[i][:, pair_idx + offset] = 1
        trg_class_code_list[i] = np.ones(src_code.shape[0], dtype=np.int64) * (pair_idx + offset)
        pair_spk = pool[pair_idx]
        cv_src_list[i] = reader(featfile, "/cvuvlogf0fil_ap_" + pair_spk)
        pair_spk_list[i] = pair_spk
    featfile_src_trg = os.path.dirname(os.path.dirname(featfile)) + "/" + pair_spk_list[0] + "/" + os.path.basename(featfile)
    if with_class:
        return cv_src_list, trg_code_list, featfile_spk, featfile_src_trg, pair_spk_list, src_class_code, trg_class_code_list
    return cv_src_list, trg_code_list, featfile_spk, featfile_src_trg, pair_spk_list


def proc_multspk_data_random(featfile, spk_src_list, spk_trg_list, n_cyc, src_code, reader=read_hdf5):
    """src/utils/dataset.py:101-135: fills `src_code` (one-hot of the utterance's own speaker over n_src + n_trg speakers) IN PLACE and
    draws one conversion pair per cycle."""
    return _proc(featfile, spk_src_list, spk_trg_list, n_cyc, src_code, reader, False)


def proc_multspk_data_random_cls(featfile, spk_src_list, spk_trg_list, n_cyc, src_code, reader=read_hdf5):
    """src/utils/dataset.py:290-329: the same plus the integer class codes (int64 [flen]) of the own and the paired speakers."""
    return _proc(featfile, spk_src_list, spk_trg_list, n_cyc, src_code, reader, True)


class FeatureDatasetMultTrainVAE(Dataset):
    """Dataset for training many-to-many (src/utils/dataset.py:138-186)."""

    with_class = False

    def __init__(self, file_list, pad_transform, spk_src_list, spk_trg_list, n_cyc, reader=read_hdf5):
        self.file_list, self.pad_transform = file_list, pad_transform
        self.spk_src_list, self.spk_trg_list = spk_src_list, spk_trg_list
        self.n_spk_src, self.n_spk_trg = len(spk_src_list), len(spk_trg_list)
        self.n_spk = self.n_spk_src + self.n_spk_trg
        self.n_cyc = n_cyc
        self.reader = reader

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        read, pad = self.reader, self.pad_transform
        featfile_src = self.file_list[idx]
        h_src = read(featfile_src, "/feat_org_lf0")
        flen_src = h_src.shape[0]
        src_code = np.zeros((flen_src, self.n_spk))
        got = _proc(featfile_src, self.spk_src_list, self.spk_trg_list, self.n_cyc, src_code, read, self.with_class)
        cv_src_list, src_trg_code_list, featfile_spk, featfile_src_trg, pair_spk_list = got[:5]
        spcidx_src = read(featfile_src, "/spcidx_range")[0]
        h_src_trg = read(featfile_src_trg, "/feat_org_lf0")
        spcidx_src_trg = read(featfile_src_trg, "/spcidx_range")[0]
        item = {'h_src': torch.FloatTensor(pad(h_src)), 'flen_src': flen_src, 'src_code': torch.FloatTensor(pad(src_code)),
                'src_trg_code_list': [torch.FloatTensor(pad(c)) for c in src_trg_code_list],
                'cv_src_list': [torch.FloatTensor(pad(c)) for c in cv_src_list],
                'spcidx_src': torch.LongTensor(pad(spcidx_src)), 'flen_spc_src': spcidx_src.shape[0],
                'h_src_trg': torch.FloatTensor(pad(h_src_trg)), 'flen_src_trg': h_src_trg.shape[0],
                'spcidx_src_trg': torch.LongTensor(pad(spcidx_src_trg)), 'flen_spc_src_trg': spcidx_src_trg.shape[0],
                'featfile_src': featfile_src, 'featfile_src_trg': featfile_src_trg, 'featfile_spk': featfile_spk,
                'pair_spk_list': pair_spk_list}
        if self.with_class:
            item['src_class_code'] = torch.LongTensor(pad(got[5]))
            item['trg_class_code_list'] = [torch.LongTensor(pad(c)) for c in got[6]]
        return item


class FeatureDatasetMultTrainVAECls(FeatureDatasetMultTrainVAE):
    """Dataset for training many-to-many with classifier (src/utils/dataset.py:332-382)."""

    with_class = True


class FeatureDatasetMultEvalVAE(Dataset):
    """Dataset for evaluation many-to-many (src/utils/dataset.py:189-287): a conversion pair is chosen DETERMINISTICALLY per source
    speaker -- even source speakers take target 1, 3, ... (0 when there is one target), odd ones target 0, 2, ..., each wrapping
    around -- and every evaluation utterance of the source speaker is paired with the same-numbered utterance of that target."""

    with_class = False

    def __init__(self, file_list_src_list, file_list_trg_list, pad_transform, spk_src_list, spk_trg_list, reader=read_hdf5):
        self.file_list_src_list, self.file_list_trg_list = file_list_src_list, file_list_trg_list
        self.pad_transform = pad_transform
        self.spk_src_list, self.spk_trg_list = spk_src_list, spk_trg_list
        self.n_spk_src, self.n_spk_trg = len(spk_src_list), len(spk_trg_list)
        self.n_spk = self.n_spk_src + self.n_spk_trg
        self.n_eval_utt = len(file_list_src_list[0])
        self.reader = reader
        self.file_list_src, self.file_list_src_trg = [], []
        self.count_spk_pair_cv = {s: {t: 0 for t in spk_trg_list} for s in spk_src_list}
        # two round-robins over the target speakers (:205-228): source speakers 0, 2, 4, ... draw from the odd-numbered targets
        # (target 0 when it is the only one), source speakers 1, 3, 5, ... from the even-numbered ones
        targets_of = (itertools.cycle(range(1, self.n_spk_trg, 2) if self.n_spk_trg > 1 else (0,)),
                      itertools.cycle(range(0, self.n_spk_trg, 2)))
        for s, src_files in enumerate(file_list_src_list[:self.n_spk_src]):
            t = next(targets_of[s & 1])
            self.count_spk_pair_cv[spk_src_list[s]][spk_trg_list[t]] += self.n_eval_utt
            self.file_list_src.extend(src_files[:self.n_eval_utt])
            self.file_list_src_trg.extend(file_list_trg_list[t][:self.n_eval_utt])

    def __len__(self):
        return len(self.file_list_src)

    def __getitem__(self, idx):
        read, pad = self.reader, self.pad_transform
        featfile_src, featfile_src_trg = self.file_list_src[idx], self.file_list_src_trg[idx]
        spk_src = os.path.basename(os.path.dirname(featfile_src))
        spk_trg = os.path.basename(os.path.dirname(featfile_src_trg))
        idx_src = self.spk_src_list.index(spk_src)
        idx_trg = self.n_spk_src + self.spk_trg_list.index(spk_trg)
        h_src = read(featfile_src, "/feat_org_lf0")
        flen_src = h_src.shape[0]
        src_code, src_trg_code = np.zeros((flen_src, self.n_spk)), np.zeros((flen_src, self.n_spk))
        src_code[:, idx_src] = 1
        src_trg_code[:, idx_trg] = 1
        cv_src = read(featfile_src, "/cvuvlogf0fil_ap_" + spk_trg)
        spcidx_src = read(featfile_src, "/spcidx_range")[0]
        h_src_trg = read(featfile_src_trg, "/feat_org_lf0")
        flen_src_trg = h_src_trg.shape[0]
        trg_code, trg_src_code = np.zeros((flen_src_trg, self.n_spk)), np.zeros((flen_src_trg, self.n_spk))
        trg_code[:, idx_trg] = 1
        trg_src_code[:, idx_src] = 1
        cv_trg = read(featfile_src_trg, "/cvuvlogf0fil_ap_" + spk_src)
        spcidx_src_trg = read(featfile_src_trg, "/spcidx_range")[0]
        trg_code_t = torch.FloatTensor(pad(trg_code))
        item = {'h_src': torch.FloatTensor(pad(h_src)), 'flen_src': flen_src, 'src_code': torch.FloatTensor(pad(src_code)),
                'src_trg_code': trg_code_t,            # (sic, :283: `trg_code`, not the src_trg_code array built above)
                'cv_src': torch.FloatTensor(pad(cv_src)), 'spcidx_src': torch.LongTensor(pad(spcidx_src)),
                'flen_spc_src': spcidx_src.shape[0], 'h_src_trg': torch.FloatTensor(pad(h_src_trg)), 'flen_src_trg': flen_src_trg,
                'trg_code': trg_code_t, 'trg_src_code': torch.FloatTensor(pad(trg_src_code)), 'cv_trg': torch.FloatTensor(pad(cv_trg)),
                'spcidx_src_trg': torch.LongTensor(pad(spcidx_src_trg)), 'flen_spc_src_trg': spcidx_src_trg.shape[0],
                'featfile_src': featfile_src, 'featfile_src_trg': featfile_src_trg}
        if self.with_class:      # (FeatureDatasetMultEvalVAECls, :385-492: int64 class codes per frame)
            for key, n, spk in (('src_class_code', flen_src, idx_src), ('src_trg_class_code', flen_src, idx_trg),
                                ('trg_class_code', flen_src_trg, idx_trg), ('trg_src_class_code', flen_src_trg, idx_src)):
                item[key] = torch.LongTensor(pad(np.full(n, spk, dtype=np.int64)))
        return item


class FeatureDatasetMultEvalVAECls(FeatureDatasetMultEvalVAE):
    """Dataset for evaluation many-to-many with classifier (src/utils/dataset.py:385-492): the evaluation item plus int64 class codes."""

    with_class = True
