"""Stage-4 data path of the one-to-one recipe: what feeds the hot path.

Mirrors, with the same names, arguments and returned keys,
  * `padding` and `FeatureDatasetSingleVAE` / `FeatureDatasetInit` of the reference's src/utils/dataset.py:23-98,
  * `read_hdf5` of src/utils/utils.py:38-60 (on the HDF5 C library through hdf5io.py; h5py where only that is installed),
  * `train_generator` of src/bin/train_gru_cyclevae_gauss_batch.py:45-149 -- the host half (:50-66: per-batch maxima, trimming,
    host -> device copies) plus the frame-window bookkeeping, which here is `windows.plan_windows` (closed form, no per-element
    device sync) instead of the reference's Python loops over device tensors (:78-99, :108-133).

What is different on purpose: `collate_pinned` stacks a list of items into PINNED host tensors (the reference leaves that to
`DataLoader`'s default collate in pageable memory), so the trimmed host -> device copies of `train_generator` are asynchronous
DMA transfers on the current stream.  Use it as `DataLoader(dataset, batch_size=..., collate_fn=loader.collate_pinned)`.

The on-disk format is the reference's: one HDF5 file per utterance with the datasets `/feat_org_lf0`, `/cvuvlogf0fil_ap`,
`/spcidx_range`.  A dataset object takes a `reader(path, key)` callable (default: `read_hdf5`), so tests -- and users of another
container format -- can supply their own.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

import hdf5io
import windows


def read_hdf5(hdf5_name, hdf5_path):
    """Values of dataset `hdf5_path` of file `hdf5_name` (src/utils/utils.py:38-60): through the HDF5 C library (hdf5io.py, ctypes),
    or through h5py where only that is installed.  Neither: ImportError -- there is no silent substitute for the file format."""
    try:
        return hdf5io.read_hdf5(hdf5_name, hdf5_path)
    except ImportError as no_c_library:
        try:
            import h5py
        except ImportError:
            raise ImportError("read_hdf5 needs libhdf5 (hdf5io.use_library(path)) or h5py; pass reader=... to the dataset for "
                              "another container format") from no_c_library
    if not os.path.exists(hdf5_name):
        raise FileNotFoundError("there is no such hdf5 file: %s" % hdf5_name)
    with h5py.File(hdf5_name, "r") as f:
        if hdf5_path not in f:
            raise KeyError("there is no such data in %s: %s" % (hdf5_name, hdf5_path))
        return f[hdf5_path][()]


def padding(x, flen, value=0):
    """Rows of `value` appended until `x` has `flen` rows; same contract as src/utils/dataset.py:23-31, including its dtype rule:
    an array that needed padding comes back as float64 (the reference concatenates with a float64 block), one that did not is
    returned as it is."""
    missing = flen - x.shape[0]
    if missing <= 0:
        return x
    out = np.full((flen,) + tuple(x.shape[1:]), value, dtype=np.result_type(x.dtype, np.float64))
    out[:x.shape[0]] = x
    return out


class FeatureDatasetInit(Dataset):
    """src/utils/dataset.py:34-51."""

    def __init__(self, file_list, pad_transform, reader=read_hdf5):
        self.file_list, self.pad_transform, self.reader = file_list, pad_transform, reader

    def __len__(self):
        return len(self.file_list)

    def __getitem__(self, idx):
        featfile = self.file_list[idx]
        h = torch.FloatTensor(self.pad_transform(self.reader(featfile, "/feat_org_lf0")))
        return {'h': h, 'flen': h.shape[0], 'featfile': featfile}


class FeatureDatasetSingleVAE(Dataset):
    """Dataset for one-to-one conversion (src/utils/dataset.py:54-98): per utterance the source features, the one-hot speaker
    codes of the source and of the conversion target, the converted-F0 features, the speech-frame indices, and the same
    utterance's features / speech-frame indices of the paired speaker."""

    def __init__(self, file_list_src, file_list_src_trg, pad_transform, spk_src, reader=read_hdf5):
        self.file_list_src = file_list_src
        self.file_list_src_trg = file_list_src_trg
        self.pad_transform = pad_transform
        self.spk_src = spk_src
        self.reader = reader

    def __len__(self):
        return len(self.file_list_src)

    def __getitem__(self, idx):
        read = self.reader
        featfile_src = self.file_list_src[idx]
        featfile_src_trg = self.file_list_src_trg[idx]
        h_src = read(featfile_src, "/feat_org_lf0")
        flen_src = h_src.shape[0]
        src_code = np.zeros((flen_src, 2))
        trg_code = np.zeros((flen_src, 2))
        own = 0 if os.path.basename(os.path.dirname(featfile_src)) == self.spk_src else 1     # :77-82
        src_code[:, own] = 1
        trg_code[:, 1 - own] = 1
        cv_src = read(featfile_src, "/cvuvlogf0fil_ap")
        spcidx_src = read(featfile_src, "/spcidx_range")[0]
        h_src_trg = read(featfile_src_trg, "/feat_org_lf0")
        spcidx_src_trg = read(featfile_src_trg, "/spcidx_range")[0]
        pad = self.pad_transform
        return {'h_src': torch.FloatTensor(pad(h_src)), 'flen_src': flen_src, 'src_code': torch.FloatTensor(pad(src_code)),
                'trg_code': torch.FloatTensor(pad(trg_code)), 'cv_src': torch.FloatTensor(pad(cv_src)),
                'spcidx_src': torch.LongTensor(pad(spcidx_src)), 'flen_spc_src': spcidx_src.shape[0],
                'h_src_trg': torch.FloatTensor(pad(h_src_trg)), 'flen_src_trg': h_src_trg.shape[0],
                'spcidx_src_trg': torch.LongTensor(pad(spcidx_src_trg)), 'flen_spc_src_trg': spcidx_src_trg.shape[0],
                'featfile_src': featfile_src, 'featfile_src_trg': featfile_src_trg}


def collate_pinned(items):
    """What `DataLoader`'s default collate returns for a list of dataset items (tensors stacked, ints as int64 tensors, strings
    as lists), with every tensor in pinned host memory when a GPU is present.

    Pinning needs the HIP context, which a forked DataLoader worker must not touch: inside a worker process this function stacks
    into pageable memory (what default collate does) and the parent pins -- run the loader as
    `DataLoader(..., num_workers=N, collate_fn=loader.collate_pinned, pin_memory=True)` for N > 0; with num_workers=0 the
    tensors are pinned here directly."""
    pin = torch.cuda.is_available() and torch.utils.data.get_worker_info() is None
    out = {}
    for k in items[0]:
        v0 = items[0][k]
        if torch.is_tensor(v0):
            t = torch.empty((len(items),) + tuple(v0.shape), dtype=v0.dtype, pin_memory=pin)
            torch.stack([it[k] for it in items], 0, out=t)
            out[k] = t
        elif isinstance(v0, (int, np.integer)):
            out[k] = torch.tensor([int(it[k]) for it in items], dtype=torch.int64)
        else:
            out[k] = [it[k] for it in items]
    return out


def _trimmed_to_device(t, n_max, device, async_copy):
    """t[:, :n_max] on `device`.  A column slice of a pinned tensor is NOT contiguous, and torch stages a non-contiguous source
    through a pageable temporary (a synchronous copy); so the slice is first packed into a contiguous pinned staging tensor, and
    that is what the DMA engine reads."""
    src = t[:, :n_max]
    if not async_copy:
        return src.to(device)
    if not src.is_contiguous() or not src.is_pinned():
        stage = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        stage.copy_(src)
        src = stage
    return src.to(device, non_blocking=True)


def train_generator(dataloader, device, batch_size=80):
    """The reference's generator (train_gru_cyclevae_gauss_batch.py:45-149), same yields in the same order:

      batch_size != 0:  (hs_src, src_codes[:, s:e+1], trg_codes[:, s:e+1], hs_src_trg, cvs_src, s, e, spcidcs_src_s_idx,
                         spcidcs_src_e_idx, c_idx, idx, spcidcs_src, spcidcs_src_trg, featfiles_src, featfiles_src_trg, flens,
                         flens_src_trg, flens_spc_src, flens_spc_src_trg, select_utt_idx, flen_acc, n_batch_utt) per frame window,
      batch_size == 0:  one yield per dataloader batch with the whole utterances,
    then the end-of-pass sentinel (c_idx = idx = -1).  Like the reference it takes ONE dataloader batch per pass (:142-146).
    The window bookkeeping comes from windows.plan_windows on the host copy of the speech-frame indices: bit-identical
    values, no device -> host sync per comparison."""
    nb = device.type == "cuda"
    keys_by_len = (("flen_src", ("h_src", "src_code", "trg_code", "cv_src")), ("flen_spc_src", ("spcidx_src",)),
                   ("flen_src_trg", ("h_src_trg",)), ("flen_spc_src_trg", ("spcidx_src_trg",)))
    while True:
        c_idx = 0
        for idx, batch in enumerate(dataloader):
            # :50-66 -- every padded array is cut to the longest utterance of the batch before it crosses PCIe
            lens, dev = {}, {}
            for lk, names in keys_by_len:
                lens[lk] = batch[lk].data.numpy()
                n_max = int(lens[lk].max())
                for name in names:
                    dev[name] = _trimmed_to_device(batch[name], n_max, device, nb)
            spc_host = batch["spcidx_src"][:, :int(lens["flen_spc_src"].max())]
            files = (batch["featfile_src"], batch["featfile_src_trg"])
            tail = (lens["flen_src"], lens["flen_src_trg"], lens["flen_spc_src"], lens["flen_spc_src_trg"])
            n_utt = dev["h_src"].size(0)
            if batch_size != 0:
                for s, e, s_idx, e_idx, sel, facc in windows.iter_windows(lens["flen_src"], spc_host, lens["flen_spc_src"], batch_size):
                    yield (dev["h_src"], dev["src_code"][:, s:e + 1], dev["trg_code"][:, s:e + 1], dev["h_src_trg"], dev["cv_src"], s, e,
                           s_idx, e_idx, c_idx, idx, dev["spcidx_src"], dev["spcidx_src_trg"]) + files + tail + (sel, facc, n_utt)
            else:
                yield (dev["h_src"], dev["src_code"], dev["trg_code"], dev["h_src_trg"], dev["cv_src"], c_idx, idx, dev["spcidx_src"],
                       dev["spcidx_src_trg"]) + files + tail + (n_utt,)
            c_idx += 1
            break                                   # :142-146: one dataloader batch per pass
        n_end = 22 if batch_size > 0 else 16        # the end-of-pass sentinel: empty fields, c_idx = idx = -1
        end = [[] for _ in range(n_end)]
        end[9 if batch_size > 0 else 5] = -1
        end[10 if batch_size > 0 else 6] = -1
        yield tuple(end)
