"""Test infrastructure: the per-device worker of stage6.convert_files on the HOST build of the library (tests/emu), so that the
file-list fan-out -- split, one process per "device", gather in caller order, draws keyed by list position -- runs without a GPU.
Same C-ABI calls as stage6._encode_pairs / _decode_pairs (cvae_gru_rnn_forward_stacked over stacked single-row cells)."""
import numpy as np


def convert_pair_np(lib, enc, dec, fs, ft, y_pp, y_src, y_trg, L, n, seed, pair_id):
    import _cabi
    from emu_util import ptr
    fl = _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3
    lens = (fs.shape[0], ft.shape[0])
    T, Cin, Co = max(lens), fs.shape[1], dec.d.out_dim
    lat = [np.zeros((T, 2 * L), np.float32) for _ in range(2)]
    ws = np.zeros(lib.pass_workspace_bytes(enc.d, 2, T) // 4, np.float32)
    lib.gru_rnn_forward_stacked(enc.d, ptr(enc.prepared), [lib.pass_input((ptr(fs), Cin, Cin), frames=lens[0]),
                                                            lib.pass_input((ptr(ft), Cin, Cin), frames=lens[1])],
                                [ptr(y_pp)] * 2, 1, T, L, [ptr(a) for a in lat], ptr(ws), ws.nbytes, fl)
    assert lib.workspace_status(ptr(ws))[0] == 0
    codes = np.eye(2, dtype=np.float32)
    d0 = 2 * n * pair_id
    out = [np.zeros((T, Co), np.float32) for _ in range(3)]

    def cell(code_row, lat_row, frames, draw0):
        return lib.pass_input((ptr(codes[code_row]), 2, 0), lat=ptr(lat[lat_row]), lat_dim=L, eps=None, seed=seed, draw_id=draw0,
                              frames=frames, n_draws=n)
    ws = np.zeros(lib.pass_workspace_bytes(dec.d, 3, T) // 4, np.float32)
    lib.gru_rnn_forward_stacked(dec.d, ptr(dec.prepared), [cell(1, 0, lens[0], d0), cell(0, 0, lens[0], d0), cell(1, 1, lens[1], d0 + n)],
                                [ptr(y_trg), ptr(y_src), ptr(y_trg)], 1, T, -1, [ptr(o) for o in out], ptr(ws), ws.nbytes, fl)
    assert lib.workspace_status(ptr(ws))[0] == 0
    return (out[0][:lens[0]].copy(), out[1][:lens[0]].copy(), out[2][:lens[1]].copy(), lat[0][:lens[0]].copy(), lat[1][:lens[1]].copy())


def emu_files_worker(rank, device, first, items, cfg, queue):
    try:
        import hdf5io
        from emu_util import NpNet, emu_lib
        lib = emu_lib()
        (ekw, esd), (dkw, dsd) = cfg["enc"], cfg["dec"]
        enc = NpNet(lib, esd, ekw["in_dim"], ekw["out_dim"], ekw["hidden_units"])
        dec = NpNet(lib, dsd, dkw["in_dim"], dkw["out_dim"], dkw["hidden_units"])
        y = [np.ascontiguousarray(v.reshape(1, -1), np.float32) for v in cfg["y_in"]]
        res = []
        for i, (a, b) in enumerate(items):
            if "boom" in a:
                raise RuntimeError("cannot read " + a)
            fs = np.ascontiguousarray(hdf5io.read_hdf5(a, cfg["key"]), np.float32)
            ft = np.ascontiguousarray(hdf5io.read_hdf5(b, cfg["key"]), np.float32)
            res.append(convert_pair_np(lib, enc, dec, fs, ft, y[0], y[1], y[2], cfg["lat_dim"], cfg["n_smpl_dec"], cfg["seed"], first + i))
        queue.put((rank, first, res, None))
    except BaseException as e:
        import traceback
        queue.put((rank, first, None, "%s\n%s" % (e, traceback.format_exc())))
