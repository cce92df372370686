"""ABI 6: the library keeps no process-wide state -- status sink, draw origin / parts, options, side stream, profiling brackets and the
train-image registry live in a cvae_ctx, and every entry point takes it ("no global state besides a per-handle
descriptor; independent across handles").  Two contexts with different draw origins and options, used INTERLEAVED
in one process (and from two threads), must each give exactly what they give alone.  Runs on the host build of the library."""
import threading

import numpy as np
import pytest

import _cabi
import synth
from emu_util import NpNet, emu_lib, ptr


@pytest.fixture(scope="module")
def base():
    return emu_lib()


def philox_eps(lib, rows=64, L=4, seed=9, draw=3):
    lat = np.zeros((rows, 2 * L), np.float32)
    z, e = np.zeros((rows, L), np.float32), np.zeros((rows, L), np.float32)
    lib.sample(ptr(lat), rows, L, None, seed, draw, ptr(z), ptr(e))
    return e


def two_row_pass(lib, P):
    """An encoder pass of two rows: the word-exchange kernel, or -- option no_ll = 1 -- the row-tile kernel (other rounding)."""
    enc = NpNet(lib, P.enc, 6, 8, 64)
    fl = _cabi.FLAG_PERSISTENT | _cabi.FLAG_EXACT3
    return enc.forward(P.x, P.y_in_enc, clamp_lat_dim=4, flags=fl)[0]


def test_contexts_do_not_share_draw_origin_or_options(base):
    P = synth.CycleVAEProblem(B=2, T=5, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="ctx")
    # what each configuration gives ALONE, in a context of its own
    alone = {}
    for name, origin, no_ll in (("a", (0, 0, 0), 0), ("b", (5, 12, 16), 1)):
        c = base.new_context()
        c.set_draw_origin(*origin)
        c.set_option("no_ll", no_ll)
        alone[name] = (philox_eps(c), two_row_pass(c, P))
        c.close()
    assert not np.array_equal(alone["a"][0], alone["b"][0])          # another draw origin: other Philox numbers
    assert not np.array_equal(alone["a"][1], alone["b"][1])          # another kernel: other rounding
    assert np.abs(alone["a"][1] - alone["b"][1]).max() < 1e-5
    # the same two configurations side by side, calls interleaved
    a, b = base.new_context(), base.new_context()
    a.set_draw_origin(0, 0, 0)
    b.set_draw_origin(5, 12, 16)
    b.set_option("no_ll", 1)
    assert a.get_option("no_ll") == 0 and b.get_option("no_ll") == 1
    for _ in range(2):
        ea, eb = philox_eps(a), philox_eps(b)
        ob, oa = two_row_pass(b, P), two_row_pass(a, P)
        assert np.array_equal(ea, alone["a"][0]) and np.array_equal(eb, alone["b"][0])
        assert np.array_equal(oa, alone["a"][1]) and np.array_equal(ob, alone["b"][1])
    b.reset_options()
    assert b.get_option("no_ll") == 0 and np.array_equal(two_row_pass(b, P), alone["a"][1])
    # the module-level context of the other tests never saw any of it
    assert base.get_option("no_ll") == 0 and np.array_equal(philox_eps(base), alone["a"][0])
    a.close()
    b.close()


def test_two_threads_two_contexts(base):
    P = synth.CycleVAEProblem(B=2, T=4, in_dim=6, out_dim=4, lat_dim=4, hidden=64, n_cyc=1, bias_scale=0.1, tag="ctxthr")
    ref = {}
    for no_ll in (0, 1):
        c = base.new_context()
        c.set_option("no_ll", no_ll)
        ref[no_ll] = two_row_pass(c, P)
        c.close()
    out, err = {0: [], 1: []}, []
    turn = threading.Lock()         # (the host-fiber emulator runs one kernel at a time; the GPU build has no such limit.  Calls of the
                                    #  two threads still alternate, each thread with its own context and its own thread-local state)

    def work(no_ll):
        try:
            c = base.new_context()
            c.set_option("no_ll", no_ll)
            for _ in range(2):
                with turn:
                    out[no_ll].append(two_row_pass(c, P))
            c.close()
        except Exception as e:          # noqa: BLE001
            err.append(e)
    th = [threading.Thread(target=work, args=(v,)) for v in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not err, err
    for v in (0, 1):
        assert all(np.array_equal(o, ref[v]) for o in out[v])


def test_a_null_context_is_refused_and_a_train_image_belongs_to_its_context(base):
    import ctypes as C
    raw = base.cdll
    d = base.desc(6, 8, 64, 3, 2, True, False)
    assert raw.cvae_net_prepared_bytes(None, C.byref(d)) == 0          # size queries: 0
    assert raw.cvae_set_draw_origin(None, 0, 0, 0) == -1 and b"null context" in raw.cvae_last_error_string()
    assert raw.cvae_set_option(None, b"no_ll", 1) == -1
    assert raw.cvae_ctx_destroy(None) == 0
    # ABI 5's process-wide setters are gone: every exported entry point but the four below takes the handle
    import re
    import os
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "cyclevae_hip.h")).read()
    protos = re.findall(r"^(?:int|size_t|const char\*|cvae_ctx\*)\s+(cvae_[a-z0-9_]+)\(([^;]*?)\);", hdr, re.M | re.S)
    assert len(protos) == len(_cabi.EXPORTS)
    for name, args in protos:
        if name in _cabi.NO_CONTEXT:
            continue
        assert args.lstrip().startswith("cvae_ctx* ctx"), name
