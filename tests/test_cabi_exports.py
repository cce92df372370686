"""The C-ABI boundary on a machine without a GPU: the real libcyclevae_hip.so (hipcc, gfx950) loads, exports every function
include/cyclevae_hip.h declares, the ctypes binding lists exactly those, and an ABI/size query answers (no compute calls)."""
import ctypes
import os
import re

import _cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cyclevae-vc_amd", "libcyclevae_hip.so")


def declared_functions():
    text = open(os.path.join(ROOT, "include", "cyclevae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cvae_[a-z0-9_]+)\s*\(", text)))


def test_header_binding_and_library_agree():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(_cabi.EXPORTS) == names, (sorted(set(names) - set(_cabi.EXPORTS)), sorted(set(_cabi.EXPORTS) - set(names)))
    lib = ctypes.CDLL(LIB)
    for n in names:
        assert hasattr(lib, n), "libcyclevae_hip.so does not export " + n
    lib.cvae_abi_version.restype = ctypes.c_int
    assert lib.cvae_abi_version() == _cabi.ABI_VERSION


def test_size_queries_answer_without_a_gpu():
    import pytest
    lib = _cabi.CvaeLib(LIB)
    d = lib.desc(54, 64, 1024, 3, 2, True, False)
    assert lib.prepared_bytes(d) > 4 * 5_000_000                # at least the folded weights of a 5.2 M-parameter net
    assert lib.pass_workspace_bytes(d, 64, 80) > 64 * 80 * 1024 * 4
    bad = lib.desc(54, 64, 1000, 3, 2, True, False)             # H not a multiple of 16: refused, with a message
    with pytest.raises(_cabi.CvaeError):
        lib.prepared_bytes(bad)
    assert lib.lib.cvae_last_error_string() not in (None, b"")
