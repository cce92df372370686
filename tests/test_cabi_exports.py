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


def test_recurrent_kernels_do_not_spill():
    """hipcc's kernel-resource-usage remarks of the shipped build: the all-resident recurrent kernels of the default paths use no
    scratch (a private segment puts a ~240 us gap in front of every launch on MI355X and turns register traffic into memory
    traffic), and run at one wave per SIMD with the 512 registers that assumes."""
    import __graft_entry__
    if not os.path.exists(__graft_entry__.RESOURCES) or os.path.getmtime(__graft_entry__.RESOURCES) < os.path.getmtime(LIB):
        __graft_entry__.build(force=True)
    text = open(__graft_entry__.RESOURCES).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
    assert len(blocks) > 100, "no resource remarks in " + __graft_entry__.RESOURCES
    seen = set()
    for b in blocks:
        name = b.split()[0]
        hot = [k for k in ("k_gru_steps_v6", "k_gru_steps_v5", "k_gru_steps_ll", "k_train_fwd_steps_x3", "k_train_fwd_steps_w3", "k_train_fwd_steps_h", "k_train_fwd_steps_ll",
                           "k_train_bwd_steps", "k_outproj_v6", "k_prologue") if k in name]
        if not hot:
            continue
        seen.add(hot[0])
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        assert scratch == 0, "%s spills %d bytes per lane" % (name, scratch)
    assert {"k_gru_steps_v6", "k_gru_steps_ll", "k_train_fwd_steps_x3", "k_train_fwd_steps_w3", "k_train_bwd_steps"} <= seen
    assert any("k_train_bwd_steps_w3" in b.split()[0] for b in blocks)          # (covered by the k_train_bwd_steps prefix above)
