import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cyclevae-vc_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`: host-fiber emulator runs of the real library, oracle checks, 2-rank gloo tests) takes ~10 min
    in one process and ~4 min on four pytest-xdist workers; the tests are independent of each other, so a run that deselects the
    GPU tests and does not say `-n` itself is spread over four workers.  GPU runs stay in one process (one device, persistent
    kernels that want the whole chip).  CYCLEVAE_TEST_WORKERS=0 turns this off, =N picks another count."""
    if os.environ.get("PYTEST_XDIST_WORKER") or hasattr(config, "workerinput"):
        return None
    n = os.environ.get("CYCLEVAE_TEST_WORKERS", "4")
    if not n.isdigit() or int(n) < 2 or not config.pluginmanager.hasplugin("xdist"):
        return None
    if "not gpu" not in (getattr(config.option, "markexpr", "") or "") or getattr(config.option, "numprocesses", None):
        return None
    if getattr(config.option, "collectonly", False) or getattr(config.option, "usepdb", False):
        return None
    # (pytest-xdist's own pytest_cmdline_main has run already: set what it derives from -n)
    config.option.numprocesses = int(n)
    config.option.dist = "load"
    config.option.tx = ["popen"] * int(n)
    return None


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


@pytest.fixture
def options(request):
    """options(name=value, ...): set tuning / diagnostic switches (cvae_set_option) on the test's library for this test only."""
    libs = []

    def setter(**kw):
        lib = request.getfixturevalue("lib")
        if lib not in libs:
            libs.append(lib)
        for k, v in kw.items():
            lib.set_option(k, v)
    yield setter
    for lib in libs:
        lib.reset_options()


@pytest.fixture(scope="session", autouse=True)
def _library_options_from_env():
    """CYCLEVAE_TEST_LIB_OPTIONS="name=value name=value": library switches for a whole `-m gpu` run (A/B of kernel variants through
    the real tests; measurement runs only -- the `options` fixture resets them after any test that uses it)."""
    spec = os.environ.get("CYCLEVAE_TEST_LIB_OPTIONS", "").split()
    if spec:
        import gru_vae
        for kv in spec:
            gru_vae._lib().set_option(kv.split("=")[0], int(kv.split("=")[1]))
    yield


def have_hdf5():
    """True when hdf5io finds an HDF5 C library (the image ships one under /opt/conda/lib); tests of the file format skip otherwise."""
    try:
        import hdf5io
        hdf5io.library_version()
        return True
    except ImportError:
        return False
