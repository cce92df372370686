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


def have_hdf5():
    """True when hdf5io finds an HDF5 C library (the image ships one under /opt/conda/lib); tests of the file format skip otherwise."""
    try:
        import hdf5io
        hdf5io.library_version()
        return True
    except ImportError:
        return False
