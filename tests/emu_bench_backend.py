"""TEST INFRASTRUCTURE (never imported by the package): lets `bench.py` run end to end on a machine without a GPU, with the host-fiber
build of the library (tests/emu) behind the drop-in module, so that the N > 1 path of the bench -- self-spawn under
torch.distributed.run, rendezvous, barrier, max-over-ranks timing, per-rank times, the `ranks` block of the JSON line -- is exercised
by the CPU suite on two gloo ranks (tests/test_distributed_shards.py).  Selected with CYCLEVAE_BENCH_BACKEND=emu, read by bench.py's
device selection only.  The workload is a SMALL model (hidden 64) at a few frames: a timing of the emulator says nothing about the
product and the JSON line says so (`config.backend`)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# model dimensions of the emulated run (the emulator executes one fiber per GPU thread: hu1024 is out of reach)
DIMS = {"in_dim": 10, "lat_dim": 4, "out_dim": 6, "hidden": 64}


def install():
    """Point the binding at the emulator build and neutralise the few CUDA-runtime calls of the eval path.  Returns the torch device
    the bench should use ("cpu")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import emu_util
    import _cabi
    _cabi.DEFAULT_LIB = emu_util.build_emu()
    import gru_vae
    gru_vae._need_cuda = lambda t, what: None          # "device" memory is host memory under emulation
    gru_vae._stream = lambda: 0

    class _NoStream(object):
        cuda_stream = 0

        def synchronize(self):
            pass

        def wait_event(self, ev):
            pass

    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _NoStream()
    return torch.device("cpu")
