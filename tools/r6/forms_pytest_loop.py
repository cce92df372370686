"""Round 6 diagnostic: call test_stage4step_forms_agree's body for (1024, 6, 12) then (64, 20, 9) N times in one process, per geometry."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "cyclevae-vc_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import gru_vae as gv
import test_gpu_train as tg
tg.note = lambda msg: None
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for geom in (1, 0):
    gv._lib().set_option("train_bwd_geom", geom)
    fails = 0
    for it in range(N):
        for hid, B, T in ((1024, 6, 12), (64, 20, 9)):
            try:
                tg.test_stage4step_forms_agree(gv, dev, hid, B, T)
            except AssertionError as e:
                fails += 1
                print("geom", geom, "iter", it, "hid", hid, "FAILED:", str(e)[:300].replace("\n", " "), flush=True)
    print("geom", geom, "failures", fails, "of", 2 * N, flush=True)
