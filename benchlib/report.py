"""What every leg of bench.py shares: the algorithmic work figures of SURVEY.md 8(d), the roofline peak, logging and the ONE JSON line."""
import json
import sys

# algorithmic MACs per frame per pass
MAC_ENC, MAC_DEC = 5166220, 4397100
# what the dominant kernel (k_gru_steps_v5 / v4: front-end + recurrence of one pass) computes, in the reference's terms:
# conv0 + conv1 + W_ih[:, :9C].x_conv + W_ih[:, 9C:].y + W_hh.h  (everything of a pass but scale_in, out_1, scale_out)
MAC_KERN_ENC = 26244 + 236196 + 1492992 + 196608 + 3145728
MAC_KERN_DEC = 10404 + 93636 + 940032 + 153600 + 3145728
PEAK_F32_MFMA_TFLOPS = 157.3                                      # MI355X_MICROARCH.md chip table



def log(msg):
    sys.stderr.write("[bench] %s\n" % msg)
    sys.stderr.flush()


def flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def emit(res):
    """The ONE JSON line, as the last line of stdout (whatever C libraries still hold in their stdio buffers goes out first)."""
    flush_c_stdio()
    print(json.dumps(res), flush=True)
