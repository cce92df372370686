"""ctypes binding of include/cyclevae_hip.h (libcyclevae_hip.so).

Pointer-level and framework-agnostic: every buffer argument is an integer device address
(`tensor.data_ptr()`); streams are integer hipStream_t handles.  There is no fallback: if the shared
library is missing or its ABI version differs, loading raises.
"""
import ctypes as C
import os

ABI_VERSION = 6
FLAG_PERSISTENT = 1
FLAG_PROFILE = 2
FLAG_GENERIC_STEP = 4
FLAG_STEP_TIMING = 8
FLAG_HOISTED_FRONTEND = 32
FLAG_SPLIT_F16 = 256
FLAG_EXACT3 = 512
CLAMP_LAPLACE = 1 << 30     # OR into a clamp_lat_dim argument: clamp_vae_laplace's floor (gru_vae.py:417) instead of ln(1e-6)

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.environ.get("CYCLEVAE_LIB") or os.path.join(_HERE, "libcyclevae_hip.so")   # (variable: A/B builds of the same library)

_fp = C.c_void_p


class NetDesc(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("out_dim", C.c_int32), ("hidden", C.c_int32), ("kernel_size", C.c_int32),
                ("layers", C.c_int32), ("has_scale_in", C.c_int32), ("has_scale_out", C.c_int32)]


WEIGHT_FIELDS = ("scale_in_w", "scale_in_b", "conv0_w", "conv0_b", "conv1_w", "conv1_b", "w_ih", "w_hh", "b_ih",
                 "b_hh", "out_w", "out_b", "scale_out_w", "scale_out_b")

# C field -> reference state_dict key
STATE_KEYS = {"scale_in_w": "scale_in.weight", "scale_in_b": "scale_in.bias", "conv0_w": "conv.conv.0.weight",
              "conv0_b": "conv.conv.0.bias", "conv1_w": "conv.conv.1.weight", "conv1_b": "conv.conv.1.bias",
              "w_ih": "gru.weight_ih_l0", "w_hh": "gru.weight_hh_l0", "b_ih": "gru.bias_ih_l0",
              "b_hh": "gru.bias_hh_l0", "out_w": "out_1.weight", "out_b": "out_1.bias",
              "scale_out_w": "scale_out.weight", "scale_out_b": "scale_out.bias"}


class NetWeights(C.Structure):
    _fields_ = [(f, _fp) for f in WEIGHT_FIELDS]


GRAD_FIELDS = ("conv0_w", "conv0_b", "conv1_w", "conv1_b", "w_ih", "w_hh", "b_ih", "b_hh", "out_w", "out_b")


class NetGrads(C.Structure):
    _fields_ = [(f, _fp) for f in GRAD_FIELDS]


class Seg(C.Structure):
    _fields_ = [("ptr", _fp), ("width", C.c_int32), ("row_stride", C.c_int32)]


class PassInput(C.Structure):
    _fields_ = [("seg0", Seg), ("seg1", Seg), ("lat", _fp), ("lat_dim", C.c_int32), ("eps", _fp),
                ("seed", C.c_uint64), ("draw_id", C.c_uint64), ("frames", C.c_int32), ("n_draws", C.c_int32),
                ("ctx_before", C.c_int32), ("ctx_after", C.c_int32), ("draw_frame0", C.c_int64), ("eps_draw_stride", C.c_int64)]


class CvaeError(RuntimeError):
    pass


class CycleState(C.Structure):
    _fields_ = [("y_enc", _fp), ("y_dec", _fp), ("h_enc", _fp), ("h_dec", _fp)]


class _Bound(object):
    """The shared library's entry points with THIS object's context as their first argument (ABI 6: every entry point but
    cvae_last_error_string / cvae_abi_version / cvae_ctx_create takes the handle).  `lib.lib.cvae_xyz(args...)` therefore reads
    like the C prototype minus the context."""

    def __init__(self, cdll, ctx):
        self._cdll, self._ctx, self._cache = cdll, ctx, {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._cdll, name)            # AttributeError for a missing export, as with the CDLL itself
            if name in NO_CONTEXT:
                fn = raw
            else:
                ctx = self._ctx

                def fn(*args, _raw=raw, _ctx=ctx):
                    return _raw(_ctx, *args)
            self._cache[name] = fn
        return fn


NO_CONTEXT = ("cvae_last_error_string", "cvae_abi_version", "cvae_ctx_create", "cvae_ctx_destroy")
_CDLLS = {}     # path -> (CDLL with argtypes declared): one load per process, any number of contexts


class CvaeLib(object):
    """One CONTEXT (cvae_ctx) of the library: its own status sink, draw origin, options, side stream, profiling brackets.
    gru_vae keeps one per device; new_context() gives another one on the same loaded library."""

    def __init__(self, path=None, _cdll=None):
        path = path or DEFAULT_LIB
        if not os.path.exists(path):
            raise CvaeError("HIP library %s not found: build it with `python __graft_entry__.py` (hipcc "
                            "--offload-arch=gfx950); there is no CPU fallback" % path)
        self.path = path
        if _cdll is None:
            _cdll = _CDLLS.get(path)
        if _cdll is None:
            _cdll = _CDLLS[path] = self._declare(C.CDLL(path), path)
        self.cdll = _cdll
        self.ctx = C.c_void_p(_cdll.cvae_ctx_create())
        if not self.ctx.value:
            raise CvaeError("cvae_ctx_create failed: %s" % _cdll.cvae_last_error_string().decode())
        self.lib = _Bound(_cdll, self.ctx)

    def new_context(self):
        """Another context on the same loaded library (its own options, draw origin, status sink, side stream)."""
        return CvaeLib(self.path, _cdll=self.cdll)

    def close(self):
        """Destroy the context (after the streams it enqueued on have been synchronised)."""
        if getattr(self, "ctx", None) is not None and self.ctx.value:
            self.cdll.cvae_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p(None)

    @staticmethod
    def _declare(L, path):
        L.cvae_last_error_string.restype = C.c_char_p
        L.cvae_abi_version.restype = C.c_int
        for fn in ("cvae_net_prepared_bytes", "cvae_net_prepare_scratch_bytes"):
            getattr(L, fn).restype = C.c_size_t
            getattr(L, fn).argtypes = [C.POINTER(NetDesc)]
        L.cvae_net_prepare.restype = C.c_int
        L.cvae_net_prepare.argtypes = [C.POINTER(NetDesc), C.POINTER(NetWeights), _fp, C.c_size_t, _fp, C.c_size_t, _fp]
        L.cvae_pass_workspace_bytes.restype = C.c_size_t
        L.cvae_pass_workspace_bytes.argtypes = [C.POINTER(NetDesc), C.c_int, C.c_int]
        L.cvae_gru_rnn_forward.restype = C.c_int
        L.cvae_gru_rnn_forward.argtypes = [C.POINTER(NetDesc), _fp, C.POINTER(PassInput), _fp, _fp, C.c_int, C.c_int,
                                           C.c_int, _fp, _fp, _fp, _fp, C.c_size_t, C.c_int, _fp]
        L.cvae_gru_rnn_forward_stacked.restype = C.c_int
        L.cvae_gru_rnn_forward_stacked.argtypes = [C.POINTER(NetDesc), _fp, C.c_int, C.POINTER(PassInput), C.POINTER(C.c_void_p),
                                                   C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), _fp, C.c_size_t, C.c_int, _fp]
        L.cvae_gru_rnn_forward_stacked_carry.restype = C.c_int
        L.cvae_gru_rnn_forward_stacked_carry.argtypes = [C.POINTER(NetDesc), _fp, C.c_int, C.POINTER(PassInput), C.POINTER(C.c_void_p),
                                                         C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                                         C.POINTER(C.c_void_p), _fp, C.c_size_t, C.c_int, _fp]
        L.cvae_sample.restype = C.c_int
        L.cvae_sample.argtypes = [_fp, C.c_int, C.c_int, _fp, C.c_uint64, C.c_uint64, _fp, _fp, _fp]
        L.cvae_sample_laplace.restype = C.c_int
        L.cvae_sample_laplace.argtypes = [_fp, C.c_int, C.c_int, _fp, C.c_uint64, C.c_uint64, _fp, _fp, _fp]
        L.cvae_sample_laplace_backward.restype = C.c_int
        L.cvae_sample_laplace_backward.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp]
        L.cvae_cycle_workspace_bytes.restype = C.c_size_t
        L.cvae_cycle_workspace_bytes.argtypes = [C.POINTER(NetDesc), C.POINTER(NetDesc), C.c_int, C.c_int, C.c_int]
        L.cvae_cycle_forward.restype = C.c_int
        L.cvae_cycle_forward.argtypes = [C.POINTER(NetDesc), _fp, C.POINTER(NetDesc), _fp, _fp, _fp, C.c_int, _fp, _fp,
                                         C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, C.c_uint64,
                                         _fp, _fp, _fp, _fp, _fp, _fp, C.c_size_t, C.c_int, _fp]
        L.cvae_cycle_forward_carry.restype = C.c_int
        L.cvae_cycle_forward_carry.argtypes = L.cvae_cycle_forward.argtypes + [C.POINTER(CycleState), C.POINTER(CycleState)]
        L.cvae_workspace_status.restype = C.c_int
        L.cvae_workspace_status.argtypes = [_fp, C.POINTER(C.c_int32 * 4), _fp]
        L.cvae_train_image_bytes.restype = C.c_size_t
        L.cvae_train_image_bytes.argtypes = [C.POINTER(NetDesc)]
        L.cvae_net_prepare_train.restype = C.c_int
        L.cvae_net_prepare_train.argtypes = [C.POINTER(NetDesc), C.POINTER(NetWeights), _fp, C.c_size_t, C.c_float, _fp]
        L.cvae_net_prepare_train_v.restype = C.c_int
        L.cvae_net_prepare_train_v.argtypes = [C.POINTER(NetDesc), C.POINTER(NetWeights), _fp, C.c_size_t, C.c_float, C.c_int, _fp]
        L.cvae_train_variants_needed.restype = C.c_int
        L.cvae_train_variants_needed.argtypes = [C.POINTER(NetDesc), C.c_int, C.c_int]
        for fn in ("cvae_train_tape_bytes", "cvae_train_scratch_bytes"):
            getattr(L, fn).restype = C.c_size_t
            getattr(L, fn).argtypes = [C.POINTER(NetDesc), C.c_int, C.c_int]
        L.cvae_gru_rnn_forward_train.restype = C.c_int
        L.cvae_gru_rnn_forward_train.argtypes = [C.POINTER(NetDesc), _fp, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp,
                                                 C.c_uint64, C.c_float, _fp, _fp, _fp, _fp, C.c_size_t, _fp, C.c_size_t, _fp]
        L.cvae_gru_rnn_backward.restype = C.c_int
        L.cvae_gru_rnn_backward.argtypes = [C.POINTER(NetDesc), _fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_size_t, _fp,
                                            C.POINTER(NetGrads), C.c_int, _fp]
        L.cvae_train_debug_counters.restype = C.c_int
        L.cvae_train_debug_counters.argtypes = [C.POINTER(NetDesc), C.c_int, C.c_int, _fp, C.POINTER(C.c_longlong * 8), _fp]
        L.cvae_adam_step.restype = C.c_int
        L.cvae_adam_step.argtypes = [_fp, _fp, _fp, _fp, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _fp, _fp]
        L.cvae_adam_step_counted.restype = C.c_int
        L.cvae_adam_step_counted.argtypes = [_fp, _fp, _fp, _fp, C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp]
        L.cvae_sample_cat.restype = C.c_int
        L.cvae_sample_cat.argtypes = [_fp, _fp, _fp, _fp, _fp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, _fp, _fp, _fp]
        L.cvae_sample_cat_backward.restype = C.c_int
        L.cvae_sample_cat_backward.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp]
        L.cvae_stage4_loss.restype = C.c_int
        L.cvae_stage4_loss.argtypes = [_fp, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp, _fp, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                       _fp, _fp, _fp, _fp, _fp, _fp, C.c_int, _fp]
        L.cvae_mcd_l1.restype = C.c_int
        L.cvae_mcd_l1.argtypes = [_fp, C.c_long, _fp, C.c_long, C.c_int, C.c_int, _fp, _fp, _fp]
        L.cvae_mcd_l1_backward.restype = C.c_int
        L.cvae_mcd_l1_backward.argtypes = [_fp, C.c_long, _fp, C.c_long, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp]
        L.cvae_kl_gauss.restype = C.c_int
        L.cvae_kl_gauss.argtypes = [_fp, C.c_long, C.c_int, C.c_int, _fp, _fp]
        L.cvae_kl_gauss_backward.restype = C.c_int
        L.cvae_kl_gauss_backward.argtypes = [_fp, C.c_long, C.c_int, C.c_int, _fp, _fp, _fp]
        L.cvae_gv_postfilter.restype = C.c_int
        L.cvae_gv_postfilter.argtypes = [_fp, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp, _fp]
        L.cvae_mc2e.restype = C.c_int
        L.cvae_mc2e.argtypes = [_fp, C.c_int, C.c_long, C.c_int, C.c_int, C.c_double, C.c_int, _fp, _fp]
        L.cvae_mcd_aligned.restype = C.c_int
        L.cvae_mcd_aligned.argtypes = [_fp, C.c_long, _fp, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]
        L.cvae_dtw_work_bytes.restype = C.c_size_t
        L.cvae_dtw_work_bytes.argtypes = [C.c_int, C.c_int]
        L.cvae_dtw_org_to_trg.restype = C.c_int
        L.cvae_dtw_org_to_trg.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, C.c_size_t, _fp]
        L.cvae_step_timing.restype = C.c_int
        L.cvae_step_timing.argtypes = [C.POINTER(NetDesc), C.c_int, C.c_int, _fp, C.POINTER(C.c_double * 8), _fp]
        L.cvae_profile_collect.restype = C.c_int
        L.cvae_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int)]
        L.cvae_profile_collect_launches.restype = C.c_int
        L.cvae_profile_collect_launches.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
        L.cvae_train_profile_collect.restype = C.c_int
        L.cvae_train_profile_collect.argtypes = [C.POINTER(C.c_double * 4), C.POINTER(C.c_int * 4), C.POINTER(C.c_double * 4)]
        L.cvae_set_status_sink.restype = C.c_int
        L.cvae_set_status_sink.argtypes = [_fp]
        L.cvae_status_latch.restype = C.c_int
        L.cvae_status_latch.argtypes = [_fp, _fp]
        L.cvae_set_draw_origin.restype = C.c_int
        L.cvae_set_draw_origin.argtypes = [C.c_int64, C.c_int64, C.c_int64]
        L.cvae_set_side_stream.restype = C.c_int
        L.cvae_set_side_stream.argtypes = [C.c_void_p]
        L.cvae_join_side_stream.restype = C.c_int
        L.cvae_join_side_stream.argtypes = [C.c_void_p]
        L.cvae_selftest_limbs.restype = C.c_int
        L.cvae_selftest_limbs.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.cvae_selftest_occupy.restype = C.c_int
        L.cvae_selftest_occupy.argtypes = [C.c_int, C.c_size_t, C.c_int64, C.c_void_p]
        L.cvae_set_draw_parts.restype = C.c_int
        L.cvae_set_draw_parts.argtypes = [C.c_int32]
        L.cvae_set_option.restype = C.c_int
        L.cvae_set_option.argtypes = [C.c_char_p, C.c_int64]
        L.cvae_get_option.restype = C.c_int
        L.cvae_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
        L.cvae_reset_options.restype = C.c_int
        L.cvae_reset_options.argtypes = []
        v = L.cvae_abi_version()
        if v != ABI_VERSION:
            raise CvaeError("%s has ABI version %d, binding expects %d" % (path, v, ABI_VERSION))
        # ABI 6: the context handle in front of every other argument
        L.cvae_ctx_create.restype = C.c_void_p
        L.cvae_ctx_create.argtypes = []
        L.cvae_ctx_destroy.restype = C.c_int
        L.cvae_ctx_destroy.argtypes = [C.c_void_p]
        for name in EXPORTS:
            if name not in NO_CONTEXT:
                fn = getattr(L, name)
                fn.argtypes = [C.c_void_p] + list(fn.argtypes or [])
        return L

    # -- helpers ------------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise CvaeError("%s failed (%d): %s" % (what, rc, self.lib.cvae_last_error_string().decode()))

    @staticmethod
    def desc(in_dim, out_dim, hidden, kernel_size=3, layers=2, has_scale_in=False, has_scale_out=False):
        return NetDesc(in_dim, out_dim, hidden, kernel_size, layers, int(bool(has_scale_in)), int(bool(has_scale_out)))

    # -- entry points -------------------------------------------------------------------------------
    def prepared_bytes(self, d):
        n = self.lib.cvae_net_prepared_bytes(C.byref(d))
        if n == 0:
            raise CvaeError("bad net descriptor: %s" % self.lib.cvae_last_error_string().decode())
        return n

    def prepare_scratch_bytes(self, d):
        return self.lib.cvae_net_prepare_scratch_bytes(C.byref(d))

    def net_prepare(self, d, weight_ptrs, prepared, prepared_bytes, scratch, scratch_bytes, stream=0):
        w = NetWeights(**{f: weight_ptrs.get(f) or None for f in WEIGHT_FIELDS})
        self._check(self.lib.cvae_net_prepare(C.byref(d), C.byref(w), prepared, prepared_bytes, scratch, scratch_bytes,
                                              stream or None), "cvae_net_prepare")

    def pass_workspace_bytes(self, d, B, T):
        n = self.lib.cvae_pass_workspace_bytes(C.byref(d), B, T)
        if n == 0:
            raise CvaeError("cvae_pass_workspace_bytes: bad arguments (B=%d, T=%d)" % (B, T))
        return n

    @staticmethod
    def pass_input(seg0, seg1=None, lat=None, lat_dim=0, eps=None, seed=0, draw_id=0, frames=0, n_draws=0, ctx_before=0, ctx_after=0,
                   draw_frame0=0, eps_draw_stride=0):
        """seg = (ptr, width, row_stride).  ctx_* / draw_frame0 / eps_draw_stride: the input is a window of a longer utterance
        (cvae_pass_input, ABI 5)."""
        s0 = Seg(seg0[0], seg0[1], seg0[2])
        s1 = Seg(seg1[0], seg1[1], seg1[2]) if seg1 else Seg(None, 0, 0)
        return PassInput(s0, s1, lat or None, lat_dim, eps or None, seed, draw_id, frames, n_draws, ctx_before, ctx_after, draw_frame0,
                         eps_draw_stride)

    def gru_rnn_forward_stacked(self, d, prepared, pins, y_ins, B, T, clamp_lat_dim, trj_outs, ws, ws_bytes, flags=0, stream=0):
        n = len(pins)
        arr = (PassInput * n)(*pins)
        ys = (C.c_void_p * n)(*y_ins)
        outs = (C.c_void_p * n)(*trj_outs)
        self._check(self.lib.cvae_gru_rnn_forward_stacked(C.byref(d), prepared, n, arr, ys, B, T, clamp_lat_dim, outs, ws,
                                                          ws_bytes, flags, stream or None), "cvae_gru_rnn_forward_stacked")

    def gru_rnn_forward_stacked_carry(self, d, prepared, pins, y_ins, h_ins, B, T, clamp_lat_dim, trj_outs, h_lasts, ws, ws_bytes,
                                      flags=0, stream=0):
        """y_ins[c] None with h_ins[c] set: the window continues the recurrence that left h_ins[c]."""
        n = len(pins)
        arr = (PassInput * n)(*pins)
        ys = (C.c_void_p * n)(*[y or None for y in y_ins])
        hi = (C.c_void_p * n)(*[h or None for h in h_ins])
        outs = (C.c_void_p * n)(*trj_outs)
        hl = (C.c_void_p * n)(*[h or None for h in h_lasts])
        self._check(self.lib.cvae_gru_rnn_forward_stacked_carry(C.byref(d), prepared, n, arr, ys, hi, B, T, clamp_lat_dim, outs, hl,
                                                                ws, ws_bytes, flags, stream or None),
                    "cvae_gru_rnn_forward_stacked_carry")

    def gru_rnn_forward(self, d, prepared, pin, y_in, h_in, B, T, clamp_lat_dim, trj_out, y_last, h_last, ws, ws_bytes,
                        flags=0, stream=0):
        self._check(self.lib.cvae_gru_rnn_forward(C.byref(d), prepared, C.byref(pin), y_in, h_in or None, B, T,
                                                  clamp_lat_dim, trj_out, y_last or None, h_last or None, ws, ws_bytes,
                                                  flags, stream or None), "cvae_gru_rnn_forward")

    def sample(self, lat, rows, lat_dim, eps, seed, draw_id, z, eps_out=None, stream=0):
        self._check(self.lib.cvae_sample(lat, rows, lat_dim, eps or None, seed, draw_id, z, eps_out or None,
                                         stream or None), "cvae_sample")

    def sample_laplace(self, lat, rows, lat_dim, eps, seed, draw_id, z, eps_out=None, stream=0):
        self._check(self.lib.cvae_sample_laplace(lat, rows, lat_dim, eps or None, seed, draw_id, z, eps_out or None, stream or None),
                    "cvae_sample_laplace")

    def sample_laplace_backward(self, dz, lat, z, rows, lat_dim, dlat, stream=0):
        self._check(self.lib.cvae_sample_laplace_backward(dz, lat, z, rows, lat_dim, dlat, stream or None), "cvae_sample_laplace_backward")

    def cycle_workspace_bytes(self, de, dd, B, T, n_cyc):
        n = self.lib.cvae_cycle_workspace_bytes(C.byref(de), C.byref(dd), B, T, n_cyc)
        if n == 0:
            raise CvaeError("cvae_cycle_workspace_bytes: bad arguments")
        return n

    def cycle_forward(self, de, enc_prep, dd, dec_prep, x, cvx, stdim, code_src, code_trg, ncode, y_in_enc, y_in_dec,
                      B, T, n_cyc, lat_dim, eps, seed, out_lat, out_rec, out_cv, out_latcv, out_reccyc, ws, ws_bytes,
                      flags=0, stream=0):
        self._check(self.lib.cvae_cycle_forward(C.byref(de), enc_prep, C.byref(dd), dec_prep, x, cvx, stdim, code_src,
                                                code_trg, ncode, y_in_enc, y_in_dec, B, T, n_cyc, lat_dim, eps or None,
                                                seed, out_lat or None, out_rec or None, out_cv or None,
                                                out_latcv or None, out_reccyc or None, ws, ws_bytes, flags,
                                                stream or None), "cvae_cycle_forward")

    def cycle_forward_carry(self, de, enc_prep, dd, dec_prep, x, cvx, stdim, code_src, code_trg, ncode, y_in_enc, y_in_dec,
                            B, T, n_cyc, lat_dim, eps, seed, out_lat, out_rec, out_cv, out_latcv, out_reccyc, ws, ws_bytes,
                            flags=0, stream=0, state_in=None, state_out=None):
        """state_in / state_out: None or (y_enc, y_dec, h_enc, h_dec) device pointers (cvae_cycle_state)."""
        si = CycleState(*state_in) if state_in else None
        so = CycleState(*state_out) if state_out else None
        self._check(self.lib.cvae_cycle_forward_carry(C.byref(de), enc_prep, C.byref(dd), dec_prep, x, cvx, stdim, code_src,
                                                      code_trg, ncode, y_in_enc, y_in_dec, B, T, n_cyc, lat_dim, eps or None,
                                                      seed, out_lat or None, out_rec or None, out_cv or None,
                                                      out_latcv or None, out_reccyc or None, ws, ws_bytes, flags,
                                                      stream or None, C.byref(si) if si else None, C.byref(so) if so else None),
                    "cvae_cycle_forward_carry")

    # -- training ------------------------------------------------------------------------------------
    def train_image_bytes(self, d):
        return self.lib.cvae_train_image_bytes(C.byref(d))

    def net_prepare_train(self, d, weight_ptrs, image, image_bytes, stream=0, gru_drop_p=0.0, variants=7):
        """variants: OR of 1 (exact-operand tile kernels), 2 (fp16-pair kernels), 4 (fp32-MFMA forward): the MFMA-order images to build."""
        w = NetWeights(**{f: weight_ptrs.get(f) or None for f in WEIGHT_FIELDS})
        self._check(self.lib.cvae_net_prepare_train_v(C.byref(d), C.byref(w), image, image_bytes, gru_drop_p, int(variants),
                                                      stream or None), "cvae_net_prepare_train_v")

    def train_variants_needed(self, d, B, T):
        return self.lib.cvae_train_variants_needed(C.byref(d), B, T)

    def train_tape_bytes(self, d, B, T):
        return self.lib.cvae_train_tape_bytes(C.byref(d), B, T)

    def train_scratch_bytes(self, d, B, T):
        return self.lib.cvae_train_scratch_bytes(C.byref(d), B, T)

    def forward_train(self, d, image, x, y_in, h_in, B, T, clamp_lat_dim, cmask, gmask, seed, p_drop, trj_out, y_last, h_last,
                      tape, tape_bytes, scratch, scratch_bytes, stream=0):
        self._check(self.lib.cvae_gru_rnn_forward_train(C.byref(d), image, x, y_in, h_in or None, B, T, clamp_lat_dim,
                                                        cmask or None, gmask or None, seed, p_drop, trj_out, y_last or None,
                                                        h_last or None, tape, tape_bytes, scratch, scratch_bytes,
                                                        stream or None), "cvae_gru_rnn_forward_train")

    def backward(self, d, image, dout, B, T, clamp_lat_dim, tape, scratch, scratch_bytes, dx, grad_ptrs, accumulate=False,
                 stream=0):
        g = NetGrads(**{f: grad_ptrs[f] for f in GRAD_FIELDS})
        self._check(self.lib.cvae_gru_rnn_backward(C.byref(d), image, dout, B, T, clamp_lat_dim, tape, scratch, scratch_bytes,
                                                   dx or None, C.byref(g), int(bool(accumulate)), stream or None),
                    "cvae_gru_rnn_backward")

    def train_debug_counters(self, d, B, T, scratch, stream=0):
        out = (C.c_longlong * 8)()
        self._check(self.lib.cvae_train_debug_counters(C.byref(d), B, T, scratch, C.byref(out), stream or None),
                    "cvae_train_debug_counters")
        return list(out)

    def adam_step(self, p, g, m, v, n, lr, b1, b2, eps, step, stream=0, gate=None):
        self._check(self.lib.cvae_adam_step(p, g, m, v, n, lr, b1, b2, eps, step, gate or None, stream or None), "cvae_adam_step")

    def adam_step_counted(self, p, g, m, v, n, lr, b1, b2, eps, state, stream=0, gate=None):
        self._check(self.lib.cvae_adam_step_counted(p, g, m, v, n, lr, b1, b2, eps, state, gate or None, stream or None),
                    "cvae_adam_step_counted")

    def sample_cat(self, lat, codes, eps, seed, draws, B, T, lat_dim, ncode, out, eps_out, stream=0):
        """codes / eps / draws: one entry per stacked part (1 or 2); eps entries may be None (Philox)."""
        parts = len(codes)
        c = list(codes) + [None] * (2 - parts)
        e = list(eps) + [None] * (2 - parts)
        d = list(draws) + [0] * (2 - parts)
        self._check(self.lib.cvae_sample_cat(lat, c[0], c[1] or None, e[0] or None, e[1] or None, seed, d[0], d[1], B, T, lat_dim, ncode,
                                             parts, out, eps_out, stream or None), "cvae_sample_cat")

    def sample_cat_backward(self, dout, lat, eps, B, T, lat_dim, ncode, parts, dlat, stream=0):
        self._check(self.lib.cvae_sample_cat_backward(dout, lat, eps, B, T, lat_dim, ncode, parts, dlat, stream or None),
                    "cvae_sample_cat_backward")

    def stage4_loss(self, rec, reccyc, lat, latcv, x, x_stride, stdim, w, latcv_w, kl_scale, B, T, D, lat_dim, d_rec, d_reccyc, d_lat,
                    d_latcv, frame_loss, loss, accumulate, stream=0):
        self._check(self.lib.cvae_stage4_loss(rec, reccyc or None, lat, latcv or None, x, x_stride, stdim, w, latcv_w or None, kl_scale,
                                              B, T, D, lat_dim, d_rec, d_reccyc or None, d_lat, d_latcv or None, frame_loss, loss,
                                              int(bool(accumulate)), stream or None), "cvae_stage4_loss")

    def mcd_l1(self, x, sx, y, sy, frames, D, frame_mcd, out3, stream=0):
        self._check(self.lib.cvae_mcd_l1(x, sx, y, sy, frames, D, frame_mcd, out3, stream or None), "cvae_mcd_l1")

    def mcd_l1_backward(self, x, sx, y, sy, frames, D, frame_mcd, out3, g3, dx, stream=0):
        self._check(self.lib.cvae_mcd_l1_backward(x, sx, y, sy, frames, D, frame_mcd, out3, g3, dx, stream or None), "cvae_mcd_l1_backward")

    def kl_gauss(self, param, stride, frames, lat_dim, out1, stream=0):
        self._check(self.lib.cvae_kl_gauss(param, stride, frames, lat_dim, out1, stream or None), "cvae_kl_gauss")

    def kl_gauss_backward(self, param, stride, frames, lat_dim, g1, dparam, stream=0):
        self._check(self.lib.cvae_kl_gauss_backward(param, stride, frames, lat_dim, g1, dparam, stream or None), "cvae_kl_gauss_backward")

    def gv_postfilter(self, c, T, D, dpow, gv_trg, cvgv, out, out_var, work, stream=0):
        self._check(self.lib.cvae_gv_postfilter(c, T, D, dpow or None, gv_trg, cvgv, out, out_var or None, work, stream or None),
                    "cvae_gv_postfilter")

    def mc2e(self, mc, is_f64, ld, T, D, alpha, irlen, e_out, stream=0):
        self._check(self.lib.cvae_mc2e(mc, 1 if is_f64 else 0, ld, T, D, alpha, irlen, e_out, stream or None), "cvae_mc2e")

    def mcd_aligned(self, a, lda, b, ldb, rows, D, d0, l2, frames, stats, stream=0):
        self._check(self.lib.cvae_mcd_aligned(a, lda, b, ldb, rows, D, d0, 1 if l2 else 0, frames, stats or None, stream or None),
                    "cvae_mcd_aligned")

    def dtw_work_bytes(self, T1, T2):
        return self.lib.cvae_dtw_work_bytes(T1, T2)

    def dtw_org_to_trg(self, org, trg, T1, T2, D, mcd, aligned, twf, frames, mean_out, work, work_bytes, stream=0):
        self._check(self.lib.cvae_dtw_org_to_trg(org, trg, T1, T2, D, mcd, aligned, twf, frames, mean_out, work, work_bytes,
                                                 stream or None), "cvae_dtw_org_to_trg")

    def step_timing(self, d, B, T, ws, stream=0):
        out = (C.c_double * 8)()
        self._check(self.lib.cvae_step_timing(C.byref(d), B, T, ws, C.byref(out), stream or None), "cvae_step_timing")
        return list(out)

    def set_status_sink(self, ptr):
        self._check(self.lib.cvae_set_status_sink(ptr or None), "cvae_set_status_sink")

    def status_latch(self, latch, stream=0):
        self._check(self.lib.cvae_status_latch(latch, stream or None), "cvae_status_latch")

    def set_draw_origin(self, row0, global_rows, frames_per_row=0):
        self._check(self.lib.cvae_set_draw_origin(row0, global_rows, frames_per_row), "cvae_set_draw_origin")

    def selftest_limbs(self, x_ptr, y_ptr, n, stream=None):
        self._check(self.lib.cvae_selftest_limbs(x_ptr, y_ptr, n, stream), "cvae_selftest_limbs")

    def selftest_occupy(self, blocks, lds_bytes, cycles, stream=None):
        self._check(self.lib.cvae_selftest_occupy(blocks, lds_bytes, cycles, stream), "cvae_selftest_occupy")

    def set_side_stream(self, stream):
        self._check(self.lib.cvae_set_side_stream(stream), "cvae_set_side_stream")

    def join_side_stream(self, stream):
        self._check(self.lib.cvae_join_side_stream(stream), "cvae_join_side_stream")

    def set_draw_parts(self, parts):
        self._check(self.lib.cvae_set_draw_parts(parts), "cvae_set_draw_parts")

    def set_option(self, name, value):
        """Tuning / diagnostic switches by name (include/cyclevae_hip.h lists them); the library reads no environment variable."""
        self._check(self.lib.cvae_set_option(name.encode(), int(value)), "cvae_set_option")

    def get_option(self, name):
        v = C.c_int64(0)
        self._check(self.lib.cvae_get_option(name.encode(), C.byref(v)), "cvae_get_option")
        return v.value

    def reset_options(self):
        self._check(self.lib.cvae_reset_options(), "cvae_reset_options")

    def profile_collect(self):
        ms, n = C.c_double(0.0), C.c_int(0)
        self._check(self.lib.cvae_profile_collect(C.byref(ms), C.byref(n)), "cvae_profile_collect")
        return ms.value, n.value

    def profile_collect_launches(self, cap=4096):
        """[(ms, stacked rows, input channels)] of the eval launches bracketed since the previous collect (FLAG_PROFILE)."""
        ms, rows, cin = (C.c_double * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
        n = self.lib.cvae_profile_collect_launches(ms, rows, cin, cap)
        if n < 0:
            raise CvaeError("cvae_profile_collect_launches failed (%d): %s" % (n, self.lib.cvae_last_error_string().decode()))
        return [(ms[i], rows[i], cin[i]) for i in range(n)]

    TRAIN_PROFILE_CLASSES = ("fwd_recurrence", "bwd_recurrence", "forward_and_dgrad_gemms", "wgrad_gemms")

    def train_profile_collect(self):
        """{class: (summed ms, brackets, summed GEMM flop)} of the launches bracketed since the previous call (option train_profile)."""
        ms, n, fl = (C.c_double * 4)(), (C.c_int * 4)(), (C.c_double * 4)()
        self._check(self.lib.cvae_train_profile_collect(C.byref(ms), C.byref(n), C.byref(fl)), "cvae_train_profile_collect")
        return {name: (ms[i], n[i], fl[i]) for i, name in enumerate(self.TRAIN_PROFILE_CLASSES)}

    def workspace_status(self, ws, stream=0):
        st = (C.c_int32 * 4)()
        self._check(self.lib.cvae_workspace_status(ws, C.byref(st), stream or None), "cvae_workspace_status")
        return list(st)


EXPORTS = ("cvae_last_error_string", "cvae_abi_version", "cvae_ctx_create", "cvae_ctx_destroy", "cvae_set_status_sink", "cvae_status_latch", "cvae_set_draw_origin", "cvae_set_draw_parts",
           "cvae_set_option", "cvae_get_option", "cvae_reset_options", "cvae_selftest_limbs", "cvae_selftest_occupy", "cvae_set_side_stream", "cvae_join_side_stream", "cvae_net_prepared_bytes", "cvae_net_prepare_scratch_bytes",
           "cvae_net_prepare", "cvae_pass_workspace_bytes", "cvae_gru_rnn_forward", "cvae_gru_rnn_forward_stacked", "cvae_gru_rnn_forward_stacked_carry", "cvae_sample", "cvae_sample_laplace", "cvae_sample_laplace_backward",
           "cvae_cycle_workspace_bytes", "cvae_cycle_forward", "cvae_cycle_forward_carry", "cvae_profile_collect", "cvae_profile_collect_launches", "cvae_train_profile_collect", "cvae_step_timing", "cvae_workspace_status",
           "cvae_train_image_bytes", "cvae_net_prepare_train", "cvae_net_prepare_train_v", "cvae_train_variants_needed", "cvae_train_tape_bytes", "cvae_train_scratch_bytes",
           "cvae_gru_rnn_forward_train", "cvae_gru_rnn_backward", "cvae_adam_step", "cvae_adam_step_counted", "cvae_train_debug_counters",
           "cvae_sample_cat", "cvae_sample_cat_backward", "cvae_stage4_loss", "cvae_mcd_l1", "cvae_mcd_l1_backward", "cvae_kl_gauss",
           "cvae_kl_gauss_backward",
           "cvae_gv_postfilter", "cvae_mcd_aligned", "cvae_mc2e", "cvae_dtw_work_bytes", "cvae_dtw_org_to_trg")
