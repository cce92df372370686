"""Drop-in for the reference's src/nets/gru_vae.py on MI355X.

Put this directory on PYTHONPATH instead of $PRJ_ROOT/src/nets (egs/one-to-one/path.sh:11) and the stage-4/5/6
scripts import the same names: GRU_RNN, TwoSidedDilConv1d, sampling_vae_batch, loss_vae, TWFSEloss, initialize.
Constructor arguments, submodule names and therefore state_dict keys are the reference's, so
its checkpoints load unchanged.  GRU_RNN.forward runs on libcyclevae_hip.so (hand-written gfx950 kernels behind
the C ABI in include/cyclevae_hip.h); there is NO eager/CPU fallback -- a CPU tensor or a missing library raises.

SURVEY 8(f) row 4, first variant: the Laplace posterior of the sibling recipes -- sampling_vae_laplace, loss_vae_laplace and the
clamp_vae_laplace flag of GRU_RNN.forward (reference gru_vae.py:101-145, :415-417).
The VQ helpers of the same file (nn_search, nn_search_batch, weighted_ctr: gru_vae.py:148-195) are torch ops, as in the reference;
sampling_vae (:69-82) is the 2-D form of sampling_vae_batch.  Not carried over (dead code): GMM, the relu_vae
(variance-parameter) branches and the forward flags noise/res/softmax/sigmoid/exp/scale_in_out; they raise NotImplementedError.
"""
import torch
from torch import nn

import _cabi

_LIB = None      # the context of the first device used (kept under this name for callers that only ever see one GPU)
_SINK = None     # its status sink: pinned host int32[4] the kernels write when a hand-off spin times out (cvae_set_status_sink)
_LIBS = {}       # CUDA device index -> _cabi.CvaeLib: ONE library context (cvae_ctx, ABI 6) per device -- its own status sink, draw
_SINKS = {}      # origin, options and side stream, so two modules on two devices never share a setting


def _lib():
    """The library context of the CURRENT device (torch.cuda.current_device(); -1 = no GPU: import / size queries only)."""
    global _LIB, _SINK
    dev = torch.cuda.current_device() if torch.cuda.is_available() else -1
    lib = _LIBS.get(dev)
    if lib is None:
        lib = _LIBS[dev] = _cabi.CvaeLib() if _LIB is None else _LIB.new_context()      # raises when libcyclevae_hip.so is absent
        if dev >= 0:
            sink = _SINKS[dev] = torch.zeros(4, dtype=torch.int32).pin_memory()
            lib.set_status_sink(sink.data_ptr())
        if _LIB is None:
            _LIB, _SINK = lib, _SINKS.get(dev)
    return lib


def _sink():
    """The status sink of the current device's context (None before the first library call on it)."""
    return _SINKS.get(torch.cuda.current_device()) if torch.cuda.is_available() else None


_status_owned = 0   # > 0 while a stage4.Stage4Step(sync=False) call is enqueuing: the status word is read through its device latch only


def check_status(sync=False, overflow_ok=False):
    """Raise if a persistent kernel gave up waiting for another block (bounded spin, cvae_kernels.h: it then runs to the end
    on whatever it had, so everything computed since is garbage), or (status 5) if a gate gradient left the range of the limb
    exchange of the persistent reverse recurrence.  Every entry point of this module calls it before enqueuing new work, which
    costs one host read of pinned memory; sync=True first waits for the current stream, for callers that are about to consume
    results on the host.  overflow_ok: leave status 5 standing for the caller that handles it (stage4.Stage4Step repeats such a
    step on the fp32 reverse recurrence).  While a Stage4Step that does not synchronise per step is enqueuing (_status_owned) this
    is a no-op: earlier steps may still be running, and a host-side clear would race with the device-side gate of their update."""
    sink = _sink()
    if sink is None or _status_owned:
        return
    if sync:
        torch.cuda.current_stream().synchronize()
    code = int(sink[0])
    if code == 5 and overflow_ok:
        return
    if code != 0:
        sink.zero_()
        if code == 5:
            raise _cabi.CvaeError("a gate gradient of the persistent reverse recurrence left the range of its limb exchange (|v| >= "
                                  "~234, status 5): the gradients of this backward are invalid; run it again with the library option "
                                  "train_bwd_per_step = 1 (fp32 exchange), as stage4.Stage4Step does by itself")
        raise _cabi.CvaeError("a hand-off spin of a persistent recurrent kernel timed out (status %d): the results of the "
                              "passes enqueued since the previous check are invalid" % code)


def set_draw_origin(row0, global_rows, frames_per_row=0):
    """Data-parallel ranks: this process holds batch rows row0 .. of a job with global_rows rows; the on-device Philox streams
    (latent draws, dropout masks) are keyed by GLOBAL row so that results do not depend on the number of ranks.
    frames_per_row = T makes stand-alone sampling_vae_batch calls on [B,T,2L] follow the same numbering."""
    _lib().set_draw_origin(int(row0), int(global_rows), int(frames_per_row))


_side_pending = []   # tapes of backward passes whose weight-gradient GEMMs may still be running on the side stream


def set_side_stream(stream):
    """stream: a torch.cuda.Stream, or None to switch the overlap off.  Modules with `_grad_sink = True` then accumulate their
    parameter gradients straight into `p.grad` (which must exist) and run the weight-gradient GEMMs of a backward pass on
    `stream`, under the next pass's reverse recurrence; join_side_stream() must be called before the gradients are used."""
    _lib().set_side_stream(None if stream is None else stream.cuda_stream)


def join_side_stream():
    _lib().join_side_stream(_stream())
    del _side_pending[:]


_concurrent = {}     # (device index, handle of the stream it was probed against, slot) -> torch.cuda.Stream


def concurrent_stream(device=None, slot=0, beside=None):
    """A stream whose kernels REALLY run beside the current stream's (and beside `beside`, a list of streams already taken).
    HIP multiplexes its streams onto a few hardware queues, and two streams that land on the same queue serialise: measured on
    MI355X (tools/r6/stream_queues.py) the 7th stream torch hands out shares the null stream's queue -- a stage-4 step whose
    weight-gradient GEMMs sit on it takes 27.0 instead of 22.7 ms, every step, with nothing else different.  So candidates are
    PROBED: a one-block spin kernel (cvae_selftest_occupy, ~0.2 ms) on each of two streams takes the time of one when they are
    concurrent and of two when they are not; the first candidate that overlaps with all of them is kept (cached per device, probed
    stream and slot).  Costs a millisecond once; called where a side stream is created, never inside a step."""
    import time
    cur = torch.cuda.current_stream(device)
    idx = cur.device.index
    key = (idx, cur.cuda_stream, slot)
    got = _concurrent.get(key)
    if got is not None:
        return got
    lib = _lib()
    others = [cur] + list(beside or [])
    spin = 400000

    def wall(a, b):
        torch.cuda.synchronize(idx)
        t0 = time.perf_counter()
        lib.selftest_occupy(1, 1024, spin, a.cuda_stream)
        lib.selftest_occupy(1, 1024, spin, b.cuda_stream)
        torch.cuda.synchronize(idx)
        return time.perf_counter() - t0

    wall(cur, cur)                                 # (first launch of the kernel: module load)
    alone = min(wall(cur, cur) for _ in range(2)) / 2.0
    best = None
    for _ in range(12):
        cand = torch.cuda.Stream(cur.device)
        worst = max(min(wall(o, cand) for _ in range(2)) for o in others)
        if best is None or worst < best[0]:
            best = (worst, cand)
        if worst < 1.4 * alone:
            break
    got = _concurrent[key] = best[1]
    return got


def set_draw_parts(parts):
    """The batch of the following train-mode passes is `parts` stacked copies of this process's rows (stage4: rec || cv as one
    decoder launch); keeps the Philox dropout masks keyed by global row per copy."""
    _lib().set_draw_parts(int(parts))


def _stream():
    return torch.cuda.current_stream().cuda_stream


_flags_extra = 0   # bench.py ORs in _cabi.FLAG_PROFILE for its timed region
# Kernel selection of the eval passes: module attributes, set through set_kernel(); NO environment variable is read.
_force_kernel = None       # "exact3" (default) | "split2" | "fp32"
_persistent = True         # False: per-step launches instead of the all-resident recurrent kernels (tests, bench --no-persistent)
_hoisted_frontend = False  # True: front-end as a GEMM before the recurrence (measurement)


def set_kernel(kernel=None, persistent=None, hoisted_frontend=None):
    """Matrix products of the persistent eval kernel:
         exact3 (default)  fp32 operands carried exactly as three fp16 limbs, six f16 MFMAs per product (k_gru_steps_v6)
         split2            (hi, lo) fp16 pairs = 22-bit operands, three f16 MFMAs per product (k_gru_steps_v5)
         fp32              v_mfma_f32_16x16x4_f32 on the fp32 operands themselves (k_gru_steps_v4)
    None leaves a setting as it is.  Returns the previous (kernel, persistent, hoisted_frontend)."""
    global _force_kernel, _persistent, _hoisted_frontend
    prev = (_force_kernel, _persistent, _hoisted_frontend)
    if kernel is not None:
        if kernel not in ("exact3", "split2", "fp32"):
            raise ValueError("kernel must be exact3, split2 or fp32, got %r" % (kernel,))
        _force_kernel = kernel
    if persistent is not None:
        _persistent = bool(persistent)
    if hoisted_frontend is not None:
        _hoisted_frontend = bool(hoisted_frontend)
    return prev


def _flags():
    f = _cabi.FLAG_PERSISTENT if _persistent else 0
    if _hoisted_frontend:
        f |= _cabi.FLAG_HOISTED_FRONTEND
    kern = _force_kernel or "exact3"
    if kern not in ("exact3", "split2", "fp32"):
        raise ValueError("kernel must be exact3, split2 or fp32, got %r" % (kern,))
    if kern == "exact3":
        f |= _cabi.FLAG_EXACT3 | _cabi.FLAG_SPLIT_F16
    elif kern == "split2":
        f |= _cabi.FLAG_SPLIT_F16
    return f | _flags_extra


def _need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: tensor is on %s; this build of gru_vae runs on the HIP device only "
                           "(no CPU fallback)" % (what, t.device))


def initialize(m):
    """Xavier-uniform weights, zero biases (reference gru_vae.py:21-33)."""
    for name, param in m.named_parameters():
        if "weight" in name:
            nn.init.xavier_uniform_(param)
        elif "bias" in name:
            nn.init.constant_(param, 0.0)


class TwoSidedDilConv1d(nn.Module):
    """Parameter container with the reference's layout (gru_vae.py:36-51): conv.0 is k=ks, dilation 1, zero-padded by
    (ks^layers-1)/2 on both sides; conv.i has dilation ks^i and no padding.  The arithmetic happens inside
    GRU_RNN.forward, folded into the input-gate GEMM."""

    def __init__(self, in_dim=39, kernel_size=3, layers=2):
        super(TwoSidedDilConv1d, self).__init__()
        self.in_dim, self.kernel_size, self.layers = in_dim, kernel_size, layers
        self.rec_field = kernel_size ** layers
        self.padding = int((self.rec_field - 1) / 2)
        self.conv = nn.ModuleList()
        for i in range(layers):
            self.conv += [nn.Conv1d(in_dim * kernel_size ** i, in_dim * kernel_size ** (i + 1), kernel_size, stride=1,
                                    dilation=kernel_size ** i, padding=self.padding if i == 0 else 0)]

    def forward(self, x):
        """x [B, C, T] -> [B, C * ks^layers, T], the reference's own statements (gru_vae.py:53-66): conv.0 then conv.1 ... as torch
        convolutions on whatever device x lives on.  GRU_RNN.forward never calls this (its front-end is the folded 9-tap GEMM
        inside the HIP pass); it exists so that the class is usable on its own, as in the reference."""
        x = self.conv[0](x)
        for i in range(1, self.layers):
            x = self.conv[i](x)
        return x


def _weight_fields(mod, device):
    """{C field: fp32 tensor} of the module's parameters in the library's naming (_cabi.STATE_KEYS), looked up attribute by attribute
    -- a state_dict() walk per pass cost the host ~80 us, twenty times per training step.  Parameters may be REPLACED
    between calls (train...:344-347 assigns new nn.Parameter objects to scale_in / scale_out), so nothing is cached across calls."""
    fields = {}
    for f, k in _cabi.STATE_KEYS.items():
        head, _, leaf = k.rpartition(".")
        m = mod
        try:
            for part in head.split("."):
                m = m[int(part)] if part.isdigit() else getattr(m, part)
        except (AttributeError, IndexError):
            continue
        t = m._parameters.get(leaf)
        if t is None:
            continue
        if t.device != device or t.dtype != torch.float32:
            raise RuntimeError("parameter %s is %s/%s, expected float32 on %s" % (k, t.device, t.dtype, device))
        fields[f] = t.detach() if t.is_contiguous() else t.detach().contiguous()
    return fields


class _Prepared(object):
    """Device weight image of one GRU_RNN, rebuilt when any parameter's storage or version changes."""

    def __init__(self):
        self.key = None
        self.image = None
        self.desc = None
        self.ws = {}

    def get(self, mod, device):
        lib = _lib()
        fields = _weight_fields(mod, device)
        key = tuple((f, t.data_ptr(), t._version) for f, t in sorted(fields.items()))
        if key != self.key:
            d = lib.desc(mod.in_dim, mod.out_dim, mod.hidden_units, mod.kernel_size, mod.dilation_size,
                         mod.scale_in_flag, mod.scale_out_flag)
            image = torch.empty(lib.prepared_bytes(d), dtype=torch.uint8, device=device)
            scratch = torch.empty(lib.prepare_scratch_bytes(d), dtype=torch.uint8, device=device)
            lib.net_prepare(d, {f: t.data_ptr() for f, t in fields.items()}, image.data_ptr(), image.numel(),
                            scratch.data_ptr(), scratch.numel(), _stream())
            self.key, self.image, self.desc, self._keep = key, image, d, (fields, scratch)
        return self.desc, self.image

    def workspace(self, B, T, device):
        """One buffer per device, grown to the largest (B, T) seen; a replaced buffer goes back to the caching allocator, which
        orders its reuse after the work already queued on the stream that used it (the eval passes run on one stream)."""
        need = _lib().pass_workspace_bytes(self.desc, B, T)
        buf = self.ws.get(device)
        if buf is None or buf.numel() < need:
            buf = self.ws[device] = torch.empty(need, dtype=torch.uint8, device=device)
        return buf


_TRAIN_PARAMS = (("conv0_w", "conv.conv.0.weight"), ("conv0_b", "conv.conv.0.bias"), ("conv1_w", "conv.conv.1.weight"),
                 ("conv1_b", "conv.conv.1.bias"), ("w_ih", "gru.weight_ih_l0"), ("w_hh", "gru.weight_hh_l0"),
                 ("b_ih", "gru.bias_ih_l0"), ("b_hh", "gru.bias_hh_l0"), ("out_w", "out_1.weight"), ("out_b", "out_1.bias"))


class _PreparedTrain(object):
    """Train-mode weight image (rebuilt after every optimiser step) and the scratch buffer shared by a net's passes."""

    def __init__(self):
        self.key = None
        self.image = None
        self.desc = None
        self.scratch = None     # {(device, slot): buffer}, grown to the largest shape seen
        self.variants = 0       # MFMA-order weight images the train image holds (cvae_net_prepare_train_v), grown on demand
        self.ready = None       # event behind the image when it was built on ANOTHER stream (stage4: the side stream); the next get() waits for it

    def get(self, mod, device, p_drop=0.0, rows_frames=None):
        """p_drop: the dropout probability of the passes that will run on the image (folded into the feedback weights of the
        exact-operand forward recurrence, cvae_net_prepare_train).  rows_frames = (B, T) of the pass about to run: the image is
        (re)built with the MFMA-order weight images that shape needs on top of those already in it (a net that only ever sees
        passes of at most three rows -- the recipe's batch_size_utt = 1 -- never builds any); None: whatever it held last."""
        lib = _lib()
        if self.ready is not None:
            # whoever uses the image next -- a Stage4Step pass, a direct GRU_RNN forward, another step object sharing the module --
            # runs behind the stream that built it (only Stage4Step._run used to wait)
            torch.cuda.current_stream().wait_event(self.ready)
            self.ready = None
        fields = _weight_fields(mod, device)
        d = self.desc
        if d is None:
            d = lib.desc(mod.in_dim, mod.out_dim, mod.hidden_units, mod.kernel_size, mod.dilation_size,
                         mod.scale_in_flag, mod.scale_out_flag)
        if rows_frames is not None:
            self.variants |= lib.train_variants_needed(d, int(rows_frames[0]), int(rows_frames[1]))
        key = tuple((f, t.data_ptr(), t._version) for f, t in sorted(fields.items())) + (float(p_drop), self.variants)
        if key != self.key:
            image = torch.empty(lib.train_image_bytes(d), dtype=torch.uint8, device=device)
            lib.net_prepare_train(d, {f: t.data_ptr() for f, t in fields.items()}, image.data_ptr(), image.numel(), _stream(),
                                  gru_drop_p=float(p_drop), variants=self.variants)
            self.key, self.image, self.desc, self._keep = key, image, d, fields
        return self.desc, self.image

    def scratch_for(self, B, T, device, slot=0):
        """slot 1: the second buffer consecutive backward passes alternate with while their weight-gradient GEMMs are still
        running on the side stream (set_side_stream).  ONE buffer per slot, grown to the largest (B, T) seen and reused for every
        smaller shape (the contents need not survive a call; the layout is recomputed per call): real training has a different
        window length at the end of every utterance batch, and the passes of a step alternate between B and 2B rows (the stacked
        rec || cv pass).  A buffer is replaced only after the launch stream has joined the side stream, so no queued
        weight-gradient GEMM can still read the old one when the caching allocator hands it out again."""
        need = _lib().train_scratch_bytes(self.desc, B, T)
        if self.scratch is None:
            self.scratch = {}
        k = (device, slot)
        buf = self.scratch.get(k)
        if buf is None or buf.numel() < need:
            if buf is not None:
                _lib().join_side_stream(_stream())
            buf = self.scratch[k] = torch.empty(need, dtype=torch.uint8, device=device)
        return buf


class _TrainPass(torch.autograd.Function):
    """One train-mode GRU_RNN pass on the HIP kernels with a hand-written backward (cvae_gru_rnn_backward), so the caller's
    `batch_loss.backward()` (reference train_gru_cyclevae_gauss_batch.py:1419) reaches x and the trainable parameters.
    y_last / h_last are carries the reference detaches (train...:1301): non-differentiable outputs."""

    @staticmethod
    def forward(ctx, mod, x, y_in, h_in, clamp_lat_dim, p_drop, masks, *params):
        lib = _lib()
        dev = x.device
        B, T, _ = x.shape
        d, image = mod._prep_train.get(mod, dev, p_drop, (B, T))
        scratch = mod._prep_train.scratch_for(B, T, dev)
        tape = torch.empty(lib.train_tape_bytes(d, B, T), dtype=torch.uint8, device=dev)
        trj = torch.empty(B, T, mod.out_dim, dtype=torch.float32, device=dev)
        y_last = torch.empty(B, 1, mod.out_dim, dtype=torch.float32, device=dev)
        h_last = torch.empty(1, B, mod.hidden_units, dtype=torch.float32, device=dev)
        cm = gm = None
        if masks is not None:
            cm, gm = (m.to(torch.float32).contiguous() for m in masks)
        lib.forward_train(d, image.data_ptr(), x.data_ptr(), y_in.data_ptr(), None if h_in is None else h_in.data_ptr(), B, T,
                          clamp_lat_dim, None if cm is None else cm.data_ptr(), None if gm is None else gm.data_ptr(),
                          _draw_seed(), float(p_drop), trj.data_ptr(), y_last.data_ptr(), h_last.data_ptr(), tape.data_ptr(),
                          tape.numel(), scratch.data_ptr(), scratch.numel(), _stream())
        ctx.mod, ctx.tape, ctx.image, ctx.desc = mod, tape, image, d
        ctx.dims = (B, T, clamp_lat_dim)
        ctx.param_shapes = [p.shape for p in params]
        ctx.x_shape = x.shape
        ctx.mark_non_differentiable(y_last, h_last)
        # (the engine would hand backward() freshly zero-filled tensors for the two carries: two fill launches per pass)
        ctx.set_materialize_grads(False)
        return trj, y_last, h_last

    @staticmethod
    def backward(ctx, dtrj, _dy, _dh):
        lib = _lib()
        check_status(overflow_ok=True)
        B, T, clamp = ctx.dims
        mod = ctx.mod
        if dtrj is None:         # (nothing downstream used the trajectory)
            dtrj = torch.zeros(B, T, mod.out_dim, dtype=torch.float32, device=ctx.tape.device)
        dev = dtrj.device
        dout = dtrj.to(torch.float32).contiguous()
        dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        if getattr(mod, "_grad_sink", False):
            # accumulate into p.grad (stage4.Stage4Step: views of one flat buffer); the recurrent weight-gradient GEMMs go to the
            # side stream, so consecutive backward passes of this net alternate between two scratch buffers
            plist = _train_param_list(mod)
            mod._bwd_slot = 1 - getattr(mod, "_bwd_slot", 1)
            scratch = mod._prep_train.scratch_for(B, T, dev, mod._bwd_slot)
            lib.backward(ctx.desc, ctx.image.data_ptr(), dout.data_ptr(), B, T, clamp, ctx.tape.data_ptr(), scratch.data_ptr(),
                         scratch.numel(), None if dx is None else dx.data_ptr(),
                         {f: p.grad.data_ptr() for (f, _), p in zip(_TRAIN_PARAMS, plist)}, True, _stream())
            _side_pending.append((ctx.tape, dout))
            ctx.tape = None
            return (None, dx, None, None, None, None, None) + (None,) * len(ctx.param_shapes)
        if _auto_sink_ok(ctx, mod):
            # plain `loss.backward()` of an unchanged training script (train...:1419): the same direct accumulation into p.grad with the
            # weight-gradient GEMMs on a side stream, joined by a callback the autograd engine runs before backward() returns
            plist = _train_param_list(mod)
            for p in plist:
                if p.grad is None:
                    p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
            _auto_sink_begin(dev)
            mod._bwd_slot = 1 - getattr(mod, "_bwd_slot", 1)
            scratch = mod._prep_train.scratch_for(B, T, dev, mod._bwd_slot)
            lib.backward(ctx.desc, ctx.image.data_ptr(), dout.data_ptr(), B, T, clamp, ctx.tape.data_ptr(), scratch.data_ptr(),
                         scratch.numel(), None if dx is None else dx.data_ptr(),
                         {f: p.grad.data_ptr() for (f, _), p in zip(_TRAIN_PARAMS, plist)}, True, _stream())
            _side_pending.append((ctx.tape, dout))
            ctx.tape = None
            return (None, dx, None, None, None, None, None) + (None,) * len(ctx.param_shapes)
        scratch = mod._prep_train.scratch_for(B, T, dev)
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.param_shapes]
        lib.backward(ctx.desc, ctx.image.data_ptr(), dout.data_ptr(), B, T, clamp, ctx.tape.data_ptr(), scratch.data_ptr(),
                     scratch.numel(), None if dx is None else dx.data_ptr(),
                     {f: g.data_ptr() for (f, _), g in zip(_TRAIN_PARAMS, grads)}, False, _stream())
        ctx.tape = None
        out = [g if need else None for g, need in zip(grads, ctx.needs_input_grad[7:])]
        return (None, dx, None, None, None, None, None) + tuple(out)


# ---- plain autograd flows (the unchanged training script): parameter gradients straight into p.grad, weight-gradient GEMMs on a side stream
_backward_overlap = True    # set_backward_overlap(False): every backward pass returns its parameter gradients to autograd (round 4's flow)
_backward_overlap_dist = False   # True: also with an initialised process group of more than one rank (see set_backward_overlap)
_auto_side = {}             # CUDA device index -> the side stream of that device's plain-autograd backward passes
_auto_task = [None]         # the autograd graph task whose end-of-backward callback has been queued


def set_backward_overlap(on, with_process_group=False):
    """Plain `loss.backward()` through GRU_RNN passes (no stage4.Stage4Step): True (default) lets every pass add its ten parameter
    gradients straight into p.grad and run its weight-gradient GEMMs on a second stream, under the next pass's reverse recurrence,
    joined before backward() returns (what Stage4Step(overlap_wgrad=True) does for its own flow); False returns them to autograd
    as tensors (one allocation + one AccumulateGrad add per parameter and pass).  Returns the previous setting.
    RESTRICTION: p.grad is complete only when backward() has returned.  Anything that reads it from a hook on the parameter's
    AccumulateGrad node -- the DDP / FSDP reducer, an optimizer-in-backward -- would see it early, so the direct accumulation
    switches itself off for parameters with post-accumulate-grad hooks and whenever a process group of more than one rank is
    initialised; with_process_group=True keeps it on there, for data-parallel loops that all-reduce after backward()."""
    global _backward_overlap, _backward_overlap_dist
    prev, _backward_overlap = _backward_overlap, bool(on)
    _backward_overlap_dist = bool(with_process_group)
    return prev


def _train_param_list(mod):
    """The ten trainable parameters in _TRAIN_PARAMS order, attribute by attribute: `dict(mod.named_parameters())` walks the module
    tree (~60 us), and a one-utterance step of the unchanged script did that thirty times."""
    return [mod.conv.conv[0].weight, mod.conv.conv[0].bias, mod.conv.conv[1].weight, mod.conv.conv[1].bias, mod.gru.weight_ih_l0,
            mod.gru.weight_hh_l0, mod.gru.bias_ih_l0, mod.gru.bias_hh_l0, mod.out_1.weight, mod.out_1.bias]


def _accumulate_nodes(mod):
    """The AccumulateGrad nodes of the module's trainable parameters (cached; looked up under enable_grad: backward runs without)."""
    cache = getattr(mod, "_acc_nodes", None)
    params = _train_param_list(mod)
    key = tuple(id(p) for p in params)
    if cache is None or cache[0] != key:
        with torch.enable_grad():
            nodes = [p.view_as(p).grad_fn.next_functions[0][0] if p.requires_grad else None for p in params]
        cache = mod._acc_nodes = (key, nodes, params)
    return cache[1], cache[2]


def _auto_sink_ok(ctx, mod):
    """True when this backward pass may add into p.grad itself: the engine is running a plain backward() that WILL accumulate into every
    trainable parameter of the module (not torch.autograd.grad / backward(inputs=...), whose callers expect the gradients back), the
    parameters are ordinary fp32 leaves without hooks, and the switch is on."""
    if not _backward_overlap or not all(ctx.needs_input_grad[7:]):
        return False
    # The engine still runs every parameter's AccumulateGrad node (with an undefined gradient) as soon as this pass returns, and with
    # it every hook on that node: a DDP / FSDP reducer or an optimizer-in-backward would then read p.grad BEFORE the side-stream
    # GEMMs of this pass have finished.  Hooks added from C++ (the DDP reducer's) cannot be listed from Python, so the
    # direct accumulation is only used where nothing of that kind can be attached: no post-accumulate-grad hooks on the parameters,
    # and no initialised process group of more than one rank (set_backward_overlap(True, with_process_group=True) lifts the second
    # condition for callers that all-reduce the gradients themselves AFTER backward() has returned, as stage4.Stage4Step does).
    if not _backward_overlap_dist and torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1:
        return False
    nodes, params = _accumulate_nodes(mod)
    for n, p in zip(nodes, params):
        if n is None or p.dtype != torch.float32 or not p.is_contiguous() or p._backward_hooks:
            return False
        if getattr(p, "_post_accumulate_grad_hooks", None):
            return False
        if p.grad is not None and (p.grad.dtype != torch.float32 or not p.grad.is_contiguous() or p.grad.requires_grad):
            return False
        if not torch._C._will_engine_execute_node(n):
            return False
    return True


def _auto_sink_begin(dev):
    """Side stream of the device set in the library; ONE callback per autograd run joins it and gives the launch stream back."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    side = _auto_side.get(idx)
    if side is None:
        with torch.cuda.device(idx):
            side = _auto_side[idx] = concurrent_stream(dev)
    _lib().set_side_stream(side.cuda_stream)
    task = torch._C._current_graph_task_id()
    if _auto_task[0] != task:
        _auto_task[0] = task

        def done():
            _auto_task[0] = None
            join_side_stream()
            _lib().set_side_stream(None)
        torch.autograd.Variable._execution_engine.queue_callback(done)


class GRU_RNN(nn.Module):
    """Conv front-end -> frame-stepped autoregressive GRU -> 1x1 projection (reference gru_vae.py:265-455)."""

    def __init__(self, in_dim=39, out_dim=35, hidden_units=1024, hidden_layers=1, kernel_size=3, dilation_size=2,
                 do_prob=0, scale_in_flag=True, scale_out_flag=True, scale_in_out_flag=False):
        super(GRU_RNN, self).__init__()
        if hidden_layers != 1:
            raise NotImplementedError("hidden_layers=%d: the recipe uses a single GRU layer" % hidden_layers)
        if scale_in_out_flag:
            raise NotImplementedError("scale_in_out_flag is dead code in this recipe")
        self.in_dim, self.out_dim = in_dim, out_dim
        self.hidden_units, self.hidden_layers = hidden_units, hidden_layers
        self.kernel_size, self.dilation_size, self.do_prob = kernel_size, dilation_size, do_prob
        self.scale_in_flag, self.scale_out_flag, self.scale_in_out_flag = scale_in_flag, scale_out_flag, False
        if scale_in_flag:
            self.scale_in = nn.Conv1d(in_dim, in_dim, 1)
        self.conv = TwoSidedDilConv1d(in_dim=in_dim, kernel_size=kernel_size, layers=dilation_size)
        self.receptive_field = self.conv.rec_field
        self.tot_in_dim = in_dim * self.receptive_field + out_dim
        if do_prob > 0:
            self.conv_drop = nn.Dropout(p=do_prob)
        self.gru = nn.GRU(self.tot_in_dim, hidden_units, hidden_layers, batch_first=True)  # parameter container
        if do_prob > 0:
            self.gru_drop = nn.Dropout(p=do_prob)
        self.out_1 = nn.Conv1d(hidden_units, out_dim, 1)
        if scale_out_flag:
            self.scale_out = nn.Conv1d(out_dim, out_dim, 1)
        self._prep = _Prepared()
        self._prep_train = _PreparedTrain()
        self._debug_masks = None   # tests: (conv_mask [B,T,9Cin], gru_mask [T,B,H]) consumed by the next train-mode pass

    def prepared(self, device):
        return self._prep.get(self, device)

    def weights_changed(self):
        """Tell the module that its parameters were rewritten behind torch's version counters (a library kernel updating them
        through raw pointers, stage4.Stage4Step's flat Adam): the device weight images are rebuilt at the next pass."""
        self._prep.key = None
        self._prep_train.key = None

    def forward(self, x, y_in, softmax=False, sigmoid=False, exp=False, h_in=None, noise=0, res=False, res_stdim=0,
                res_endim=35, do=False, clamp_vae=False, relu_vae=False, lat_dim=16, clamp_vae_laplace=False):
        if softmax or sigmoid or exp or noise > 0 or res or relu_vae:
            raise NotImplementedError("forward flag outside the CycleVAE recipe (dead code in the reference)")
        # clamp of the second half of the outputs (gru_vae.py:408-417): clamp_vae wins over clamp_vae_laplace, as in the reference
        clamp = lat_dim if clamp_vae else ((lat_dim | _cabi.CLAMP_LAPLACE) if clamp_vae_laplace else -1)
        _need_cuda(x, "GRU_RNN.forward(x)")
        _lib()
        check_status()
        # nn.Dropout is the identity after model.eval() (reference conv_drop / gru_drop, gru_vae.py:355,380): `do` alone
        # does not switch it on
        p_drop = float(self.do_prob) if (self.do_prob > 0 and do and self.training) else 0.0
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad or p_drop > 0:
            return self._forward_train(x, y_in, h_in, p_drop, clamp)
        two_d = x.dim() == 2
        if two_d:
            x = x.unsqueeze(0)
        x = x.to(torch.float32).contiguous()
        B, T, Cin = x.shape
        if Cin != self.in_dim:
            raise ValueError("input has %d features, network expects %d" % (Cin, self.in_dim))
        dev = x.device
        d, image = self._prep.get(self, dev)
        ws = self._prep.workspace(B, T, dev)
        y0 = y_in.to(torch.float32).reshape(B, self.out_dim).contiguous()
        h0 = None if h_in is None else h_in.to(torch.float32).reshape(B, self.hidden_units).contiguous()
        trj = torch.empty(B, T, self.out_dim, dtype=torch.float32, device=dev)
        y_last = torch.empty(B, 1, self.out_dim, dtype=torch.float32, device=dev)
        h_last = torch.empty(1, B, self.hidden_units, dtype=torch.float32, device=dev)
        lib = _lib()
        pin = lib.pass_input((x.data_ptr(), Cin, Cin))
        lib.gru_rnn_forward(d, image.data_ptr(), pin, y0.data_ptr(), None if h0 is None else h0.data_ptr(), B, T,
                            clamp, trj.data_ptr(), y_last.data_ptr(), h_last.data_ptr(),
                            ws.data_ptr(), ws.numel(), _flags(), _stream())
        if two_d:
            trj = trj.squeeze(0)
        return trj, y_last, h_last


def _forward_train(self, x, y_in, h_in, p_drop, clamp_lat_dim):
    """Train-mode pass (dropout and/or autograd): reference gru_vae.py:353-355, :376-382 on the HIP training kernels."""
    two_d = x.dim() == 2
    if two_d:
        x = x.unsqueeze(0)
    x = x.to(torch.float32).contiguous()
    B, T, Cin = x.shape
    if Cin != self.in_dim:
        raise ValueError("input has %d features, network expects %d" % (Cin, self.in_dim))
    y0 = y_in.detach().to(torch.float32).reshape(B, self.out_dim).contiguous()
    h0 = None if h_in is None else h_in.detach().to(torch.float32).reshape(B, self.hidden_units).contiguous()
    params = _train_param_list(self)
    masks, self._debug_masks = self._debug_masks, None
    trj, y_last, h_last = _TrainPass.apply(self, x, y0, h0, clamp_lat_dim, p_drop, masks, *params)
    if two_d:
        trj = trj.squeeze(0)
    return trj, y_last, h_last


GRU_RNN._forward_train = _forward_train


def _draw_seed():
    # one 62-bit seed per call from torch's CPU generator: torch.manual_seed() keeps runs reproducible
    return int(torch.randint(0, 2 ** 62, (1,)).item())


class _SampleVAE(torch.autograd.Function):
    """z = mu + exp(s/2) * eps with eps drawn on the device, ONE launch forward (cvae_sample) and one backward
    (cvae_sample_cat_backward with no code columns): d mu = dz, d s = dz * eps * exp(s/2) / 2.  As torch ops the same map is a slice,
    an exp, a mul and an add plus their autograd nodes: ~14 small launches per draw in the unchanged training script."""

    @staticmethod
    def forward(ctx, param, lat_dim, seed):
        lib = _lib()
        p = param.detach().to(torch.float32).contiguous()
        rows = p.numel() // p.shape[-1]
        z = torch.empty(p.shape[:-1] + (lat_dim,), dtype=torch.float32, device=p.device)
        eps = torch.empty_like(z)
        lib.sample(p.data_ptr(), rows, lat_dim, None, seed, 0, z.data_ptr(), eps.data_ptr(), _stream())
        ctx.save_for_backward(p, eps)
        ctx.dims = (rows, lat_dim, param.shape, param.dtype)
        return z

    @staticmethod
    def backward(ctx, dz):
        p, eps = ctx.saved_tensors
        rows, L, shape, dtype = ctx.dims
        dlat = torch.empty_like(p)
        _lib().sample_cat_backward(dz.to(torch.float32).contiguous().data_ptr(), p.data_ptr(), eps.data_ptr(), rows, 1, L, 0, 1,
                                   dlat.data_ptr(), _stream())
        return dlat.view(shape).to(dtype), None, None


def sampling_vae_batch(param, lat_dim=None, training=False, relu_vae=False):
    """z = mu + exp(log_var/2) * eps with eps ~ N(0,1) drawn on device by Philox4x32-10 (reference gru_vae.py:85-98
    draws eps on the CPU generator and copies it over)."""
    if relu_vae:
        raise NotImplementedError("relu_vae is dead code in this recipe")
    _need_cuda(param, "sampling_vae_batch(param)")
    if lat_dim is None:
        lat_dim = int(param.shape[-1] / 2)
    p = param.to(torch.float32).contiguous()
    rows = p.numel() // p.shape[-1]
    z = torch.empty(p.shape[:-1] + (lat_dim,), dtype=torch.float32, device=p.device)
    lib = _lib()
    if torch.is_grad_enabled() and param.requires_grad:
        if param.shape[-1] != 2 * lat_dim:
            raise ValueError("sampling_vae_batch: the last axis has %d entries, expected 2 * lat_dim = %d" % (param.shape[-1], 2 * lat_dim))
        return _SampleVAE.apply(param, lat_dim, _draw_seed())
    lib.sample(p.data_ptr(), rows, lat_dim, None, _draw_seed(), 0, z.data_ptr(), None, _stream())
    return z


def sampling_vae(param, lat_dim=None, training=False, relu_vae=False):
    """The 2-D form of sampling_vae_batch (reference gru_vae.py:69-82, used by the sibling many-to-many recipes): param [N, 2L]
    -> z [N, L], the same device draw."""
    if lat_dim is None:
        lat_dim = int(param.shape[1] / 2)
    return sampling_vae_batch(param, lat_dim=lat_dim, training=training, relu_vae=relu_vae)


# ---- SURVEY 8(f) row 4, VQ helpers of the sibling recipes (reference gru_vae.py:148-195): plain torch ops on whatever device the
# tensors live on, the reference's statements without its [T, K, D] `repeat` copies (broadcasting gives the same numbers).  Off the
# hot path (nothing in egs/one-to-one calls them); pinned by tests/golden/vq.npz, recorded from the reference.  (The reference's GMM class, :202-262, is
# not carried over: its forward fails on a freshly constructed module -- `wghts.weight` is [K, 1] where the code repeats a [K] vector.)
def nn_search(encoding, centroids):
    """Index of the L1-nearest centroid per frame: encoding [T, D], centroids [K, D] -> [T] int64 (gru_vae.py:148-160)."""
    return torch.argmin(torch.sum((encoding.unsqueeze(1) - centroids.unsqueeze(0)).abs(), 2), dim=-1)


def nn_search_batch(encoding, centroids):
    """encoding [B, T, D], centroids [K, D] -> [B, T] int64 (gru_vae.py:163-176)."""
    return torch.argmin(torch.sum((encoding.unsqueeze(2) - centroids.unsqueeze(0).unsqueeze(0)).abs(), 3), dim=-1)


def weighted_ctr(encoding, centroids):
    """Posterior-weighted centroid per frame and the mean weighted L1 distance (gru_vae.py:179-195)."""
    dist = torch.sum((encoding.unsqueeze(1) - centroids.unsqueeze(0)).abs(), 2)       # T x K
    score = torch.exp(-dist)
    post = score / torch.sum(score, 1).unsqueeze(1)
    weighted_centroids = torch.sum(post.unsqueeze(2) * centroids.unsqueeze(0), 1)     # T x D
    return weighted_centroids, torch.sum(dist * post, 1).mean()


def sampling_with_eps(param, eps, lat_dim=None):
    """Same draw with eps supplied by the caller (parity tests inject the reference's eps)."""
    _need_cuda(param, "sampling_with_eps(param)")
    if lat_dim is None:
        lat_dim = int(param.shape[-1] / 2)
    p, e = param.to(torch.float32).contiguous(), eps.to(torch.float32).contiguous()
    z = torch.empty_like(e)
    _lib().sample(p.data_ptr(), p.numel() // p.shape[-1], lat_dim, e.data_ptr(), 0, 0, z.data_ptr(), None, _stream())
    return z


def _rows(t):
    """(tensor, row stride) of a 2-D float32 device tensor whose rows are contiguous (slices of a batch along frames / channels are)."""
    if t.stride(1) != 1:
        t = t.contiguous()
    return t, t.stride(0) if t.shape[0] > 1 else t.shape[1]


class _KLGauss(torch.autograd.Function):
    """loss_vae on a device tensor as one launch each way (cvae_kl_gauss); the torch form is ~8 launches and as many autograd nodes."""

    @staticmethod
    def forward(ctx, param, lat_dim):
        p, sp = _rows(param.detach())
        out = torch.empty(1, dtype=torch.float32, device=p.device)
        _lib().kl_gauss(p.data_ptr(), sp, p.shape[0], lat_dim, out.data_ptr(), _stream())
        ctx.save_for_backward(p)
        ctx.dims = (sp, lat_dim)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        sp, L = ctx.dims
        d = torch.empty(p.shape[0], 2 * L, dtype=torch.float32, device=p.device)
        g1 = g.to(torch.float32).reshape(1).contiguous()
        _lib().kl_gauss_backward(p.data_ptr(), sp, p.shape[0], L, g1.data_ptr(), d.data_ptr(), _stream())
        return d, None


class _MCDL1(torch.autograd.Function):
    """TWFSEloss(x, y, twf=None, GV=False, rmse=False, L2=False) on device tensors as one launch each way (cvae_mcd_l1)."""

    @staticmethod
    def forward(ctx, x, y):
        xs, sx = _rows(x.detach())
        ys, sy = _rows(y.detach())
        n, D = xs.shape
        out = torch.empty(3, dtype=torch.float32, device=xs.device)
        frame = torch.empty(n, dtype=torch.float32, device=xs.device)
        _lib().mcd_l1(xs.data_ptr(), sx, ys.data_ptr(), sy, n, D, frame.data_ptr(), out.data_ptr(), _stream())
        ctx.save_for_backward(xs, ys, frame, out)
        ctx.dims = (sx, sy, n, D)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, g_sum, g_mean, g_std):
        xs, ys, frame, out = ctx.saved_tensors
        sx, sy, n, D = ctx.dims
        g3 = torch.stack([g.to(torch.float32).reshape(()) if g is not None else torch.zeros((), device=xs.device)
                          for g in (g_sum, g_mean, g_std)])
        dx = torch.empty(n, D, dtype=torch.float32, device=xs.device)
        _lib().mcd_l1_backward(xs.data_ptr(), sx, ys.data_ptr(), sy, n, D, frame.data_ptr(), out.data_ptr(), g3.data_ptr(), dx.data_ptr(),
                               _stream())
        return dx, None


def _fused_ok(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] > 0 for t in ts) and _LIB is not None


def loss_vae(param, lat_dim=None, relu_vae=False):
    """KL(N(mu, exp(s)) || N(0, I)) averaged over frames (reference gru_vae.py:117-123); param [T, 2L]."""
    if relu_vae:
        raise NotImplementedError("relu_vae is dead code in this recipe")
    if lat_dim is None:
        lat_dim = int(param.shape[1] / 2)
    if _fused_ok(param) and param.shape[1] == 2 * lat_dim:
        return _KLGauss.apply(param, lat_dim)           # device tensor: one launch each way (same number to fp32 rounding)
    mu, s = param[:, :lat_dim], param[:, lat_dim:]
    return (0.5 * (s.exp() + mu * mu - s - 1.0).sum(1)).mean()


class _SampleLaplace(torch.autograd.Function):
    """z = mu - exp(s) * sign(eps) * log1p(-2|eps|) with eps ~ U(-0.4999, 0.5) drawn on the device (cvae_sample_laplace), one launch
    each way; d mu = dz, d s = dz * (z - mu)."""

    @staticmethod
    def forward(ctx, param, lat_dim, seed, eps):
        lib = _lib()
        p = param.detach().to(torch.float32).contiguous()
        rows = p.numel() // p.shape[-1]
        z = torch.empty(p.shape[:-1] + (lat_dim,), dtype=torch.float32, device=p.device)
        e = None if eps is None else eps.detach().to(torch.float32).contiguous()
        lib.sample_laplace(p.data_ptr(), rows, lat_dim, None if e is None else e.data_ptr(), seed, 0, z.data_ptr(), None, _stream())
        ctx.save_for_backward(p, z)
        ctx.dims = (rows, lat_dim, param.shape, param.dtype)
        return z

    @staticmethod
    def backward(ctx, dz):
        p, z = ctx.saved_tensors
        rows, L, shape, dtype = ctx.dims
        dlat = torch.empty_like(p)
        _lib().sample_laplace_backward(dz.to(torch.float32).contiguous().data_ptr(), p.data_ptr(), z.data_ptr(), rows, L, dlat.data_ptr(),
                                       _stream())
        return dlat.view(shape).to(dtype), None, None, None


def sampling_vae_laplace(param, lat_dim=None, training=False, relu_vae=False, eps=None):
    """Reference gru_vae.py:101-114 (the log-scale branch): z = mu - exp(log_scale) * sign(eps) * log1p(-2|eps|) with
    eps ~ U(-0.4999, 0.5), param [N, 2L].  The reference draws eps with torch's CUDA generator; here it comes from the on-device
    Philox stream seeded from torch's CPU generator (torch.manual_seed keeps runs reproducible), or from `eps` [N, L] (parity tests).
    training=False draws under no_grad -- the same values (reference :106-110)."""
    if relu_vae:
        raise NotImplementedError("relu_vae is dead code in this recipe")
    _need_cuda(param, "sampling_vae_laplace(param)")
    if lat_dim is None:
        lat_dim = int(param.shape[1] / 2)
    if param.shape[-1] != 2 * lat_dim:
        raise ValueError("param has %d columns, expected 2 * lat_dim = %d" % (param.shape[-1], 2 * lat_dim))
    check_status()
    return _SampleLaplace.apply(param, lat_dim, 0 if eps is not None else _draw_seed(), eps)


def loss_vae_laplace(param, lat_dim=None, relu_vae=False):
    """KL(Laplace(mu, exp(s)) || Laplace(0, 1)) averaged over frames (reference gru_vae.py:130-139, the log-scale branch);
    param [T, 2L].  Torch ops on the device the tensor lives on, like the reference."""
    if relu_vae:
        raise NotImplementedError("relu_vae is dead code in this recipe")
    if lat_dim is None:
        lat_dim = int(param.shape[1] / 2)
    mu_abs, s = param[:, :lat_dim].abs(), param[:, lat_dim:]
    scale = torch.exp(s)
    return torch.mean(torch.sum(-s + scale * torch.exp(-mu_abs / scale) + mu_abs - 1, 1))


class TWFSEloss(nn.Module):
    """Mel-cepstral distortion loss / metric (reference gru_vae.py:466-534), every branch: `twf` (time-warping indices
    into x), `rmse` (per-dimension RMSE / L1 + correlation), `L2`, `GV`.  Plain torch ops on whatever device x lives on,
    like the reference; the stage-4 step uses twf=None, GV=False, L2=False (train...:1366-1368)."""

    K = 10.0 / 2.3025850929940456840179914546844

    def forward(self, x, y, twf=None, GV=True, rmse=False, L2=True):
        if twf is None and not rmse and not L2 and not GV and _fused_ok(x, y) and x.shape == y.shape and not y.requires_grad:
            return _MCDL1.apply(x, y)                   # the training script's call (train...:1366-1368): one launch each way
        xs = x if twf is None else torch.index_select(x, 0, twf)      # gru_vae.py:472-475 / :517
        if rmse:
            err = torch.sqrt(torch.mean((xs - y) ** 2, 0)) if L2 else torch.mean(torch.abs(y - xs), 0)
            out_diff = (x - torch.mean(x, 0)) if twf is None else torch.index_select(x - torch.mean(x, 0), 0, twf)
            trg_diff = y - torch.mean(y, 0)
            corr = torch.sum(out_diff * trg_diff, 0) / (torch.sqrt(torch.sum(out_diff * out_diff, 0)) *
                                                        torch.sqrt(torch.sum(trg_diff * trg_diff, 0)))
            return torch.mean(err), torch.mean(corr)
        d = xs - y
        if L2:
            mcd = self.K * torch.sqrt(2.0 * (d * d).sum(1))
        else:
            mcd = self.K * 1.4142135623730950488016887242097 * d.abs().sum(1)
        out = (mcd.sum(), mcd.mean(), mcd.std())
        if GV:
            lx, ly = torch.log(torch.var(xs, 0)), torch.log(torch.var(y, 0))
            # with twf the reference always takes the squared-error form (:506), without it L1 follows L2=False (:530-533)
            if L2 or twf is not None:
                out += (torch.sqrt((lx - ly) ** 2).mean(),)
            else:
                out += ((lx - ly).abs().mean(),)
        return out


class CycleChain(object):
    """Fused n_cyc reconversion loop in eval form (the four hand-written copies of it in the reference's
    train_gru_cyclevae_gauss_batch.py:1299-1338, 1491-1510, 856-885 collapse to one C-ABI call)."""

    def __init__(self, model_encoder, model_decoder, lat_dim, n_cyc=2):
        self.enc, self.dec, self.lat_dim, self.n_cyc = model_encoder, model_decoder, lat_dim, n_cyc
        self._ws = None

    def __call__(self, x, cvx, code_src, code_trg, y_in_enc, y_in_dec, eps=None, seed=None, outputs=True, state=None,
                 return_state=False):
        """x [B,T,Cin]; cvx [B,T,stdim]; codes [B,T,ncode]; y_in_* [B,1,C]; eps None (Philox) or [n_cyc,3,B,T,L].
        Returns dict lat, rec, cv, latcv, reccyc, each [n_cyc,B,T,C].
        state: None (fresh window) or the dict a previous call returned with return_state=True: every pass of every cycle then
        continues from its own (y_last, h) of the previous window, as the reference's windowed loop does (train...:1299-1311).
        With return_state the result is (outputs, state); state = {"y_enc" [n_cyc,2,B,2L], "y_dec" [n_cyc,3,B,Cout],
        "h_enc" [n_cyc,2,B,H], "h_dec" [n_cyc,3,B,H]} (encoder slots lat, latcv; decoder slots rec, cv, reccyc)."""
        _need_cuda(x, "CycleChain(x)")
        lib = _lib()
        check_status()
        dev = x.device
        f = lambda t: t.to(torch.float32).contiguous()
        x, cvx, code_src, code_trg = f(x), f(cvx), f(code_src), f(code_trg)
        B, T, _ = x.shape
        de, ie = self.enc.prepared(dev)
        dd, idd = self.dec.prepared(dev)
        need = lib.cycle_workspace_bytes(de, dd, B, T, self.n_cyc)
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        ye, yd = f(y_in_enc.reshape(B, -1)), f(y_in_dec.reshape(B, -1))
        n, L, Co = self.n_cyc, self.lat_dim, self.dec.out_dim
        out = {}
        if outputs:
            for k, c in (("lat", 2 * L), ("rec", Co), ("cv", Co), ("latcv", 2 * L), ("reccyc", Co)):
                out[k] = torch.empty(n, B, T, c, dtype=torch.float32, device=dev)
        p = lambda k: out[k].data_ptr() if k in out else None
        e = None if eps is None else f(eps)
        if state is None and not return_state:
            lib.cycle_forward(de, ie.data_ptr(), dd, idd.data_ptr(), x.data_ptr(), cvx.data_ptr(), cvx.shape[2],
                              code_src.data_ptr(), code_trg.data_ptr(), code_src.shape[2], ye.data_ptr(), yd.data_ptr(),
                              B, T, n, L, None if e is None else e.data_ptr(), _draw_seed() if seed is None else seed,
                              p("lat"), p("rec"), p("cv"), p("latcv"), p("reccyc"), self._ws.data_ptr(), self._ws.numel(),
                              _flags(), _stream())
            return out
        keys = ("y_enc", "y_dec", "h_enc", "h_dec")
        shapes = {"y_enc": (n, 2, B, 2 * L), "y_dec": (n, 3, B, Co), "h_enc": (n, 2, B, self.enc.hidden_units),
                  "h_dec": (n, 3, B, self.dec.hidden_units)}
        s_in = None
        if state is not None:
            s_in = {k: f(state[k]) for k in keys}
            for k in keys:
                if tuple(s_in[k].shape) != shapes[k]:
                    raise ValueError("state[%r] has shape %s, expected %s" % (k, tuple(s_in[k].shape), shapes[k]))
        s_out = {k: torch.empty(shapes[k], dtype=torch.float32, device=dev) for k in keys} if return_state else None
        ptrs = lambda d_: None if d_ is None else tuple(d_[k].data_ptr() for k in keys)
        lib.cycle_forward_carry(de, ie.data_ptr(), dd, idd.data_ptr(), x.data_ptr(), cvx.data_ptr(), cvx.shape[2],
                                code_src.data_ptr(), code_trg.data_ptr(), code_src.shape[2], ye.data_ptr(), yd.data_ptr(),
                                B, T, n, L, None if e is None else e.data_ptr(), _draw_seed() if seed is None else seed,
                                p("lat"), p("rec"), p("cv"), p("latcv"), p("reccyc"), self._ws.data_ptr(), self._ws.numel(),
                                _flags(), _stream(), ptrs(s_in), ptrs(s_out))
        return (out, s_out) if return_state else out

    def status(self):
        """Synchronises; [0] != 0 = a hand-off spin timed out somewhere since the last check."""
        torch.cuda.current_stream().synchronize()
        return [int(v) for v in _sink()] if _sink() is not None else _lib().workspace_status(self._ws.data_ptr(), _stream())
