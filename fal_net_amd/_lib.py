"""ctypes binding of libfalnet_hip.so (the C-ABI declared in include/falnet_hip.h).

There is deliberately no fallback: if the library is missing or a call fails, this raises.
`import torch` happens first so that the HIP runtime the library binds to (soname
libamdhip64.so.7) is the one PyTorch already loaded -- device pointers and streams are then
shared between torch's caching allocator and these kernels.
"""
import ctypes as C
import os
import threading

import torch  # noqa: F401  (must precede CDLL: see module docstring)

F32, BF16, F16 = 0, 1, 2
ACT_NONE, ACT_ELU, ACT_RELU = 0, 1, 2
OUT_NHWC, OUT_PLANAR_F32 = 0, 1
CPAD = 32  # channel padding granule of NHWC tensors / packed weights (falnet_channel_pad)

# FALNET_DETERMINISTIC=1: bit-identical results from run to run (include/falnet_hip.h: falnet_set_deterministic; host side in ops.py)
DETERMINISTIC = os.environ.get("FALNET_DETERMINISTIC") == "1"

def ab(name, default):
    """Value of an EXPERIMENT switch (kernel-selection / scheduling A/B knobs used by tools/ab_*.sh and the tuning notes in DESIGN.md).
    They are honoured only when FALNET_AB=1 is set: an ordinary process (training, bench, tests) always runs the tuned defaults and
    cannot be steered off them by a stray environment variable.  Product switches (FALNET_LIB, FALNET_AUTOTUNE*, FALNET_DETERMINISTIC,
    FALNET_F16_*, FALNET_FORCE_DIST, FALNET_VGG19_WEIGHTS, FALNET_HW_QUEUES, FALNET_FUSED_STEP) are read directly."""
    if os.environ.get("FALNET_AB") != "1":
        return default
    return os.environ.get(name, default)


# FALNET_LIB: alternative build of the same C-ABI (kernel A/B experiments); default = the in-tree library
_LIB_PATH = os.environ.get("FALNET_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libfalnet_hip.so")


class Src(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("sb", C.c_int64), ("sy", C.c_int64), ("sx", C.c_int64)]


class Conv(C.Structure):
    _fields_ = [("src", Src * 2), ("nsrc", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32),
                ("weight", C.c_void_p), ("cin_total", C.c_int32), ("ntaps", C.c_int32),
                ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9), ("tap_w", C.c_int32 * 9),
                ("w_taps", C.c_int32), ("w_rows", C.c_int32), ("isy", C.c_int32), ("isx", C.c_int32),
                ("B", C.c_int32), ("TH", C.c_int32), ("TW", C.c_int32),
                ("osy", C.c_int32), ("osx", C.c_int32), ("ooy", C.c_int32), ("oox", C.c_int32),
                ("out", C.c_void_p), ("OH", C.c_int32), ("OW", C.c_int32), ("Cout", C.c_int32),
                ("out_cstride", C.c_int32), ("out_layout", C.c_int32), ("bias", C.c_void_p),
                ("addend", C.c_void_p), ("act", C.c_int32), ("actout", C.c_void_p),
                ("actout_kind", C.c_int32), ("dtype", C.c_int32), ("ksplit", C.c_int32), ("splitk_ws", C.c_void_p),
                ("splitk_ws_bytes", C.c_int64), ("variant", C.c_int32), ("pool_out", C.c_void_p),
                ("pool_mode", C.c_int32), ("pool_actout_kind", C.c_int32), ("pool_actout", C.c_void_p), ("weight_up2", C.c_void_p),
                ("scratch", C.c_void_p), ("scratch_bytes", C.c_int64)]


class Wgrad(C.Structure):
    _fields_ = [("src", Src * 2), ("nsrc", C.c_int32), ("IH", C.c_int32), ("IW", C.c_int32),
                ("gout", C.c_void_p), ("gC", C.c_int32), ("ntaps", C.c_int32),
                ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9), ("isy", C.c_int32), ("isx", C.c_int32),
                ("B", C.c_int32), ("TH", C.c_int32), ("TW", C.c_int32), ("cin_total", C.c_int32),
                ("nsplit", C.c_int32), ("partial", C.c_void_p), ("dtype", C.c_int32), ("variant", C.c_int32), ("bias_grad", C.c_void_p),
                ("cout", C.c_int32), ("up2", C.c_int32)]


class ReduceDesc(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("grad", C.c_void_p), ("nsplit", C.c_int32), ("ntaps", C.c_int32),
                ("w_rows", C.c_int32), ("cin_total", C.c_int32), ("cout", C.c_int32), ("cin", C.c_int32),
                ("c0_real", C.c_int32), ("c0_pad", C.c_int32), ("groups", C.c_int32), ("block_begin", C.c_int32)]


class PackDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wf", C.c_void_p), ("wd", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32),
                ("taps", C.c_int32), ("c0_real", C.c_int32), ("c0_pad", C.c_int32), ("cin_pad", C.c_int32),
                ("cout_pad", C.c_int32), ("block_begin", C.c_int32), ("no_update", C.c_int32), ("reserved", C.c_int32)]


class PackUp2Desc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wu", C.c_void_p), ("cout", C.c_int32), ("cin", C.c_int32), ("cin_pad", C.c_int32), ("cout_pad", C.c_int32),
                ("block_begin", C.c_int32), ("reserved", C.c_int32), ("wdd", C.c_void_p)]


class Cmd(C.Structure):  # falnet_cmd_t
    _fields_ = [("op", C.c_int32), ("stream", C.c_int32), ("event", C.c_int32), ("nint", C.c_int32), ("nflt", C.c_int32), ("reserved", C.c_int32),
                ("iarg", C.c_uint64 * 18), ("farg", C.c_double * 8)]


CMD_RECORD, CMD_WAIT = -1, -2


class BiasGradDesc(C.Structure):
    _fields_ = [("g", C.c_void_p), ("db", C.c_void_p), ("npix", C.c_int64), ("gC", C.c_int32), ("cout", C.c_int32),
                ("blocks", C.c_int32), ("block_begin", C.c_int32)]


_P, _I, _L, _F, _D = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
# name -> argtypes (restype int unless listed in _RESTYPES); mirrors include/falnet_hip.h one to one
SIGNATURES = {
    "falnet_version": [],
    "falnet_last_error": [],
    "falnet_channel_pad": [_I],
    "falnet_set_device": [_I],
    "falnet_set_deterministic": [_I],
    "falnet_get_deterministic": [],
    "falnet_conv2d": [C.POINTER(Conv), _P],
    "falnet_conv3x3_c3": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "falnet_conv2d_multi": [C.POINTER(Conv), _I, _P],
    "falnet_conv2d_kernel_name": [C.POINTER(Conv), C.c_char_p, _I],
    "falnet_wgrad_workspace_bytes": [C.POINTER(Wgrad)],
    "falnet_wgrad": [C.POINTER(Wgrad), _P],
    "falnet_wgrad_fuses_bias": [C.POINTER(Wgrad)],
    "falnet_wgrad_reduce": [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P],
    "falnet_bias_grad": [_P, _L, _I, _I, _P, _I, _I, _P],
    "falnet_pack_weights_batched": [_P, _I, _I, _I, _P],
    "falnet_adam_pack_batched": [_P, _I, _I, _I, _L, _L, _L, _P, _F, _F, _F, _F, _P, _P],
    "falnet_adam_ranges": [_P, _L, _L, _L, _P, _I, _P, _F, _F, _F, _F, _P, _P],
    "falnet_adam_tick": [_P, _P, _P],
    "falnet_pack_up2_batched": [_P, _I, _I, _I, _P],
    "falnet_wgrad_reduce_batched": [_P, _I, _I, _I, _P],
    "falnet_wgrad_reduce_blocks": [_I, _I, _I],
    "falnet_bias_grad_batched": [_P, _I, _I, _I, _P],
    "falnet_bias_grad_batched_det": [_P, _I, _I, _I, _P, _L, _P],
    "falnet_pack_weights": [_P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "falnet_nchw_to_nhwc": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "falnet_nhwc_to_nchw": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "falnet_upsample_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "falnet_wgrad_const_plane": [_P, _P, _L, _P, _L, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "falnet_maxpool2_fwd": [_P, _P, _I, _I, _I, _I, _I, _P],
    "falnet_maxpool2_bwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "falnet_act_bwd": [_P, _P, _P, _L, _I, _I, _P],
    "falnet_med_head_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "falnet_med_head_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "falnet_med_head_bwd_nhwc": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "falnet_med_head_kernel_name": [_I, _I, _I, _I, C.c_char_p, _I],
    "falnet_med_masks_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "falnet_med_maskr_acfalse_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "falnet_l1_fwd": [_P, _P, _P, _I, _I, _L, _F, _P, _I, _P],
    "falnet_l1_bwd": [_P, _P, _P, _I, _I, _L, _F, _P, _P, _I, _P],
    "falnet_mse_fwd": [_P, _P, _L, _I, _F, _P, _I, _I, _P],
    "falnet_mse_bwd": [_P, _P, _L, _I, _F, _P, _P, _I, _P],
    "falnet_l1_fwd_bwd": [_P, _P, _I, _I, _L, _F, _P, _P, _P, _P],
    "falnet_l1_fwd_bwd_add": [_P, _P, _I, _I, _L, _F, _P, _P, _P, _P, _P],
    "falnet_mse_fwd_bwd": [_P, _P, _L, _I, _F, _P, _F, _P, _P, _I, _P],
    "falnet_mse3_fwd_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "falnet_smooth_fwd_bwd": [_P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P],
    "falnet_smooth_fwd": [_P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _I, _P],
    "falnet_smooth_bwd": [_P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _I, _P],
    "falnet_step_scalars": [_P, _F, _P, _P],
    "falnet_mask_mix": [_P, _P, _P, _P, _I, _I, _L, _P],
    "falnet_adam_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _I, _F, _P],
    "falnet_adam_step_dev": [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P],
    "falnet_adam_step_guarded": [_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P, _P],
    "falnet_grad_guard": [_P, _L, _P, _P],
    "falnet_loss_scale_update": [_P, _F, _F, _I, _F, _F, _P],
    "falnet_loss_seeds": [_P, _P, _P, _I, _P],
    "falnet_hflip": [_P, _P, _L, _I, _P],
    "falnet_resample_u8": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "falnet_augment_normalize": [_P, _I, _I, _I, _I, _I, _I, _I, _D, _D, _D, _D, _D, _F, _F, _F, _P, _P],
    "falnet_rowmax": [_P, _P, _I, _L, _P],
    "falnet_gemm_f32_small": [_P, _L, _L, _P, _L, _L, _P, _I, _I, _I, _I, _P],
    "falnet_resize_planar": [_P, _P, _L, _I, _I, _I, _I, _I, _F, _P],
    "falnet_disp_prologue": [_P, _P, _F, _F, _P, _P, _P, _I, _I, _I, _P],
    "falnet_occlusion_mask": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "falnet_mirror_weight": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "falnet_replay_op_index": [C.c_char_p],
    "falnet_replay_op_args": [_I, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "falnet_replay": [_P, _I, _P, _I, _P, _I, C.POINTER(C.c_int)],
    "falnet_fill_f32": [_P, _L, _F, _P],
    "falnet_copy_bytes": [_P, _P, _L, _P],
    "falnet_spin": [_I, _P],
    "falnet_mfma_probe": [_P, _P, _I, _I, _P],
}
_RESTYPES = {"falnet_last_error": C.c_char_p, "falnet_wgrad_workspace_bytes": C.c_int64}

_lib = None
_TLS = threading.local()  # per-thread launch state: the pinned stream (stream_scope) and the active Recorder
# falnet_version() of the library this binding was written against (api.cpp; bumped with every struct / entry-point change): a stale
# FALNET_LIB build with the same symbols but another descriptor layout must not load
EXPECTED_VERSION = 600


def lib():
    """The loaded library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} is missing: build it with `python -m fal_net_amd._build` "
                "(there is no CPU or PyTorch fallback for the FAL_netB kernels)")
        l = C.CDLL(_LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError here = header/library drift: fail loudly
            fn.argtypes = args
            fn.restype = _RESTYPES.get(name, C.c_int)
        got = l.falnet_version()
        if got != EXPECTED_VERSION:
            raise RuntimeError(f"{_LIB_PATH} reports falnet_version() = {got}, this binding expects {EXPECTED_VERSION}: rebuild it "
                               "(`python -m fal_net_amd._build`; experiment builds: `--ab <tag>`)")
        if DETERMINISTIC:
            l.falnet_set_deterministic(1)
        _lib = _LibProxy(l)
    return _lib


# ---- host launch path in C (csrc/replay.cpp) -------------------------------------------------------------------------------------------
# Every launch entry point (a function of the header that ends in `void* stream`) is reached through _LibProxy: normally a plain call; while
# a Recorder is active on this thread the call is NOT made but appended to the recorder's command list, with its stream replaced by an
# index.  Event records / stream waits go through ev_record / ev_wait for the same reason.  Recorder.finalize() -> Segment, whose run()
# issues the whole list from one falnet_replay call.
_MASK64 = (1 << 64) - 1


def _int_arg(a):
    if a is None:
        return 0
    if isinstance(a, int):
        return a & _MASK64
    if isinstance(a, C.c_void_p):
        return a.value or 0
    if isinstance(a, (C.Structure, C.Array)):
        return C.addressof(a)
    obj = getattr(a, "_obj", None)  # C.byref(x)
    if obj is not None:
        return C.addressof(obj)
    if isinstance(a, C._SimpleCData):
        return int(a.value or 0) & _MASK64
    raise TypeError(f"cannot record argument {a!r}")


class Segment:
    """A recorded launch sequence: run(main_stream_pointer) issues it through falnet_replay (stream index 0 = the caller's stream)."""

    def __init__(self, cmds, streams, events, keep):
        self.n = len(cmds)
        self.cmds = (Cmd * max(self.n, 1))(*cmds)
        self.streams = (C.c_void_p * len(streams))(*streams)
        self.events = (C.c_void_p * max(len(events), 1))(*events)
        self.ns, self.ne = len(streams), len(events)
        self.failed = C.c_int(-1)
        self._keep = keep
        self._fn = _lib._cdll.falnet_replay

    def run(self, main_ptr):
        self.streams[0] = main_ptr
        rc = self._fn(self.cmds, self.n, self.streams, self.ns, self.events, self.ne, C.byref(self.failed))
        if rc != 0:
            msg = _lib._cdll.falnet_last_error().decode(errors="replace")
            raise RuntimeError(f"libfalnet_hip replay failed at command {self.failed.value} (rc={rc}): {msg}")


class SegmentChain:
    """Recorded launch sequences with Python calls between them (Recorder.cut): run() issues segment, call, segment, ... in order.  The parts share
    ONE stream / event table, so a wait in a later segment may name an event recorded in an earlier one."""

    def __init__(self, parts):
        self.parts = parts  # Segment | zero-argument callable

    def run(self, main_ptr):
        for p in self.parts:
            if isinstance(p, Segment):
                p.run(main_ptr)
            else:
                p()


class Recorder:
    """`with Recorder(main_stream_ptr) as r: <launches>` -> r.finalize() is the Segment of everything the block would have launched."""

    def __init__(self, main_ptr):
        self.streams = [int(main_ptr or 0)]
        self.events, self.cmds, self.keep = [], [], []
        self.parts = []
        self._ops = {}

    def cut(self, fn):
        """Something that cannot be replayed from C happens HERE in the sequence (a torch.distributed collective issued from Python): the commands
        recorded so far become one segment, `fn` is called between it and the next one on every replay."""
        self.parts.append(self.cmds)
        self.parts.append(fn)
        self.cmds = []

    def __enter__(self):
        assert getattr(_TLS, "rec", None) is None, "recorders do not nest"
        _TLS.rec = self
        return self

    def __exit__(self, *exc):
        _TLS.rec = None
        return False

    def _stream(self, ptr):
        v = int((ptr.value if isinstance(ptr, C.c_void_p) else ptr) or 0)
        if v not in self.streams:
            self.streams.append(v)
        return self.streams.index(v)

    def _event(self, ev):
        h = int(ev.cuda_event)
        if not h:
            raise RuntimeError("recording an event that was never recorded eagerly (no handle yet)")
        if h not in self.events:
            self.events.append(h)
            self.keep.append(ev)
        return self.events.index(h)

    def add(self, name, argtypes, args):
        op = self._ops.get(name)
        if op is None:
            op = self._ops[name] = _lib._cdll.falnet_replay_op_index(name.encode())
            if op < 0:
                raise RuntimeError(f"{name} is not a replayable entry point")
        c = Cmd()
        c.op, c.stream = op, self._stream(args[-1])
        ni = nf = 0
        for t, a in zip(argtypes[:-1], args[:-1]):
            if t is _F or t is _D:
                c.farg[nf] = float(a)
                nf += 1
            else:
                c.iarg[ni] = _int_arg(a)
                ni += 1
        c.nint, c.nflt = ni, nf
        self.cmds.append(c)
        self.keep.append(args)
        return 0

    def record(self, ev, stream_ptr):
        c = Cmd()
        c.op, c.stream, c.event = CMD_RECORD, self._stream(stream_ptr), self._event(ev)
        self.cmds.append(c)

    def wait(self, stream_ptr, ev):
        c = Cmd()
        c.op, c.stream, c.event = CMD_WAIT, self._stream(stream_ptr), self._event(ev)
        self.cmds.append(c)

    def finalize(self):
        if not self.parts:
            return Segment(self.cmds, self.streams, self.events, self.keep)
        parts = [p if callable(p) else Segment(p, self.streams, self.events, self.keep) for p in self.parts + [self.cmds] if callable(p) or p]
        return SegmentChain(parts)


class _LibProxy:
    """The loaded library; launch entry points become recordable wrappers (cached as instance attributes on first use)."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        raw = getattr(self._cdll, name)
        argtypes = SIGNATURES.get(name)
        if not argtypes or argtypes[-1] is not _P or name in ("falnet_replay", "falnet_conv2d_kernel_name", "falnet_med_head_kernel_name"):
            fn = raw
        else:
            def fn(*args, _raw=raw, _name=name, _at=argtypes):
                rec = getattr(_TLS, "rec", None)
                if rec is None:
                    return _raw(*args)
                return rec.add(_name, _at, args)
        object.__setattr__(self, name, fn)
        return fn


def recording():
    return getattr(_TLS, "rec", None) is not None


def cut(fn):
    """Inside a Recorder block: call `fn` at this point of every replay (Recorder.cut)."""
    _TLS.rec.cut(fn)


def ev_record(ev, stream):
    """ev.record(stream) -- or its recorded form; `stream`: a torch.cuda.Stream."""
    rec = getattr(_TLS, "rec", None)
    if rec is None:
        ev.record(stream)
    else:
        rec.record(ev, stream.cuda_stream)


def ev_wait(stream, ev):
    """stream.wait_event(ev) -- or its recorded form."""
    rec = getattr(_TLS, "rec", None)
    if rec is None:
        stream.wait_event(ev)
    else:
        rec.wait(stream.cuda_stream, ev)


def record_calls(calls, main_ptr=None):
    """Segment of the launches `calls` (zero-argument callables) would issue on the current launch stream."""
    mp = stream_ptr().value if main_ptr is None else main_ptr
    with Recorder(mp) as r:
        for c in calls:
            c()
    return r.finalize()


REPLAY = os.environ.get("FALNET_REPLAY", "1") == "1"  # plans replay their recorded launch sequences through falnet_replay (0: every launch from Python)


def check(rc, what=""):
    if rc != 0:
        msg = lib().falnet_last_error().decode(errors="replace")
        raise RuntimeError(f"libfalnet_hip {what} failed (rc={rc}): {msg}")


# Launch stream.  Every kernel launch names its stream explicitly (C-ABI), so a plan replay does not need torch's notion of a current
# stream per launch: `stream_scope` pins the pointer once for a whole forward / backward replay (torch.cuda.current_stream() builds a
# Stream object per call: ~1.5 us x 300 launches), and `on_stream` redirects the launches of a `with` block to another stream (the
# weight-gradient side stream) without torch.cuda.stream()'s context switch (~10 us per launch group).  Outside such scopes the current
# torch stream is looked up per call, as before.
# (the pin is per THREAD: autograd's backward thread pins its own stream without redirecting the main thread's launches)


def stream_ptr():
    p = getattr(_TLS, "pinned", None)
    if p is not None:
        return p
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class stream_scope:
    """Pin stream_ptr() to the CURRENT torch stream (or to `stream`) for the duration of the block; scopes nest."""

    def __init__(self, stream=None):
        self._s = stream

    def __enter__(self):
        self._old = getattr(_TLS, "pinned", None)
        s = self._s if self._s is not None else torch.cuda.current_stream()
        _TLS.pinned = C.c_void_p(s.cuda_stream)
        return self

    def __exit__(self, *exc):
        _TLS.pinned = self._old
        return False


on_stream = stream_scope  # `with L.on_stream(side): launch()` -- launches of the block go to `side`


def ptr(t):
    """Raw device pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise ValueError(f"unsupported compute dtype {dt}")
