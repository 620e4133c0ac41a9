"""Host-side launch descriptors for the libfalnet_hip.so kernels.

Everything here is plumbing: it fills the C structs of include/falnet_hip.h from torch tensors
(device memory + current stream come from PyTorch) and returns zero-argument callables, so a
network is turned once into a static *plan* (a list of launches over pre-allocated buffers) and a
step is just that list replayed -- no per-step allocation, no Python tensor math on the hot path.
"""
import ctypes as C
import os
import sys

import torch

from . import _lib as L

CPAD = L.CPAD
H16 = (torch.bfloat16, torch.float16)  # 16-bit operand types (bf16 / f16 MFMA, f32 accumulate)
TYPE_SYM = {torch.bfloat16: "DF16b", torch.float16: "DF16_", torch.float32: "f"}  # Itanium-mangled template argument, as rocprofv3 prints it

# Optional launch timer (bench.py / profiling): object with run(tag, flops, bytes, launch).  None on the
# product path: launches are then plain C-ABI calls.
TIMER = None
# Plan-build-time kernel selection: every conv launch tries the applicable kernel variants a few times on its
# own (scratch) buffers and keeps the fastest (static plan => decided once).  FALNET_AUTOTUNE=0 disables.
AUTOTUNE = os.environ.get("FALNET_AUTOTUNE", "1") != "0"
# Deterministic mode (FALNET_DETERMINISTIC=1): kernel choices come from the autotune cache or the library's heuristic -- never from
# timing launches, whose winner can differ from run to run --, no split-K (f32 atomics), one writer per element in the slab reduce,
# bias gradients through the two-pass ordered form; the library side is falnet_set_deterministic (fal_net_amd/_lib.py sets it).
DETERMINISTIC = L.DETERMINISTIC
UP2W = L.ab("FALNET_UP2W", "0") == "1"  # deconv weight gradients on the low-resolution grid (falnet_wgrad_t::up2): correct and tested, NOT faster -- the
# row-streaming kernel's step is bound by its DMA / barrier cadence, not by its MFMAs (profiles/r05_ab_up2w.txt) -- so off by default
UP2D = L.ab("FALNET_UP2D", "1") == "1"  # deconv data gradients on the low-resolution grid (falnet_conv2d variant 26) among the plan's candidates


# ---- persistent autotune choices --------------------------------------------------------------------------------------
# Every plan-build-time choice (conv variant / split-K factor, fused vs separate stride-2 data gradients) is remembered in a JSON
# file next to the library, keyed by the launch's shape signature: a later process replays the same variants WITHOUT timing
# launches -- bench, profile and test runs then execute the same kernels (reproducible traces, no autotune launches inside a
# rocprofv3 collection, two ranks of one job on the same choices).
#   * The file carries a header (`_meta`): the hash of the conv kernel sources it was tuned with, falnet_version() and the GPU
#     architecture.  A file whose header does not match the running library is IGNORED (every entry re-tuned): a stale entry can
#     no longer pin a variant after a kernel rewrite.
#   * The packaged file is READ-ONLY for ordinary processes (tests, training, bench): new choices live in memory only.  It is
#     written when FALNET_AUTOTUNE_CACHE=<path> names a file explicitly (tools/regen_cache.sh) or FALNET_AUTOTUNE_CACHE_WRITE=1.
#   * FALNET_AUTOTUNE_CACHE=0 disables it; so does any non-default candidate-gating experiment switch (FALNET_AB=1 with FALNET_NO_DMA,
#     FALNET_WS2, FALNET_S2F_DMA, FALNET_S2D_DMA, FALNET_S2_SPLITK, FALNET_GATHER_NARROW): an A/B run must tune, not replay.
_PKG_CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "autotune_cache.json")
_CACHE_PATH = os.environ.get("FALNET_AUTOTUNE_CACHE", _PKG_CACHE)
_CACHE_WRITABLE = ("FALNET_AUTOTUNE_CACHE" in os.environ and _CACHE_PATH != "0") or os.environ.get("FALNET_AUTOTUNE_CACHE_WRITE") == "1"
_GATES = {"FALNET_NO_DMA": "0", "FALNET_WS2": "1", "FALNET_S2F_DMA": "1", "FALNET_S2D_DMA": "1", "FALNET_S2_SPLITK": "1",
          "FALNET_GATHER_NARROW": "0", "FALNET_S2_MULTI": None, "FALNET_SMALL_TILE_DMA": "1", "FALNET_UP2": "1", "FALNET_DEEP": "1", "FALNET_TILE8": "1"}
if os.environ.get("FALNET_AB") == "1" and any(os.environ.get(k, v) != v for k, v in _GATES.items()):
    _CACHE_PATH = "0"
_CACHE = None
_CACHE_DIRTY = False
_TUNE_SOURCES = ("conv.hip", "conv_dma.hip", "conv_wave.hip", "conv_epilogue.h", "common.h")  # what a cached conv choice depends on


def cache_meta():
    """Header a cache file must carry to be replayed by this library build."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    for name in _TUNE_SOURCES:
        path = os.path.join(csrc, name)
        if os.path.isfile(path):
            with open(path, "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return {"src_hash": h.hexdigest()[:16], "version": int(L.lib().falnet_version()), "arch": "gfx950"}


def _cache():
    global _CACHE
    if _CACHE is None:
        _CACHE = {}
        if _CACHE_PATH != "0" and os.path.isfile(_CACHE_PATH):
            try:
                import json
                with open(_CACHE_PATH) as f:
                    data = json.load(f)
                if data.pop("_meta", None) == cache_meta():
                    _CACHE = data
                elif os.environ.get("FALNET_AUTOTUNE_LOG") == "1":
                    print(f"[autotune] {_CACHE_PATH}: tuned with another build of the kernels -- ignored", file=sys.stderr)
            except (OSError, ValueError):
                _CACHE = {}
    return _CACHE


def cache_get(key):
    return _cache().get(key) if _CACHE_PATH != "0" else None


def cache_put(key, value):
    global _CACHE_DIRTY
    if _CACHE_PATH != "0":
        _cache()[key] = value
        _CACHE_DIRTY = True


def cache_flush():
    """Write new choices back (atomic rename) -- only to a file the user named or with FALNET_AUTOTUNE_CACHE_WRITE=1."""
    global _CACHE_DIRTY
    if not _CACHE_DIRTY or _CACHE_PATH == "0" or not _CACHE_WRITABLE:
        return
    try:
        import json
        tmp = _CACHE_PATH + f".{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump({"_meta": cache_meta(), **dict(sorted(_cache().items()))}, f, indent=0)
        os.replace(tmp, _CACHE_PATH)
        _CACHE_DIRTY = False
    except OSError:
        pass


import atexit  # noqa: E402
atexit.register(cache_flush)


def conv_signature(d):
    """Shape signature of a falnet_conv_t: everything kernel selection and speed depend on, nothing run-specific (no pointers)."""
    srcs = ";".join(f"{d.src[i].C},{d.src[i].H},{d.src[i].W},{int(d.src[i].sy == 0)}" for i in range(d.nsrc))
    taps = ",".join(f"{d.tap_dy[t]}:{d.tap_dx[t]}:{d.tap_w[t]}" for t in range(d.ntaps))
    flags = f"{int(bool(d.bias))}{int(bool(d.addend))}{d.act}{int(bool(d.actout))}{d.actout_kind}{int(bool(d.pool_out))}{d.pool_mode}{int(bool(d.pool_actout))}{int(bool(d.out))}"
    return (f"conv|t{d.dtype}|{srcs}|{d.IH}x{d.IW}|k{d.cin_total}|{taps}|w{d.w_taps}x{d.w_rows}|s{d.isy}|B{d.B}|{d.TH}x{d.TW}|"
            f"o{d.osy},{d.ooy},{d.oox}|{d.OH}x{d.OW}|c{d.Cout},{d.out_cstride},{d.out_layout}|f{flags}|ws{int(d.splitk_ws_bytes > 0)}"
            + ("|up2" if d.weight_up2 else ""))


# What-if probe (experiments only, FALNET_AB=1): launches whose name contains one of these comma-separated substrings are NOT issued -- wrong
# results, valid timing: how much of the step does a launch (or a family) really cost beside everything that overlaps it?
_PROBE_SKIP = [x for x in L.ab("FALNET_PROBE_SKIP", "").split(",") if x]


def _timed(tag, flops, nbytes, launch, name=""):
    if _PROBE_SKIP and any(x in name for x in _PROBE_SKIP):
        launch = lambda *a: None  # noqa: E731
    def call(*a):
        t = TIMER
        if t is None:
            launch(*a)
        else:
            t.run(tag, flops, nbytes, lambda: launch(*a), name)
    call.tag, call.flops, call.nbytes, call.name = tag, flops, nbytes, name
    return call


def conv_kernel_tag(dtype, w_rows, Cout, planar, variant=1):
    """Kernel family falnet_conv2d dispatches to (conv.hip: falnet_conv2d), for the bench's per-family totals."""
    dn = {torch.bfloat16: 'bf16', torch.float16: 'f16'}.get(dtype, 'f32')
    if variant >= 2:
        return f"conv3x3_patch_kernel<{dn},{ {2: 'kcb128', 3: 'kcb64', 4: 'single-stage', 5: 'double-stage', 6: 'kcb64,M512', 7: 'single-stage,M512'}[variant]}>"
    bn = 128 if (w_rows % 128 == 0 and Cout > 64) else (64 if (w_rows % 64 == 0 and Cout > 32) else 32)
    return f"conv_igemm_kernel<{dn},{bn},{'planar' if planar else 'nhwc'}>"


def pad_c(c):
    return (c + CPAD - 1) // CPAD * CPAD


def nhwc_src(t, C_used=None):
    """falnet_src_t of a contiguous NHWC tensor (B,H,W,C)."""
    B, H, W, Ct = t.shape
    assert t.is_contiguous()
    s = L.Src()
    s.ptr, s.C, s.H, s.W = t.data_ptr(), (Ct if C_used is None else C_used), H, W
    s.sb, s.sy, s.sx = H * W * Ct, W * Ct, Ct
    return s


def planar_src(t):
    """Planar f32 [B][3][H][W] image as the source of the first-layer weight gradient (falnet_wgrad variant 6)."""
    B, Cc, H, W = t.shape
    assert Cc == 3 and t.dtype == torch.float32 and t.is_contiguous()
    s = L.Src()
    s.ptr, s.C, s.H, s.W = t.data_ptr(), 3, H, W
    s.sb, s.sy, s.sx = 3 * H * W, W, 1
    return s


def bcast_src(t, H, W):
    """Per-sample constant (B,C) presented as an HxW image (pixel strides 0): the `flow` plane."""
    B, Ct = t.shape
    s = L.Src()
    s.ptr, s.C, s.H, s.W = t.data_ptr(), Ct, H, W
    s.sb, s.sy, s.sx = Ct, 0, 0
    return s


class PackedConv:
    """One convolution's parameters plus its packed MFMA operands.

    groups_real / groups_pad: input-channel groups (concat sources) and their padded sizes.
    wf [cout_pad][taps][cin_pad]  forward operand; wd [cin_pad][taps][cout_pad]  dgrad operand.
    """

    def __init__(self, name, weight, bias, groups_real, stride=1):
        self.name, self.weight, self.bias, self.stride = name, weight, bias, stride
        self.cout, self.cin, kh, kw = weight.shape
        assert sum(groups_real) == self.cin and len(groups_real) <= 2
        self.taps = kh * kw
        self.ksize = kh if kh == kw else (kh, kw)  # (3, 1) / (1, 3): FAL_netA's separable residual convs
        self.groups_real = list(groups_real)
        self.groups_pad = [pad_c(c) for c in groups_real]
        self.cin_pad = sum(self.groups_pad)
        self.cout_pad = pad_c(self.cout)
        self.wf = self.wd = None
        self.wu = None      # sub-pixel weights [cout_pad][16][cin_pad] of a `deconv` layer (set `up2 = True` before alloc): falnet_conv2d variant 18
        self.wdd = None     # the same layer's data-gradient weights on the low-resolution grid [cin_pad][4][4 cout_pad]: falnet_conv2d variant 26
        self.up2 = False
        self._packed_version = None
        self._dtype = None

    def alloc(self, dtype, device):
        if self.wf is None or self._dtype != dtype or self.wf.device != device:
            self.wf = torch.zeros(self.cout_pad, self.taps, self.cin_pad, dtype=dtype, device=device)
            self.wd = torch.zeros(self.cin_pad, self.taps, self.cout_pad, dtype=dtype, device=device)
            self.wu = torch.zeros(self.cout_pad, 16, self.cin_pad, dtype=dtype, device=device) if (self.up2 and dtype in H16 and self.taps == 9) else None
            self.wdd = torch.zeros(self.cin_pad, 4, 4 * self.cout_pad, dtype=dtype, device=device) if (self.wu is not None and UP2D) else None
            self._dtype = dtype
            self._packed_version = None

    def pack_call(self, need_dgrad=True):
        """Callable that (re)packs the f32 OIHW master weight into wf / wd."""
        lib = L.lib()
        c0_real = self.groups_real[0]
        c0_pad = self.groups_pad[0] if len(self.groups_real) == 2 else self.cin_pad
        if len(self.groups_real) == 1:
            c0_real = self.cin
        args = (L.ptr(self.weight), self.cout, self.cin, self.taps, c0_real, c0_pad, self.cin_pad, self.cout_pad,
                L.ptr(self.wf), L.ptr(self.wd if need_dgrad else None), L.dtype_code(self._dtype))

        def call():
            L.check(lib.falnet_pack_weights(*args, L.stream_ptr()), "pack_weights " + self.name)
        return call

    def group_channels(self):
        c0_real = self.groups_real[0] if len(self.groups_real) == 2 else self.cin
        c0_pad = self.groups_pad[0] if len(self.groups_real) == 2 else self.cin_pad
        return c0_real, c0_pad


def conv_c3_call(dtype, x_planar, pc, out, act, name="conv0(c3)"):
    """First-layer launch (falnet_conv3x3_c3): planar f32 image + f32 OIHW master weights -> NHWC activation.
    `call.set_input(t)` re-points the launch at another contiguous f32 image of the same shape (no staging copy)."""
    lib = L.lib()
    B, C, H, W = x_planar.shape
    assert C == 3 and pc.cin == 3 and pc.cout in (32, 64) and out.shape == (B, H, W, pc.cout)
    tail = (L.ptr(pc.weight), L.ptr(pc.bias), L.ptr(out), B, H, W, pc.cout, act, L.dtype_code(dtype))
    tn, nt = TYPE_SYM[dtype], pc.cout // 32
    cur = [x_planar]

    def launch(_keep=(pc, out)):
        L.check(lib.falnet_conv3x3_c3(L.ptr(cur[0]), *tail, L.stream_ptr()), name)
    call = _timed(f"_Z17conv3x3_c3_kernelI{tn}Li{nt}EEvPKfS1_13falnet_conv_tii", 2 * B * H * W * pc.cout * 27, 0, launch, name)

    def set_input(t):
        assert t.shape == x_planar.shape and t.dtype == torch.float32 and t.is_contiguous() and t.device == x_planar.device
        cur[0] = t
    call.set_input = set_input
    return call


def pack_all_call(pcs, dtype, device):
    """ONE launch that re-packs every layer's f32 OIHW master weight into its wf / wd operands."""
    lib = L.lib()
    descs = (L.PackDesc * len(pcs))()
    blk = 0
    for i, pc in enumerate(pcs):
        c0_real, c0_pad = pc.group_channels()
        d = descs[i]
        d.w, d.wf, d.wd = pc.weight.data_ptr(), pc.wf.data_ptr(), pc.wd.data_ptr()
        d.cout, d.cin, d.taps, d.c0_real, d.c0_pad, d.cin_pad, d.cout_pad, d.block_begin = (
            pc.cout, pc.cin, pc.taps, c0_real, c0_pad, pc.cin_pad, pc.cout_pad, blk)
        blk += (pc.cout_pad // 32) * (pc.cin_pad // 32)
    dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
    n, total, code = len(pcs), blk, L.dtype_code(dtype)

    def launch(_keep=(dev, pcs)):
        L.check(lib.falnet_pack_weights_batched(L.ptr(dev), n, total, code, L.stream_ptr()), "pack_weights_batched")
    return _timed("pack_weights_batched", 0, 0, launch, "pack_weights_batched")


def adam_pack_call(pcs, dtype, device, derived=()):
    """ONE launch = the Adam update of every layer in `pcs` (f32 masters inside the model's flat buffer) + its re-pack into wf / wd
    (falnet_adam_pack_batched); the layers in `derived` (masters outside the flat buffer, rebuilt from updated factors before this launch)
    are only packed.  Returns launch(g_off, m_off, v_off, state, b1, b2, eps, grad_scale, scaler_ptr_or_None)."""
    lib = L.lib()
    n_own = len(pcs)
    pcs = list(pcs) + list(derived)
    descs = (L.PackDesc * len(pcs))()
    blk = 0
    for i, pc in enumerate(pcs):
        descs[i].no_update = int(i >= n_own)
        c0_real, c0_pad = pc.group_channels()
        d = descs[i]
        d.w, d.wf, d.wd = pc.weight.data_ptr(), pc.wf.data_ptr(), pc.wd.data_ptr()
        d.cout, d.cin, d.taps, d.c0_real, d.c0_pad, d.cin_pad, d.cout_pad, d.block_begin = (
            pc.cout, pc.cin, pc.taps, c0_real, c0_pad, pc.cin_pad, pc.cout_pad, blk)
        blk += (pc.cout_pad // 32) * (pc.cin_pad // 32)
    dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
    n, total, code = len(pcs), blk, L.dtype_code(dtype)

    def launch(g_off, m_off, v_off, state, b1, b2, eps, grad_scale, scaler, _keep=(dev, pcs)):
        L.check(lib.falnet_adam_pack_batched(L.ptr(dev), n, total, code, g_off, m_off, v_off, L.ptr(state), b1, b2, eps, float(grad_scale),
                                             L.ptr(scaler), L.stream_ptr()), "adam_pack_batched")
    return launch


def pack_up2_call(pcs, dtype, device):
    """ONE launch that rebuilds the sub-pixel weights (PackedConv.wu) of every `deconv` layer from its f32 OIHW master weight."""
    lib = L.lib()
    pcs = [pc for pc in pcs if pc.wu is not None]
    if not pcs:
        return None
    descs = (L.PackUp2Desc * len(pcs))()
    blk = 0
    for i, pc in enumerate(pcs):
        d = descs[i]
        d.w, d.wu, d.cout, d.cin, d.cin_pad, d.cout_pad, d.block_begin = pc.weight.data_ptr(), pc.wu.data_ptr(), pc.cout, pc.cin, pc.cin_pad, pc.cout_pad, blk
        d.wdd = 0 if pc.wdd is None else pc.wdd.data_ptr()
        blk += (pc.cout_pad // 32) * (pc.cin_pad // 32)
    dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
    n, total, code = len(pcs), blk, L.dtype_code(dtype)

    def launch(_keep=(dev, pcs)):
        L.check(lib.falnet_pack_up2_batched(L.ptr(dev), n, total, code, L.stream_ptr()), "pack_up2_batched")
    return _timed("pack_up2_batched", 0, 0, launch, "pack_up2_batched")


def fwd_taps(ksize):
    """(dy, dx, packed-weight tap index) of a 'same'-padded kernel: 1, 3 (3x3) or (kh, kw) for the 3x1 / 1x3 convs."""
    if ksize == 1:
        return [(0, 0, 0)]
    KH, KW = (3, 3) if ksize == 3 else ksize
    return [(kh - KH // 2, kw - KW // 2, kh * KW + kw) for kh in range(KH) for kw in range(KW)]


def dgrad_taps_s1(ksize):
    return [(-dy, -dx, t) for dy, dx, t in fwd_taps(ksize)]


def dgrad_taps_s2(py, px):
    """Taps feeding input pixels of parity (py, px) for a 3x3 stride-2 pad-1 conv: iy = 2*oy + kh - 1."""
    taps = []
    for kh in range(3):
        if (py + 1 - kh) % 2:
            continue
        for kw in range(3):
            if (px + 1 - kw) % 2:
                continue
            taps.append(((py + 1 - kh) // 2, (px + 1 - kw) // 2, kh * 3 + kw))
    return taps


def _fill_taps(d, taps):
    d.ntaps = len(taps)
    for i, (dy, dx, w) in enumerate(taps):
        d.tap_dy[i], d.tap_dx[i] = dy, dx
        if hasattr(d, "tap_w"):
            d.tap_w[i] = w


def conv_call(dtype, srcs, IH, IW, weight, cin_total, taps, w_taps, w_rows, stride_in, B, TH, TW, out, OH, OW,
              Cout, out_cstride, out_layout=L.OUT_NHWC, out_step=(1, 1, 0, 0), bias=None, addend=None,
              act=L.ACT_NONE, actout=None, actout_kind=L.ACT_NONE, weight_offset_elems=0, name="conv", flops=0,
              autotune=True, ws_owner=None, pool_out=None, pool_mode=0, pool_actout=None, pool_actout_kind=L.ACT_NONE, weight_up2=None,
              variant=None):
    """Build one falnet_conv2d launch; returns a zero-argument callable.  `pool_out`: fused 2x2 max pool of the output
    (halo-patch kernels only; `out` may then be None when only the pooled map is needed).  `variant`: this kernel and no other (no autotune;
    ValueError when it does not apply)."""
    lib = L.lib()
    d = L.Conv()
    d.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        d.src[i] = s
    d.IH, d.IW = IH, IW
    d.weight = weight.data_ptr() + weight_offset_elems * weight.element_size()
    d.cin_total, d.w_taps, d.w_rows = cin_total, w_taps, w_rows
    _fill_taps(d, taps)
    d.isy = d.isx = stride_in
    d.B, d.TH, d.TW = B, TH, TW
    d.osy, d.osx, d.ooy, d.oox = out_step
    d.out, d.OH, d.OW, d.Cout, d.out_cstride, d.out_layout = (0 if out is None else out.data_ptr()), OH, OW, Cout, out_cstride, out_layout
    d.pool_out = 0 if pool_out is None else pool_out.data_ptr()
    d.pool_mode, d.pool_actout_kind = pool_mode, pool_actout_kind
    d.pool_actout = 0 if pool_actout is None else pool_actout.data_ptr()
    d.weight_up2 = 0 if weight_up2 is None else weight_up2.data_ptr()
    dev_t = out if out is not None else pool_out
    d.bias = 0 if bias is None else bias.data_ptr()
    d.addend = 0 if addend is None else addend.data_ptr()
    d.act = act
    d.actout = 0 if actout is None else actout.data_ptr()
    d.actout_kind = actout_kind
    d.dtype = L.dtype_code(dtype)
    d.variant = 0
    d.ksplit = 1
    ws = _splitk_workspace(dev_t.device, ws_owner) if dev_t.is_cuda and out_layout == L.OUT_NHWC and pool_out is None else None
    d.splitk_ws = 0 if ws is None else ws.data_ptr()
    d.splitk_ws_bytes = 0 if ws is None else ws.numel() * 4
    scratch = None
    if ws is not None and dtype in H16 and TH * TW <= 128 and len(taps) == 9:  # the deepest levels: variant 19's partial tiles
        scratch = _deep_scratch(dev_t.device, ws_owner)
        d.scratch, d.scratch_bytes = scratch.data_ptr(), scratch.numel() * 4
    ref = C.byref(d)
    keep = (d, srcs, weight, out, bias, addend, actout, ws, pool_out, pool_actout, weight_up2, scratch)
    if variant is not None:
        d.variant = variant
        if lib.falnet_conv2d_kernel_name(ref, C.create_string_buffer(160), 160) != 0:
            raise ValueError(f"falnet_conv2d variant {variant} does not apply to this launch: {lib.falnet_last_error().decode(errors='replace')}")
    elif AUTOTUNE and autotune and dev_t.is_cuda:
        key = conv_signature(d)
        hit = cache_get(key)
        if hit is not None:
            d.variant, d.ksplit = int(hit[0]), int(hit[1])
            if DETERMINISTIC and d.ksplit > 1 and d.variant != 19:  # (variant 19 sums its K slices in a fixed order)
                d.variant, d.ksplit = 1, 1  # the cached choice was the gather kernel with split-K: same kernel, one K pass
            if lib.falnet_conv2d_kernel_name(ref, C.create_string_buffer(160), 160) != 0:  # stale entry (kernel table changed): re-tune
                hit = None
        if hit is None and DETERMINISTIC:
            d.variant, d.ksplit = 0, 1  # the library's heuristic: reproducible, no timing
        elif hit is None:
            d.variant, d.ksplit = _autotune_conv(lib, d, ref, B * TH * TW, w_rows, Cout)
            cache_put(key, [int(d.variant), int(d.ksplit)])
    if pool_out is not None and lib.falnet_conv2d_kernel_name(ref, C.create_string_buffer(160), 160) != 0:
        raise ValueError("no fused-pool kernel applies to this launch")  # the caller falls back to falnet_maxpool2_fwd

    def launch(_keep=keep):
        L.check(lib.falnet_conv2d(ref, L.stream_ptr()), name)
    buf = C.create_string_buffer(160)
    tag = buf.value.decode() if lib.falnet_conv2d_kernel_name(ref, buf, 160) == 0 and buf.value else \
        conv_kernel_tag(dtype, w_rows, Cout, out_layout == L.OUT_PLANAR_F32, d.variant)
    tag = buf.value.decode() or tag
    call = _timed(tag, flops, 0, launch, f"{name} v{d.variant} k{d.ksplit}")
    call.desc, call.ref = d, ref
    return call


UP2D_TAPS = [(0, 0, 0), (0, 1, 1), (1, 0, 2), (1, 1, 3)]  # (du, dv, weight tile) of falnet_conv2d variant 26: pairs u = i, i + 1 of upstream rows / columns


def deconv_dgrad_call(dtype, gout, pc, B, gin, actout, name="", flops=0, ws_owner=None):
    """Data gradient of a `deconv` layer (nearest 2x upsampling + 3x3 convolution, FAL_netB.py:52-58) on the low-resolution grid:
    gin[B, H, W, cin_pad] = (4x4 / stride-2 convolution of gout[B, 2H, 2W, cout_pad] with the summed taps pc.wdd) * elu'(actout) -- falnet_conv2d
    variant 26, 16 instead of 36 tap-MACs per position.  ValueError when the kernel does not apply (maps below 16 x 32, f32)."""
    if pc.wdd is None:
        raise ValueError("no low-resolution data-gradient weights for this layer")
    GH, GW = gout.shape[1], gout.shape[2]
    OH, OW = GH // 2, GW // 2
    assert (2 * OH, 2 * OW) == (GH, GW) and gout.shape[3] == pc.cout_pad and gin.shape[1:3] == (OH, OW)
    return conv_call(dtype, [nhwc_src(gout)], GH, GW, pc.wdd, 4 * pc.cout_pad, UP2D_TAPS, 4, pc.cin_pad, 1, B, OH, OW, gin, OH, OW, pc.cin_pad, gin.shape[3],
                     actout=actout, actout_kind=L.ACT_ELU if actout is not None else L.ACT_NONE, name="dgrad(low-res) " + name, flops=flops,
                     ws_owner=ws_owner, variant=26)


_SPLITK_WS = {}


def _splitk_workspace(device, owner=None, nbytes=32 << 20):
    """f32 scratch for split-K conv launches.  One buffer per (device, owner): launches that share a buffer must be
    stream-ordered, and different plans (backbone, each VGG plan instance) run concurrently on different streams."""
    key = (device.type, device.index, owner)
    if key not in _SPLITK_WS:
        _SPLITK_WS[key] = torch.zeros(nbytes // 4, dtype=torch.float32, device=device)  # zero on entry by contract; launches leave it zero
    return _SPLITK_WS[key]


def _deep_scratch(device, owner=None, nbytes=32 << 20):
    """Uninitialised scratch for the K-slice partial tiles of falnet_conv2d variant 19 (falnet_conv_t::scratch); one per (device, owner) like
    the split-K workspace: launches that share it are stream-ordered."""
    key = (device.type, device.index, owner, "deep")
    if key not in _SPLITK_WS:
        _SPLITK_WS[key] = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    return _SPLITK_WS[key]


def best_of(*calls, reps=5, key=None):
    """Plan-build-time choice between equivalent launch sequences (zero-argument callables); ties go to the earlier one.
    `key`: remember the winner's index in the autotune cache (and replay it without timing when it is there)."""
    if key is not None:
        hit = cache_get(key)
        if hit is not None and 0 <= int(hit) < len(calls):
            return calls[int(hit)]
    if DETERMINISTIC:
        return calls[0]  # no timing-based choice

    def t(c):
        c()
        best = None
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                c()
            e1.record()
            e1.synchronize()
            best = e0.elapsed_time(e1) if best is None else min(best, e0.elapsed_time(e1))
        return best
    times = [t(c) for c in calls]
    if key is not None:
        cache_put(key, times.index(min(times)))
    return calls[times.index(min(times))]


GATHER_NARROW = L.ab("FALNET_GATHER_NARROW", "0") == "1"  # opt-in: autotune also tries 32 / 64-channel gather workgroups on small layers (same-box A/B: no gain)


def gather_bn(w_rows, Cout):
    """Output channels per workgroup the gather kernel picks by itself (conv.hip: choose_conv_kernel)."""
    return 128 if (w_rows % 128 == 0 and Cout > 64) else (64 if (w_rows % 64 == 0 and Cout > 32) else 32)


def conv_multi_call(calls, name="conv multi", bn=None, ksplit=1, s2d=False):
    """Fuse up to four conv_call launches (built with autotune off: gather kernel) into one falnet_conv2d_multi.
    bn = 32 / 64: narrower workgroups than the default (variants 11 / 12: more workgroups for small layers).
    ksplit > 1: split-K over blockIdx.z with one fused epilogue; the members share the plan's split-K workspace, each with its
    own region (raises ValueError when it does not fit)."""
    lib = L.lib()
    if ksplit > 1 and DETERMINISTIC:
        raise ValueError("split-K is not available in deterministic mode")
    n = len(calls)
    arr = (L.Conv * n)()
    off = 0
    for i, c in enumerate(calls):
        C.memmove(C.byref(arr[i]), C.byref(c.desc), C.sizeof(L.Conv))
        arr[i].variant, arr[i].ksplit = (14 if s2d else {32: 11, 64: 12}.get(bn, 1)), ksplit
        if ksplit > 1:
            need = (arr[i].B * arr[i].TH * arr[i].TW * arr[i].w_rows * 4 + 255) // 256 * 256
            if not c.desc.splitk_ws or off + need > c.desc.splitk_ws_bytes or arr[i].Cout % 8:
                raise ValueError("split-K workspace too small for a fused multi launch")
            arr[i].splitk_ws = c.desc.splitk_ws + off
            arr[i].splitk_ws_bytes = need
            off += need
    keep = (arr, calls)
    dn = {L.BF16: "DF16b", L.F16: "DF16_"}.get(arr[0].dtype, "f")
    bn = bn or gather_bn(arr[0].w_rows, arr[0].Cout)

    def launch(_keep=keep):
        L.check(lib.falnet_conv2d_multi(arr, n, L.stream_ptr()), name)
    sym = f"_Z22conv3x3_s2d_dma_kernelI{dn}Ev13falnet_conv_tiiiii" if s2d else f"_Z23conv_igemm_multi_kernelI{dn}Li{bn}EEv14falnet_conv4_t"
    return _timed(sym, sum(c.flops for c in calls), 0, launch, name)


_AUTOTUNE_LOG = os.environ.get("FALNET_AUTOTUNE_LOG") == "1"


def _autotune_conv(lib, d, ref, M, w_rows, Cout, reps=3):
    """Fastest (variant, ksplit) for this launch.  Candidates: gather (with split-K when the launch would
    otherwise occupy only a fraction of the 256 CUs), and the halo-patch variants where applicable."""
    st = L.stream_ptr()
    cands = [(1, 1)]
    bn = 128 if (w_rows % 128 == 0 and Cout > 64) else (64 if (w_rows % 64 == 0 and Cout > 32) else 32)
    wgs = ((M + 127) // 128) * ((Cout + bn - 1) // bn)
    if wgs < 256 and M * w_rows * 4 <= d.splitk_ws_bytes and Cout % 8 == 0:
        cands += [(1, k) for k in (2, 4, 8, 16) if wgs * k <= 2048]
    if wgs < 256 and GATHER_NARROW:  # narrower workgroups (gather variants 11 / 12 = 32 / 64 channels): 2-4x the workgroups without split-K's second pass
        for v, nb in ((12, 64), (11, 32)):
            if nb < bn and w_rows % nb == 0:
                w2 = ((M + 127) // 128) * ((Cout + nb - 1) // nb)
                cands += [(v, 1)]
                if M * w_rows * 4 <= d.splitk_ws_bytes and Cout % 8 == 0:
                    cands += [(v, k) for k in (2, 4) if w2 * k <= 2048]
    cands += [(2, 1), (3, 1), (4, 1), (6, 1), (7, 1), (10, 1)]
    if L.ab("FALNET_WS2", "1") == "1":
        cands += [(16, 1)]  # two-phase weight-stationary
    if L.ab("FALNET_NO_DMA", "0") != "1":
        cands += [(13, 1)]
        if L.ab("FALNET_DMA2", "0") == "1":
            # four rows per wave, 16-channel chunks: 16x32 tiles on two four-wave workgroups per CU / 32x32 tiles on eight waves.  NOT candidates by
            # default: equal within 1-5 % in isolation (fewer cycles, returned by the chip as clock: profiles/r05_sq_counters.txt), +0.6 % on the step
            cands += [(21, 1), (22, 1)]
        if L.ab("FALNET_DMA16", "1") == "1":
            cands += [(23, 1)]  # variant 13's tile on v_mfma_f32_16x16x32
            if wgs < int(L.ab("FALNET_SMALL_TILE_MAXWGS", "1024")):
                cands += [(24, 1), (25, 1)]  # ... variants 17 / 20 (4x32 / 8x32 tiles)
        if d.cin_total == 32 and w_rows == 32 and L.ab("FALNET_CONV_WAVE", "1") == "1":
            cands += [(27, 1)]  # wave-streaming kernel: 32 -> 32 channels at full resolution (HBM-bound layers)
        if d.cin_total == 64 and d.out_layout == L.OUT_PLANAR_F32 and Cout <= 4 and L.ab("FALNET_CONV_WAVE", "1") == "1":
            cands += [(29, 1)]  # wave-streaming kernel: 64 -> (<= 4) planar-f32 channels (the VGG adjoint's last launch)
        if d.weight_up2:
            cands += [(18, 1)]  # deconv forward in sub-pixel form
        if wgs < int(L.ab("FALNET_SMALL_TILE_MAXWGS", "1024")) and L.ab("FALNET_SMALL_TILE_DMA", "1") == "1":
            cands += [(17, 1)]  # LDS-DMA on 4x32 tiles: the 16x32 maps of level 4
            if L.ab("FALNET_TILE8", "1") == "1":
                cands += [(20, 1)]  # ... on 8x32 tiles: the 32x64 maps of level 3
        if d.isy == 2 and L.ab("FALNET_S2F_DMA", "1") == "1":
            cands += [(15, 1)]  # LDS-DMA forward 3x3 stride-2
    if wgs < 512:
        cands += [(8, 1), (9, 1)]
    if d.TH * d.TW <= 128 and d.ntaps == 9 and L.ab("FALNET_DEEP", "1") == "1":
        # levels 5-6: K-sliced one-shot LDS-DMA kernel (variant 19; its ksplit is the number of 32- or 64-channel K slices)
        cands += [(19, d.cin_total // kc) for kc in (32, 64) if d.cin_total % (4 * kc) == 0]
    best, best_t = (1, 1), None
    for v, k in cands:
        d.variant, d.ksplit = v, k
        if lib.falnet_conv2d(ref, st) != 0:  # -2: variant not applicable to this launch
            if _AUTOTUNE_LOG:
                print(f"[autotune] M={M} w_rows={w_rows} Cout={Cout} v{v} k{k}: n/a ({lib.falnet_last_error().decode()}) "
                      f"src={[(d.src[i].C, d.src[i].H, d.src[i].W) for i in range(d.nsrc)]} cin_total={d.cin_total}", file=sys.stderr)
            continue
        lib.falnet_conv2d(ref, st)  # second warm-up: caches / clocks settled before timing
        t = None
        for _ in range(3):  # min over three short batches: robust against one-off stalls
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                lib.falnet_conv2d(ref, st)
            e1.record()
            e1.synchronize()
            tt = e0.elapsed_time(e1)
            t = tt if t is None else min(t, tt)
        if _AUTOTUNE_LOG:
            print(f"[autotune] M={M} w_rows={w_rows} Cout={Cout} ntaps={d.ntaps} nsrc={d.nsrc} v{v} k{k}: {t / reps * 1e3:.1f} us", file=sys.stderr)
        if best_t is None or t < best_t:
            best, best_t = (v, k), t
    return best


def _wgrad_plan(dtype, srcs, taps, stride_in, B, TH, TW, IH, IW, cin_pad, cout_pad, max_slabs, target_wgs=1536, up2=False):
    """(variant, nsplit, kernel symbol) of one weight-gradient launch -- the host-side mirror of conv.hip's kernel choice
    (falnet_wgrad re-checks the variant and fails loudly; whether the bias gradient is fused is ASKED from the library,
    falnet_wgrad_fuses_bias, never re-derived here)."""
    M = B * TH * TW
    h16 = dtype in (torch.bfloat16, torch.float16)
    tn = TYPE_SYM[dtype]
    dense = len(taps) == 9 and stride_in == 1 and TW >= 16  # halo-patch kernel (conv.hip: falnet_wgrad)
    c3 = len(srcs) == 1 and srcs[0].C == 3  # planar f32 image source (ops.planar_src): first-layer kernel, variant 6
    npatch = B * ((TH + 3) // 4) * ((TW + 31) // 32)
    if c3:
        assert h16 and dense and cout_pad == 32 and cin_pad == 32, "variant 6: 16-bit first layer, Cout 32"
        variant, nsplit, sym = 6, max(1, min(_WGRAD_WGS, npatch)), f"_Z18wgrad3x3_c3_kernelI{tn}Ev14falnet_wgrad_tiiii"
        if IW % 4 == 0 and TW >= 32 and not DETERMINISTIC:  # the wave-streaming form (csrc/wgrad_wave.hip: falnet_wgrad_c3wave_applicable)
            nsplit = max(1, min(_WGRAD_WAVE_WGS, B * ((TW + 31) // 32) * TH // (8 * _WGRAD_WAVE_MIN_ROWS)))
            sym = f"_Z22wgrad3x3_c3wave_kernelI{tn}Ev14falnet_wgrad_ti"
    elif _wgrad_s2(dtype, taps, stride_in, TH, TW, IH, IW, srcs) and _wgrad_wave_s2(srcs, IH, IW, TW, cin_pad, cout_pad):
        # wave-streaming kernel, stride-2 form (conv1's image source): a workgroup = ONE pixel range x 4 parity planes x 2 output-channel halves
        units = B * ((TW + 31) // 32) * TH
        nsplit = max(1, min(_WGRAD_WAVE_WGS, units // _WGRAD_WAVE_MIN_ROWS))
        variant, sym = 9, f"_Z22wgrad3x3_wave32_kernelI{tn}Li2ELb1EEv14falnet_wgrad_ti"
    elif _wgrad_s2(dtype, taps, stride_in, TH, TW, IH, IW, srcs) and _wgrad_rows_s2(srcs, IH, IW, TW, cin_pad, cout_pad):
        tiles = ((cin_pad + 63) // 64) * ((cout_pad + 63) // 64)
        units = B * ((TW + 31) // 32) * TH
        nsplit = max(1, min(_WGRAD_ROWS_WGS // tiles, units // _WGRAD_ROWS_MIN_ROWS))
        if nsplit >= 8:
            nsplit -= nsplit % 8
        variant, sym = 8, f"_Z23wgrad3x3_rows8s2_kernelI{tn}Li2EEv14falnet_wgrad_tiiii"
    elif _wgrad_s2(dtype, taps, stride_in, TH, TW, IH, IW, srcs):
        tiles = (cin_pad // 32) * (cout_pad // (64 if cout_pad % 64 == 0 else 32))
        variant, nsplit = 5, max(1, min((_WGRAD_WGS + tiles - 1) // tiles, npatch))
        sym = f"_Z18wgrad3x3_s2_kernelI{tn}Li{2 if cout_pad % 64 == 0 else 1}EEv14falnet_wgrad_tiiii"
    elif _wgrad_wave(dtype, dense, srcs, IH, IW, TW, cin_pad, cout_pad) and not up2:
        # wave-streaming kernel (32-channel input): one workgroup = 8 waves = 8 / (cout_pad / 32) pixel ranges, one slab per workgroup; every
        # range gets at least _WGRAD_WAVE_MIN_ROWS strip rows (two of its steps are halo rows)
        npw = 8 // (cout_pad // 32)
        units = B * ((TW + 31) // 32) * TH
        nsplit = max(1, min(_WGRAD_WAVE_WGS, units // (npw * _WGRAD_WAVE_MIN_ROWS)))
        variant, sym = 9, f"_Z22wgrad3x3_wave32_kernelI{tn}Li{cout_pad // 32}ELb0EEv14falnet_wgrad_ti"
    elif _wgrad_rows(dtype, dense, srcs, IH, IW, TW, cin_pad, cout_pad):
        tiles = ((cin_pad + 63) // 64) * ((cout_pad + 63) // 64)
        units = B * ((TW + 31) // 32) * TH * (4 if up2 else 1)  # up2: every parity class walks the whole low-resolution grid
        nsplit = max(1, min(_WGRAD_ROWS_WGS // tiles, units // _WGRAD_ROWS_MIN_ROWS))
        if nsplit >= 8:
            nsplit -= nsplit % 8  # multiples of 8: the channel tiles of one pixel range then share an XCD
        if up2:
            nsplit = max(4, nsplit - nsplit % 4)  # whole groups of the four parity classes
            max_slabs -= max_slabs % 4
            if max_slabs < 4:
                raise ValueError("up2 weight gradient: the slab budget holds fewer than four slabs")
        variant, sym = 7, f"_Z22wgrad3x3_rows16_kernelI{tn}Li4ELi2ELb{int(up2)}EEv14falnet_wgrad_tiiii"
    elif up2:
        raise ValueError("up2 weight gradient: the row-streaming kernel does not apply")
    elif dense:
        co2 = _wgrad_co2(dtype, dense, cin_pad, cout_pad)
        tiles = (cin_pad // 32) * (cout_pad // 32) // (2 if co2 else 1)
        nsplit = max(1, min((_WGRAD_WGS + tiles - 1) // tiles, npatch))
        variant = 3 if co2 else 0
        sym = f"_Z21wgrad3x3_patch_kernelI{tn}Li1ELi{2 if co2 else 1}EEv14falnet_wgrad_tiiii"
    else:
        tiles = ((cin_pad + 63) // 64) * ((cout_pad + 63) // 64) * len(taps)
        variant, nsplit, sym = 0, max(1, min((target_wgs + tiles - 1) // tiles, (M + 255) // 256)), f"_Z12wgrad_kernelI{tn}Ev14falnet_wgrad_ti"
    return variant, max(1, min(nsplit, max_slabs)), sym


def _fill_wgrad(d, dtype, srcs, IH, IW, gout, taps, stride_in, B, TH, TW, pc):
    d.nsrc = len(srcs)
    for i, s in enumerate(srcs):
        d.src[i] = s
    d.IH, d.IW = IH, IW
    gC = gout.shape[-1]
    d.gout, d.gC, d.cout = gout.data_ptr(), gC, pc.cout
    _fill_taps(d, taps)
    d.isy = d.isx = stride_in
    d.B, d.TH, d.TW = B, TH, TW
    d.cin_total = pc.cin_pad
    d.dtype = L.dtype_code(dtype)


def _fuse_bias(lib, d, grad_b):
    """Point the launch at the bias gradient when -- and only when -- the kernel the LIBRARY selects for it sums it."""
    if grad_b is None or not _FUSED_BIAS:
        return False
    if lib.falnet_wgrad_fuses_bias(C.byref(d)) != 1:
        return False
    d.bias_grad = grad_b.data_ptr()
    return True



def wgrad_calls(dtype, srcs, IH, IW, gout, taps, stride_in, B, TH, TW, pc, grad_w, grad_b, ws, target_wgs=1536,
                name="wgrad", flops=0, up2=False):
    """Weight (+bias) gradient of one conv: split-K partial slabs in `ws`, then a reduce into the
    OIHW f32 views `grad_w` / `grad_b`.  Returns a callable taking (accumulate).  up2: as WgradBatch.add."""
    lib = L.lib()
    d = L.Wgrad()
    _fill_wgrad(d, dtype, srcs, IH, IW, gout, taps, stride_in, B, TH, TW, pc)
    gC = gout.shape[-1]
    M = B * TH * TW * (4 if up2 else 1)
    slab = len(taps) * pad_c(gC) * pc.cin_pad * 4
    d.variant, nsplit, sym = _wgrad_plan(dtype, srcs, taps, stride_in, B, TH, TW, IH, IW, pc.cin_pad, pad_c(gC), ws.numel() * 4 // slab,
                                         target_wgs, up2=up2)
    d.nsplit = nsplit
    d.up2 = int(up2)
    d.partial = ws.data_ptr()
    assert lib.falnet_wgrad_workspace_bytes(C.byref(d)) <= ws.numel() * 4, "wgrad workspace too small"
    ref = C.byref(d)
    c0_real, c0_pad = pc.group_channels()
    npix = M
    keep = (d, srcs, gout, grad_w, grad_b, ws)

    def k_wgrad(accumulate=0, _keep=keep):
        L.check(lib.falnet_wgrad(ref, L.stream_ptr()), name)

    fused_bias = _fuse_bias(lib, d, grad_b)

    def k_reduce(accumulate=0):
        st = L.stream_ptr()
        L.check(lib.falnet_wgrad_reduce(L.ptr(ws), d.nsplit, len(taps), pad_c(gC), pc.cin_pad, L.ptr(grad_w), pc.cout, pc.cin, c0_real, c0_pad,
                                        int(accumulate), st), name + " reduce")
        if grad_b is not None and not fused_bias:
            L.check(lib.falnet_bias_grad(L.ptr(gout), npix, gC, pc.cout, L.ptr(grad_b), int(accumulate),
                                         L.dtype_code(dtype), st), name + " bias")
    t_wgrad = _timed(sym, flops, 0, k_wgrad)
    t_reduce = _timed("wgrad_reduce+bias_grad", 0, 0, k_reduce)

    def call(accumulate=0):
        if fused_bias and not accumulate:
            grad_b.zero_()  # the fused bias gradient ADDS (atomics), like the batched plan path into its pre-zeroed buffer
        t_wgrad(accumulate)
        t_reduce(accumulate)
    call.desc = d
    return call


_FUSED_BIAS = L.ab("FALNET_FUSED_BIAS", "1") == "1"  # bias gradients inside the halo weight-gradient kernels
_WGRAD_WGS = int(L.ab("FALNET_WGRAD_WGS", "256"))  # workgroups per dense weight-gradient launch (split-K factor = this / channel tiles): one per CU.  512 was the
# setting of rounds 2-4; with the round-5 kernels 256 is 0.4 % faster on the step on two boxes and halves these layers' slab bytes (profiles/r05_ab_wgrad_wgs.txt)


_WGRAD_ROWS_WGS = int(L.ab("FALNET_WGRAD_ROWS_WGS", "128"))  # workgroups (one per CU, eight waves) per row-streaming weight-gradient launch: half the chip,
# the other half runs the data-gradient chain beside it (same-box A/B: 128 beats 256 by 3 % on the step and halves the slab bytes)
_REDUCE_PER_LAYER = L.ab("FALNET_REDUCE_PER_LAYER", "0") == "1"
_REDUCE_BLOCKS = int(L.ab("FALNET_REDUCE_BLOCKS", "64"))  # blocks per layer of the batched slab reduce (split over slab groups)
_WGRAD_ROWS_MIN_ROWS = int(L.ab("FALNET_WGRAD_ROWS_MIN_ROWS", "8"))  # image rows per split-K range, at least


_WGRAD_WAVE_WGS = int(L.ab("FALNET_WGRAD_WAVE_WGS", "256"))  # workgroups of a wave-streaming weight-gradient launch (HBM-bound: the whole chip)
_WGRAD_WAVE_MIN_ROWS = int(L.ab("FALNET_WGRAD_WAVE_MIN_ROWS", "8"))


def _wgrad_wave(dtype, dense, srcs, IH, IW, TW, cin_pad, cout_pad):
    """Wave-streaming weight-gradient kernel (falnet_wgrad variant 9, csrc/wgrad_wave.hip): 16-bit dense 3x3 stride-1 layers over ONE
    32-channel NHWC source at the launch size with 32 or 64 (padded) output channels -- conv0_1's two convolutions, the skip group of the
    logits convolution."""
    if not (dense and dtype in (torch.bfloat16, torch.float16) and TW >= 32 and cin_pad == 32 and cout_pad in (32, 64)):
        return False
    if len(srcs) != 1 or srcs[0].C != 32 or srcs[0].H != IH or srcs[0].W != IW:
        return False
    return L.ab("FALNET_WGRAD_WAVE", "1") == "1"


def _wgrad_wave_s2(srcs, IH, IW, TW, cin_pad, cout_pad):
    """The stride-2 form of the wave-streaming kernel (falnet_wgrad variant 9 with isy = 2): ONE 32-channel NHWC source at the input size, 64 output
    channels -- conv1's image source (its constant `flow` channel goes through falnet_wgrad_const_plane)."""
    return (len(srcs) == 1 and srcs[0].C == 32 and srcs[0].H == IH and srcs[0].W == IW and srcs[0].sx != 0 and cin_pad == 32 and cout_pad == 64
            and TW >= 32 and L.ab("FALNET_WGRAD_WAVE", "1") == "1" and L.ab("FALNET_WGRAD_WAVE_S2", "1") == "1")


def _wgrad_rows(dtype, dense, srcs, IH, IW, TW, cin_pad, cout_pad):
    """Row-streaming weight-gradient kernel (falnet_wgrad variant 7, csrc/wgrad_rows.hip): 16-bit dense 3x3 stride-1 layers with
    at least 64 channels on one side (a 64 x 64 block per workgroup; 32 -> 32 layers would leave three of its four waves idle),
    32-pixel strips, sources at the launch size or exactly half of it."""
    if not (dense and dtype in (torch.bfloat16, torch.float16) and TW >= 32 and max(cin_pad, cout_pad) >= 64):
        return False
    if cin_pad == 32:  # a 32-channel source under a 64 x 64 block: half of the block's waves idle -> the 32 x 64 halo-patch kernel (variant 3)
        return False
    if any(not ((s.H == IH or 2 * s.H == IH) and (s.W == IW or 2 * s.W == IW)) or s.C % 32 for s in srcs):
        return False
    if str(cin_pad) in L.ab("FALNET_WGRAD_ROWS_SKIP_CIN", "").split(","):  # A/B: leave layers of this input width on the patch kernel
        return False
    return L.ab("FALNET_WGRAD_ROWS", "1") == "1"


def _wgrad_rows_s2(srcs, IH, IW, TW, cin_pad, cout_pad):
    """Row-streaming stride-2 weight gradient (falnet_wgrad variant 8, csrc/wgrad_rows.hip): one NHWC source at the input size with at
    least 64 input channels (64 x 64 blocks per workgroup: a 32-channel source would idle half of its waves), 32-pixel strips."""
    return (len(srcs) == 1 and srcs[0].H == IH and srcs[0].W == IW and srcs[0].sx != 0 and cin_pad >= int(L.ab("FALNET_WGRAD_ROWS_S2_MINCIN", "64")) and cout_pad >= 64 and TW >= 32
            and L.ab("FALNET_WGRAD_ROWS_S2", "1") == "1")


def _wgrad_s2(dtype, taps, stride_in, TH, TW, IH, IW, srcs):
    """Parity-plane halo kernel for 3x3 stride-2 weight gradients (falnet_wgrad variant 5): bf16, canonical taps, sources at
    the input size (or per-sample constants)."""
    return (dtype in H16 and len(taps) == 9 and stride_in == 2 and TW >= 32 and TH == (IH + 1) // 2 and TW == (IW + 1) // 2
            and all((s.H == IH and s.W == IW) or (s.sy == 0 and s.sx == 0) for s in srcs)
            and L.ab("FALNET_WGRAD_S2", "1") == "1")


def _wgrad_co2(dtype, dense, cin_pad, cout_pad):
    """32 (cin) x 64 (cout) channels per workgroup (falnet_wgrad variant 3): one gout fragment pair feeds twice the MFMAs
    (1.7 instead of 2.7 transposed LDS reads per MFMA).  In isolation -17 % on 64 -> 64 channel layers and -4 % on 128 -> 128;
    in the step (weight gradients are the longest chain of backward) +2.1 % pairs/s with the cut at 128 input channels, a little
    less with no cut (same-box A/B)."""
    return dense and dtype in H16 and cin_pad <= int(L.ab("FALNET_WGRAD_CO2_MAXCIN", "128")) and cout_pad % 64 == 0 and L.ab("FALNET_WGRAD_CO2", "1") == "1"


class WgradBatch:
    """All weight / bias gradients of one backward pass: per-layer split-K wgrad launches (each with its OWN slab
    region, so they can run back to back) followed by ONE batched slab reduce and ONE batched bias-gradient launch
    that add into the (pre-zeroed or accumulating) flat gradient buffer."""

    SLAB_CAP = int(L.ab("FALNET_SLAB_CAP_MB", "32")) << 20  # bytes of partial slabs per layer

    def __init__(self, dtype, device):
        self.dtype, self.device = dtype, device
        self.items, self.bias = [], []
        self.ws = None
        self.accumulate = 1  # set per backward by the plan: 0 = the step's gradient buffer was zeroed, single-writer entries may overwrite

    def add(self, srcs, IH, IW, gout, taps, stride_in, B, TH, TW, pc, grad_w, grad_b, name="wgrad", flops=0, bucket=0, up2=False):
        """up2: the weight gradient of a `deconv` layer on the LOW-resolution grid (falnet_wgrad_t::up2): `gout` at [B][2 TH][2 TW][gC], the one
        source at TH x TW = IH x IW; ValueError when the row-streaming kernel does not apply."""
        lib = L.lib()
        d = L.Wgrad()
        _fill_wgrad(d, self.dtype, srcs, IH, IW, gout, taps, stride_in, B, TH, TW, pc)
        gC = gout.shape[-1]
        M = B * TH * TW * (4 if up2 else 1)
        slab = len(taps) * pad_c(gC) * pc.cin_pad * 4
        d.variant, nsplit, sym = _wgrad_plan(self.dtype, srcs, taps, stride_in, B, TH, TW, IH, IW, pc.cin_pad, pad_c(gC), max(1, self.SLAB_CAP // slab),
                                             up2=up2)
        d.nsplit = nsplit
        d.up2 = int(up2)
        ref = C.byref(d)
        keep = (d, srcs, gout, grad_w, grad_b)

        post = []  # per-layer mode: this layer's slab reduce, launched right behind the wgrad (finalize() fills it in)

        def launch(_keep=keep):
            L.check(lib.falnet_wgrad(ref, L.stream_ptr()), name)
        c0_real, c0_pad = pc.group_channels()
        self.items.append(dict(bucket=bucket, post=post, d=d, bytes=nsplit * slab, nsplit=nsplit, ntaps=len(taps), w_rows=pad_c(gC), cin_total=pc.cin_pad,
                               cout=pc.cout, cin=pc.cin, c0_real=c0_real, c0_pad=c0_pad, grad=grad_w))
        if not _fuse_bias(lib, d, grad_b) and grad_b is not None:
            self.bias.append(dict(bucket=bucket, g=gout, npix=M, gC=gC, cout=pc.cout, db=grad_b))
        t_launch = _timed(sym, flops, 0, launch, name)

        def launch_and_reduce():
            t_launch()
            for c in post:
                c()
        launch_and_reduce.tag, launch_and_reduce.flops, launch_and_reduce.name = t_launch.tag, flops, name
        return launch_and_reduce

    def finalize(self):
        """Allocate the slab arena, point every wgrad descriptor at its region, upload the descriptor tables.
        Returns {bucket: (reduce_call, bias_call)}: one batched slab reduce and one batched bias-gradient launch per
        gradient bucket (buckets are finalised -- and all-reduced -- as backward passes them)."""
        lib = L.lib()
        total = sum((it["bytes"] + 255) // 256 * 256 for it in self.items)
        self.ws = torch.empty(max(total, 256) // 4, dtype=torch.float32, device=self.device)
        off = 0
        for it in self.items:
            it["d"].partial = self.ws.data_ptr() + off
            it["partial"] = self.ws.data_ptr() + off
            off += (it["bytes"] + 255) // 256 * 256
        out, self._tables = {}, []
        code = L.dtype_code(self.dtype)
        for bucket in sorted({it["bucket"] for it in self.items}):
            all_items = [it for it in self.items if it["bucket"] == bucket]
            # per-layer mode: a layer's slabs (<= 19 MB) are reduced right behind its wgrad launch, while they still sit in
            # L2 / the 256 MiB Infinity Cache, instead of by one launch per bucket that re-reads 100-300 MB from HBM
            items = [] if _REDUCE_PER_LAYER else all_items
            for it in (all_items if _REDUCE_PER_LAYER else []):
                one = (L.ReduceDesc * 1)()
                base = lib.falnet_wgrad_reduce_blocks(it["cout"], it["cin_total"], 1)
                groups = 1 if DETERMINISTIC else max(1, min(it["nsplit"] // 4, (_REDUCE_BLOCKS + base - 1) // base))
                r = one[0]
                r.partial, r.grad = it["partial"], it["grad"].data_ptr()
                r.nsplit, r.ntaps, r.w_rows, r.cin_total = it["nsplit"], it["ntaps"], it["w_rows"], it["cin_total"]
                r.cout, r.cin, r.c0_real, r.c0_pad, r.groups, r.block_begin = it["cout"], it["cin"], it["c0_real"], it["c0_pad"], groups, 0
                one_dev = torch.frombuffer(bytearray(bytes(one)), dtype=torch.uint8).to(self.device)
                self._tables.append(one_dev)

                def reduce_one(one_dev=one_dev, blocks=base * groups):
                    L.check(lib.falnet_wgrad_reduce_batched(L.ptr(one_dev), 1, blocks, int(self.accumulate), L.stream_ptr()), "wgrad_reduce")
                it["post"].append(_timed("wgrad_reduce_batched", 0, 0, reduce_one, "wgrad_reduce(layer)"))
            red = (L.ReduceDesc * max(len(items), 1))()
            blk = 0
            for i, it in enumerate(items):
                # slab groups: enough blocks to pull the slabs at the HBM rate (a block keeps <= 18 16-B loads per thread in
                # flight), at least four slabs per group; groups > 1 makes the kernel add with atomics
                base = lib.falnet_wgrad_reduce_blocks(it["cout"], it["cin_total"], 1)
                groups = 1 if DETERMINISTIC else max(1, min(it["nsplit"] // 4, (_REDUCE_BLOCKS + base - 1) // base))
                blocks = base
                r = red[i]
                r.partial, r.grad = it["partial"], it["grad"].data_ptr()
                r.nsplit, r.ntaps, r.w_rows, r.cin_total = it["nsplit"], it["ntaps"], it["w_rows"], it["cin_total"]
                r.cout, r.cin, r.c0_real, r.c0_pad, r.groups, r.block_begin = it["cout"], it["cin"], it["c0_real"], it["c0_pad"], groups, blk
                blk += blocks * groups
            red_dev = torch.frombuffer(bytearray(bytes(red)), dtype=torch.uint8).to(self.device)
            biases = [b for b in self.bias if b["bucket"] == bucket]
            bias = (L.BiasGradDesc * max(len(biases), 1))()
            bblk = 0
            for i, it in enumerate(biases):
                segs = it["gC"] // 8
                rows = 256 // min(segs, 256)
                blocks = int(max(1, min(256, (it["npix"] + rows * 64 - 1) // (rows * 64))))
                b = bias[i]
                b.g, b.db, b.npix, b.gC, b.cout, b.blocks, b.block_begin = it["g"].data_ptr(), it["db"].data_ptr(), it["npix"], it["gC"], it["cout"], blocks, bblk
                bblk += blocks
            bias_dev = torch.frombuffer(bytearray(bytes(bias)), dtype=torch.uint8).to(self.device)
            self._tables += [red_dev, bias_dev]

            def reduce_all(red_dev=red_dev, n=len(items), blocks=blk):
                if n:
                    L.check(lib.falnet_wgrad_reduce_batched(L.ptr(red_dev), n, blocks, int(self.accumulate), L.stream_ptr()), "wgrad_reduce_batched")

            bias_ws = torch.empty(max(bblk, 1) * 512, dtype=torch.float32, device=self.device) if DETERMINISTIC and biases else None
            self._tables.append(bias_ws)

            def bias_all(bias_dev=bias_dev, n=len(biases), blocks=bblk, bias_ws=bias_ws):
                if n and bias_ws is not None:  # deterministic mode: per-block partials, added in block order
                    L.check(lib.falnet_bias_grad_batched_det(L.ptr(bias_dev), n, blocks, code, L.ptr(bias_ws), bias_ws.numel(), L.stream_ptr()),
                            "bias_grad_batched_det")
                elif n:
                    L.check(lib.falnet_bias_grad_batched(L.ptr(bias_dev), n, blocks, code, L.stream_ptr()), "bias_grad_batched")
            out[bucket] = (_timed("wgrad_reduce_batched", 0, 0, reduce_all, "wgrad_reduce_batched"),
                           _timed("bias_grad_batched", 0, 0, bias_all, "bias_grad_batched"))
        return out


def replay_ok():
    """May a plan issue its recorded launch sequence through falnet_replay right now?  Not inside bench.py's instrumented pass (per-launch
    events), not while another sequence is being recorded, not under hipGraph capture."""
    return L.REPLAY and TIMER is None and not L.recording() and not torch.cuda.is_current_stream_capturing()


class ReplayList:
    """A fixed list of launches on ONE stream (a VGG plan's forward / backward, a stretch of the backbone's forward).  Called like the
    loop over the list it replaces; after two eager passes (autotuning done, every launch has run once) the list is recorded and later
    calls are one falnet_replay.  `eager_head`: leading launches with per-call inputs (conv_c3_call.set_input) stay eager."""

    def __init__(self, calls, eager_head=0):
        self.calls, self.head, self.seg, self.runs = list(calls), eager_head, None, 0

    def __call__(self):
        if not replay_ok():
            for c in self.calls:
                c()
            return
        for c in self.calls[:self.head]:
            c()
        if self.seg is None:
            if self.runs < 2:
                self.runs += 1
                for c in self.calls[self.head:]:
                    c()
                return
            self.seg = L.record_calls(self.calls[self.head:])
        self.seg.run(L.stream_ptr().value)


def simple_call(fn_name, *args, name=None, nbytes=0):
    lib = L.lib()
    fn = getattr(lib, fn_name)

    def launch():
        L.check(fn(*args, L.stream_ptr()), name or fn_name)
    return _timed(fn_name, 0, nbytes, launch, name or fn_name)
