"""Loss surface of the reference (loss_functions.py) over libfalnet_hip.so kernels.

Same names and argument meaning: `vgg(x, full=False)`, `rec_loss_fnc(mask, synth, label, vgg_label, a_p)`,
`perceptual_loss(out_vgg, label_vgg, layer=None)`, `smoothness(img, disp, gamma=1)`, `realEPE`.
Unlike the reference, importing this module needs no GPU, no torchvision and no download
(loss_functions.py:4,10-11,48 do all three at import): the VGG19 slices are built lazily, with weights
from a torchvision-format state_dict file if `FALNET_VGG19_WEIGHTS` names one, else seeded (SURVEY.md 8c:
perceptual parity with the ImageNet weights is unpinned).

VGG feature maps are returned as logical NCHW tensors that are channels-last in memory (the compute
layout), in the compute dtype.  All reductions are f32.
"""
import os
import weakref

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import ops, synthetic
from .ops import PackedConv

_MEAN = (0.411, 0.432, 0.45)
# conv index in torchvision's features -> (Cin, Cout); slices end with MaxPool (loss_functions.py:21-29)
_SLICES = ((0, 2), (5, 7), (10, 12, 14, 16))


_GOUT_OF = {}  # data_ptr of a grad-plan's output map -> [that plan's gradient buffer for it, claimed-in-this-backward flag]
_MAX_HELD_PLANS = 4  # grad-enabled VGG calls per shape whose backward has not run yet; beyond it the oldest plan is recycled


def _forget_gouts(keys):
    for k in keys:
        _GOUT_OF.pop(k, None)


def _release_plan(plan, gen):
    if plan.gen == gen:
        plan.busy = False


class _VggPlan:
    """Static launch plan of VGG19 features[0:19] forward + data-gradient for one (B,H,W,dtype)."""

    def __init__(self, owner, B, H, W, dtype, device, need_grad=True):
        """need_grad=False (label features, Train_Stage1_K.py:241-244 under no_grad): the convs in front of a pool keep only
        their pooled map (fused pool, no full-resolution store) and no backward launches are built."""
        self.B, self.H, self.W, self.dtype, self.device = B, H, W, dtype, device
        self.busy, self.need_grad, self.gen = False, need_grad, 0
        code = L.dtype_code(dtype)
        self.fwd, self.bwd = [], []
        _conv = lambda *a, **kw: ops.conv_call(*a, ws_owner=("vgg", id(self)), **kw)  # own split-K scratch per plan instance
        pcs = owner._packed
        x_in = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
        x0 = x_in  # the first conv reads the planar f32 image directly (falnet_conv3x3_c3): no layout conversion
        self.x_in, self.keep = x_in, []
        cur, h, w = x0, H, W
        self.outs, acts = [], []  # acts: (pc, input tensor, relu output tensor, h, w)
        for convs in _SLICES:
            for idx in convs:
                pc = pcs[idx]
                last = idx == convs[-1]
                pooled = torch.empty(B, h // 2, w // 2, pc.cout, dtype=dtype, device=device) if last else None
                y = torch.empty(B, h, w, pc.cout, dtype=dtype, device=device) if (need_grad or not last) else None
                kw = dict(bias=pc.bias, act=L.ACT_RELU, name=f"vgg conv{idx}", flops=2 * B * h * w * pc.cout * pc.cin * 9)
                args = (dtype, [ops.nhwc_src(cur)], h, w, pc.wf, pc.cin_pad, ops.fwd_taps(3), 9, pc.cout_pad, 1, B, h, w)
                fused = False
                if cur is x0:
                    self.c3_call = ops.conv_c3_call(dtype, x_in, pc, y, L.ACT_RELU, name="vgg conv0(c3)")
                    self.fwd.append(self.c3_call)
                elif last and L.ab("FALNET_FUSED_POOL", "1") == "1":
                    try:  # 2x2 max pool in the conv epilogue: the full-resolution map is not re-read (nor written at all for labels)
                        self.fwd.append(_conv(*args, y, h, w, pc.cout, pc.cout, pool_out=pooled, **kw))
                        fused = True
                    except ValueError:
                        pass
                if not fused and cur is not x0:
                    if y is None:
                        y = torch.empty(B, h, w, pc.cout, dtype=dtype, device=device)
                    self.fwd.append(_conv(*args, y, h, w, pc.cout, pc.cout, **kw))
                if last and not fused:
                    self.fwd.append(ops.simple_call("falnet_maxpool2_fwd", L.ptr(y), L.ptr(pooled), B, h, w, pc.cout, code))
                acts.append((pc, cur, y, h, w))
                cur = y
            acts.append(("pool", cur, pooled, h, w))
            self.outs.append(pooled)
            cur, h, w = pooled, h // 2, w // 2
        self.run_fwd = ops.ReplayList(self.fwd, eager_head=1)  # (the first conv reads the caller's image: set_input per call)
        self.run_bwd = None
        if not need_grad:
            return
        # ---- backward: gradients of the three pooled outputs -> gradient of the planar f32 input ----
        self.gouts = [torch.empty_like(o) for o in self.outs]
        for o, g in zip(self.outs, self.gouts):
            _GOUT_OF[o.data_ptr()] = [g, False]  # lets the perceptual loss write d loss / d feature straight into the plan
        weakref.finalize(self, _forget_gouts, [o.data_ptr() for o in self.outs])  # a dropped plan leaves no stale entries
        g_next = None  # gradient wrt `cur` of the step being undone (post-pool tensor of the slice below)
        slice_i = len(self.outs) - 1
        for entry in reversed(acts):
            if entry[0] == "pool":
                _, x, pooled, h, w = entry
                # deeper slices feed back into this pooled output: their data gradient already holds the sum (the conv below
                # this pool took the slice's own gradient as its epilogue addend), the deepest slice has only its own
                gy = self.gouts[slice_i] if g_next is None else g_next
                gx = torch.empty_like(x)  # gradient wrt the pre-ReLU conv output feeding the pool (relu' fused)
                self.bwd.append(ops.simple_call("falnet_maxpool2_bwd", L.ptr(x), L.ptr(pooled), L.ptr(gy), L.ptr(gx), B, h, w,
                                                x.shape[3], code))
                g_next, slice_i = gx, slice_i - 1
                self.keep.append(gx)
            else:
                pc, x, y, h, w = entry
                if x is x0:
                    # first conv: data gradient wrt the 3-channel image, written straight into the planar f32 gradient the
                    # caller gets (the halo kernels' epilogue has the pixels on the lanes: no NHWC intermediate, no conversion)
                    self.g_in = torch.empty(B, 3, H, W, dtype=torch.float32, device=device)
                    self.bwd.append(_conv(dtype, [ops.nhwc_src(g_next)], h, w, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(3), 9,
                                          ops.pad_c(3), 1, B, h, w, self.g_in, h, w, 3, 0, out_layout=L.OUT_PLANAR_F32, name="vgg dgrad0",
                                          flops=2 * B * h * w * pc.cout * pc.cin * 9))
                else:
                    # x is the ReLU output of the previous conv (or a pooled map: relu' = 1 where > 0 holds there too,
                    # but a pooled map's gradient must NOT be masked -> only mask when x came from a conv)
                    from_pool = any(x is o for o in self.outs)
                    gin = torch.empty_like(x)
                    own = next((g for o, g in zip(self.outs, self.gouts) if x is o), None)  # perceptual gradient of that slice
                    self.bwd.append(_conv(dtype, [ops.nhwc_src(g_next)], h, w, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(3), 9,
                                                  pc.cin_pad, 1, B, h, w, gin, h, w, pc.cin_pad, pc.cin_pad, addend=own,
                                                  actout=None if from_pool else x,
                                                  actout_kind=L.ACT_NONE if from_pool else L.ACT_RELU, name="vgg dgrad",
                                                  flops=2 * B * h * w * pc.cout * pc.cin * 9))
                    self.keep.append(gin)
                    g_next = gin
        self.run_bwd = ops.ReplayList(self.bwd)


class _VggFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, plan, x):
        plan.c3_call.set_input(x)  # the first conv reads the caller's image in place (it is only read during this forward)
        plan.run_fwd()
        ctx.plan, ctx.owner, ctx.gen = plan, owner, plan.gen
        # no clone: the plan (and its output buffers) stays reserved for this call until its backward has run -- or until the graph
        # is dropped without one (validation under grad mode, an exception): the finalizer frees it then
        weakref.finalize(ctx, _release_plan, plan, plan.gen)
        return tuple(o.permute(0, 3, 1, 2) for o in plan.outs)

    @staticmethod
    def backward(ctx, *gouts):
        plan = ctx.plan
        if plan.gen != ctx.gen:
            raise RuntimeError(f"this VGG call's plan was recycled: more than {_MAX_HELD_PLANS} grad-enabled vgg() calls of one shape "
                               "were alive without a backward")
        for buf, g in zip(plan.gouts, gouts):
            if g is None:
                buf.zero_()
            elif g.data_ptr() != buf.data_ptr():  # _PerceptualMse writes its gradient straight into this buffer
                buf.copy_(g.permute(0, 2, 3, 1))
        plan.run_bwd()
        out = plan.g_in.clone()
        plan.busy = False
        return None, None, out


class Vgg19_pc(nn.Module):
    """VGG19 `features[0:19]` in three slices (loss_functions.py:7-44); frozen."""

    def __init__(self, requires_grad=False, compute_dtype=None, state_dict=None):
        super().__init__()
        self.features = nn.ModuleDict({str(i): nn.Conv2d(cin, cout, 3, padding=1) for i, (cin, cout) in synthetic.VGG19_PC_CONVS.items()})
        sd = state_dict
        path = os.environ.get("FALNET_VGG19_WEIGHTS")
        if sd is None and path:
            sd = torch.load(path, map_location="cpu")
        self.weights_source = "file" if sd is not None else "seeded"
        if sd is None:
            sd = synthetic.seeded_vgg19_state_dict()
        with torch.no_grad():
            for i, conv in self.features.items():
                conv.weight.copy_(sd[f"features.{i}.weight"])
                conv.bias.copy_(sd[f"features.{i}.bias"])
        for p in self.parameters():
            p.requires_grad = requires_grad
        self.compute_dtype = compute_dtype
        self._plans, self._packed, self._packed_key = {}, None, None

    def _prepare(self, device, dtype):
        key = (device, dtype)
        if self._packed_key != key:
            if next(self.parameters()).device != device:
                self.to(device)
            self._packed = {}
            for i, conv in self.features.items():
                pc = PackedConv(f"vgg{i}", conv.weight, conv.bias, [conv.weight.shape[1]], 1)
                pc.alloc(dtype, device)
                pc.pack_call()()  # frozen weights: packed once
                self._packed[int(i)] = pc
            self._packed_key, self._plans = key, {}

    def _plan(self, B, H, W, dtype, device, hold):
        pool = self._plans.setdefault((B, H, W, hold), [])  # hold = the call needs a backward (plan busy until it ran)
        p = next((q for q in pool if not q.busy), None)
        if p is None and hold and len(pool) >= _MAX_HELD_PLANS:
            p = min(pool, key=lambda q: q.gen)  # bounded HBM: recycle the oldest held plan (its backward, if it ever comes, raises)
        if p is None:
            p = _VggPlan(self, B, H, W, dtype, device, need_grad=hold)
            pool.append(p)
        self._gen = getattr(self, "_gen", 0) + 1
        p.busy, p.gen = hold, self._gen
        if hold:
            for o in p.outs:
                _GOUT_OF[o.data_ptr()][1] = False  # nobody has written this call's feature gradients yet
        return p

    def forward(self, x, full=False, borrow=False):
        """borrow=True (no-grad calls): return views of the plan's output buffers instead of copies -- valid until the next
        no-grad call with the same input shape (the trainer's per-step label features, Train_Stage1_K.py:241-244)."""
        if full:
            raise NotImplementedError("slice4 (relu4_4) is never used on the FAL_net hot path (loss_functions.py:40-42)")
        if not x.is_cuda:
            raise RuntimeError("fal_net_amd Vgg19_pc runs on an MI355X only (no CPU fallback)")
        dtype = self.compute_dtype or _default_dtype()
        self._prepare(x.device, dtype)
        B, C, H, W = x.shape
        assert C == 3 and H % 8 == 0 and W % 8 == 0, "VGG19 slices need a 3-channel input with H, W divisible by 8"
        need_grad = torch.is_grad_enabled() and x.requires_grad
        plan = self._plan(B, H, W, dtype, x.device, hold=need_grad)
        xs = x.float().contiguous() if need_grad else x.detach().float().contiguous()
        if need_grad:
            return _VggFunction.apply(self, plan, xs)
        with torch.no_grad():
            plan.c3_call.set_input(xs)
            plan.run_fwd()
            return tuple((o if borrow else o.clone()).permute(0, 3, 1, 2) for o in plan.outs)


_DEFAULT_DTYPE = [torch.float32]


def set_compute_dtype(dtype):
    """Compute dtype of the module-global `vgg` (float32 = parity path, bfloat16 = throughput path)."""
    _DEFAULT_DTYPE[0] = dtype


def _default_dtype():
    return _DEFAULT_DTYPE[0]


class _LazyVgg:
    """Module-global `vgg` of the reference (loss_functions.py:48), constructed on first use."""

    def __init__(self):
        self._m = None

    def _get(self):
        if self._m is None:
            self._m = Vgg19_pc()
        return self._m

    def __call__(self, x, full=False, **kw):
        return self._get()(x, full, **kw)

    def __getattr__(self, name):
        return getattr(self._get(), name)


vgg = _LazyVgg()


# ------------------------------------------------------------------------------------------ scalar losses
class _L1Mean(torch.autograd.Function):
    """scale * sum(mask * |synth - label|); scale defaults to 1 / numel (the mean of loss_functions.py:53)."""

    @staticmethod
    def forward(ctx, synth, label, mask, scale=None):
        B, C, H, W = synth.shape
        a, b = synth.contiguous(), label.contiguous()
        m = None if mask is None else mask.expand(B, 1, H, W).contiguous()
        out = torch.empty(1, device=synth.device)
        sc = 1.0 / (B * C * H * W) if scale is None else float(scale)
        L.check(L.lib().falnet_l1_fwd(L.ptr(a), L.ptr(b), L.ptr(m), B, C, H * W, sc, L.ptr(out), 0, L.stream_ptr()), "l1_fwd")
        ctx.save_for_backward(a, b, m if m is not None else torch.empty(0))
        ctx.has_mask, ctx.sc = m is not None, sc
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b, m = ctx.saved_tensors
        B, C, H, W = a.shape
        ga = torch.empty_like(a)
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().falnet_l1_bwd(L.ptr(a), L.ptr(b), L.ptr(m if ctx.has_mask else None), B, C, H * W, ctx.sc, L.ptr(gs),
                                      L.ptr(ga), 0, L.stream_ptr()), "l1_bwd")
        return ga, None, None, None


def occlusion_mask(a, b, x0, x1):
    """Stage-2 occlusion mask (Train_Stage2_K.py:296-302): a * b with the column window [x0, x1) forced to 1; no gradient."""
    B, _, H, W = a.shape
    a, b = a.detach().contiguous(), b.detach().contiguous()
    out = torch.empty_like(a)
    L.check(L.lib().falnet_occlusion_mask(L.ptr(a), L.ptr(b), L.ptr(out), B, H, W, int(x0), int(x1), L.stream_ptr()), "occlusion_mask")
    return out


def mirror_loss_fnc(disp, teacher_disp, occ, x0, x1):
    """Mirror loss of one view (Train_Stage2_K.py:319-324): mean over the window [x0, x1) of
    (1 / max teacher disparity of the sample) * (1 - occ) * |disp - teacher_disp|; gradient wrt `disp` only."""
    B, _, H, W = disp.shape
    t = teacher_disp.detach().contiguous()
    rmax = torch.empty(B, device=disp.device)
    L.check(L.lib().falnet_rowmax(L.ptr(t), L.ptr(rmax), B, H * W, L.stream_ptr()), "rowmax")  # F.max_pool2d(kernel=(H, W)), :319
    w = torch.empty_like(t)
    occ = occ.detach().contiguous()
    L.check(L.lib().falnet_mirror_weight(L.ptr(occ), L.ptr(rmax), L.ptr(w), B, H, W, int(x0), int(x1), L.stream_ptr()), "mirror_weight")
    return _L1Mean.apply(disp, t, w, 1.0 / (B * H * (x1 - x0)))


class _MaskMix(torch.autograd.Function):
    """mask*synth + (1-mask)*label (loss_functions.py:55); gradient wrt synth only (masks carry no grad)."""

    @staticmethod
    def forward(ctx, synth, label, mask):
        B, C, H, W = synth.shape
        a, b = synth.contiguous(), label.contiguous()
        m = mask.expand(B, 1, H, W).contiguous()
        out = torch.empty_like(a)
        L.check(L.lib().falnet_mask_mix(L.ptr(a), L.ptr(b), L.ptr(m), L.ptr(out), B, C, H * W, L.stream_ptr()), "mask_mix")
        ctx.save_for_backward(m)
        return out

    @staticmethod
    def backward(ctx, g):
        (m,) = ctx.saved_tensors
        return g * m, None, None


def _nhwc(t):
    """Memory view (B,H,W,C) of a logical-NCHW tensor, made channels-last contiguous if it is not."""
    p = t.permute(0, 2, 3, 1)
    return p if p.is_contiguous() else p.contiguous()


class _PerceptualMse(torch.autograd.Function):
    """sum_i mean((a_i - b_i)^2) over the VGG maps (loss_functions.py:61-65)."""

    @staticmethod
    def forward(ctx, n, *maps):
        outs, labels = maps[:n], maps[n:]
        out = torch.empty(1, device=outs[0].device)
        saved, scales = [], []
        for i, (a, b) in enumerate(zip(outs, labels)):
            an, bn = _nhwc(a), _nhwc(b.to(a.dtype))
            sc = 1.0 / a.numel()
            B, H, W, Cc = an.shape
            L.check(L.lib().falnet_mse_fwd(L.ptr(an), L.ptr(bn), B * H * W, Cc, sc, L.ptr(out), int(i > 0), L.dtype_code(an.dtype),
                                           L.stream_ptr()), "mse_fwd")
            saved += [an, bn]
            scales.append(sc)
        ctx.save_for_backward(*saved)
        ctx.scales, ctx.n = scales, n
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        gs = g.reshape(1).contiguous().float()
        grads = []
        for i in range(ctx.n):
            an, bn = ctx.saved_tensors[2 * i], ctx.saved_tensors[2 * i + 1]
            slot = _GOUT_OF.get(an.data_ptr())  # `an` is a VGG plan's output: write into the plan's own gradient buffer,
            ga = None                            # once -- a second loss on the same map gets its own (autograd sums them)
            if slot is not None and not slot[1] and slot[0].shape == an.shape and slot[0].dtype == an.dtype:
                ga, slot[1] = slot[0], True
            if ga is None:
                ga = torch.empty_like(an)
            B, H, W, Cc = an.shape
            L.check(L.lib().falnet_mse_bwd(L.ptr(an), L.ptr(bn), B * H * W, Cc, ctx.scales[i], L.ptr(gs), L.ptr(ga),
                                           L.dtype_code(an.dtype), L.stream_ptr()), "mse_bwd")
            grads.append(ga.permute(0, 3, 1, 2))
        return (None, *grads, *([None] * ctx.n))


def perceptual_loss(out_vgg, label_vgg, layer=None):
    if layer is not None:
        return _PerceptualMse.apply(1, out_vgg[layer], label_vgg[layer])
    return _PerceptualMse.apply(3, *out_vgg[:3], *label_vgg[:3])


def rec_loss_fnc(mask, synth, label, vgg_label, a_p):
    """mean(mask*|synth-label|) + a_p * perceptual(vgg(mask*synth + (1-mask)*label), vgg_label)  (loss_functions.py:52-56)."""
    mt = mask if torch.is_tensor(mask) else None
    if mt is None and mask != 1:
        mt = torch.full((synth.shape[0], 1, synth.shape[2], synth.shape[3]), float(mask), device=synth.device)
    loss = _L1Mean.apply(synth, label, mt)
    if a_p > 0 and vgg_label is not None:
        mixed = synth if mt is None else _MaskMix.apply(synth, label, mt)
        loss = loss + a_p * perceptual_loss(vgg(mixed), vgg_label)
    return loss


class _Smoothness(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, disp, gamma, x0, x1):
        B, _, H, W = disp.shape
        out = torch.empty(1, device=disp.device)
        sc = 1.0 / (B * H * (x1 - x0))
        L.check(L.lib().falnet_smooth_fwd(L.ptr(img), L.ptr(disp), B, H, W, x0, x1, float(gamma), sc, L.ptr(out), 0, L.stream_ptr()),
                "smooth_fwd")
        ctx.save_for_backward(img, disp)
        ctx.args = (float(gamma), x0, x1, sc)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        img, disp = ctx.saved_tensors
        gamma, x0, x1, sc = ctx.args
        B, _, H, W = disp.shape
        gd = torch.empty_like(disp)
        gs = g.reshape(1).contiguous().float()
        L.check(L.lib().falnet_smooth_bwd(L.ptr(img), L.ptr(disp), B, H, W, x0, x1, gamma, sc, L.ptr(gs), L.ptr(gd), 0,
                                          L.stream_ptr()), "smooth_bwd")
        return None, gd, None, None, None


def _column_window(t):
    """If `t` is a column slice t_full[..., x0:x1] of a contiguous tensor, return (t_full, x0, x1)."""
    base = t._base if t._base is not None else None
    if base is not None and base.dim() == 4 and base.is_contiguous() and base.shape[:3] == t.shape[:3] \
            and t.stride() == base.stride():
        x0 = (t.storage_offset() - base.storage_offset())
        if 0 <= x0 and x0 + t.shape[3] <= base.shape[3]:
            return base, x0, x0 + t.shape[3]
    return None


def smoothness(img, disp, gamma=1):
    """Edge-aware smoothness (loss_functions.py:70-101).  The callers pass column crops
    (`left[:, :, :, c:]`, `disp[:, :, :, c:]`, Train_Stage1_K.py:255); when both are views of contiguous
    tensors the kernel works on the parents with a column window, without copies."""
    wi, wd = _column_window(img), _column_window(disp)
    if wi is not None and wd is not None and wi[1:] == wd[1:] and wi[0].shape[3] == wd[0].shape[3]:
        full = _Smoothness.apply(wi[0].detach(), wd[0], gamma, wd[1], wd[2])
        return full
    return _Smoothness.apply(img.detach().contiguous(), disp.contiguous(), gamma, 0, disp.shape[3])


# ------------------------------------------------------------------------------------------ validation metric (plumbing)
def EPE(net_out, target, sparse=False, disp=True, mean=True):
    """loss_functions.py:124-138 (validation metric, not on the timed path)."""
    EPE_map = torch.norm(target - net_out, p=2, dim=1)
    batch_size = EPE_map.size(0)
    if sparse:
        mask = target[:, 0] == 0 if disp else (target[:, 0] == 0) & (target[:, 1] == 0)
        EPE_map = EPE_map[~mask]
    return EPE_map.mean() if mean else EPE_map.sum() / batch_size


def realEPE(output, target, sparse=False):
    """loss_functions.py:170-173."""
    b, _, h, w = target.size()
    up = F.interpolate(output, size=(h, w), mode="bilinear", align_corners=True)
    return EPE(up, target, sparse, mean=True)
