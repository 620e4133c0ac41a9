"""FAL_netB on MI355X: the reference's nn.Module surface over hand-written HIP kernels.

Drop-in for `models.FAL_netB` of JuanLuisGonzalez/FAL_net (models/FAL_netB.py:28-32,179-297):
same factory signature, same `forward(input_left, min_disp, max_disp, ret_disp, ret_subocc, ret_pan)`
contract and return ordering, same `state_dict` keys (51 tensors), same
`weight_parameters()` / `bias_parameters()`.  Underneath there is no aten compute: the module owns a
static *plan* per input shape -- pre-allocated NHWC activations and a list of libfalnet_hip.so
launches (implicit-GEMM MFMA convolutions with fused bias/ELU/residual/concat/upsample, the fused MED
head) -- and one autograd.Function whose backward replays the hand-written adjoint plan and deposits
weight gradients straight into a flat f32 buffer that `p.grad` views alias (one RCCL all-reduce and one
fused Adam launch operate on that buffer, see fal_net_amd/train.py).

`compute_dtype`: torch.float32 (exact-f32 MFMA; the parity path, 1e-4 vs the reference) or
torch.bfloat16 (bf16 MFMA with f32 accumulation; the throughput path).  Parameters, the MED head,
losses and Adam stay f32 in both.
"""
import torch
import torch.nn as nn

from .. import _lib as L
from ..arch import ARCHS
from ..ops import PackedConv

__all__ = ["FAL_netB"]


def FAL_netB(data=None, no_levels=49, compute_dtype=None):
    """Factory with the reference's signature (models/FAL_netB.py:28-32)."""
    return _make("B", data, no_levels, compute_dtype)


def _make(arch, data, no_levels, compute_dtype):
    model = FAL_net(batchNorm=False, no_levels=no_levels, compute_dtype=compute_dtype, arch=arch)
    if data is not None:
        model.load_state_dict(data["state_dict"])
    return model


# ---- parameter holders mirroring the reference module tree (so state_dict keys match) ----
def conv_elu(batchNorm, in_planes, out_planes, kernel_size=3, stride=1, pad=1):
    assert not batchNorm, "FAL_netB is built with batchNorm=False (models/FAL_netB.py:29)"
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size, stride, pad, bias=True), nn.ELU(inplace=True))


class deconv(nn.Module):
    def __init__(self, in_planes, out_planes):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, out_planes, 3, 1, 1, bias=False)


class residual_block(nn.Module):
    def __init__(self, in_planes, separable=False):
        super().__init__()
        if separable:  # FAL_netA.py:73-76: a 3x1 then a 1x3 convolution
            self.conv1 = nn.Conv2d(in_planes, in_planes, (3, 1), padding=(1, 0), bias=False)
            self.conv2 = nn.Conv2d(in_planes, in_planes, (1, 3), padding=(0, 1), bias=False)
        else:
            self.conv1 = nn.Conv2d(in_planes, in_planes, 3, padding=1, bias=False)
            self.conv2 = nn.Conv2d(in_planes, in_planes, 3, padding=1, bias=False)


def predict_amask(in_planes, out_planes):
    # constructed but never executed by the reference (FAL_netB.py:128); kept for checkpoint compatibility
    return nn.Sequential(nn.Conv2d(in_planes, in_planes // 2, 3, 1, 1, bias=True), nn.ELU(inplace=True),
                         nn.Conv2d(in_planes // 2, out_planes, 3, 1, 1, bias=False), nn.Sigmoid())


class BackBone(nn.Module):
    """Parameter tree of the reference BackBone (FAL_netB.py:92-138, FAL_netA.py:92-135, FAL_netC.py:96-137), built in the
    reference's registration order from the variant's layer table (fal_net_amd/arch.py); compute lives in FalnetPlan."""

    def __init__(self, batchNorm=False, no_in=3, no_flow=1, no_out=64, arch="B"):
        super().__init__()
        t = ARCHS[arch]
        enc, dec = t["enc"], t["dec"]
        for i, ch in enumerate(enc):
            cin = no_in if i == 0 else enc[i - 1] + (no_flow if i == 1 else 0)
            setattr(self, f"conv{i}", conv_elu(batchNorm, cin, ch, stride=1 if i == 0 else 2))
            setattr(self, f"conv{i}_1", residual_block(ch, t["separable"]))
        below = enc[6]
        for lvl in range(6, 1, -1):
            dch, ich = dec[lvl]
            setattr(self, f"deconv{lvl}", deconv(below, dch))
            setattr(self, f"iconv{lvl}", conv_elu(batchNorm, dch + enc[lvl - 1], ich))
            below = ich
        self.deconv1 = deconv(below, 64)
        self.iconv1 = nn.Conv2d(enc[0] + 64, no_out, 3, 1, 1, bias=False)
        if t["amask"]:
            self.amask_conv = predict_amask(32 + 64, 1)
        for m in self.modules():  # FAL_netB.py:131-138
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()


from ..plan import FalnetPlan, _TAIL_LEVELS  # noqa: E402  (the launch plan / stream scheduler of this module's forward and backward)


class _FalnetFunction(torch.autograd.Function):
    """One autograd node for the whole network: forward = plan replay, backward = adjoint plan replay.
    Parameter gradients are deposited directly into the model's flat gradient buffer."""

    @staticmethod
    def forward(ctx, model, plan, left, min_disp, max_disp, ret_disp, ret_subocc, ret_pan, anchor):
        gen = plan.run_forward(left, min_disp, max_disp, ret_disp, ret_subocc, ret_pan)
        ctx.plan, ctx.gen = plan, gen
        b = plan.buf
        outs = [b["p_im0"].clone() if ret_pan else None, b["disp"].clone() if ret_disp else None]
        if ret_subocc:
            outs += [b["maskL"].clone(), b["maskR"].clone()]
        else:
            outs += [None, None]
        nd = [o for i, o in enumerate(outs) if o is not None and i >= 2]
        ctx.mark_non_differentiable(*nd)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_pan, g_disp, g_ml, g_mr):
        plan = ctx.plan
        if plan.generation != ctx.gen:
            raise RuntimeError("FAL_netB: the plan's activations were overwritten by a later forward of the same "
                               "module and shape before backward ran")
        if g_pan is not None or g_disp is not None:
            plan.run_backward(None if g_disp is None else g_disp.contiguous(), None if g_pan is None else g_pan.contiguous())
        return (None,) * 9


class FAL_net(nn.Module):
    def __init__(self, batchNorm, no_levels, compute_dtype=None, arch="B"):
        super().__init__()
        self.no_levels = no_levels
        self.no_fac = 1
        self.arch = arch
        # the attribute name is part of the checkpoint keys: backbone (FAL_netB.py:184) / BackBone (FAL_netA.py:183) / synth (FAL_netC.py:185)
        setattr(self, ARCHS[arch]["prefix"], BackBone(batchNorm, no_in=3, no_flow=1, no_out=self.no_levels, arch=arch))
        self.softmax = nn.Softmax(dim=1)
        self.elu = nn.ELU(inplace=True)
        self.sigmoid = nn.Sigmoid()
        self.conv0 = nn.Conv2d(self.no_levels, self.no_fac * self.no_levels, 1, 1, 0, bias=True)  # FAL_netB.py:190
        nn.init.kaiming_normal_(self.conv0.weight.data)
        self.conv0.bias.data.zero_()
        self.compute_dtype = compute_dtype or torch.float32
        self._plans = {}
        self._flat = self._flat_grad = None
        self._anchor = None

    @property
    def _bb(self):
        return getattr(self, ARCHS[self.arch]["prefix"])

    def weight_parameters(self):
        return [param for name, param in self.named_parameters() if "weight" in name]

    def bias_parameters(self):
        return [param for name, param in self.named_parameters() if "bias" in name]

    # ---- flat parameter / gradient storage ----
    def _trainable_named(self):
        """Parameters that receive gradients (amask_conv never does: FAL_netB.py:128).  Cached: the module tree is fixed after
        construction (the Parameter objects survive .to() / load_state_dict), and a step asks for this list half a dozen times
        (named_parameters() walks the whole tree: ~1 ms of host time per step)."""
        c = getattr(self, "_trainable_cache", None)
        if c is None:
            c = self._trainable_cache = [(n, p) for n, p in self.named_parameters() if "amask_conv" not in n]
        return c

    def _ensure_flat(self, device):
        named = self._trainable_named()
        ok = (self._flat is not None and self._flat.device == device and
              all(p.data_ptr() == self._flat.data_ptr() + off * 4 for (n, p), off in zip(named, self._offsets)))
        if ok:
            return
        sizes = [(p.numel() + 3) // 4 * 4 for _, p in named]  # 16-B aligned slices
        self._offsets = [sum(sizes[:i]) for i in range(len(sizes))]
        total = sum(sizes)
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        for (n, p), off in zip(named, self._offsets):
            flat[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = flat[off:off + p.numel()].view(p.shape)
        for n, p in self.named_parameters():
            if "amask_conv" in n and p.device != device:
                p.data = p.data.to(device)
        self._flat = flat
        self._flat_grad = torch.zeros(total, dtype=torch.float32, device=device)
        self._gviews = {id(p): self._flat_grad[off:off + p.numel()].view(p.shape) for (n, p), off in zip(named, self._offsets)}
        self._plans = {}
        self._build_packed()

    def _packed_is_fresh(self):
        """True when the packed compute-dtype weight copies already match the f32 masters: FlatAdam re-packs right behind its
        update (`repack_after_optimizer`), and nothing has written the flat buffer in place since (shared version counter of
        the parameter views; writes through `.data` / raw pointers by other code are not tracked -> call `mark_weights_changed`)."""
        return (getattr(self, "_packed_version", None) is not None and self._flat is not None
                and self._packed_version == self._weights_version() and not torch.cuda.is_current_stream_capturing())

    def _weights_version(self):
        # in-place writes bump the version counter of the tensor they go through: the flat buffer's or a parameter's own
        return self._flat._version + sum(p._version for _, p in self._trainable_named())

    def mark_weights_changed(self):
        self._packed_version = None

    def repack_after_optimizer(self):
        """Re-pack the weights NOW (current stream, behind the optimizer's kernel) instead of at the start of the next forward:
        the ~0.1 ms launch then runs while the host is still issuing the next step's prologue."""
        plan = next(iter(self._plans.values()), None)
        if plan is None or torch.cuda.is_current_stream_capturing():
            self._packed_version = None
            return
        for call in plan.pack:
            call()
        self._packed_version = self._weights_version()

    def adam_and_repack(self, grad, m, v, state, b1, b2, eps, grad_scale, scaler_state):
        """torch.optim.Adam's update of the whole flat buffer AND the re-pack of the compute-dtype weight copies: one pass over the masters
        of the packed layers (falnet_adam_pack_batched), a range list for the rest, then the derived weights.  Returns False when no plan
        exists yet (the caller runs the stand-alone update)."""
        plan = next(iter(self._plans.values()), None)
        ap = getattr(plan, "adam_pack", None)
        if ap is None or torch.cuda.is_current_stream_capturing():
            return False
        flat = self._flat
        g_off, m_off, v_off = ((t.data_ptr() - flat.data_ptr()) // 4 for t in (grad, m, v))
        lib, st = L.lib(), L.stream_ptr()
        if ap["rest"] is not None:
            L.check(lib.falnet_adam_ranges(L.ptr(flat), g_off, m_off, v_off, L.ptr(ap["rest"]), ap["n_rest"], L.ptr(state), b1, b2, eps, float(grad_scale),
                                           L.ptr(scaler_state), st), "adam_ranges")
        for call in ap["before"]:
            call()
        if ap["owned"] is not None:
            ap["owned"](g_off, m_off, v_off, state, b1, b2, eps, grad_scale, scaler_state)
        L.check(lib.falnet_adam_tick(L.ptr(state), L.ptr(scaler_state), st), "adam_tick")
        for call in ap["after"]:
            call()
        self._packed_version = self._weights_version()
        return True

    def gradient_buckets(self):
        """Contiguous element ranges of the flat gradient buffer, in the order backward finalises them."""
        named = self._trainable_named()
        pre = ARCHS[self.arch]["prefix"]
        first = {}
        for (n, p), off in zip(named, self._offsets):
            if n.startswith(pre + ".deconv6"):
                first.setdefault("dec", off)
            if n.startswith(pre + ".conv4."):
                first.setdefault("enc4", off)
            if n.startswith(pre + f".conv{_TAIL_LEVELS}."):
                first.setdefault("enc1", off)
        total = self._flat.numel()
        return [(first["dec"], total), (first["enc4"], first["dec"]), (first["enc1"], first["enc4"]), (0, first["enc1"])]

    def ensure_flat(self, device=None):
        """Move the trainable parameters into ONE flat f32 buffer (and create the flat gradient buffer) now instead of at the
        first forward: the start-up parameter broadcast of a multi-GPU run and the gradient-bucket tests need it.  Works on any
        device (no kernels involved)."""
        self._ensure_flat(torch.device(device) if device is not None else next(self.parameters()).device)
        return self._flat

    def _bucket_ready(self, bucket):
        hook = getattr(self, "bucket_hook", None)
        # an ACCUMULATING backward (p.grad kept from the previous micro-batch) must not start the all-reduce: the running sum
        # would be reduced once per micro-batch; train.allreduce_gradients then falls back to ONE sum over the whole buffer
        if hook is not None and not getattr(self, "_accumulating", False):
            lo, hi = self.gradient_buckets()[bucket]
            view = self._flat_grad[lo:hi]
            if L.recording():
                # a recorded backward (plan.run_backward) is CUT here: the collective is issued from Python between two replayed segments, on the
                # stream that was current when the sequence was recorded (the weight-gradient side stream)
                stream = torch.cuda.current_stream()

                def fire(bucket=bucket, view=view, stream=stream):
                    h = getattr(self, "bucket_hook", None)
                    if h is not None:
                        with torch.cuda.stream(stream):
                            h(bucket, view)
                L.cut(fire)
            else:
                hook(bucket, view)

    def flat_parameters(self):
        return self._flat

    def flat_gradients(self):
        return self._flat_grad

    def _grad_view(self, p):
        return self._gviews[id(p)]

    def _begin_grad_accumulation(self):
        """Decide whether this backward accumulates into existing grads (p.grad already aliases the flat
        buffer and was not reset) or starts fresh.  Mixed states are normalised to 'fresh + add'."""
        named = self._trainable_named()
        aliased = [p.grad is not None and p.grad.data_ptr() == self._gviews[id(p)].data_ptr() for _, p in named]
        self._foreign = None
        if all(aliased):
            return 1
        if any(p.grad is not None for _, p in named) and not all(aliased):
            self._foreign = {id(p): p.grad for _, p in named if p.grad is not None}
        return 0

    def _end_grad_accumulation(self):
        for _, p in self._trainable_named():
            v = self._gviews[id(p)]
            if self._foreign and id(p) in self._foreign and self._foreign[id(p)].data_ptr() != v.data_ptr():
                v.add_(self._foreign[id(p)])
            p.grad = v

    def _build_packed(self):
        bb = self._bb
        enc, dec = ARCHS[self.arch]["enc"], ARCHS[self.arch]["dec"]
        P = {}

        def add(key, conv, groups, stride=1):
            P[key] = PackedConv(key, conv.weight, conv.bias, groups, stride)
        add("conv0", bb.conv0[0], [3])
        add("conv1", bb.conv1[0], [enc[0], 1], 2)
        for i in range(2, 7):
            add(f"conv{i}", getattr(bb, f"conv{i}")[0], [enc[i - 1]], 2)
        for i, ch in enumerate(enc):
            rb = getattr(bb, f"conv{i}_1")
            add(f"conv{i}_1.conv1", rb.conv1, [ch])
            add(f"conv{i}_1.conv2", rb.conv2, [ch])
        below = enc[6]
        for lvl in range(6, 1, -1):
            dch, ich = dec[lvl]
            add(f"deconv{lvl}", getattr(bb, f"deconv{lvl}").conv1, [below])
            add(f"iconv{lvl}", getattr(bb, f"iconv{lvl}")[0], [dch, enc[lvl - 1]])
            below = ich
        add("deconv1", bb.deconv1.conv1, [below])
        add("iconv1", bb.iconv1, [64, enc[0]])
        add("conv0_1x1", self.conv0, [self.no_levels])
        # iconv1 (3x3, no bias, no activation; FAL_netB.py:127,174) followed by the 1x1 conv0 (FAL_netB.py:190,215) is ONE linear
        # map: the plans run a single 3x3 convolution with the composed weights Wc = W1x1 . W3x3 straight into the planar f32
        # logits (no NHWC intermediate, no 1x1 launches in forward / dgrad / wgrad); the weight gradients are split back by
        # two small matrix products (dW3x3 = W1x1^T dWc, dW1x1 = dWc W3x3^T).  FALNET_COMPOSE_LOGITS=0 keeps the two launches.
        self._compose_logits = L.ab("FALNET_COMPOSE_LOGITS", "1") == "1"
        if self._compose_logits:
            w3, w1 = bb.iconv1.weight, self.conv0.weight
            dev = w3.device
            self._wc = torch.zeros(w1.shape[0], w3.shape[1], 3, 3, dtype=torch.float32, device=dev)
            self._gwc = torch.zeros_like(self._wc)
            self._gviews[id(self._wc)] = self._gwc
            P["logits"] = PackedConv("logits", self._wc, self.conv0.bias, [64, enc[0]], 1)
        self._packed = P

    def _plan(self, B, H, W, device):
        self._ensure_flat(device)
        key = (B, H, W, self.compute_dtype)
        if key not in self._plans:
            self._plans[key] = FalnetPlan(self, B, H, W, self.compute_dtype, device)
        return self._plans[key]

    def forward(self, input_left, min_disp, max_disp, ret_disp=True, ret_subocc=False, ret_pan=False):
        """FAL_net.forward (FAL_netB.py:200-297).  Returns bare `disp` when only ret_disp, else the list
        [p_im0][disp][maskL, maskR] in the reference's order."""
        if not input_left.is_cuda:
            raise RuntimeError("fal_net_amd.FAL_netB runs on an MI355X only (no CPU fallback); input is on " + str(input_left.device))
        B, C, H, W = input_left.shape
        plan = self._plan(B, H, W, input_left.device)
        left = input_left.detach().to(torch.float32).contiguous()
        if self._anchor is None or self._anchor.device != input_left.device:
            self._anchor = torch.zeros(1, device=input_left.device, requires_grad=True)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for _, p in self._trainable_named())
        anchor = self._anchor if need_grad else self._anchor.detach()
        pan, disp, maskL, maskR = _FalnetFunction.apply(self, plan, left, min_disp.detach().float(), max_disp.detach().float(),
                                                        ret_disp, ret_subocc, ret_pan, anchor)
        if ret_disp and not ret_subocc and not ret_pan:
            return disp
        output = []
        if ret_pan:
            output.append(pan)
        if ret_disp:
            output.append(disp)
        if ret_subocc:
            output.append(maskL)
            output.append(maskR)
        return output
