"""Step bodies of the reference's training scripts on the HIP-backed module, plus the two pieces the
reference delegates to torch: the optimiser (torch.optim.Adam, Train_Stage1_K.py:180) as ONE fused launch
over the model's flat parameter buffer, and data parallelism (nn.DataParallel, Train_Stage1_K.py:172) as
one process per GPU with ONE RCCL all-reduce of the flat gradient buffer per step.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from . import _lib as L
from . import ops
from .loss_functions import mirror_loss_fnc, occlusion_mask, rec_loss_fnc, smoothness, vgg


import os as _os
_FORCE_DIST = _os.environ.get("FALNET_FORCE_DIST") == "1"  # exercise the collective path with world_size 1 (tests)


class FlatAdam:
    """Adam(betas, eps, weight_decay=0) over `model.flat_parameters()` (torch.optim.Adam semantics).

    Both of the reference's param groups (biases / weights, Train_Stage1_K.py:177-178) use weight_decay 0,
    so one flat update is identical to the two-group optimiser.  amask_conv never receives gradients and
    is skipped exactly like torch skips `grad is None` parameters."""

    def __init__(self, model, lr=1e-4, betas=(0.5, 0.999), eps=1e-8):
        self.model, self.betas, self.eps = model, betas, eps
        self.param_groups = [{"lr": lr}]
        self.t = 0
        self.m = self.v = None

    def zero_grad(self, set_to_none=True):
        for p in self.model.parameters():
            p.grad = None

    def step(self, grad_scale=1.0, scaler=None):
        """`scaler`: the model's LossScaler (f16 path) -- the update then divides the gradients by the device-resident scale, is
        skipped as a whole when the all-reduced gradient holds an inf / NaN, and the scale backs off / grows (GradScaler semantics)."""
        flat, grad = self.model.flat_parameters(), self.model.flat_gradients()
        if flat is None:
            raise RuntimeError("FlatAdam.step before any forward/backward of the model")
        if self.m is None or self.m.numel() != flat.numel() or self.m.device != flat.device:
            self.m, self.v = torch.zeros_like(flat), torch.zeros_like(flat)
            # hyper-state {lr, t} lives on the device so that a captured hipGraph advances the step count
            self.state = torch.tensor([float(self.param_groups[0]["lr"]), float(self.t)], device=flat.device)
            self._lr_dev = float(self.param_groups[0]["lr"])
        if self._lr_dev != float(self.param_groups[0]["lr"]):
            self._lr_dev = float(self.param_groups[0]["lr"])
            self.state[0] = self._lr_dev
        self.t += 1
        b1, b2 = self.betas
        if _ADAM_PACK and hasattr(self.model, "adam_and_repack"):
            # the update of the packed layers rides in the launch that re-packs them (one pass over the f32 masters instead of two)
            if scaler is not None:
                L.check(L.lib().falnet_grad_guard(L.ptr(grad), grad.numel(), L.ptr(scaler.state), L.stream_ptr()), "grad_guard")
            if self.model.adam_and_repack(grad, self.m, self.v, self.state, b1, b2, self.eps, grad_scale, None if scaler is None else scaler.state):
                if scaler is not None:
                    scaler.update()
                return
            if scaler is not None:  # (no plan yet: fall through to the stand-alone update; the guard has run)
                L.check(L.lib().falnet_adam_step_guarded(L.ptr(flat), L.ptr(grad), L.ptr(self.m), L.ptr(self.v), flat.numel(), L.ptr(self.state),
                                                         b1, b2, self.eps, float(grad_scale), L.ptr(scaler.state), L.stream_ptr()), "adam_step_guarded")
                scaler.update()
                self.model.mark_weights_changed()
                return
        if scaler is None:
            L.check(L.lib().falnet_adam_step_dev(L.ptr(flat), L.ptr(grad), L.ptr(self.m), L.ptr(self.v), flat.numel(), L.ptr(self.state),
                                                 b1, b2, self.eps, float(grad_scale), L.stream_ptr()), "adam_step_dev")
        else:
            st = L.stream_ptr()
            L.check(L.lib().falnet_grad_guard(L.ptr(grad), grad.numel(), L.ptr(scaler.state), st), "grad_guard")
            L.check(L.lib().falnet_adam_step_guarded(L.ptr(flat), L.ptr(grad), L.ptr(self.m), L.ptr(self.v), flat.numel(), L.ptr(self.state),
                                                     b1, b2, self.eps, float(grad_scale), L.ptr(scaler.state), st), "adam_step_guarded")
            scaler.update()
        if L.ab("FALNET_PACK_AFTER_ADAM", "1") == "1":
            self.model.repack_after_optimizer()  # the raw-pointer update is invisible to autograd's version counters
        else:
            self.model.mark_weights_changed()


# f16 compute path: activation gradients are stored in IEEE half (normal range >= 6.1e-5) while a mean loss over B*3*H*W
# elements seeds them at ~1e-7 -- they would flush to subnormals / zero.  Loss scaling: backward runs on S * loss (every node of
# the step is linear in its upstream gradient), the f32 weight gradients come out S times too large and 1/S is folded into
# the fused Adam launch beside 1/world.  bf16 / f32 have the f32 exponent range: S = 1, no scaler.
# S is DYNAMIC and lives on the device (GradScaler semantics without a host sync): a gradient that overflows f16 (|x| > 65504)
# somewhere in backward puts inf / NaN into the flat f32 gradient; falnet_grad_guard (after the all-reduce, so every rank
# decides alike) raises a flag, the guarded Adam skips the whole update, and the scale halves; it doubles again after
# FALNET_F16_GROWTH_INTERVAL clean steps.  A run whose scale has collapsed to 1 and still overflows is broken: LossScaler.check()
# (called at --print-freq by the training scripts) raises.
_F16_LOSS_SCALE = float(_os.environ.get("FALNET_F16_LOSS_SCALE", "8192"))
_F16_GROWTH_INTERVAL = int(_os.environ.get("FALNET_F16_GROWTH_INTERVAL", "2000"))


class LossScaler:
    """Device-resident dynamic loss scale: state = [scale, clean steps, overflow flag, skipped steps] (f32[4])."""

    def __init__(self, device, init=None, growth=2.0, backoff=0.5, interval=None, min_scale=1.0, max_scale=65536.0):
        self.state = torch.tensor([float(init or _F16_LOSS_SCALE), 0.0, 0.0, 0.0], device=device)
        self.growth, self.backoff, self.interval = growth, backoff, int(interval or _F16_GROWTH_INTERVAL)
        self.min_scale, self.max_scale = min_scale, max_scale
        self._seeds = {}

    def seeds(self, *coef):
        """Device tensor seeds[i] = scale * coef[i] for THIS step (one tiny launch): the `gscale` operands of the loss kernels."""
        key = tuple(float(c) for c in coef)
        if key not in self._seeds:
            self._seeds[key] = (torch.tensor(key, device=self.state.device), torch.empty(len(key), device=self.state.device))
        c, out = self._seeds[key]
        L.check(L.lib().falnet_loss_seeds(L.ptr(self.state), L.ptr(c), L.ptr(out), len(key), L.stream_ptr()), "loss_seeds")
        return out

    def scale_tensor(self):
        return self.state[0]

    def state_dict(self):
        """For checkpoints: a resumed f16 run continues at the scale it had reached instead of re-discovering it from the initial 8192."""
        return {"state": [float(x) for x in self.state.tolist()]}

    def load_state_dict(self, sd):
        self.state.copy_(torch.tensor([float(x) for x in sd["state"]], device=self.state.device))

    def update(self):
        L.check(L.lib().falnet_loss_scale_update(L.ptr(self.state), self.growth, self.backoff, self.interval, self.min_scale, self.max_scale,
                                                 L.stream_ptr()), "loss_scale_update")

    def check(self):
        """Host-side health check (synchronises: call it at logging frequency, not per step).  Returns (scale, skipped steps);
        raises when the scale has backed off to its floor and the last step still overflowed -- the run is diverging."""
        scale, clean, flag, skipped = (float(x) for x in self.state.tolist())
        if clean < 0:  # (set by falnet_loss_scale_update when a step overflowed at a scale that was already at its floor)
            raise FloatingPointError(f"f16 step: gradients are non-finite even at loss scale {scale} ({int(skipped)} steps skipped): "
                                     "the run diverged (use --dtype bf16 / f32 or a lower learning rate)")
        return scale, int(skipped)


def loss_scaler(model):
    """The model's LossScaler (created on first use) on the f16 compute path, None otherwise."""
    if getattr(model, "compute_dtype", None) != torch.float16:
        return None
    sc = getattr(model, "_loss_scaler", None)
    dev = next(model.parameters()).device
    if sc is None or sc.state.device != dev:
        sc = model._loss_scaler = LossScaler(dev)
    return sc


def scaled_backward(loss, model):
    """loss.backward() under the model's loss scale; returns the LossScaler the optimiser must be given (None: no scaling)."""
    sc = loss_scaler(model)
    if sc is None:
        loss.backward()
        return None
    (loss * sc.scale_tensor()).backward()
    return sc


def sync_parameters(model, check=True):
    """Start-up rank consistency of a data-parallel run: ONE broadcast of rank 0's flat parameter buffer (the reference's
    nn.DataParallel re-broadcasts the parameters every step, Train_Stage1_K.py:172; here once, outside every timed region --
    the per-step collective stays the single gradient all-reduce), then an all-reduced checksum that every rank holds the same
    bytes.  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _FORCE_DIST):
        return False
    flat = model.ensure_flat()
    dist.broadcast(flat, 0)
    for n, p in model.named_parameters():  # amask_conv never trains (FAL_netB.py:128) and lives outside the flat buffer
        if "amask_conv" in n:
            dist.broadcast(p.data, 0)
    model.mark_weights_changed()
    if check:
        d = flat.double()
        mine = torch.stack([d.sum(), (d * d).sum()])
        lo, hi = mine.clone(), mine.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise RuntimeError("sync_parameters: ranks hold different parameters after the broadcast")
    return True


def enable_overlapped_allreduce(model):
    """Install the bucket hook: as backward finalises each contiguous range of the flat gradient buffer (decoder first,
    deep encoder levels next, the small shallow levels last) its share of the step's all-reduce is launched
    asynchronously (RCCL stream) while backward continues.  Logically still ONE sum over the flat buffer per step."""
    if getattr(model, "bucket_hook", None) is not None or getattr(model, "_no_overlap", False):
        return
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _FORCE_DIST):
        return
    model._pending_reduces = []
    model._comm_event = None
    on_device = model.flat_gradients() is not None and model.flat_gradients().is_cuda
    sync_on_stream = on_device and dist.get_backend() == "nccl" and _COMM_ON_AUX

    def hook(bucket, flat_slice):
        if _SKIP_ALLREDUCE:
            return
        if sync_on_stream:
            # The collective runs ON one of the step's own streams: torch >= 2.8 issues a SYNC collective (async_op=False) of the NCCL / RCCL
            # backend on the caller's current stream, and the auxiliary stream (the label image's VGG pass in forward) is idle throughout
            # backward.  No fifth busy hardware queue: the compute pipes of the chip run four queues side by side, a fifth time-slices with one
            # of them (profiles/r05_ab_dist_queues.txt: the more queues the slower; r06_ab_dist.txt: world-1 exposure 0.31 -> see there).
            # Order: slab reduce of the bucket (caller's stream) -> event -> aux: all_reduce -> event -> the optimiser's stream (allreduce_gradients).
            from .plan import aux_stream
            aux = aux_stream(flat_slice.device)
            cur = torch.cuda.current_stream(flat_slice.device)
            ev = torch.cuda.Event()
            ev.record(cur)
            aux.wait_event(ev)
            with torch.cuda.stream(aux):
                dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=False)
                done = torch.cuda.Event()
                done.record(aux)
            model._comm_event = done  # (stream order on aux: the last bucket's event covers the earlier ones)
            return
        model._pending_reduces.append(dist.all_reduce(flat_slice, op=dist.ReduceOp.SUM, async_op=True))
    model.bucket_hook = hook
    # The hooked form of the stream self-test issues collectives: run it HERE, where every rank stands at the same point of its program (the
    # first step of a data-parallel run), not lazily inside some plan's first backward.  Gloo groups on CPU tensors have no
    # device stream to test.
    p0 = model.flat_gradients()
    if p0 is not None and p0.is_cuda and L.ab("FALNET_STREAM_SELFTEST", "1") == "1":
        from .plan import StepStreams
        st = StepStreams.for_device(p0.device)
        if not st.hooked_tested:
            with torch.cuda.device(p0.device):
                st.selftest(torch.cuda.current_stream(p0.device), hooked=True)


_COMM_ON_AUX = L.ab("FALNET_COMM_ON_AUX", "1") == "1"  # bucket all-reduces as sync collectives on the auxiliary stream (0: async on the backend's own stream)
_SKIP_ALLREDUCE = False  # bench.py only: time the same steps without the collective (exposed communication = the difference)


def allreduce_gradients(model):
    """The step's single collective: sum of the flat f32 gradient buffer over ranks (RCCL over xGMI), pipelined in
    buckets behind backward when the hook is installed.  Returns the scale (1/world) to fold into the optimiser."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not _FORCE_DIST):
        return 1.0
    pending = getattr(model, "_pending_reduces", None)
    if _SKIP_ALLREDUCE:
        if pending:
            pending.clear()
        return 1.0 / dist.get_world_size()
    done = getattr(model, "_comm_event", None)
    if done is not None:  # buckets reduced on the auxiliary stream: the optimiser's stream waits for the last of them
        torch.cuda.current_stream().wait_event(done)
        model._comm_event = None
        return 1.0 / dist.get_world_size()
    if pending:
        for w in pending:
            w.wait()  # current stream waits for the collective
        pending.clear()
    else:
        dist.all_reduce(model.flat_gradients(), op=dist.ReduceOp.SUM)
    return 1.0 / dist.get_world_size()


def _aux_stream(device):
    from .plan import aux_stream  # (one per device: plan.StepStreams, whose self-test checks it too)
    return aux_stream(device)


def vgg_label_async(label, borrow=True):
    """VGG features of a label image on an auxiliary HIP stream: independent of the network forward, so it runs
    concurrently with it and fills the CUs the small backbone layers leave idle.  Returns join() -> features.
    `borrow`: the features alias the label plan's buffers (valid until the next label pass of that shape)."""
    if ops.TIMER is not None:  # instrumented pass (bench.py roofline): serial launches so per-kernel times are uncontended
        feats_serial = vgg(label, borrow=borrow)
        return lambda: feats_serial
    main = torch.cuda.current_stream()
    aux = _aux_stream(label.device)
    aux.wait_stream(main)
    with torch.cuda.stream(aux), L.on_stream(aux):  # (torch stream for the allocator / record_stream, pinned pointer for the launches)
        feats = vgg(label, borrow=borrow)  # consumed by this step's perceptual loss only

    def join():
        main.wait_stream(aux)
        if not torch.cuda.is_current_stream_capturing():
            for t in feats:
                t.record_stream(main)
        return feats
    return join


_FUSED_STEP = _os.environ.get("FALNET_FUSED_STEP", "1") == "1"
_ADAM_PACK = L.ab("FALNET_ADAM_PACK", "1") == "1"  # FlatAdam.step: optimiser update fused with the weight re-pack (falnet_adam_pack_batched)
_SEEDS = {}


def _seed(device, value):
    """Cached 1-element device tensors (upstream-gradient scalars of the loss kernels)."""
    key = (device, float(value))
    if key not in _SEEDS:
        _SEEDS[key] = torch.tensor([float(value)], device=device)
    return _SEEDS[key]


def _stage1_fused(model, opt, left, right, max_disp, a_p, a_sm, min_disp_arg, max_disp_arg, optimize):
    """The same iteration as `stage1_step`'s autograd form, as a static launch sequence: the plan's forward, the loss kernels
    accumulating into one device scalar pair, their adjoints seeded by cached device scalars and writing straight into the
    plans' gradient buffers, the plan's backward.  No autograd graph, no scalar aten launches (the autograd form spends ~30
    launches of 4-5 us on `l1 + a_p * perc`, `rec + a_sm * sm`, output clones, gradient sums and seed fills).  `rpan` /
    `ldisp` in the result alias the plan's output buffers: valid until the model's next forward of this shape."""
    with L.stream_scope():  # one stream lookup for the ~330 launches of the step
        try:
            return _stage1_fused_body(model, opt, left, right, max_disp, a_p, a_sm, min_disp_arg, max_disp_arg, optimize)
        except BaseException:
            # the loss accumulators are zero on entry by contract (falnet_step_scalars re-zeroes them at the END of a step): a step that
            # fails in between must not leave its partial sums for the next one
            for plan in getattr(model, "_plans", {}).values():
                S = plan.buf.get("step_scalars")
                if S is not None:
                    S.zero_()
            raise


PHASE_MARKS = None  # bench.py: a list -> every fused Stage-1 step appends five HIP events recorded on the step's stream at its phase boundaries
# (start | forward + MED head done | label join, VGG(synth), losses, VGG adjoint done | backward joined | optimiser done)


def _mark(marks):
    if marks is not None:
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)


def _stage1_fused_body(model, opt, left, right, max_disp, a_p, a_sm, min_disp_arg, max_disp_arg, optimize):
    from . import loss_functions as LF
    lib = L.lib()
    marks = None if PHASE_MARKS is None else []
    _mark(marks)
    B, C, H, W = left.shape
    dev = left.device
    plan = model._plan(B, H, W, dev)
    plan._ensure_backward()
    b = plan.buf
    scaler = loss_scaler(model)
    if scaler is None:
        seed_l1, seed_p, seed_sm = _seed(dev, 1.0), _seed(dev, a_p), _seed(dev, a_sm)
    else:  # f16: the upstream scalars carry the device-resident loss scale
        sd3 = scaler.seeds(1.0, a_p, a_sm)
        seed_l1, seed_p, seed_sm = sd3[0:], sd3[1:], sd3[2:]
    joins = []
    if a_p > 0:
        if L.ab("FALNET_LABEL_VGG_MID", "1") == "1" and ops.TIMER is None:
            model._mid_forward_hook = lambda: joins.append(vgg_label_async(right))
        else:
            joins.append(vgg_label_async(right))
    lf = left.detach()
    lf = lf if lf.dtype == torch.float32 else lf.float()
    plan.run_forward(lf.contiguous(), None, max_disp.detach().float().contiguous(), True, False, True, min_from=(min_disp_arg, max_disp_arg))
    if a_p > 0 and not joins:
        model._mid_forward_hook = None
        joins.append(vgg_label_async(right))
    _mark(marks)
    rpan, ldisp = b["p_im0"], b["disp"]
    rt = right.detach().contiguous()
    st = L.stream_ptr()
    S = plan.buf.get("step_scalars")  # [rec = L1 + a_p * perceptual, sm]: zero on entry (falnet_step_scalars re-zeroes it at the end of every step)
    if S is None:
        S = plan.buf["step_scalars"] = torch.zeros(2, device=dev)
    n_img = B * C * H * W
    # every loss term AND its adjoint in one pass over its operands (the upstream scalars -- loss scale, a_p, a_sm -- are known now)
    g_pan, g_disp = b["g_pan"], b["g_disp"]
    if a_p <= 0:  # no perceptual term: the L1 gradient is the whole gradient of the synthesised view
        L.check(lib.falnet_l1_fwd_bwd(L.ptr(rpan), L.ptr(rt), B, C, H * W, 1.0 / n_img, L.ptr(S), L.ptr(seed_l1), L.ptr(g_pan), st),
                "l1_fwd_bwd")  # loss_functions.py:53
    vplan = None
    if a_p > 0:
        vm = LF.vgg._get()
        vdt = vm.compute_dtype or LF._default_dtype()
        vm._prepare(dev, vdt)
        vplan = vm._plan(B, H, W, vdt, dev, hold=True)
        vplan.c3_call.set_input(rpan)
        vplan.run_fwd()
        labels = joins[0]()
        code = L.dtype_code(vdt)
        feats = []
        for o, lab in zip(vplan.outs, labels):  # loss_functions.py:61-65, a_p folded into the scale
            ln = LF._nhwc(lab.to(o.dtype))
            Bo, Ho, Wo, Co = o.shape
            sc = 1.0 / o.numel()
            feats.append((o, ln, Bo * Ho * Wo, Co, sc))
        if len(feats) == 3 and all(o.numel() % 8 == 0 and (o.data_ptr() | ln.data_ptr() | go.data_ptr()) % 32 == 0
                                   for (o, ln, _, _, _), go in zip(feats, vplan.gouts)):
            # the three slices' value + gradient in ONE launch (the small slices are latency-bound as launches of their own)
            import ctypes as _C
            P3, L3, F3 = _C.c_void_p * 3, _C.c_int64 * 3, _C.c_float * 3
            L.check(lib.falnet_mse3_fwd_bwd(P3(*[f[0].data_ptr() for f in feats]), P3(*[f[1].data_ptr() for f in feats]), L3(*[f[0].numel() for f in feats]),
                                            F3(*[a_p * f[4] for f in feats]), L.ptr(S), F3(*[f[4] for f in feats]), L.ptr(seed_p),
                                            P3(*[g.data_ptr() for g in vplan.gouts]), code, st), "mse3_fwd_bwd")
        else:
            for (o, ln, npix, Co, sc), go in zip(feats, vplan.gouts):
                L.check(lib.falnet_mse_fwd_bwd(L.ptr(o), L.ptr(ln), npix, Co, a_p * sc, L.ptr(S), sc, L.ptr(seed_p), L.ptr(go), code, st),
                        "mse_fwd_bwd")
    x0 = int(0.20 * W)
    sc_sm = 1.0 / (B * H * (W - x0))
    if a_sm > 0:  # Train_Stage1_K.py:255
        L.check(lib.falnet_smooth_fwd_bwd(L.ptr(b["left"]), L.ptr(ldisp), B, H, W, x0, W, 2.0, sc_sm, L.ptr(S[1:]), L.ptr(seed_sm),
                                          L.ptr(g_disp), st), "smooth_fwd_bwd")
    # ---- the VGG adjoint (every node is linear in its upstream scalar: s = loss scale) joins the L1 gradient ----
    if vplan is not None:
        vplan.run_bwd()
        # L1 term (loss_functions.py:53) and its gradient, with the VGG gradient of the synthesised view added in the same pass
        L.check(lib.falnet_l1_fwd_bwd_add(L.ptr(rpan), L.ptr(rt), B, C, H * W, 1.0 / n_img, L.ptr(S), L.ptr(seed_l1), L.ptr(vplan.g_in), L.ptr(g_pan), st),
                "l1_fwd_bwd_add")
        vplan.busy = False
    _mark(marks)
    plan.run_backward(g_disp if a_sm > 0 else None, g_pan, in_place=True)
    _mark(marks)
    Sc = torch.empty(3, device=dev)  # a fresh triple per step (callers keep loss tensors across steps); no launch: caching allocator
    L.check(lib.falnet_step_scalars(L.ptr(S), float(a_sm), L.ptr(Sc), st), "step_scalars")  # {rec + a_sm sm, rec, sm}; S -> 0
    out = {"loss": Sc[0], "rec": Sc[1], "sm": Sc[2] if a_sm > 0 else 0, "rpan": rpan, "ldisp": ldisp, "scaler": scaler}
    if optimize:
        opt.step(allreduce_gradients(model), scaler=scaler)
    _mark(marks)
    if marks is not None:
        PHASE_MARKS.append(marks)
    return out


def stage1_step(model, opt, left, right, max_disp, a_p=0.01, a_sm=0.2 * 2 / 512, min_disp_arg=2.0, max_disp_arg=300.0,
                optimize=True):
    """One iteration of Train_Stage1_K.py:233-262 (forward, VGG, losses, backward, all-reduce, Adam).
    Returns device scalars (no host sync)."""
    opt.zero_grad()
    enable_overlapped_allreduce(model)
    if (_FUSED_STEP and left.is_cuda and torch.is_grad_enabled() and hasattr(model, "_plan")
            and all(p.requires_grad for _, p in model._trainable_named())):
        return _stage1_fused(model, opt, left, right, max_disp, a_p, a_sm, min_disp_arg, max_disp_arg, optimize)
    W = left.shape[3]
    min_disp = max_disp * min_disp_arg / max_disp_arg  # :237
    # :241-244 label features, overlapped with the model forward: started from the plan's mid-forward hook, i.e. when the
    # main stream reaches the small deep layers (beside the chip-filling shallow ones the overlap would only time-slice)
    joins = []
    if a_p > 0:
        if L.ab("FALNET_LABEL_VGG_MID", "1") == "1" and ops.TIMER is None:
            model._mid_forward_hook = lambda: joins.append(vgg_label_async(right))
        else:
            joins.append(vgg_label_async(right))
    rpan, ldisp = model(left, min_disp, max_disp, ret_disp=True, ret_pan=True, ret_subocc=False)  # :238
    if a_p > 0 and not joins:  # the plan did not reach the hook (no plan replay on this path): start it now
        model._mid_forward_hook = None
        joins.append(vgg_label_async(right))
    vgg_right = joins[0]() if joins else None
    rec_loss = rec_loss_fnc(1, rpan, right, vgg_right, a_p)  # :248
    sm_loss = 0
    if a_sm > 0:
        c = int(0.20 * W)
        sm_loss = smoothness(left[:, :, :, c:], ldisp[:, :, :, c:], gamma=2)  # :255
    loss = rec_loss + a_sm * sm_loss  # :258
    sc = scaled_backward(loss, model)
    out = {"loss": loss.detach(), "rec": rec_loss.detach(), "sm": sm_loss.detach() if torch.is_tensor(sm_loss) else sm_loss,
           "rpan": rpan, "ldisp": ldisp, "scaler": sc}
    if optimize:
        opt.step(allreduce_gradients(model), scaler=sc)
    return out


def stage1_slow_step(model, opt, left, right, max_disp, a_p=0.01, a_sm=0.2 * 2 / 512, min_disp_arg=2.0,
                     max_disp_arg=300.0):
    """One iteration of Train_Stage1_Kslow.py:236-284: both views in one 2B batch (left | flip(right)), the second
    half of the outputs flipped back, reconstruction and smoothness averaged over the two views."""
    opt.zero_grad()
    enable_overlapped_allreduce(model)
    B, C, H, W = left.shape
    min_disp = max_disp * min_disp_arg / max_disp_arg  # :244
    mn2, mx2 = torch.cat((min_disp, min_disp), 0), torch.cat((max_disp, max_disp), 0)
    joins = [vgg_label_async(right, borrow=False), vgg_label_async(left)] if a_p > 0 else []  # :259-264
    pan, disp = model(torch.cat((left, hflip(right)), 0), mn2, mx2, ret_disp=True, ret_pan=True, ret_subocc=False)  # :245-248
    rpan, lpan = pan[0:B], flip(pan[B:])  # :249-256
    ldisp, rdisp = disp[0:B], flip(disp[B:])
    vgg_right, vgg_left = (joins[0](), joins[1]()) if joins else (None, None)
    rec_loss = (rec_loss_fnc(1, rpan, right, vgg_right, a_p) + rec_loss_fnc(1, lpan, left, vgg_left, a_p)) / 2  # :268-269
    sm_loss = 0
    if a_sm > 0:  # :274-278
        c2, c8 = int(0.20 * W), int(0.80 * W)
        sm_loss = (smoothness(left[:, :, :, c2:], ldisp[:, :, :, c2:], gamma=2) +
                   smoothness(right[:, :, :, 0:c8], rdisp[:, :, :, 0:c8], gamma=2)) / 2
    loss = rec_loss + a_sm * sm_loss  # :281
    sc = scaled_backward(loss, model)
    opt.step(allreduce_gradients(model), scaler=sc)
    return {"loss": loss.detach(), "rec": rec_loss.detach(), "sm": sm_loss.detach() if torch.is_tensor(sm_loss) else sm_loss,
            "rpan": rpan, "lpan": lpan, "ldisp": ldisp, "rdisp": rdisp, "scaler": sc}


class GraphedStage1Step:
    """The Stage-1 step captured ONCE into a hipGraph and replayed: the ~300 launches of a step become one graph
    launch (no per-launch host cost, dependent-kernel gaps shrink to the hardware boundary).  Possible because the
    plan is static (fixed buffers, fixed descriptors) and Adam's step count lives on the device.  The inputs are the
    tensors given here (copy new batches into them).  With world_size > 1 the graph ends after backward; the single
    RCCL all-reduce and the Adam launch stay eager (collectives are not captured)."""

    def __init__(self, model, opt, left, right, max_disp, warmup=3, **kw):
        self.model, self.opt, self.kw = model, opt, kw
        self.left, self.right, self.max_disp = left, right, max_disp
        self.multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        model._no_overlap, model.bucket_hook = True, None  # collectives are not captured: one eager all-reduce after the graph
        s = torch.cuda.Stream(device=left.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):  # warm-up off the default stream: builds plans, autotunes, allocates optimiser state
            for _ in range(warmup):
                stage1_step(model, opt, left, right, max_disp, **kw)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = stage1_step(model, opt, left, right, max_disp, optimize=not self.multi, **kw)

    def __call__(self):
        self.graph.replay()
        if self.multi:
            self.opt.step(allreduce_gradients(self.model), scaler=self.out["scaler"])
        else:
            self.opt.t += 1  # host mirror of the device step count
        return self.out


def validate(model, val_loader, max_disp=300.0, min_disp=2.0, rel_baset=1.0, sparse=True, print_freq=100, log=print):
    """Train_Stage1_K.py:279-347 / Train_Stage2_K.py validate(): full-size forward (disp + synthesised right view + masks), RMSE of
    the synthesised view, EPE and the KITTI depth errors against the ground-truth disparity.  `val_loader` yields lists of
    (left_u8, right_u8, disp) from fal_net_amd.datasets.StereoValDataset (batch size 1 in the reference, :152).
    Returns {'rmse', 'epe', 'kitti': {name: value}} (the reference returns the RMSE, :345)."""
    import numpy as np
    from . import datasets as DS
    from . import myUtils as utils
    from .loss_functions import realEPE
    dev = next(model.parameters()).device
    was_training = model.training
    model.eval()
    rmses, epes, kitti = utils.AverageMeter(), utils.AverageMeter(), utils.multiAverageMeter(utils.kitti_error_names)
    with torch.no_grad():
        for i, batch in enumerate(val_loader):
            for left_u8, right_u8, disp_gt in batch:
                left, right = DS.to_model_input(left_u8, dev), DS.to_model_input(right_u8, dev)
                mx = torch.full((1, 1, 1), float(max_disp) * rel_baset, device=dev)
                mn = mx * min_disp / max_disp
                p_im, disp, maskL, maskRL = model(left, mn, mx, ret_disp=True, ret_pan=True, ret_subocc=True)
                rmses.update(float(utils.get_rmse(p_im, right)))
                if disp_gt is not None:
                    target = disp_gt.to(dev).view(1, 1, *disp_gt.shape)
                    epes.update(float(realEPE(disp, target, sparse=sparse)), 1)
                    gt_depth, pred_depth = utils.disps_to_depths_kitti2015(target.squeeze(1).cpu().numpy(), disp.squeeze(1).cpu().numpy())
                    kitti.update(utils.compute_kitti_errors(gt_depth[0], pred_depth[0]), 1)
            if log is not None and i % print_freq == 0:
                log('Test: [{0}/{1}]\t RMSE {2:.3f}'.format(i, len(val_loader), rmses.avg))
    model.train(was_training)
    return {"rmse": rmses.avg, "epe": epes.avg, "kitti": dict(zip(utils.kitti_error_names, [float(a) for a in kitti.avg]))}


def hflip(x):
    """Flip via index reversal; the reference's affine_grid + grid_sample flip equals it to <=8.4e-7
    (Train_Stage2_K.py:248-253)."""
    x = x.contiguous()
    out = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    L.check(L.lib().falnet_hflip(L.ptr(x), L.ptr(out), rows, x.shape[-1], L.stream_ptr()), "hflip")
    return out


class _Flip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return hflip(x)

    @staticmethod
    def backward(ctx, g):
        return hflip(g)


def flip(x):
    return _Flip.apply(x) if x.requires_grad else hflip(x)


def stage2_step(model, fix_model, opt, left, right, max_disp, a_p=0.01, a_sm=0.4 * 2 / 512, a_mr=1.0,
                min_disp_arg=2.0, max_disp_arg=300.0):
    """One iteration of Train_Stage2_K.py:233-331: frozen teacher on (flip(left) | right), student on
    (left | flip(right)) with occlusion masks, masked reconstruction, smoothness and mirror losses."""
    opt.zero_grad()
    B, C, H, W = left.shape
    min_disp = max_disp * min_disp_arg / max_disp_arg
    mn2, mx2 = torch.cat((min_disp, min_disp), 0), torch.cat((max_disp, max_disp), 0)
    if a_mr > 0:
        with torch.no_grad():  # :256-264
            tdisp = fix_model(torch.cat((hflip(left), right), 0), mn2, mx2, ret_disp=True, ret_pan=False, ret_subocc=False)
            mldisp, mrdisp = hflip(tdisp[0:B]), tdisp[B:].contiguous()
    pan, disp, mask0, mask1 = model(torch.cat((left, hflip(right)), 0), mn2, mx2, ret_disp=True, ret_pan=True, ret_subocc=True)
    rpan, lpan = pan[0:B], flip(pan[B:])
    ldisp, rdisp = disp[0:B], flip(disp[B:])
    lmask, rmask = mask0[0:B], hflip(mask0[B:])
    rlmask, lrmask = mask1[0:B], hflip(mask1[B:])
    vgg_right, vgg_left = (vgg(right), vgg(left)) if a_p > 0 else (None, None)
    c2, c8 = int(0.20 * W), int(0.80 * W)
    if a_mr == 0:
        O_L = O_R = 1  # :300-302
    else:
        O_L = occlusion_mask(lmask, lrmask, 0, c2)   # lmask * lrmask; O_L[:, :, :, 0:c2] = 1
        O_R = occlusion_mask(rmask, rlmask, c8, W)   # rmask * rlmask; O_R[:, :, :, c8:] = 1
    rec_loss = (rec_loss_fnc(O_R, rpan, right, vgg_right, a_p) + rec_loss_fnc(O_L, lpan, left, vgg_left, a_p)) / 2
    sm_loss = 0
    if a_sm > 0:
        sm_loss = (smoothness(left[:, :, :, c2:], ldisp[:, :, :, c2:], gamma=2) +
                   smoothness(right[:, :, :, 0:c8], rdisp[:, :, :, 0:c8], gamma=2)) / 2
    mirror_loss = 0
    if a_mr > 0:  # :316-324: per-sample 1 / max(teacher disparity), (1 - O) weight, windowed masked L1 -- three launches per view
        mirror_loss = (mirror_loss_fnc(ldisp, mldisp, O_L, c2, W) + mirror_loss_fnc(rdisp, mrdisp, O_R, 0, c8)) / 2
    loss = rec_loss + a_sm * sm_loss + a_mr * mirror_loss
    sc = scaled_backward(loss, model)
    opt.step(allreduce_gradients(model), scaler=sc)
    return {"loss": loss.detach(), "rec": rec_loss.detach(), "sm": sm_loss, "mirror": mirror_loss, "ldisp": ldisp, "rdisp": rdisp,
            "O_L": O_L, "O_R": O_R, "scaler": sc}
