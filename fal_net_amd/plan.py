"""Static launch plan of the FAL_netB step on MI355X: buffers, launch order, the three-stream scheduler of backward and the gradient buckets.

One `FalnetPlan` per (batch, height, width, compute dtype, device) holds every activation / gradient tensor (allocated once) and every C-ABI
launch descriptor of `FAL_net.forward` (models/FAL_netB.py:200-297 of the reference) and of its adjoint.  The nn.Module surface that owns the
plans is fal_net_amd/models/FAL_netB.py; the kernels are fal_net_amd/csrc/.  What lives here:

  * `_build`: forward launch list (encoder, decoder, logits conv, MED head) and the lazily built backward list (data gradients on the main stream,
    weight gradients handed to a side stream and -- below full resolution -- a third stream, slab reduces per gradient bucket);
  * `run_forward` / `run_backward`: eager issue for the first two passes, then recorded launch sequences replayed through `falnet_replay`
    (csrc/replay.cpp).  With a gradient-bucket hook (N > 1) the recorded backward is CUT at the bucket boundaries: the hook -- an asynchronous
    `torch.distributed.all_reduce` of that range of the flat gradient buffer -- is called from Python between two replayed segments;
  * `StepStreams`: the side / third / auxiliary streams of a DEVICE (shared by all plans) and their self-test: two spin kernels on every pair of
    the step's streams must overlap; HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues by creation order, and two busy streams of the step
    on ONE queue serialise it (5.5 -> 8.4 ms, profiles/r04_ab_dist_third_stream.txt).
"""
import torch

from . import _lib as L
from . import ops
from .arch import ARCHS
from .ops import pad_c

_TAIL_MAIN = "conv1_1.conv1"  # default of FALNET_TAIL_MAIN: extra weight gradients for the main stream's tail beside level 0's.  Re-tuned whenever a kernel
# change moves work between the chains: "" while the main stream was the longer one; with the deconv data gradients on the low-resolution grid (main
# stream -69 us) the side streams are, and one level-1 weight gradient on the main stream is worth 0.7 % (profiles/r05_ab_up2d.txt)
_TAIL_LEVELS = int(L.ab("FALNET_TAIL_LEVELS", "2"))  # encoder levels (from level 0) in the LAST gradient bucket


def aux_stream(device):
    """The auxiliary stream of a device (the label image's VGG pass beside the network forward, train.vgg_label_async): one per device, kept
    beside the other streams of the step (StepStreams) so that the stream self-test sees every stream the step keeps busy."""
    return StepStreams.for_device(device).aux


def _spin_pair_ms(a, b, us=200):
    """Wall time (ms) of two `us`-microsecond spin kernels issued to streams a and b behind a common gate: ~us when they overlap, ~2 us when HIP
    has mapped both streams onto one hardware queue."""
    lib = L.lib()
    gate, e0 = torch.cuda.Event(), torch.cuda.Event(enable_timing=True)
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(lib.falnet_spin(50, a.cuda_stream), "spin")  # (both spins are queued by the time the gate opens)
    gate.record(a)
    b.wait_event(gate)
    e0.record(a)
    L.check(lib.falnet_spin(us, a.cuda_stream), "spin")
    ea.record(a)
    L.check(lib.falnet_spin(us, b.cuda_stream), "spin")
    eb.record(b)
    ea.synchronize()
    eb.synchronize()
    return max(e0.elapsed_time(ea), e0.elapsed_time(eb))


def _spin_vs_call_ms(x, probe, fn, us=300):
    """ms from the start of a `us` spin on stream x to the completion of fn() (a collective issued from stream `probe`, which is known not to
    collide with x): ~the call's own time when its internal stream runs beside x, > us when it is queued behind the spin."""
    lib = L.lib()
    gate, e0, e1 = torch.cuda.Event(), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    L.check(lib.falnet_spin(50, x.cuda_stream), "spin")
    gate.record(x)
    probe.wait_event(gate)
    e0.record(x)
    L.check(lib.falnet_spin(us, x.cuda_stream), "spin")
    with torch.cuda.stream(probe):
        fn()
        e1.record(probe)
    e1.synchronize()
    x.synchronize()
    return e0.elapsed_time(e1)


def _spin_all_ms(streams, fn=None, us=200, rounds=3):
    """Wall time (ms) of `rounds` rounds of one `us` spin on EVERY stream of `streams` at once (behind a common gate), fn() -- a collective --
    issued from the LAST stream (the auxiliary one) in every round: ~rounds x us when the hardware runs all of them side by side."""
    lib = L.lib()
    a = streams[0]
    gate, e0 = torch.cuda.Event(), torch.cuda.Event(enable_timing=True)
    ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
    L.check(lib.falnet_spin(50, a.cuda_stream), "spin")
    gate.record(a)
    for s in streams[1:]:
        s.wait_event(gate)
    e0.record(a)
    for _ in range(rounds):
        for i, s in enumerate(streams):
            L.check(lib.falnet_spin(us, s.cuda_stream), "spin")
            if fn is not None and i == len(streams) - 1:  # (the auxiliary stream: where the step's bucket all-reduces run)
                with torch.cuda.stream(s):
                    fn()
    for s, e in zip(streams, ends):
        e.record(s)
    for e in ends:
        e.synchronize()
    return max(e0.elapsed_time(e) for e in ends)


class StepStreams:
    """The HIP streams a step keeps busy on ONE device besides the caller's -- side (full-resolution weight gradients, slab reduces, bucket
    hooks), third (weight gradients below full resolution), auxiliary (the label image's VGG pass) -- shared by EVERY plan and model on that
    device (Stage 2's teacher and student, several crop shapes), and their hardware-queue self-test, run once per (caller's stream, hook
    state) and device, not per plan.

    HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues in creation order; two BUSY streams of the step on one queue serialise it
    (5.5 -> 8.4 ms, profiles/r04_ab_dist_third_stream.txt) and nothing in the API says which queue a stream got.  So measure it: on every
    pair of (main, side, third, auxiliary) two 200-us spin kernels must take ~200 us, not ~400; an offending side / third / auxiliary stream
    is replaced by a fresh one until the pair overlaps.  The chip's compute pipes run FOUR queues side by side; a fifth busy queue
    time-slices with one of them (profiles/r05_ab_dist_queues.txt: a world-1 RCCL group's own stream made the step 5.5 -> 5.8 ms at 5 queues,
    9.1 at 8).  So the data-parallel step has no fifth stream: its bucket all-reduces are SYNC collectives issued on the auxiliary stream,
    idle during backward (train.enable_overlapped_allreduce; torch >= 2.8 runs a sync NCCL / RCCL collective on the caller's stream).  With a
    process group (`hooked`) the self-test probes exactly that: the collective on the auxiliary stream beside a spin on each other stream
    must not wait for the spin.  That part issues collectives, so it is ALIGNED ACROSS RANKS: it runs at an explicit collective point
    (train.enable_overlapped_allreduce, which every rank calls at the same place), every round starts behind a barrier, the times are
    MAX-reduced over the ranks, and nothing is decided on them.  `generation` counts stream replacements:
    a plan whose recorded launch sequences were made under another generation drops them (they hold raw stream handles)."""

    _BY_DEVICE = {}

    def __init__(self, device):
        self.device = device
        self.side = torch.cuda.Stream(device=device)
        self.third = torch.cuda.Stream(device=device)
        self.aux = torch.cuda.Stream(device=device)
        self.generation = 0
        self.tested = set()       # (handle of the caller's stream, hooked) states already measured under the current generation
        self.result = None        # last self-test record (bench.py prints it)
        self.hooked_tested = False
        self.third_with_hook = L.ab("FALNET_DEEP_WITH_HOOK", "1") == "1"  # third stream beside a bucket hook -- when the self-test found it a queue of its own

    @classmethod
    def for_device(cls, device):
        device = torch.device(device)
        key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
        if key not in cls._BY_DEVICE:
            cls._BY_DEVICE[key] = cls(torch.device(*key))
        return cls._BY_DEVICE[key]

    def ensure_tested(self, main, hooked):
        """Run the self-test for this (caller's stream, hook state) unless it already ran under the current set of streams.  Flipping between
        states (gradient accumulation in a data-parallel run: hooked, not hooked, hooked, ...) costs nothing after the first visit of each."""
        if L.ab("FALNET_STREAM_SELFTEST", "1") != "1":
            return
        if (main.cuda_stream, bool(hooked)) in self.tested:
            return
        # the collective part runs once per device (normally at train.enable_overlapped_allreduce); a later visit from another caller's stream, or
        # after a local replacement, repeats only the LOCAL pair part -- no rank ever issues a probe collective its peers do not
        again = bool(hooked) and self.hooked_tested
        self.selftest(main, hooked and not again)
        if again:
            self.tested.add((main.cuda_stream, True))

    def selftest(self, main, hooked, tries=12):
        dev = self.device
        import os
        dist = torch.distributed
        with_group = bool(hooked) and dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if with_group else 1
        roles = [("main", lambda: main, None), ("side", lambda: self.side, lambda s: setattr(self, "side", s)),
                 ("third", lambda: self.third, lambda s: setattr(self, "third", s)), ("aux", lambda: self.aux, lambda s: setattr(self, "aux", s))]
        coll = None
        if with_group:
            buf = torch.zeros(1 << 18, device=dev)

            def coll():  # (on the caller's CURRENT stream: a sync collective of the NCCL / RCCL backend, torch >= 2.8)
                dist.all_reduce(buf, async_op=False)
        log, accepted, replaced = [], [], 0
        torch.cuda.synchronize(dev)
        for name, get, put in roles:  # local part: no collectives, any rank may loop as long as it needs
            n, worst = 0, 0.0
            while True:
                s = get()
                worst = max([_spin_pair_ms(o, s) for _, o in accepted] + [0.0])
                if worst < 0.3 or put is None or n >= tries:
                    break
                put(torch.cuda.Stream(device=dev))
                n += 1
                replaced += 1
            log.append({"stream": name, "handle": hex(get().cuda_stream), "replaced": n, "worst_pair_ms": round(worst, 3)})
            accepted.append((name, get()))
        collective = None
        if coll is not None:
            # The bucket all-reduces run as SYNC collectives on the auxiliary stream (train.enable_overlapped_allreduce): no stream of the
            # backend's own beside the step's four.  Probe exactly that: a spin on each other stream, the collective on aux beside it must not
            # wait for the spin.  Every rank issues the SAME collectives whatever it measures -- three fixed rounds, each behind a barrier, the
            # times MAX-reduced over the ranks -- and NO stream is replaced on these numbers (the pair test above has already placed aux).
            aux_s = dict(accepted)["aux"]
            with torch.cuda.stream(aux_s):
                coll()  # (first call: communicator set-up outside the measurement)
            torch.cuda.synchronize(dev)
            others = [(n, s_) for n, s_ in accepted if n != "aux"]
            for _ in range(3):
                if world > 1:
                    dist.barrier()
                collective = {name: _spin_vs_call_ms(s_, aux_s, coll) for name, s_ in others}
                if world > 1:
                    t = torch.tensor([collective[name] for name, _ in others], device=dev)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    collective = {name: float(v) for (name, _), v in zip(others, t.tolist())}
                collective = {k: round(v, 3) for k, v in collective.items()}
            self.hooked_tested = True
        torch.cuda.synchronize(dev)
        if coll is not None and world > 1:
            dist.barrier()
        all_ms = _spin_all_ms([s for _, s in accepted], coll)  # every stream of the step busy at once (three rounds of 200 us), the collective among them
        torch.cuda.synchronize(dev)
        hwq = os.environ.get("GPU_MAX_HW_QUEUES")
        self.result = {"pairs_overlap": all(e["worst_pair_ms"] < 0.3 for e in log), "streams": log, "collective_beside_spin_ms": collective,
                       "all_streams_3x200us_ms": round(all_ms, 3), "streams_replaced": replaced,
                       "third_stream_with_hook": bool(self.third_with_hook) if hooked else None, "hw_queues": hwq, "world": world,
                       "decisions_max_reduced_over_ranks": bool(with_group and world > 1)}
        if hooked and hwq is not None and hwq.isdigit() and int(hwq) > 5:
            # the pair / collective probes do NOT see the many-queue cliff (profiles/r05_ab_dist_queues.txt: 9.07 ms at 8 queues with a green
            # self-test, 5.77 at 5; world-1 RCCL group): say so in the record and on stderr
            self.result["hw_queues_note"] = (f"GPU_MAX_HW_QUEUES={hwq} with a gradient-bucket hook: measured 1.6x slower than 5 queues on a world-1 RCCL "
                                             "group (9.07 vs 5.77 ms, profiles/r05_ab_dist_queues.txt); this self-test does not detect it")
            import warnings
            warnings.warn(self.result["hw_queues_note"])
        if replaced:
            self.generation += 1
            self.tested.clear()
        self.tested.add((main.cuda_stream, bool(hooked)))
        if hooked:
            self.tested.add((main.cuda_stream, False))  # (the pair part is the same measurement)
        return self.result


class _WgradPart:
    """One input-channel group of a two-source convolution presented to ops.WgradBatch as a layer of its own: `cin` stays the row stride
    of the full OIHW gradient, the group's channels are the packed columns [0, c_pad) -> real columns [0, c_real) of the gradient VIEW
    the caller passes (offset to the group's first input channel)."""

    def __init__(self, pc, c_real, c_pad):
        self.cout, self.cin, self.cin_pad, self.bias, self.taps, self.ksize, self.stride = pc.cout, pc.cin, c_pad, pc.bias, pc.taps, pc.ksize, pc.stride
        self._g = (c_real, c_pad)

    def group_channels(self):
        return self._g


class FalnetPlan:
    """Static launch plan of FAL_net.forward / backward for one (B, H, W, dtype, device)."""

    def __init__(self, model, B, H, W, dtype, device):
        self.model, self.B, self.H, self.W, self.dtype, self.device = model, B, H, W, dtype, device
        t = ARCHS[model.arch]
        # encoder level i: (stride-2 conv name, residual block name, channels)
        self._enc = [(f"conv{i}", f"conv{i}_1", ch) for i, ch in enumerate(t["enc"])]
        # decoder level i (6..1): (deconv name, deconv Cout, iconv name, iconv Cout)
        self._dec = {lvl: (f"deconv{lvl}", dch, f"iconv{lvl}", ich) for lvl, (dch, ich) in t["dec"].items()}
        self._dec[1] = ("deconv1", 64, "iconv1", None)
        self.N = model.no_levels
        self.generation = 0
        self.use_side_stream = True
        self._streams = StepStreams.for_device(device)  # side / third / auxiliary streams: per DEVICE, shared by every plan
        self._streams_gen = self._streams.generation
        self._side_stream = None
        self._side_pending, self._side_events, self._side_ev_next = [], [], 0
        self._side_batch = max(1, int(L.ab("FALNET_SIDE_BATCH", "2")))
        # third stream of backward: every weight gradient below full resolution (levels 1-6).  The deep ones (levels 4-6) are 12-30 us
        # launches that fill a fraction of the chip and are latency-, not throughput-bound; queued behind the big full-resolution weight
        # gradients on the side stream they lengthen the LONGER chain of backward by ~0.3 ms while the chip idles.  With the side stream
        # keeping only the full-resolution layers (logits, deconv1, level 0) and this stream everything else, two weight-gradient chains
        # of 128 workgroups each run beside the data gradients: same-box A/B -1.7 % on the step (profiles/r04_ab_wgrad_streams.txt:
        # levels 4-6 only -0.7 %, levels 3-6 -1.0 %, levels 1-6 -1.7 %, everything on this stream or alternating launches: worse).
        self._deep_stream = None
        self._deep_pending, self._deep_events, self._deep_ev_next, self._deep_dirty = [], [], 0, False
        self._sync_events, self._sync_ev_next = [], 0
        self._bwd_segments, self._bwd_eager_runs = {}, {}
        self._fwd_segments, self._fwd_eager_runs = {}, {}
        self._main_stream = None
        self._deep_batch = max(1, int(L.ab("FALNET_DEEP_BATCH", "3")))
        self._deep_alt = L.ab("FALNET_DEEP_ALT", "0") == "1"  # experiment: every second larger weight gradient on the third stream as well
        self._alt_n = 0
        self._deep_max_px = int(L.ab("FALNET_DEEP_STREAM_PX", str(H * W // 4)))  # maps of at most this many positions (level 1 and below); 0 = off
        self.buf = {}
        self.fwd, self.bwd_head, self.bwd_body, self.pack = [], [], [], []
        self._build()

    # ---- helpers ----
    def _conv_call(self, *a, **kw):
        """ops.conv_call with this plan's own split-K scratch (plans run concurrently on different streams)."""
        return ops.conv_call(*a, ws_owner=("falnet", id(self)), **kw)

    def _act(self, name, h, w, c):
        t = torch.empty(self.B, h, w, c, dtype=self.dtype, device=self.device)
        self.buf[name] = t
        return t

    def _f32(self, name, *shape):
        t = torch.empty(*shape, dtype=torch.float32, device=self.device)
        self.buf[name] = t
        return t

    def _conv_fwd(self, pc, srcs, IH, IW, out, act, addend=None, name=""):
        B = self.B
        OH, OW = out.shape[1], out.shape[2]
        self.fwd.append(self._conv_call(self.dtype, srcs, IH, IW, pc.wf, pc.cin_pad, ops.fwd_taps(pc.ksize), pc.taps,
                                      pc.cout_pad, pc.stride, B, OH, OW, out, OH, OW, out.shape[3], out.shape[3],
                                      bias=pc.bias, addend=addend, act=act, name="fwd " + name,
                                      flops=2 * B * OH * OW * pc.cout * pc.cin * pc.taps,
                                      weight_up2=pc.wu if (len(srcs) == 1 and 2 * srcs[0].H == IH and 2 * srcs[0].W == IW) else None))

    def _dgrad(self, pc, group, gout, gin, IH, IW, addend=None, actout=None, name="", sum2x2_into=None, sum2x2_actout=None):
        """Append launches computing gin = dgrad_group(gout) [+ addend] [* elu'(actout)].
        gin lives on the conv's (virtual) input grid IH x IW; gout on its output grid.
        sum2x2_into: instead of storing gin, store its 2x2 block sums * elu'(sum2x2_actout) (the adjoint of the exact 2x
        nearest upsampling in front of a deconv, FAL_netB.py:58) -- raises ValueError when no fused kernel applies."""
        B = self.B
        if sum2x2_into is not None:
            assert pc.stride == 1 and addend is None and actout is None
            OH, OW = gout.shape[1], gout.shape[2]
            off = sum(pc.groups_pad[:group]) * pc.taps * pc.cout_pad
            cg = pc.groups_pad[group]
            call = self._conv_call(
                self.dtype, [ops.nhwc_src(gout)], OH, OW, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(pc.ksize), pc.taps, cg, 1, B, IH, IW,
                None, IH, IW, cg, sum2x2_into.shape[3], weight_offset_elems=off, name="dgrad+sum2x2 " + name,
                flops=2 * B * OH * OW * pc.cout * pc.groups_real[group] * pc.taps, pool_out=sum2x2_into, pool_mode=1,
                pool_actout=sum2x2_actout, pool_actout_kind=L.ACT_ELU if sum2x2_actout is not None else L.ACT_NONE)
            self.bwd_body.append(call)
            return
        OH, OW = gout.shape[1], gout.shape[2]
        off = sum(pc.groups_pad[:group]) * pc.taps * pc.cout_pad
        cg = pc.groups_pad[group]
        kind = L.ACT_ELU if actout is not None else L.ACT_NONE
        src = [ops.nhwc_src(gout)]
        fl1 = 2 * B * OH * OW * pc.cout * pc.groups_real[group]  # algorithmic flops per tap
        if pc.stride == 1:
            self.bwd_body.append(self._conv_call(
                self.dtype, src, OH, OW, pc.wd, pc.cout_pad, ops.dgrad_taps_s1(pc.ksize), pc.taps, cg, 1, B, IH, IW,
                gin, IH, IW, cg, gin.shape[3], addend=addend, actout=actout, actout_kind=kind,
                weight_offset_elems=off, name="dgrad " + name, flops=fl1 * pc.taps))
        else:
            # the four output-parity classes of the stride-2 data gradient: ONE launch (blockIdx.z = class) or, for
            # the tiny bottleneck layers where split-K matters more, four separately autotuned launches -- timed here
            members, singles = [], []
            for py in range(2):
                for px in range(2):
                    th, tw = (IH - py + 1) // 2, (IW - px + 1) // 2
                    if th <= 0 or tw <= 0:
                        continue
                    args = (self.dtype, src, OH, OW, pc.wd, pc.cout_pad, ops.dgrad_taps_s2(py, px), pc.taps, cg, 1, B, th, tw, gin, IH, IW,
                            cg, gin.shape[3])
                    kw = dict(out_step=(2, 2, py, px), addend=addend, actout=actout, actout_kind=kind, weight_offset_elems=off,
                              name=f"dgrad{py}{px} " + name, flops=fl1 * len(ops.dgrad_taps_s2(py, px)))
                    members.append(self._conv_call(*args, autotune=False, **kw))
                    singles.append(self._conv_call(*args, **kw))
            multi = ops.conv_multi_call(members, name="dgrad(s2 x4) " + name)
            multis = [multi]
            if (len(members) == 4 and self.dtype in ops.H16 and IH % 2 == 0 and IW % 2 == 0 and OH >= 16 and OW >= 32
                    and L.ab("FALNET_S2D_DMA", "1") == "1"):
                # all four classes from ONE staged gout patch (conv_dma.hip): the gather form re-reads gout per tap and channel slice
                multis.append(ops.conv_multi_call(members, name="dgrad(s2 x4, dma) " + name, s2d=True))
            if ops.AUTOTUNE and L.ab("FALNET_S2_SPLITK", "1") == "1":
                # small levels: the four classes as ONE split-K launch + ONE epilogue instead of one long-K launch or 4 x (split-K + epilogue)
                wgs = 4 * ((B * ((IH + 1) // 2) * ((IW + 1) // 2) + 127) // 128) * ((cg + ops.gather_bn(cg, cg) - 1) // ops.gather_bn(cg, cg))
                for k in (2, 4, 8):
                    if wgs * k <= 2048 and wgs < 512:
                        try:
                            multis.append(ops.conv_multi_call(members, name=f"dgrad(s2 x4, k{k}) " + name, ksplit=k))
                        except ValueError:
                            pass
            if ops.GATHER_NARROW and B * IH * IW // 4 <= 32768:  # small levels: narrower workgroups fill more of the chip
                multis += [ops.conv_multi_call(members, name="dgrad(s2 x4) " + name, bn=nb) for nb in (64, 32)
                           if nb < ops.gather_bn(cg, cg) and cg % nb == 0]

            def separate(calls=tuple(singles)):
                for c in calls:
                    c()
            s2key = f"s2dgrad|t{L.dtype_code(self.dtype)}|B{B}|{IH}x{IW}|{pc.cout_pad}>{cg}|a{int(addend is not None)}{int(actout is not None)}|n{len(multis)}d"
            chosen = ops.best_of(*multis, separate, key=s2key) if (ops.AUTOTUNE and L.ab('FALNET_S2_MULTI', None) != '1') else multi
            if chosen is separate:
                self.bwd_body.extend(singles)
            else:
                self.bwd_body.append(chosen)

    def _wgrad(self, pc, srcs, IH, IW, gout, name="", on_main=False, up2=False):
        """up2: a `deconv` layer (its one source sits at exactly half of gout's map): the gradient is taken on the LOW-resolution grid where the
        row-streaming kernel applies (falnet_wgrad_t::up2: 16 instead of 36 tap products per position), else on gout's grid as every other layer's."""
        OH, OW = gout.shape[1], gout.shape[2]
        self._buckets_seen = getattr(self, "_buckets_seen", set()) | {self._bucket}
        gw = self.model._grad_view(pc.weight)
        gb = self.model._grad_view(pc.bias) if pc.bias is not None else None
        def route(call):
            if on_main:  # tail balancing: the main stream has nothing left to do once its last data gradient is out
                self._main_tail = getattr(self, "_main_tail", [])
                self._main_tail.append(call)  # queued behind the LAST data gradient (flushed at the end of the encoder loop)
            elif OH * OW <= self._deep_max_px or (self._deep_alt and self._alt_toggle()):
                self._deep_call(call)
            else:
                self._side_call(call)
        if up2 and len(srcs) == 1 and pc.stride == 1 and pc.taps == 9 and self.dtype in ops.H16 and ops.UP2W:
            try:
                lh, lw = OH // 2, OW // 2
                route(self.wbatch.add(srcs, lh, lw, gout, [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(pc.ksize)], 1, self.B, lh, lw, pc, gw, gb,
                                      name="wgrad(low-res) " + name, flops=2 * self.B * OH * OW * pc.cout * pc.cin * pc.taps, bucket=self._bucket, up2=True))
                return
            except ValueError:
                pass
        if (len(srcs) == 2 and pc.groups_pad == [64, 32] and pc.stride == 1 and pc.taps == 9 and self.dtype in ops.H16
                and L.ab("FALNET_SPLIT_WGRAD_96", "1") == "1"):
            # 64 + 32 input channels (the logits conv over concat(deconv1, conv0_1)): the row-streaming kernel works on 64 x 64 channel
            # blocks, so its second input-channel block would be half padding -- two of every four waves idle through the full-resolution
            # pass (294 us, the longest weight gradient of the step).  One launch per source instead: the 64-channel source on the
            # row-streaming kernel, the 32-channel source on the 32 x 64 halo-patch kernel; each reduces into its own input-channel
            # columns of the same OIHW gradient (row stride = all 96 channels).
            taps = [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(pc.ksize)]
            calls = []
            for gi, (src, c_real, c_pad, col0) in enumerate(((srcs[0], pc.groups_real[0], 64, 0), (srcs[1], pc.groups_real[1], 32, pc.groups_real[0]))):
                part = _WgradPart(pc, c_real, c_pad)
                calls.append(self.wbatch.add([src], IH, IW, gout, taps, 1, self.B, OH, OW, part, gw[:, col0:], gb if gi == 0 else None,
                                             name=f"wgrad {name}[{'deconv' if gi == 0 else 'skip'}]",
                                             flops=2 * self.B * OH * OW * pc.cout * c_real * 9, bucket=self._bucket))
            for c in calls:
                self._side_call(c)
            return
        if (len(srcs) == 2 and srcs[1].sy == 0 and srcs[1].sx == 0 and pc.groups_real[1] == 1 and pc.taps == 9 and self.dtype in ops.H16
                and not ops.DETERMINISTIC and OH >= 2 and OW >= 2 and L.ab("FALNET_FLOW_WGRAD", "1") == "1"):
            # conv1 over concat(conv0_1 output, `flow`): the second source is ONE real channel, constant per sample, padded to 32 -- half of the
            # stride-2 launch's K tiles multiply zeros.  The image source alone goes through the MFMA kernel; the flow channel's nine weights
            # per output channel come from nine masked sums of the gradient (falnet_wgrad_const_plane), added straight into its column of dW.
            c_real, c_pad = pc.groups_real[0], pc.groups_pad[0]
            main_call = self.wbatch.add([srcs[0]], IH, IW, gout, [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(pc.ksize)], pc.stride, self.B, OH, OW,
                                        _WgradPart(pc, c_real, c_pad), gw, gb, name="wgrad " + name,
                                        flops=2 * self.B * OH * OW * pc.cout * pc.cin * pc.taps, bucket=self._bucket)
            flow_t, gC = self.buf["flow"], gout.shape[3]
            ws = torch.zeros(self.B * 9 * gC, dtype=torch.float32, device=self.device)  # zero on entry, left zero by the kernel
            self.buf["flow_wgrad_ws"] = ws
            plane_call = ops.simple_call("falnet_wgrad_const_plane", L.ptr(gout), L.ptr(flow_t), flow_t.stride(0), L.ptr(gw[:, c_real:]),
                                         pc.cin * 9, L.ptr(ws), self.B, OH, OW, gC, pc.cout, IH, IW, pc.stride, L.dtype_code(self.dtype),
                                         name="wgrad " + name + "[flow plane]")

            self._side_call(plane_call)  # (independent of the MFMA launch: it goes to the side stream, which idles at the end of backward)
            call = main_call
        else:
            call = self.wbatch.add(srcs, IH, IW, gout, [(dy, dx, 0) for dy, dx, _ in ops.fwd_taps(pc.ksize)], pc.stride, self.B, OH, OW,
                                   pc, gw, gb, name="wgrad " + name, flops=2 * self.B * OH * OW * pc.cout * pc.cin * pc.taps,
                                   bucket=self._bucket)
        route(call)

    def _alt_toggle(self):
        self._alt_n += 1
        return self._alt_n % 2 == 0

    def _deep_call(self, call):
        """A small weight gradient for the third stream (see __init__); handed over in groups like the side stream's launches."""
        def run(c=call):
            if self._deep_stream is None:  # no third stream in this pass (a bucket hook is installed, or the side streams are off): side stream
                if self._side_stream is None:
                    c()
                    return
                self._side_pending.append(c)
                if len(self._side_pending) >= self._side_batch:
                    self._flush_side()
                return
            self._deep_pending.append(c)
            if len(self._deep_pending) >= self._deep_batch:
                self._flush_deep()
        self.bwd_body.append(run)

    def _event(self, pool, index):
        if index == len(pool):
            pool.append(torch.cuda.Event())
        return pool[index]

    def _flush_deep(self):
        pend, deep = self._deep_pending, self._deep_stream
        if not pend:
            return
        ev = self._event(self._deep_events, self._deep_ev_next)
        self._deep_ev_next += 1
        L.ev_record(ev, self._main_stream)  # main stream: behind the producer of the group's last gradient
        L.ev_wait(deep, ev)
        with L.on_stream(deep):
            for c in pend:
                c()
        pend.clear()
        self._deep_dirty = True

    def _join_deep(self, stream):
        """`stream` waits for everything the third stream has been given so far (a bucket's slab reduce reads those slabs)."""
        if self._deep_stream is None:
            return
        self._flush_deep()
        if self._deep_dirty:
            ev = self._event(self._deep_events, self._deep_ev_next)
            self._deep_ev_next += 1
            L.ev_record(ev, self._deep_stream)
            L.ev_wait(stream, ev)
            self._deep_dirty = False

    def _side_call(self, call):
        """Weight gradients are off the data-gradient critical path: run them on a side HIP stream so they fill the
        CUs that the small (latency-bound) dgrad launches leave idle.  Ordering: the side stream waits for the producer
        of the launch's inputs (event on the main stream); run_backward joins the side stream at the end.
        An event record between two kernels of the main stream costs a ~6 us bubble there (rocprofv3 timeline), so side launches
        are handed over in groups of `_side_batch`: one record / wait pair per group, taken behind the group's LAST producer."""
        def run(c=call):
            if self._side_stream is None:
                c()
                return
            self._side_pending.append(c)
            if len(self._side_pending) >= self._side_batch or getattr(c, "needs_torch_stream", False):
                self._flush_side()
        self.bwd_body.append(run)

    def _flush_side(self):
        pend, side = self._side_pending, self._side_stream
        if not pend:
            return
        if self._side_ev_next == len(self._side_events):
            self._side_events.append(torch.cuda.Event())
        ev = self._side_events[self._side_ev_next]
        self._side_ev_next += 1
        L.ev_record(ev, self._main_stream)
        L.ev_wait(side, ev)
        for c in pend:
            if getattr(c, "needs_torch_stream", False):  # the bucket hook (torch.distributed collectives run on torch's current stream)
                with torch.cuda.stream(side), L.on_stream(side):
                    c()
            else:
                with L.on_stream(side):  # C-ABI launches name their stream: no torch stream switch needed
                    c()
        pend.clear()

    # ---- plan construction ----
    def _build(self):
        m, B, H, W, N, dt, dev = self.model, self.B, self.H, self.W, self.N, self.dtype, self.device
        lib = L.lib()
        code = L.dtype_code(dt)
        pcs = m._packed
        compose = getattr(m, "_compose_logits", False)
        if compose:
            w1_2d = m.conv0.weight.detach().view(m.conv0.weight.shape[0], -1)
            w3_2d = m._bb.iconv1.weight.detach().view(m._bb.iconv1.weight.shape[0], -1)
            wc_2d = m._wc.view(m._wc.shape[0], -1)

            n1, k3 = w1_2d.shape[0], w3_2d.shape[1]
            compose_call = ops.simple_call("falnet_gemm_f32_small", L.ptr(w1_2d), w1_2d.shape[1], 1, L.ptr(w3_2d), k3, 1, L.ptr(wc_2d),
                                           n1, k3, w1_2d.shape[1], 0, name="compose logits weights")
            self.pack.append(compose_call)  # before the re-pack below: the composed f32 master changes with every update
        packed_now = [pc for k, pc in pcs.items() if not (compose and k in ("iconv1", "conv0_1x1"))]
        for k, pc in pcs.items():
            # sub-pixel weights for the nearest-upsample + 3x3 layers whose low-resolution map is at least 32 x 64 (falnet_conv2d variant 18;
            # below that the 4 x 32-tile kernel wins and the per-step repack would be wasted): deconv<l> reads level l, H >> l
            pc.up2 = (k.startswith("deconv") and k[6:].isdigit() and (H >> int(k[6:])) >= 32 and (W >> int(k[6:])) >= 64
                      and L.ab("FALNET_UP2", "1") == "1")
        for pc in packed_now:
            pc.alloc(dt, dev)
        self.pack.append(ops.pack_all_call(packed_now, dt, dev))
        up2 = ops.pack_up2_call(packed_now, dt, dev)
        if up2 is not None:
            self.pack.append(up2)
        # Optimiser step fused with the re-pack (train.FlatAdam.step -> model.adam_and_repack): the layers whose f32 masters live in the flat
        # parameter buffer are updated by the launch that packs them; everything else in the buffer (biases, the two factors of the composed
        # logits weights) by a range list; derived weights (the composed logits conv, the sub-pixel deconv weights) are rebuilt behind them.
        flat = m._flat
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        owned = [pc for pc in packed_now if lo <= pc.weight.data_ptr() < hi]
        derived = [pc for pc in packed_now if not (lo <= pc.weight.data_ptr() < hi)]
        spans = sorted(((pc.weight.data_ptr() - lo) // 4, pc.weight.numel()) for pc in owned)
        rest, pos = [], 0
        for a, cnt in spans:
            if a > pos:
                rest += [pos, a - pos]
            pos = a + cnt
        if pos < flat.numel():
            rest += [pos, flat.numel() - pos]
        self.adam_pack = dict(
            rest=torch.tensor(rest, dtype=torch.int64, device=dev) if rest else None, n_rest=len(rest) // 2,
            before=[compose_call] if compose else [],  # (needs the factors the range update has just written)
            owned=ops.adam_pack_call(owned, dt, dev, derived) if owned else None,
            after=[up2] if up2 is not None else [])
        self.wbatch = ops.WgradBatch(dt, dev)

        # boundary tensors (planar f32)
        left = self._f32("left", B, 3, H, W)
        mn, mx = self._f32("min_disp", B), self._f32("max_disp", B)
        flow = torch.zeros(B, pad_c(1), dtype=dt, device=dev)
        self.buf["flow"] = flow
        # NHWC copy of the image: only the first layer's weight gradient reads it (the forward uses the planar image
        # directly, falnet_conv3x3_c3), so the conversion runs in backward on the side stream
        x0 = self._act("x0", H, W, pad_c(3))
        self._x0_convert = ops.simple_call("falnet_nchw_to_nhwc", L.ptr(left), L.ptr(x0), B, 3, H, W, pad_c(3), code)

        # ---- encoder ----
        sizes = [(H, W)]
        for _ in range(6):
            sizes.append(((sizes[-1][0] - 1) // 2 + 1, (sizes[-1][1] - 1) // 2 + 1))
        a, h_, c = {}, {}, {}
        for i, (cname, rname, ch) in enumerate(self._enc):
            hh, ww = sizes[i]
            a[i], h_[i], c[i] = self._act(f"a{i}", hh, ww, ch), self._act(f"h{i}", hh, ww, ch), self._act(f"c{i}", hh, ww, ch)
            if i == 0:
                # bf16: conv0's weight gradient reads the planar f32 image itself (falnet_wgrad variant 6): no NHWC copy of the image
                self._c3_wgrad = dt in ops.H16 and W >= 16 and L.ab("FALNET_WGRAD_C3", "1") == "1"
                srcs, ih, iw = [ops.planar_src(left) if self._c3_wgrad else ops.nhwc_src(x0)], H, W
            elif i == 1:
                ih, iw = sizes[0]
                srcs = [ops.nhwc_src(c[0]), ops.bcast_src(flow, ih, iw)]
            else:
                ih, iw = sizes[i - 1]
                srcs = [ops.nhwc_src(c[i - 1])]
            self._enc_srcs = getattr(self, "_enc_srcs", {})
            self._enc_srcs[i] = (srcs, ih, iw)
            if i == 0:
                self.fwd.append(ops.conv_c3_call(dt, left, pcs[cname], a[0], L.ACT_ELU, name="fwd conv0(c3)"))
            else:
                self._conv_fwd(pcs[cname], srcs, ih, iw, a[i], L.ACT_ELU, name=cname)
            self._conv_fwd(pcs[rname + ".conv1"], [ops.nhwc_src(a[i])], hh, ww, h_[i], L.ACT_ELU, name=rname + ".conv1")
            self._conv_fwd(pcs[rname + ".conv2"], [ops.nhwc_src(h_[i])], hh, ww, c[i], L.ACT_ELU, addend=a[i],
                           name=rname + ".conv2")
            if i == int(L.ab("FALNET_MID_HOOK_LEVEL", "2")):
                # from here on (levels 3-6 of the encoder, 6-3 of the decoder) the launches are small and leave most CUs
                # idle: the trainer's mid-forward hook starts independent heavy work (the label's VGG features) HERE, on
                # another stream, instead of beside the chip-filling level-0..2 layers
                self._mid_index = len(self.fwd)
        # ---- decoder ----
        d, ic = {}, {7: c[6]}
        for lvl in range(6, 0, -1):
            dname, dch, iname, ich = self._dec[lvl]
            hh, ww = sizes[lvl - 1]
            below = ic[lvl + 1]  # tensor being upsampled (c6, then iconv outputs)
            d[lvl] = self._act(f"d{lvl}", hh, ww, dch)
            self._conv_fwd(pcs[dname], [ops.nhwc_src(below)], hh, ww, d[lvl], L.ACT_ELU, name=dname)
            skip = c[lvl - 1]
            if lvl > 1:
                ic[lvl] = self._act(f"i{lvl}", hh, ww, ich)
                self._conv_fwd(pcs[iname], [ops.nhwc_src(d[lvl]), ops.nhwc_src(skip)], hh, ww, ic[lvl], L.ACT_ELU, name=iname)
            elif compose:
                logits_srcs = [ops.nhwc_src(d[lvl]), ops.nhwc_src(skip)]
            else:
                dlog = self._act("dlog", hh, ww, pad_c(N))
                self._conv_fwd(pcs[iname], [ops.nhwc_src(d[lvl]), ops.nhwc_src(skip)], hh, ww, dlog, L.ACT_NONE, name=iname)
        # ---- logits: planar f32 [B][N][H][W] for the MED head ----
        pc0 = pcs["conv0_1x1"]
        dlog0 = self._f32("dlog0", B, N, H, W)
        if compose:
            pcl = pcs["logits"]
            self.fwd.append(self._conv_call(dt, logits_srcs, H, W, pcl.wf, pcl.cin_pad, ops.fwd_taps(3), 9, pcl.cout_pad, 1, B, H, W,
                                            dlog0, H, W, N, 0, out_layout=L.OUT_PLANAR_F32, bias=pcl.bias, name="fwd logits(iconv1*conv0)",
                                            flops=2 * B * H * W * N * pcl.cin * 9))
        else:
            self.fwd.append(self._conv_call(dt, [ops.nhwc_src(dlog)], H, W, pc0.wf, pc0.cin_pad, ops.fwd_taps(1), 1, pc0.cout_pad,
                                            1, B, H, W, dlog0, H, W, N, 0, out_layout=L.OUT_PLANAR_F32, bias=pc0.bias,
                                            name="fwd conv0(1x1)", flops=2 * B * H * W * N * N))
        disp, pan, stats = self._f32("disp", B, 1, H, W), self._f32("p_im0", B, 3, H, W), self._f32("stats", B, 4, H, W)
        maskL, maskR = self._f32("maskL", B, 1, H, W), self._f32("maskR", B, 1, H, W)
        self.head_disp_only = ops.simple_call("falnet_med_head_fwd", L.ptr(dlog0), L.ptr(left), L.ptr(mn), L.ptr(mx),
                                              L.ptr(disp), L.ptr(None), L.ptr(stats), B, N, H, W)
        self.head_full = ops.simple_call("falnet_med_head_fwd", L.ptr(dlog0), L.ptr(left), L.ptr(mn), L.ptr(mx),
                                         L.ptr(disp), L.ptr(pan), L.ptr(stats), B, N, H, W,
                                         nbytes=(N + 7) * H * W * 4 * B)
        self.head_masks = ops.simple_call("falnet_med_masks_fwd", L.ptr(dlog0), L.ptr(mn), L.ptr(mx), L.ptr(stats),
                                          L.ptr(maskL), L.ptr(maskR), B, N, H, W)
        if not ARCHS[m.arch]["maskr_align_corners"]:  # FAL_netA.py:264: maskR sampled with align_corners=False
            both = self.head_masks
            fix_r = ops.simple_call("falnet_med_maskr_acfalse_fwd", L.ptr(dlog0), L.ptr(mn), L.ptr(mx), L.ptr(stats), L.ptr(maskR),
                                    B, N, H, W)

            def masks_a():
                both()
                fix_r()
            self.head_masks = masks_a

        # The backward half (activation-gradient buffers, ~40 data-gradient and ~34 weight-gradient launches with their slab
        # arena, their autotuning) is built on the FIRST run_backward: inference models, every extra ms_pp shape and the frozen
        # Stage-2 teacher never pay for it (gigabytes of HBM and seconds of first-call latency per input shape).
        def build_backward():
            # =========================== backward plan ===========================
            g_disp, g_pan = self._f32("g_disp", B, 1, H, W), self._f32("g_pan", B, 3, H, W)
            G0 = self._act("G0", H, W, pad_c(N))  # grad wrt conv0(1x1) output, written NHWC by the head backward itself

            def head_bwd(has_disp, has_pan):
                return ops.simple_call("falnet_med_head_bwd_nhwc", L.ptr(dlog0), L.ptr(left), L.ptr(mn), L.ptr(mx), L.ptr(disp),
                                       L.ptr(pan), L.ptr(stats), L.ptr(g_disp if has_disp else None),
                                       L.ptr(g_pan if has_pan else None), L.ptr(G0), pad_c(N), code, B, N, H, W,
                                       name="falnet_med_head_bwd", nbytes=(N + 7) * H * W * 4 * B + B * H * W * pad_c(N) * G0.element_size())
            self.head_bwd = {(hd, hp): head_bwd(hd, hp) for hd in (False, True) for hp in (False, True) if hd or hp}
            # Gradient buckets = contiguous ranges of the flat gradient buffer in the order backward completes them:
            # 0: decoder + logits conv (tail of the buffer), 1: encoder levels 4-6, 2: levels 2-3, 3: levels 0-1.  After a bucket's
            # last wgrad its slab reduce / bias-gradient launches run and model._bucket_ready(i) lets the trainer start that
            # bucket's share of the step's all-reduce while backward continues.
            self._bucket = 0
            self._finish = []  # placeholders in bwd_body, patched after WgradBatch.finalize()
            # NHWC copy of the image for conv0's weight gradient: converted on the side stream; that weight gradient runs on the MAIN
            # stream at the tail (tail balancing), so it waits for this event
            self._x0_event = torch.cuda.Event()

            def x0_convert_and_mark():
                if not self._c3_wgrad:
                    self._x0_convert()
                L.ev_record(self._x0_event, torch.cuda.current_stream())  # on the stream the conversion was launched on
            x0_convert_and_mark.needs_torch_stream = True  # (the event is recorded on torch's current stream: make that the side stream)
            self._side_call(x0_convert_and_mark)
            if compose:
                g_dlog = G0  # the composed conv's output gradient IS the head's gradient
            else:
                self._wgrad(pc0, [ops.nhwc_src(dlog)], H, W, G0, name="conv0(1x1)")
                g_dlog = self._act("g_dlog", H, W, pad_c(N))
                self._dgrad(pc0, 0, G0, g_dlog, H, W, name="conv0(1x1)")

            gc = {i: self._act(f"g_c{i}", sizes[i][0], sizes[i][1], self._enc[i][2]) for i in range(7)}
            # decoder, top (level 1) to bottom (level 6)
            g_ipre = {1: g_dlog}  # gradient wrt the pre-activation of iconv{lvl} (iconv1 has no activation)
            for lvl in range(1, 7):
                dname, dch, iname, ich = self._dec[lvl]
                hh, ww = sizes[lvl - 1]
                below = ic[lvl + 1]
                bh, bw = below.shape[1], below.shape[2]
                pci, pcd = pcs[iname], pcs[dname]
                if compose and lvl == 1:
                    pci = pcs["logits"]  # weight gradient lands in model._gwc and is split back after the bucket's slab reduce
                    iname = "logits"
                skip = c[lvl - 1]
                gi = g_ipre[lvl]
                self._wgrad(pci, [ops.nhwc_src(d[lvl]), ops.nhwc_src(skip)], hh, ww, gi, name=iname)
                g_dpre = self._act(f"g_d{lvl}", hh, ww, dch)
                self._dgrad(pci, 0, gi, g_dpre, hh, ww, actout=d[lvl], name=iname + "[deconv]")
                self._dgrad(pci, 1, gi, gc[lvl - 1], hh, ww, name=iname + "[skip]")  # first writer of g_c{lvl-1}
                self._wgrad(pcd, [ops.nhwc_src(below)], hh, ww, g_dpre, name=dname, up2=(2 * below.shape[1], 2 * below.shape[2]) == (hh, ww))
                below_ch = below.shape[3]
                if (bh, bw) == (hh, ww):  # degenerate: no resize
                    tgt = gc[6] if lvl == 6 else self._act(f"g_i{lvl + 1}", bh, bw, below_ch)
                    self._dgrad(pcd, 0, g_dpre, tgt, hh, ww, actout=below, name=dname)
                else:
                    tgt = gc[6] if lvl == 6 else self._act(f"g_i{lvl + 1}", bh, bw, below_ch)
                    fused = None
                    if (2 * bh, 2 * bw) == (hh, ww) and L.ab("FALNET_FUSED_UPSUM", "1") == "1":
                        try:  # exact 2x: the 2x2 block sum and elu'(below) ride in the data-gradient epilogue (no full-res g_up)
                            at = len(self.bwd_body)
                            self._dgrad(pcd, 0, g_dpre, None, hh, ww, name=dname, sum2x2_into=tgt, sum2x2_actout=below)
                            fused, self.bwd_body = self.bwd_body[at:], self.bwd_body[:at]
                        except ValueError:
                            pass
                    # two launches (plain data gradient, then the adjoint of the upsampling): on the small deep maps the plain data
                    # gradient has kernels the fused epilogue does not (variant 19 on 8 x 16 maps: deconv6 38 us fused on the halo-patch kernel)
                    plain = None
                    if fused is None or hh * ww <= 128:
                        at = len(self.bwd_body)
                        g_up = self._act(f"g_up{lvl}", hh, ww, below_ch)
                        self._dgrad(pcd, 0, g_dpre, g_up, hh, ww, name=dname)
                        self.bwd_body.append(ops.simple_call("falnet_upsample_bwd", L.ptr(g_up), L.ptr(tgt), L.ptr(below), B, hh,
                                                             ww, bh, bw, below_ch, code))
                        plain, self.bwd_body = self.bwd_body[at:], self.bwd_body[:at]
                    # the same gradient on the LOW-resolution grid: the taps that meet one upstream pixel summed beforehand (falnet_conv2d variant 26,
                    # 16 instead of 36 tap-MACs per position; maps of at least 16 x 32, 16-bit types)
                    lowres = None
                    if (2 * bh, 2 * bw) == (hh, ww) and pcd.wdd is not None:
                        try:
                            lowres = [ops.deconv_dgrad_call(self.dtype, g_dpre, pcd, B, tgt, below, name=dname, ws_owner=("falnet", id(self)),
                                                            flops=2 * B * hh * ww * pcd.cout * pcd.cin * 9)]
                        except ValueError:
                            pass
                    seqs = [q for q in (fused, plain, lowres) if q is not None]
                    if len(seqs) > 1:
                        runs = [lambda q=q: [c() for c in q] for q in seqs]
                        tags = "".join("f" if q is fused else ("p" if q is plain else "l") for q in seqs)
                        pick = ops.best_of(*runs, key=f"upsum3|{tags}|t{code}|B{B}|{hh}x{ww}|{pcd.cin_pad}>{pcd.cout_pad}")
                        self.bwd_body.extend(seqs[runs.index(pick)])
                    else:
                        self.bwd_body.extend(seqs[0])
                if lvl < 6:
                    g_ipre[lvl + 1] = tgt
            # gc[6] now holds g_z6 (pre-activation grad of the last residual block output)
            self._finish.append((0, len(self.bwd_body)))
            # encoder, bottom (level 6) to top (level 0)
            for i in range(6, -1, -1):
                if i == 3:
                    self._finish.append((1, len(self.bwd_body)))
                if i == _TAIL_LEVELS - 1:
                    self._finish.append((2, len(self.bwd_body)))
                self._bucket = 1 if i >= 4 else (2 if i >= _TAIL_LEVELS else 3)
                tail = i == 0 and L.ab("FALNET_TAIL_BALANCE", "1") == "1"
                # more weight gradients for the main stream's idle tail (names, comma separated): the side stream is the longer chain
                extra = L.ab("FALNET_TAIL_MAIN", _TAIL_MAIN).split(",") if L.ab("FALNET_TAIL_BALANCE", "1") == "1" else []
                cname, rname, ch = self._enc[i]
                hh, ww = sizes[i]
                gz = gc[i]
                pr1, pr2, pcc = pcs[rname + ".conv1"], pcs[rname + ".conv2"], pcs[cname]
                self._wgrad(pr2, [ops.nhwc_src(h_[i])], hh, ww, gz, name=rname + ".conv2", on_main=(rname + ".conv2") in extra)
                g_h = self._act(f"g_h{i}", hh, ww, ch)
                self._dgrad(pr2, 0, gz, g_h, hh, ww, actout=h_[i], name=rname + ".conv2")
                self._wgrad(pr1, [ops.nhwc_src(a[i])], hh, ww, g_h, name=rname + ".conv1", on_main=tail or (rname + ".conv1") in extra)
                g_a = self._act(f"g_a{i}", hh, ww, ch)
                self._dgrad(pr1, 0, g_h, g_a, hh, ww, addend=gz, actout=a[i], name=rname + ".conv1")
                srcs, ih, iw = self._enc_srcs[i]
                if tail:
                    self._main_tail = getattr(self, "_main_tail", [])
                    self._main_tail.append(lambda: L.ev_wait(torch.cuda.current_stream(), self._x0_event))
                self._wgrad(pcc, srcs, ih, iw, g_a, name=cname, on_main=tail or cname in extra)
                if i > 0:  # data gradient into the previous level's output (already holds the skip contribution)
                    self._dgrad(pcc, 0, g_a, gc[i - 1], ih, iw, addend=gc[i - 1], actout=c[i - 1], name=cname)
            # hand the side / third streams their pending weight gradients BEFORE the main stream's own tail launches: the hand-over event is
            # recorded behind whatever the main stream was given last, and behind the tail it would hold them back by ~130 us (traced step)
            if L.ab("FALNET_TAIL_FLUSH", "1") == "1":
                self.bwd_body.append(lambda: (self._flush_side(), self._deep_stream is not None and self._flush_deep()))
            self.bwd_body.extend(getattr(self, "_main_tail", []))
            self._finish.append((3, len(self.bwd_body)))
            # per bucket: one batched slab reduce + one batched bias-gradient launch after the bucket's last wgrad, then the
            # trainer's hook (asynchronous all-reduce of that range of the flat gradient buffer)
            finals = self.wbatch.finalize()
            body, self.bwd_body = self.bwd_body, []
            pos = 0
            for bucket, at in self._finish:
                self.bwd_body.extend(body[pos:at])
                pos = at
                if bucket in finals:
                    red, bias = finals[bucket]

                    if bucket == 3 and L.ab("FALNET_TAIL_BALANCE", "1") == "1":
                        # last bucket: the bias gradients only need the data gradients -> main stream (idle by now), beside the
                        # side stream's last weight gradients and slab reduce; run_backward fires the bucket hook after the join
                        self.bwd_body.append(bias)

                        def last_reduce(red=red):
                            self._join_deep(torch.cuda.current_stream())
                            red()
                        last_reduce.needs_torch_stream = True
                        self.bwd_body.append(lambda: self._deep_stream is not None and self._flush_deep())  # (from the main stream's context)
                        self._side_call(last_reduce)
                        self._deferred_ready = bucket
                        continue

                    def finish(red=red, bias=bias, bucket=bucket):
                        self._join_deep(torch.cuda.current_stream())  # (the side stream: needs_torch_stream below)
                        red()
                        bias()
                        if bucket == 0 and getattr(self.model, "_compose_logits", False):
                            self._split_logits_grad()
                        self.model._bucket_ready(bucket)
                    finish.needs_torch_stream = True
                    self.bwd_body.append(lambda: self._deep_stream is not None and self._flush_deep())  # (from the main stream's context)
                    self._side_call(finish)
            self.bwd_body.extend(body[pos:])

        self._backward_builder = build_backward

    def _ensure_backward(self):
        if self._backward_builder is not None:
            builder, self._backward_builder = self._backward_builder, None
            builder()

    def _split_logits_grad(self):
        """dWc (composed 3x3 logits conv) -> dW3x3 = W1x1^T dWc and dW1x1 = dWc W3x3^T, added into the flat gradient buffer."""
        m, lib = self.model, L.lib()
        w1, w3 = m.conv0.weight.detach(), m._bb.iconv1.weight.detach()
        n, k = w1.shape[0], w3.numel() // w3.shape[0]  # N planes, 9 * 96
        g, g3, g1 = m._gwc, m._grad_view(m._bb.iconv1.weight), m._grad_view(m.conv0.weight)
        st = L.stream_ptr()
        # dW3x3[n x k] += W1x1^T[n x n] dWc[n x k]   (A = W1x1 read transposed: strides (1, n))
        L.check(lib.falnet_gemm_f32_small(L.ptr(w1), 1, n, L.ptr(g), k, 1, L.ptr(g3), n, k, n, 1, st), "split logits grad (3x3)")
        # dW1x1[n x n] += dWc[n x k] W3x3^T[k x n]   (B = W3x3 read transposed: strides (1, k))
        L.check(lib.falnet_gemm_f32_small(L.ptr(g), k, 1, L.ptr(w3), 1, k, L.ptr(g1), n, n, k, 1, st), "split logits grad (1x1)")

    # ---- execution ----
    def run_forward(self, left, min_disp, max_disp, ret_disp, ret_subocc, ret_pan, repack=True, min_from=None):
        """min_disp=None: the plan's min_disp is max_disp * min_from[0] / min_from[1] (Train_Stage1_K.py:237), made by the same launch."""
        b = self.buf
        self.generation += 1
        b["left"].copy_(left)
        mx = max_disp.reshape(-1)
        mn = None if min_disp is None else min_disp.reshape(-1)
        assert mx.is_contiguous() and mx.dtype == torch.float32 and mx.numel() == self.B and (mn is None or (mn.is_contiguous() and mn.dtype == torch.float32))
        mul, div = min_from if mn is None else (0.0, 1.0)
        flow = b["flow"]  # (B, Cpad) in the compute dtype, channel 0 = max_disp / 100 (FAL_netB.py:208-209)
        L.check(L.lib().falnet_disp_prologue(L.ptr(mx), L.ptr(mn), float(mul), float(div), L.ptr(b["min_disp"]), L.ptr(b["max_disp"]),
                                             L.ptr(flow), flow.stride(0), self.B, L.dtype_code(self.dtype), L.stream_ptr()), "disp_prologue")
        if repack and not self.model._packed_is_fresh():
            for call in self.pack:
                call()
        hook, mid = getattr(self.model, "_mid_forward_hook", None), getattr(self, "_mid_index", -1)
        with L.stream_scope():  # one stream lookup for the whole replay
            if ops.replay_ok():
                # host launch path in C: the launches in front of the mid-forward hook and those behind it (with the MED head) as two
                # recorded segments (csrc/replay.cpp) -- after two eager passes of this output combination
                fk = (bool(ret_pan or ret_subocc), bool(ret_subocc))
                segs = self._fwd_segments.get(fk)
                fire = hook is not None and 0 <= mid < len(self.fwd)
                if segs is None and self._fwd_eager_runs.get(fk, 0) >= 2:
                    m = mid if 0 <= mid < len(self.fwd) else 0
                    tail = list(self.fwd[m:]) + [self.head_full if fk[0] else self.head_disp_only] + ([self.head_masks] if fk[1] else [])
                    segs = self._fwd_segments[fk] = (L.record_calls(self.fwd[:m]), L.record_calls(tail))
                if segs is not None:
                    ptr = L.stream_ptr().value
                    segs[0].run(ptr)
                    if fire:
                        self.model._mid_forward_hook = None  # one shot
                        hook()
                    segs[1].run(ptr)
                    return self.generation
                self._fwd_eager_runs[fk] = self._fwd_eager_runs.get(fk, 0) + 1
            for i, call in enumerate(self.fwd):
                if hook is not None and i == mid:
                    self.model._mid_forward_hook = None  # one shot
                    hook()  # (launches on its own stream: train.vgg_label_async redirects with L.on_stream)
                call()
            if ret_pan or ret_subocc:
                self.head_full()
            else:
                self.head_disp_only()
            if ret_subocc:
                self.head_masks()
        return self.generation

    def run_backward(self, g_disp, g_pan, in_place=False):
        """in_place: the caller has written the upstream gradients into buf["g_disp"] / buf["g_pan"] itself (the fused step)."""
        self._ensure_backward()
        b = self.buf
        if g_disp is not None and not in_place:
            b["g_disp"].copy_(g_disp)
        if g_pan is not None and not in_place:
            b["g_pan"].copy_(g_pan)
        self._accumulate = self.model._begin_grad_accumulation()
        self.wbatch.accumulate = 1 if self._accumulate else 0
        self.model._accumulating = bool(self._accumulate)
        main = torch.cuda.current_stream()
        use_side = self.use_side_stream and ops.TIMER is None
        # Host launch path in C (csrc/replay.cpp): the body below is a static sequence of C-ABI launches, event records and stream waits per
        # (which upstream gradients exist, accumulating or not); after two eager passes it is recorded once and every later backward is ONE
        # falnet_replay call -- the same launches on the same three streams.  Not with a gradient-bucket hook installed (torch.distributed
        # collectives are issued from Python between the buckets) and not inside bench.py's instrumented pass.
        hooked = getattr(self.model, "bucket_hook", None) is not None and not self._accumulate
        if use_side:
            # once per (caller's stream, hook state) and device; the hooked form normally ran already at train.enable_overlapped_allreduce
            # (an explicit collective point: its probes are collectives) -- here only for a hook installed by hand
            self._streams.ensure_tested(main, hooked)
        if self._streams_gen != self._streams.generation:  # a stream was re-created (by any plan's test): recorded sequences hold the old handles
            self.invalidate_segments()
            self._streams_gen = self._streams.generation
        key = (g_disp is not None, g_pan is not None, bool(self._accumulate), hooked)
        can = use_side and ops.replay_ok() and (not hooked or L.ab("FALNET_REPLAY_HOOKED", "1") == "1")
        seg = self._bwd_segments.get(key) if can else None
        if seg is None and can and self._bwd_eager_runs.get(key, 0) >= 2:
            try:
                with L.Recorder(main.cuda_stream) as rec:
                    self._backward_body(main, use_side, key)
            except BaseException:
                self._reset_handover_state()  # a recording that raised mid-body must not leave half-advanced hand-over lists for the next eager pass
                raise
            seg = self._bwd_segments[key] = rec.finalize()
        if seg is not None:
            try:
                seg.run(main.cuda_stream)
            except BaseException:
                # a failed replay leaves the side / third streams without their join waits: drain the device, drop the sequence, re-raise
                torch.cuda.synchronize(self.device)
                self._bwd_segments.pop(key, None)
                self._reset_handover_state()
                raise
        else:
            self._backward_body(main, use_side, key)
            if can:
                self._bwd_eager_runs[key] = self._bwd_eager_runs.get(key, 0) + 1
        if getattr(self, "_deferred_ready", None) is not None:
            self.model._bucket_ready(self._deferred_ready)
        self.model._end_grad_accumulation()

    def _backward_body(self, main, use_side, key):
        lib = L.lib()
        self._main_stream = main
        with L.stream_scope(main):  # one stream lookup for the whole replay; side calls redirect their launches with L.on_stream
            st = L.stream_ptr()
            if not self._accumulate:  # the batched reduce / bias kernels ADD into the flat gradient buffer
                fg = self.model._flat_grad
                L.check(lib.falnet_fill_f32(L.ptr(fg), fg.numel(), 0.0, st), "fill flat gradient")
            if getattr(self.model, "_compose_logits", False):  # per-backward scratch: its content is split into the two real gradients
                gw = self.model._gwc
                L.check(lib.falnet_fill_f32(L.ptr(gw), gw.numel(), 0.0, st), "fill composed-logits gradient")
            self._side_ev_next = self._deep_ev_next = self._sync_ev_next = 0
            self._deep_dirty = False
            if use_side:
                self._side_stream = self._streams.side
                self._stream_wait(self._side_stream, main)  # the previous step's Adam / repack must not be overtaken
                # with a gradient-bucket hook (N > 1: torch.distributed collectives fired from the side stream) the third stream is OFF: on a
                # world-size-1 RCCL group the step measured 8.4 ms with it against 5.5 ms without (exposed communication 3.0 vs 0.0 ms with 8 or 16
                # hardware queues, none with 4: profiles/r04_ab_dist_third_stream.txt) -- the collective's stream, the side stream it waits on
                # and the third stream the side stream waits on serialise against the main stream's data gradients
                if self._deep_max_px > 0 and (not key[3] or self._streams.third_with_hook):
                    self._deep_stream = self._streams.third
                    self._stream_wait(self._deep_stream, main)
                else:
                    self._deep_stream = None
            else:
                self._side_stream = self._deep_stream = None
            self.head_bwd[(key[0], key[1])]()
            for call in self.bwd_body:
                call()
            self._flush_side()
            self._join_deep(main)
            if self._side_stream is not None:
                self._stream_wait(main, self._side_stream)

    def _reset_handover_state(self):
        self._side_pending.clear()
        self._deep_pending.clear()
        self._side_ev_next = self._deep_ev_next = self._sync_ev_next = 0
        self._deep_dirty = False

    def invalidate_segments(self):
        """Drop every recorded launch sequence of this plan (they hold raw stream / event handles and by-value state): call it when a stream is
        re-created, a descriptor's buffers change, or a bucket hook is installed / removed."""
        self._bwd_segments.clear()
        self._fwd_segments.clear()
        self._bwd_eager_runs.clear()
        self._fwd_eager_runs.clear()
        self._reset_handover_state()

    @property
    def selftest(self):
        """Record of the device's stream self-test (StepStreams.selftest); None = not run yet."""
        return self._streams.result

    def stream_selftest(self, main, hooked):
        return self._streams.selftest(main, hooked)

    def _stream_wait(self, waiter, waited):
        """waiter.wait_stream(waited) as an explicit event pair (recordable)."""
        ev = self._event(self._sync_events, self._sync_ev_next)
        self._sync_ev_next += 1
        L.ev_record(ev, waited)
        L.ev_wait(waiter, ev)
