#!/usr/bin/env python3
"""Benchmark of the hot path named by BASELINE.json: one Stage-1 training step of FAL_netB
(Train_Stage1_K.py:233-262: model fwd -> VGG(right) -> L1+perceptual -> smoothness -> backward ->
gradient all-reduce -> Adam) at 256x512, N=49, batch 8 per GPU, synthetic data, seeded weights.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = stereo pairs/s over all ranks with inputs resident in HBM.
`roofline` = the dominant kernel's achieved rate (algorithmic FLOPs / HIP-event time, from an instrumented
pass after the timed region).  `cpu_baseline` = the CPU oracle (oracle/falnet_oracle.py: a torch-CPU
restatement of the same step) timed on this host on a bounded sample -- the only use of oracle/ here.
"""
import argparse
import json
import os
import sys
import time

# The step keeps four HIP streams busy (plus a collective's when N > 1): five hardware queues, and the plan's stream self-test re-creates any
# stream that shares a queue with another (fal_net_amd/__init__.py has the measurements: with 8 queues the data-parallel step falls off a cliff,
# 9.1 vs 5.3 ms).  An application-level, process-global choice that must be made before HIP initialises -- so it is made HERE (and in the
# Train_* / Test_KITTI scripts), not by importing the library.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


class EventTimer:
    """ops.TIMER hook: brackets every C-ABI launch with HIP events on the launch stream."""

    def __init__(self):
        self.records = []

    def run(self, tag, flops, nbytes, launch, name=""):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        self.records.append((tag, flops, nbytes, e0, e1, name))

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for tag, flops, nbytes, e0, e1, _ in self.records:
            a = agg.setdefault(tag, {"ms": 0.0, "flops": 0, "bytes": 0, "launches": 0})
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += flops
            a["bytes"] += nbytes
            a["launches"] += 1
        return agg


def _family_matcher(symbols):
    """row-name predicate for rocprofv3 CSVs: the row is one of `symbols` (mangled, as the library reports them) -- or the demangled spelling of
    one of them, which rocprofv3 prints for some template instantiations ("void conv3x3_dma16_kernel<bool _Accum, bool, E, 16, 8, false>(...)"):
    same template name and the same integer parameters in order."""
    import re
    symbols = [symbols] if isinstance(symbols, str) else list(symbols)
    keys = []
    for s in symbols:
        m = re.match(r"_Z\d+([A-Za-z_0-9]+?)I", s)
        keys.append((m.group(1) if m else s, re.findall(r"Li(\d+)E", s)))

    def match(name):
        head = name.split("(")[0]
        if any(head[:100] == s[:100] for s in symbols):
            return True
        for base, ints in keys:
            if base + "<" in name:
                got = re.findall(r"[<,] ?(\d+)(?=[,>])", name.split(">(")[0] + ">")
                if got == ints:
                    return True
        return False
    return match


def hbm_traffic_from_profiles(kernel_symbol):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/*_hbm_traffic.json, made by
    tools/profile_round.sh: separate FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950 correction).
    None when no profile of this kernel is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json")))
    if not files:
        return None
    table = json.load(open(files[-1]))
    meta = table.pop("_meta", {})
    match = _family_matcher(kernel_symbol)
    fetch = write = 0.0
    n = 0
    for name, v in table.items():
        if match(name):
            fetch += v["fetch_bytes_per_launch_corrected"] * v["launches"]
            write += v["write_bytes_per_launch"] * v["launches"]
            n += v["launches"]
    if n:
        return {"bytes_per_launch": (fetch + write) / n, "fetch_bytes_corrected": fetch / n, "write_bytes": write / n,
                "source": os.path.basename(files[-1]), "collected_at_commit": meta.get("commit", "unknown")}
    return None


def hbm_traffic_live(kernel_symbol, args):
    """HBM bytes per launch of the dominant kernel measured in THIS run on THIS box: two short child runs of this script under
    `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, counters only -- no trace domains), exactly as
    tools/profile_round.sh / MI355X_MICROARCH.md prescribe (FETCH_SIZE doubled: gfx950 reports half the bytes of wide coalesced
    reads).  Returns None when rocprofv3 is unavailable or a pass fails (the caller then falls back to the committed profile)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None
    sums = {}
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--no-roofline",
             "--workload", args.workload, "--dtype", args.dtype, "--batch", str(args.batch),
             "--height", str(args.height), "--width", str(args.width), "--levels", str(args.levels)]
    # the child is a stand-alone single-GPU run: it must not inherit this process's rendezvous (under torchrun, or with
    # FALNET_FORCE_DIST=1, it would call init_process_group as rank 0 of the PARENT's world and wait for peers that never come)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "ROLE_WORLD_SIZE",
                        "GROUP_WORLD_SIZE", "FALNET_FORCE_DIST", "FALNET_DIST_BACKEND")
           and not k.startswith(("MASTER_", "TORCHELASTIC_", "TORCH_NCCL_", "NCCL_ASYNC"))}
    env["TMPDIR"] = "/tmp"
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="falnet_pmc_", dir="/tmp")
        try:
            r = subprocess.run([rp, "--pmc", counter, "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp",
                               env=env, capture_output=True, text=True, timeout=600)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None
            tot, n = 0.0, 0
            match = _family_matcher(kernel_symbol)
            for row in csv.DictReader(open(files[0])):
                if row["Counter_Name"] == counter and match(row["Kernel_Name"]):
                    tot += float(row["Counter_Value"])
                    n += 1
            if n == 0:
                return None
            sums[counter] = (tot * 1024.0 / n, n)  # rocprofv3 reports KiB
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    fetch, write = 2.0 * sums["FETCH_SIZE"][0], sums["WRITE_SIZE"][0]
    return {"bytes_per_launch": fetch + write, "fetch_bytes_corrected": fetch, "write_bytes": write, "launches_sampled": sums["FETCH_SIZE"][1],
            "source": "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this bench run (FETCH_SIZE x2, gfx950)"}


def sustained_mfma(dtype, dev):
    """What the board sustains on the matrix pipe, measured here: falnet_mfma_probe (register-resident v_mfma_f32_16x16x32 loop, two waves per SIMD on
    every CU, no LDS / memory traffic) after ~0.25 s of back-to-back launches, on random operands (real switching activity: the board's 1 400 W cap
    sets the clock) and on zeros (the clock-bound rate).  Context for `roofline.frac`, whose `peak` stays the guide's dense figure."""
    from fal_net_amd import _lib as L
    dt = dtype if dtype in (torch.bfloat16, torch.float16) else torch.bfloat16
    out = torch.zeros(4, device=dev)
    iters = 1500
    flops = 2048 * 4 * iters * 16 * 16384.0
    res = {}
    for name, scale in (("random", 1.0), ("zeros", 0.0)):
        ab = (torch.randn(64 * 8 * 64 * 8, device=dev) * scale).to(dt)
        def launch():
            L.check(L.lib().falnet_mfma_probe(L.ptr(ab), L.ptr(out), iters, L.dtype_code(dt), L.stream_ptr()), "mfma_probe")
        for _ in range(90):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            launch()
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(flops * 40 / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
    return {"tflops_random_operands": res["random"], "tflops_zero_operands": res["zeros"], "unit": "TFLOP/s",
            "how": "falnet_mfma_probe: register-resident v_mfma_f32_16x16x32 loop, 2 waves/SIMD on all CUs, 40 launches timed behind 90 (0.25 s) of the same"}


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, height, width, levels, sample_batch=8, timed=3):
    """Reported CPU baseline: the oracle's step of the same workload (fp32, torch CPU) at the benchmark's own batch (B = 8,
    BASELINE.md section 4: >= 3 timed steps) on a bounded sample: one warm-up step, then `timed` steps (about 40 s on the GPU box's host), plus the
    single-pair forward of BASELINE configs[0].  Threads: the best of 16 / 32 / 64 / 128 (one step each, `seconds_per_step_by_threads`) for the
    default workload, 32 otherwise -- on the 256-thread GPU-box host torch's CPU convs get *slower* with every thread (measured: 235 s for B=2
    with 256 threads); `cores` is the thread count the reported sample actually ran with."""
    from fal_net_amd import synthetic
    from oracle import falnet_oracle as O
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    sweep = None
    if height * width > 256 * 512:
        sample_batch, timed = 1, 2  # 384x1280, N=96: one pair per step keeps the leg under a minute
    elif workload == "stage2":
        sample_batch = 4  # the Stage-2 step runs three 2B forwards: B=4 keeps the leg at about 30 s
    left, right, mn, mx = synthetic.synthetic_pair(sample_batch, height, width, seed=1234)
    sd = synthetic.seeded_falnetb_state_dict(levels)
    params = O.leaf_params(sd)
    vsd = synthetic.seeded_vgg19_state_dict()
    opt = O.OracleAdam(params)

    def step():
        if workload == "stage2":
            for q in params.values():
                q.grad = None
            O.stage2_losses(params, sd, vsd, left, right, mn, mx)["loss"].backward()
            opt.step()
        else:
            O.stage1_step(params, opt, vsd, left, right, mn, mx)
    step()  # warm-up (allocator, thread pool)
    if workload == "stage1" and height * width <= 256 * 512:
        # Which thread count IS the host's CPU path?  torch's CPU convolutions are not monotone in threads on the GPU box's 256-thread host
        # (VERDICT r4 #9), so one step is timed at 16 / 32 / 64 / 128 threads (outside the reported sample) and the sample runs at the best.
        sweep = {}
        for n in (16, 32, 64, 128):
            if n > (os.cpu_count() or 1):
                continue
            torch.set_num_threads(n)
            ts = time.time()
            step()
            sweep[str(n)] = round(time.time() - ts, 3)
        cores = int(min(sweep, key=sweep.get)) if sweep else cores
        torch.set_num_threads(cores)
    t0 = time.time()
    for _ in range(timed):
        step()
    dt = (time.time() - t0) / timed
    one = synthetic.synthetic_pair(1, 256, 512, seed=99)
    sd49 = sd if levels == 49 else synthetic.seeded_falnetb_state_dict(49)
    with torch.no_grad():
        O.falnet_forward(sd49, one[0], one[2], one[3])
        t1 = time.time()
        for _ in range(3):
            O.falnet_forward(sd49, one[0], one[2], one[3])
        fwd = (time.time() - t1) / 3
    name = {"stage1": "Stage-1 step (fwd+VGG+losses+bwd+Adam)", "stage2": "Stage-2 step (teacher fwd, 2B student fwd, masks, losses, bwd, Adam)",
            "highres": "Stage-1 step (fwd+VGG+losses+bwd+Adam)"}[workload]
    return {"value": sample_batch / dt, "unit": "stereo-pairs/s", "cores": cores, "kind": "port", "host_cpu": host_cpu_model(),
            "host_logical_cpus": os.cpu_count(), "seconds_per_step_by_threads": sweep,
            "sample": f"1 warm-up + {timed} timed x {name}, B={sample_batch}, {height}x{width}, N={levels}, "
                      f"fp32 torch-CPU oracle, {torch.get_num_threads()} threads, {dt:.2f} s per step",
            "configs0_forward_pairs_per_s": 1.0 / fwd,
            "configs0_sample": f"FAL_netB forward, one 256x512 pair, N=49, fp32 torch-CPU oracle, {fwd * 1e3:.0f} ms (mean of 3 after 1 warm-up)"}


def parity_vs_oracle(height, width, levels, dev):
    """SURVEY 8(d) `abs_rel vs ref`: one seeded pair through the CPU oracle (fixture-pinned restatement of the reference) and
    through the HIP path in f32 (gate 1e-4), bf16 and f16 (reported); depth = f*b/disp, abs_rel = mean(|d_ref - d_hip| / d_ref)
    (myUtils.py:225).  Part of the cpu_baseline leg: the only place bench.py touches oracle/."""
    import numpy as np
    from fal_net_amd import synthetic
    from fal_net_amd.models import FAL_netB
    from oracle import falnet_oracle as O
    left, right, mn, mx = synthetic.synthetic_pair(1, height, width, seed=4321)
    sd = synthetic.seeded_falnetb_state_dict(levels)
    with torch.no_grad():
        ref = O.falnet_forward(sd, left, mn, mx)
    d_ref = O.disp_to_depth(ref.numpy())
    out = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16), ("f16", torch.float16)):
        m = FAL_netB({"state_dict": sd}, no_levels=levels, compute_dtype=dt).to(dev).eval()
        with torch.no_grad():
            disp = m(left.to(dev), mn.to(dev), mx.to(dev)).float().cpu()
        d = O.disp_to_depth(disp.numpy())
        out[name] = {"abs_rel_depth": float(np.mean(np.abs(d_ref - d) / d_ref)),
                     "disp_max_rel": float((disp - ref).abs().max() / ref.abs().max())}
        del m
    out["gate_f32"] = 1e-4
    out["sample"] = f"1 pair, {height}x{width}, N={levels}, seeded weights, forward (disp) vs CPU oracle"
    return out


def abs_rel_vs_ground_truth(args, steps=200):
    """`abs_rel` where a ground truth exists: `steps` Stage-1 steps at the benchmark's own batch and size on STRUCTURED synthetic stereo
    (fal_net_amd.synthetic.structured_stereo: right view = left view displaced by a smooth known disparity; pool of 4 batches), from the
    seeded weights, per compute dtype; depth abs_rel (myUtils.py:225) of the trained model's disparity against the known one.  After the
    timed region; tools/trajectory.py is the long form (profiles/r04_trajectory_*.json)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("trajectory", os.path.join(ROOT, "tools", "trajectory.py"))
    traj = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(traj)
    r = traj.run(steps=steps, height=args.height, width=args.width, batch=args.batch, pool=4, levels=args.levels, dtypes=("f32", "bf16", "f16"))
    out = {k: {"abs_rel_vs_gt": r[k]["abs_rel_vs_gt"], "abs_rel_vs_gt_heldout": r[k]["abs_rel_vs_gt_heldout"], "loss_last": r[k]["loss_last"]} for k in ("f32", "bf16", "f16")}
    out["at_seeded_weights"] = r["f32"]["abs_rel_vs_gt_start"]
    out["sample"] = (f"{steps} Stage-1 steps, B={args.batch}, {args.height}x{args.width}, N={args.levels}, pool of 4 structured synthetic batches with known "
                     "disparity, Adam lr 1e-4; depth abs_rel of the trained model vs ground truth (training pool / 2 held-out batches of 2)")
    return out


def allreduce_report(model, step, args, world, rank, dev, ms_per_step):
    """The step's single collective, measured after the timed region (every rank takes part; rank 0 reports):
      * `param_checksum_equal`: the flat parameter buffers of all ranks are still bit-identical after the timed steps (a racing
        bucket of the overlapped all-reduce would make them drift) -- asserted on every rank;
      * `isolated_ms`: each gradient bucket's all-reduce and the whole 67.7 MB buffer alone (barrier, HIP events around the
        collective on the current stream, median of 5), with the achieved bus bandwidth 2 (N-1)/N x bytes / time against the xGMI
        bounds of BASELINE.md section 3 (ring, per-link 153 GB/s: 0.78 ms at N = 8; all seven links direct: 0.11 ms);
      * `exposed_ms` = step time - step time of the same steps with the collective SKIPPED (same box, same kernels): what the
        pipelined buckets do not hide behind backward."""
    import torch.distributed as dist
    from fal_net_amd import train
    flat, grad = model.flat_parameters(), model.flat_gradients()
    d = flat.double()
    mine = torch.stack([d.sum(), (d * d).sum(), d.abs().max()])
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(torch.equal(lo, hi))
    assert same, f"rank {rank}: parameters differ across ranks after the timed steps (checksum {mine.tolist()} not in [{lo.tolist()}, {hi.tolist()}])"
    seen = torch.zeros(world, device=dev)
    seen[rank] = 1
    dist.all_reduce(seen)

    def timed_allreduce(t):
        ts = []
        for _ in range(6):
            dist.barrier()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts[1:])[len(ts[1:]) // 2]
    scratch = torch.zeros_like(grad)
    buckets = model.gradient_buckets()
    iso = {"whole_buffer": timed_allreduce(scratch)}
    for i, (a, b) in enumerate(buckets):
        iso[f"bucket{i}"] = timed_allreduce(scratch[a:b])
    nbytes = grad.numel() * 4
    busbw = lambda ms_, nb: 2 * (world - 1) / max(world, 1) * nb / (ms_ * 1e-3) / 1e9 if ms_ > 0 else 0.0
    # the same steps without the collective (weights diverge across ranks from here on: nothing after this compares them)
    train._SKIP_ALLREDUCE = True
    try:
        for _ in range(2):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    finally:
        train._SKIP_ALLREDUCE = False
    ms_nocomm = float(t) * 1e3 / args.steps
    ring_ms = 2 * (world - 1) / max(world, 1) * nbytes / 153e9 * 1e3
    direct_ms = ring_ms / 7.0
    return {"bytes": nbytes, "buckets_bytes": [(b - a) * 4 for a, b in buckets], "overlapped": getattr(model, "bucket_hook", None) is not None,
            "ranks_seen": int(seen.sum().item()), "param_checksum_equal": same,
            "isolated_ms": iso, "isolated_busbw_GBps": {k: busbw(v, nbytes if k == "whole_buffer" else (buckets[int(k[6:])][1] - buckets[int(k[6:])][0]) * 4)
                                                        for k, v in iso.items()},
            "ms_per_step_without_collective": ms_nocomm, "exposed_ms": ms_per_step - ms_nocomm,
            "xgmi_bound_ms": {"ring_per_link_153GBps": ring_ms, "all_7_links_direct": direct_ms},
            "stream_selftest": next((p.selftest for p in getattr(model, "_plans", {}).values() if getattr(p, "selftest", None)), None),
            "backward_issue": "recorded sequence cut at the bucket hooks (falnet_replay + Python collectives)" if os.environ.get("FALNET_REPLAY", "1") == "1" else "eager",
            # where the buckets' all-reduces run (train.enable_overlapped_allreduce): as sync collectives on the step's auxiliary stream there is no
            # stream of the backend's own on the data path, whatever RCCL opens internally -- the step keeps four busy hardware queues at any N
            "collective_stream": ("auxiliary stream of the step (sync collectives, torch >= 2.8: issued on the caller's current stream)"
                                  if (train._COMM_ON_AUX and dist.get_backend() == "nccl") else "the backend's own stream (async collectives)"),
            "busy_hw_queues_of_a_step": 4 if (train._COMM_ON_AUX and dist.get_backend() == "nccl") else 5,
            "torch_version": torch.__version__,
            "isolated_vs_ring_bound": (ring_ms / iso["whole_buffer"]) if iso["whole_buffer"] > 0 else None}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port <free> bench.py <same arguments>` as a child process and return its exit code.  Lines are relayed as they come; a JSON result line
    is held back and printed LAST (the driver reads the last line of stdout)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool: RCCL needs it across processes
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {n} without WORLD_SIZE: starting {n} ranks as child processes: {' '.join(cmd[1:9])} bench.py ...", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    line_json = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            line_json = line
        else:
            sys.stdout.write(line)
            sys.stdout.flush()
    rc = proc.wait()
    if line_json is not None:
        sys.stdout.write(line_json)
        sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)   # >= 10 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (BASELINE configs[1]: 8)")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--levels", type=int, default=49)
    ap.add_argument("--workload", default="stage1", choices=["stage1", "stage2", "highres"],
                    help="stage1: BASELINE configs[1] (default); stage2: configs[3] (Train_Stage2_K.py:233-331, teacher = copy of the "
                         "student); highres: configs[4] (Stage-1 step at 384x1280, N=96, f16 unless --dtype says otherwise)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-trajectory", action="store_true", help="skip the abs_rel_vs_gt leg (200 training steps per dtype on structured stereo, ~15 s)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="roofline.traffic from the committed profiles/ file instead of two live rocprofv3 --pmc child passes (~25 s each)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as one captured hipGraph (measured slower than eager launches on ROCm 7.0: off by default)")
    ap.add_argument("--launch-table", default=None, help="write the per-launch time table of the instrumented pass here")
    args = ap.parse_args()
    if args.workload == "highres":
        args.height, args.width, args.levels = 384, 1280, 96
    if args.dtype is None:
        args.dtype = "f16" if args.workload == "highres" else "bf16"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` typed as is: this process becomes the launcher.  It has not touched the GPU (no HIP call above this line) and
        # never will: the N ranks are CHILD processes of torch.distributed.run (never an exec), their output is relayed, rank 0's JSON line stays last.
        return self_launch(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (or leave WORLD_SIZE unset and let bench.py start them)")
    # FALNET_DIST_BACKEND=gloo (tests): the N > 1 path of this script on a box with fewer GPUs than ranks -- ranks then share devices
    # (RCCL itself refuses two ranks on one device and stays the backend of every real run)
    backend = os.environ.get("FALNET_DIST_BACKEND", "nccl")
    if os.environ.get("FALNET_BENCH_DRYRUN") == "1":
        # launch-path check (tests, CPU box): rendezvous + one collective over gloo, no GPU call, no measurement -- the line says so
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "dry run of the launch path (no measurement)", "value": None, "dry_run": True, "n_gpus": world, "ranks_seen": int(t.item())}), flush=True)
        return 0
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1 or os.environ.get("FALNET_FORCE_DIST") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # backend "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(backend)

    from fal_net_amd import loss_functions as LF
    from fal_net_amd import ops, synthetic, train
    from fal_net_amd.models import FAL_netB

    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    LF.set_compute_dtype(dtype)
    model = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(args.levels)}, no_levels=args.levels,
                     compute_dtype=dtype).to(dev).train()
    train.sync_parameters(model)  # world > 1: one broadcast of rank 0's flat parameters + checksum, before any timed step
    opt = train.FlatAdam(model, lr=1e-4, betas=(0.5, 0.999))
    left, right, mn, mx = synthetic.synthetic_pair(args.batch, args.height, args.width, seed=1234 + rank)
    left, right, mx = left.to(dev), right.to(dev), mx.to(dev)  # inputs resident in HBM before the timed region

    fix_model = None
    if args.workload == "stage2":  # frozen Stage-1 teacher (Train_Stage2_K.py:190-198): here a copy of the student's weights
        fix_model = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(args.levels)}, no_levels=args.levels,
                             compute_dtype=dtype).to(dev).eval()
        for q in fix_model.parameters():
            q.requires_grad_(False)

    def eager_step():
        if fix_model is not None:
            return train.stage2_step(model, fix_model, opt, left, right, mx)
        return train.stage1_step(model, opt, left, right, mx)

    step, graphed = eager_step, False
    if args.graph and args.workload != "stage2":
        try:
            step = train.GraphedStage1Step(model, opt, left, right, mx)
            graphed = True
        except Exception as e:  # capture is an optimisation, not a requirement: report and run eagerly
            if rank == 0:
                print(f"[bench] hipGraph capture unavailable ({type(e).__name__}: {e}); running eagerly", file=sys.stderr)
            torch.cuda.synchronize()
            step = eager_step

    for _ in range(args.warmup):
        out = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # one event per 10 steps on the step's stream (no synchronisation inside the timed loop): min / median / max of the 10-step windows show the
    # box's jitter in the line itself
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps // 10 + 1)] if not graphed else []
    t0 = time.perf_counter()
    for i in range(args.steps):
        if marks and i % 10 == 0:
            marks[i // 10].record()
        out = step()
    if marks and args.steps % 10 == 0:
        marks[-1].record()
    issued = time.perf_counter() - t0  # host time to ISSUE the K steps (no synchronisation inside the loop): the launch-bound share of a step
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    loss = float(out["loss"])
    # host time to ISSUE a step, measured where the queues cannot push back: three steps from an idle device (inside the timed loop a
    # host that runs ahead blocks on the hardware queues, and `issued` then measures the GPU, not the host)
    torch.cuda.synchronize()
    ti = time.perf_counter()
    for _ in range(3):
        step()
    issue_idle = (time.perf_counter() - ti) / 3
    torch.cuda.synchronize()
    ms = elapsed * 1e3 / args.steps
    pairs_per_s = world * args.batch * args.steps / elapsed
    nwin = args.steps // 10 if marks else 0
    wins = sorted(marks[k].elapsed_time(marks[k + 1]) / 10 for k in range(nwin if args.steps % 10 == 0 else max(nwin - 1, 0))) if nwin else []

    stage = "Stage-2" if args.workload == "stage2" else "Stage-1"
    cfg_name = {"stage1": "BASELINE configs[1]", "stage2": "BASELINE configs[3] on one GPU", "highres": "BASELINE configs[4] on one GPU"}[args.workload]
    result = {
        "metric": f"stereo-pairs/sec {stage} step @{args.height}x{args.width} N={args.levels}",
        "value": pairs_per_s, "unit": "stereo-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{stage} training step ({cfg_name}), batch {args.batch}/GPU, "
                               f"{args.height}x{args.width}, N={args.levels}, seeded weights, seeded VGG19",
                   "global_batch": world * args.batch, "parallelism": f"dp{world}", "final_loss": loss,
                   "launch": "hipGraph replay" if graphed else ("eager launches, recorded sequences issued by falnet_replay (C)" if os.environ.get("FALNET_REPLAY", "1") == "1"
                                                                       else "eager launches from Python"),
                   "host_issue_ms_per_step": issue_idle * 1e3, "host_issue_ms_per_step_in_timed_loop": issued * 1e3 / args.steps},
    }
    st_plan = next((p for p in getattr(model, "_plans", {}).values() if getattr(p, "selftest", None)), None)
    if st_plan is not None:  # stream / hardware-queue self-test of the step's streams (plan.stream_selftest)
        t = st_plan.selftest
        result["config"]["stream_selftest"] = {"pairs_overlap": t["pairs_overlap"], "streams_replaced": t["streams_replaced"], "hw_queues": t["hw_queues"],
                                               "worst_pair_ms": max(e["worst_pair_ms"] for e in t["streams"]), "collective_beside_spin_ms": t["collective_beside_spin_ms"],
                                               "all_streams_3x200us_ms": t.get("all_streams_3x200us_ms")}
    if wins:
        result["ms_per_step_10step_windows"] = {"min": round(wins[0], 4), "median": round(wins[len(wins) // 2], 4), "max": round(wins[-1], 4), "n": len(wins),
                                                "source": "HIP events on the step's stream every 10 steps inside the timed region (rank 0's device)"}

    if world > 1 or os.environ.get("FALNET_FORCE_DIST") == "1":
        result["allreduce"] = allreduce_report(model, eager_step, args, world, rank, dev, ms)
    if world > 1:
        # Everything below runs on rank 0 ONLY (phase marks, the instrumented pass): steps issued there must not issue collectives the other ranks
        # never join -- they would wait for their peers forever.  The timed region and the all-reduce report are done; from here on the bucket
        # hooks and allreduce_gradients skip the collective on every rank (the weights may diverge now: nothing below compares them).
        train._SKIP_ALLREDUCE = True
        dist.barrier()

    phase_ms = None
    if rank == 0 and not args.no_roofline and args.workload != "stage2" and not graphed:
        # phase boundaries of the REAL step (all streams live, replayed launch sequences): five HIP events per step on the step's stream, 10 steps,
        # NOT part of the timed region.  With the per-phase algorithmic FLOPs of the instrumented pass below this is `roofline.phase_tflops`:
        # where the step delivers its FLOPs and where it does not, without a trace.
        train.PHASE_MARKS = []
        for _ in range(10):
            eager_step()
        torch.cuda.synchronize()
        names = ("forward_and_med_head", "label_join_vgg_losses_vgg_adjoint", "backward_three_streams", "allreduce_and_optimiser")
        if train.PHASE_MARKS and all(len(m) == 5 for m in train.PHASE_MARKS):
            phase_ms = {n: sorted(m[i].elapsed_time(m[i + 1]) for m in train.PHASE_MARKS)[len(train.PHASE_MARKS) // 2] for i, n in enumerate(names)}
        train.PHASE_MARKS = None
    if rank == 0 and not args.no_roofline:
        # instrumented pass (NOT part of the timed region): HIP events around every launch
        timer = EventTimer()
        ops.TIMER = timer
        for _ in range(3):
            eager_step()
        agg = timer.summary()
        ops.TIMER = None
        if args.launch_table:
            # one row per LAUNCH of a step (position in the step's launch order), averaged over the three instrumented steps;
            # FLOPs are that launch's own algorithmic FLOPs
            n_per_step = len(timer.records) // 3
            rows = []
            for i in range(n_per_step):
                tag, flops, nbytes, _, _, name = timer.records[i]
                us = sum(timer.records[i + k * n_per_step][3].elapsed_time(timer.records[i + k * n_per_step][4]) for k in range(3)) * 1e3 / 3
                same = all(timer.records[i + k * n_per_step][5] == name for k in range(3))
                rows.append((us, i, flops, nbytes, name if same else name + " (order varies)", tag))
            with open(args.launch_table, "w") as f:
                f.write(f"# {n_per_step} launches per step; us = mean of 3 instrumented steps (HIP events, side streams off); TF = the launch's own algorithmic FLOPs / us\n")
                for us, i, flops, nbytes, name, tag in sorted(rows, key=lambda r: -r[0]):
                    f.write(f"{us:8.1f} us  #{i:3d}  {flops / 1e9:8.2f} GFLOP {flops / (us * 1e-6) / 1e12 if flops else 0:7.1f} TF  "
                            f"{nbytes / (us * 1e-6) / 1e9 if nbytes else 0:7.0f} GB/s  {name:44s} {tag}\n")
        total_ms = sum(a["ms"] for a in agg.values()) / 3
        # dominant kernel FAMILY: the instantiations of one kernel template that differ only in boolean flags (fused pool / planar output / ...) are
        # one family -- same tile, same loop; its members are listed in `roofline.members`
        import re
        fam = {}
        for t, a in agg.items():
            k = re.sub(r"Lb[01]E", "LbXE", t)
            f = fam.setdefault(k, {"ms": 0.0, "flops": 0.0, "launches": 0, "members": []})
            f["ms"] += a["ms"]; f["flops"] += a["flops"]; f["launches"] += a["launches"]; f["members"].append(t)
        dom_fam = max((k for k in fam if fam[k]["flops"] > 0), key=lambda k: fam[k]["ms"])
        d = fam[dom_fam]
        dom_tag = max(d["members"], key=lambda t: agg[t]["ms"])  # (the member with the most time: the symbol the live PMC passes are read for)
        peak = 157.3 if dtype == torch.float32 else 2500.0  # exact-f32 MFMA / dense bf16 = f16 MFMA (MI355X_MICROARCH.md)
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        mfma_ms = sum(a["ms"] for a in agg.values() if a["flops"] > 0) / 3
        mfma_fl = sum(a["flops"] for a in agg.values()) / 3
        result["roofline"] = {
            "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
            # live PMC child passes only from a single-process run (world > 1: the other ranks would sit in destroy_process_group)
            "traffic": (None if (args.no_live_traffic or world > 1) else hbm_traffic_live(d["members"], args)) or hbm_traffic_from_profiles(d["members"]),
            "kernel": dom_fam, "members": sorted(d["members"]), "launches_per_step": d["launches"] // 3, "avg_launch_us": d["ms"] * 1e3 / d["launches"],
            "kernel_ms_per_step": d["ms"] / 3, "all_kernels_ms_per_step": total_ms,
            "all_mfma_kernels": {"achieved": mfma_fl / (mfma_ms * 1e-3) / 1e12, "ms_per_step": mfma_ms,
                                 "algorithmic_gflop_per_step": mfma_fl / 1e9},
        }
        if phase_ms is not None:
            # FLOPs per phase from the instrumented launches of ONE step, cut at the MED head's forward / backward launches and the optimiser
            n_per_step = len(timer.records) // 3
            tags = [r[0] for r in timer.records[:n_per_step]]
            i_hf = max((i for i, t in enumerate(tags) if t.startswith("falnet_med_head_fwd")), default=-1)
            i_hb = max((i for i, t in enumerate(tags) if t.startswith("falnet_med_head_bwd")), default=-1)
            if 0 <= i_hf < i_hb:
                fl = [sum(r[1] for r in timer.records[:i_hf + 1]), sum(r[1] for r in timer.records[i_hf + 1:i_hb]),
                      sum(r[1] for r in timer.records[i_hb:n_per_step]), 0]
                # (the label image's VGG pass is LAUNCHED first in the instrumented, serial pass but runs beside the forward: count it there)
                result["roofline"]["phase_tflops"] = {
                    n: {"ms": round(ms_, 4), "gflop": round(f / 1e9, 1), "tflops": round(f / (ms_ * 1e-3) / 1e12, 1) if ms_ > 0 else None}
                    for (n, ms_), f in zip(phase_ms.items(), fl)}
                result["roofline"]["phase_tflops"]["note"] = ("median of 10 steps outside the timed region, HIP events on the step's stream at the phase boundaries (all streams live); "
                                                              "algorithmic FLOPs of the launches issued in each phase (weight gradients count in `backward`)")
        if dtype != torch.float32:
            sm = sustained_mfma(dtype, dev)
            result["roofline"]["sustained_mfma"] = sm
            result["roofline"]["frac_of_sustained"] = round(ach / sm["tflops_random_operands"], 4)
        heads = {t: a for t, a in agg.items() if t.startswith("falnet_med_head")}
        if heads:
            hb = sum(a["bytes"] for a in heads.values()) / 3
            hms = sum(a["ms"] for a in heads.values()) / 3
            result["roofline"]["med_head_hbm"] = {"bound": "hbm", "achieved": hb / (hms * 1e-3) / 1e9, "peak": 8000.0,
                                                  "unit": "GB/s", "frac": hb / (hms * 1e-3) / 1e9 / 8000.0, "ms_per_step": hms}
        result["kernel_breakdown_ms_per_step"] = {t: round(a["ms"] / 3, 4) for t, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload, args.height, args.width, args.levels)
        result["abs_rel_vs_ref"] = parity_vs_oracle(args.height, args.width, args.levels, dev)
    if rank == 0 and world == 1 and args.workload == "stage1" and not args.no_trajectory and not args.no_cpu_baseline:
        result["abs_rel_vs_gt"] = abs_rel_vs_ground_truth(args)
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)  # RCCL's banner sits in the C stdio buffer: flush it so the JSON is the LAST line
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    sys.exit(main())
