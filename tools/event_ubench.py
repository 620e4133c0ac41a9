#!/usr/bin/env python3
"""What does an event record between two kernels of one stream cost, and does the release scope of the event matter?
200 back-to-back launches of a ~20 us kernel on one stream: (a) nothing between them, (b) torch.cuda.Event().record() after each,
(c) a raw HIP event created with hipEventDisableTiming | hipEventReleaseToDevice, (d) ... | hipEventDisableSystemFence, recorded
after each; in (b')-(d') a second stream waits for every event (as the weight-gradient streams do).  Tuning tool only."""
import ctypes, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
hip = ctypes.CDLL("libamdhip64.so")
hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
DISABLE_TIMING, REL_DEVICE, NO_SYSFENCE = 0x2, 0x40000000, 0x20000000
x = torch.zeros(8 << 20, device="cuda")
y = torch.zeros(1 << 20, device="cuda")
main, side = torch.cuda.Stream(), torch.cuda.Stream()
N = 200


def run(kind, waiter):
    evs = []
    if kind == "torch":
        evs = [torch.cuda.Event() for _ in range(N)]
    elif kind is not None:
        for _ in range(N):
            e = ctypes.c_void_p()
            assert hip.hipEventCreateWithFlags(ctypes.byref(e), kind) == 0
            evs.append(e)
    best = 1e9
    for rep in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(main):
            for i in range(N):
                x.add_(1.0)
                if kind == "torch":
                    evs[i].record(main)
                    if waiter:
                        side.wait_event(evs[i])
                elif kind is not None:
                    hip.hipEventRecord(evs[i], ctypes.c_void_p(main.cuda_stream))
                    if waiter:
                        hip.hipStreamWaitEvent(ctypes.c_void_p(side.cuda_stream), evs[i], 0)
                if waiter and i % 4 == 3:
                    with torch.cuda.stream(side):
                        y.add_(1.0)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best / N * 1e6


for waiter in (False, True):
    base = run(None, waiter)
    print(f"waiter={waiter}: no events {base:.2f} us per launch")
    for name, kind in (("torch.cuda.Event", "torch"), ("hip DisableTiming", DISABLE_TIMING), ("hip DisableTiming|ReleaseToDevice", DISABLE_TIMING | REL_DEVICE),
                       ("hip DisableTiming|DisableSystemFence", DISABLE_TIMING | NO_SYSFENCE), ("hip DisableTiming|ReleaseToDevice|DisableSystemFence", DISABLE_TIMING | REL_DEVICE | NO_SYSFENCE)):
        t = run(kind, waiter)
        print(f"   {name:55s} {t:.2f} us per launch (+{t - base:.2f})")
