#!/usr/bin/env python3
"""Host cost of one replayed command (tools): N tiny launches as one falnet_replay call vs N ctypes calls, on an idle device."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from fal_net_amd import _lib as L

dev = torch.device("cuda", 0)
S, out = torch.zeros(2, device=dev), torch.zeros(3, device=dev)
lib = L.lib()
N = 300
call = lambda: L.check(lib.falnet_step_scalars(L.ptr(S), 1.0, L.ptr(out), L.stream_ptr()))
for _ in range(10):
    call()
seg = L.record_calls([call] * N)
torch.cuda.synchronize()
for name, fn in (("python", lambda: [call() for _ in range(N)]), ("replay", lambda: seg.run(L.stream_ptr().value))):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) / N * 1e6)
        torch.cuda.synchronize()
    print(name, "us per launch (host):", [round(t, 2) for t in ts])
