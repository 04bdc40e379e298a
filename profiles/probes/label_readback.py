"""Probe: cheapest way to read the (B,2) int64 one-hot matrix back while the stream is busy."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
dev = torch.device('cuda:0')
tgt = torch.nn.functional.one_hot(torch.randint(0, 2, (256,)), 2).to(dev)
busy = torch.randn(256, 4, 5000, device=dev); out = torch.empty_like(busy)
pinned = torch.empty(256, 2, dtype=torch.int64, pin_memory=True)
ev = torch.cuda.Event()
N = 500
def a():
    return tgt.detach().cpu().numpy().argmax(axis=1)
def b():
    pinned.copy_(tgt, non_blocking=True); torch.cuda.current_stream().synchronize()
    return pinned.numpy().argmax(axis=1)
def c():
    pinned.copy_(tgt, non_blocking=True); ev.record(); ev.synchronize()
    return pinned.numpy().argmax(axis=1)
for name, fn in (("cpu()", a), ("pinned + stream sync", b), ("pinned + event sync", c)):
    for with_kernel in (False, True):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(N):
            if with_kernel:
                torch.add(busy, 1.0, out=out)       # ~10 us of GPU work queued ahead
            fn()
        torch.cuda.synchronize()
        print(f"{name:24s} kernel_ahead={with_kernel}: {(time.perf_counter() - t0) / N * 1e6:6.1f} us")
