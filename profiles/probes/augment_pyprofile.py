"""Where the Python side of the strict-signature augment() step spends its time: cProfile over 20k
calls (profiler overhead inflates every frame equally; read the shares) and a few micro-timings of
the pieces called per step.
Measured: 24.4 us per step un-profiled, of which the library call 16.4; numpy seed + beta 1.5,
torch.empty_like 1.1, boundary and input checks 0.85 each, detach 0.3-0.5, the rest call overhead."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from pcgmix_amd import augmentations

dev = torch.device("cuda:0")
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 7, dev)
args, sc = bench.Args("durratiomixup"), bench.StepCounter()


def step():
    out = augmentations.augment(args, data, tgt, frames, wav, sc, None, dev, "")
    sc.add()
    return out


for _ in range(2000):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20000):
    step()
torch.cuda.synchronize()
print(f"un-profiled: {(time.perf_counter() - t0) / 20000 * 1e6:.2f} us per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(20000):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue()[:3500])


def t(name, f, n=200000):
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e6:6.2f} us")


t("np.random.seed(i); beta(1,1)", lambda: (np.random.seed(5), np.random.beta(1.0, 1.0)))
t("torch.empty_like(data)", lambda: torch.empty_like(data), 100000)
t("np.empty(256, int64)", lambda: np.empty(256, dtype=np.int64))
t("data.data_ptr()", lambda: data.data_ptr())
t("target_ohe.detach()", lambda: tgt.detach())
