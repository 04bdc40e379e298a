"""How long the strict step takes to reach its steady state in a fresh process: 5 calls, then ten consecutive
20-step regions (each bracketed by a synchronisation), per-call times of the first region.
python profiles/probes/cold_start_regions.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pcgmix_amd  # noqa: F401
from pcgmix_amd import augmentations, hostprep, synthetic
from conftest import Args, StepCounter
dev = torch.device("cuda", 0)
print(hostprep.bind_host_threads(0), flush=True)
B, C, T = 256, 4, 5000
x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=0)
data = torch.from_numpy(x).to(dev)
tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(dev)
fr = torch.from_numpy(frames)
args, sc = Args("durratiomixup"), StepCounter(0)
a = torch.randn(4096, 4096, device=dev)
for _ in range(10):
    a = (a @ a).clamp_(-1, 1)
torch.cuda.synchronize()
k = 0
out = None
for _ in range(5):
    sc.count = k; k += 1
    out = augmentations.augment(args, data, tgt, fr, wav, sc, None, dev, "")
for r in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); st = []
    for _ in range(20):
        sc.count = k; k += 1
        out = augmentations.augment(args, data, tgt, fr, wav, sc, None, dev, "")
        st.append(time.perf_counter())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    calls = [1e6 * (b - a) for a, b in zip([t0] + st[:-1], st)]
    print("region %d: %.2f us/step; calls: first %.1f median %.1f max %.1f; drain %.1f" %
          (r, 1e6 * (t1 - t0) / 20, calls[0], sorted(calls)[10], max(calls), 1e6 * (t1 - st[-1])), flush=True)
