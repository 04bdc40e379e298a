import sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import bench
from pcgmix_amd import hostprep, _lib
import ctypes
dev = torch.device('cuda:0')
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, seed=0, device=dev)
for rep in range(3):
    tr = {}
    dt, _ = bench.run_augment_steps("durmixmagwarp(0.2,4)", data, tgt, frames, wav, dev, 20, 5, lambda: None, trace=tr)
    h = hostprep._NPDRAW[0]
    m = ctypes.c_longlong()
    hits = _lib.load().pcgmix_npdraw_stats(h, ctypes.byref(m))
    print(f"rep {rep}: {dt/20*1e6:.1f} us/step; hits {hits} misses {m.value}; trace {tr}")
import time
# per-call times of a fresh sequence without the settle
args, sc = bench.Args("durmixmagwarp(0.2,4)"), bench.StepCounter()
from pcgmix_amd import augmentations
torch.cuda.synchronize()
ts = []
for i in range(30):
    t0 = time.perf_counter()
    out = augmentations.augment(args, data, tgt, frames, wav, sc, None, dev, "")
    sc.add()
    ts.append((time.perf_counter() - t0) * 1e6)
print("per-call us:", [round(t) for t in ts])
