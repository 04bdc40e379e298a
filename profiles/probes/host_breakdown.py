"""Probe: where the ~100 us of one drop-in augment() call go on the host (no profiler)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
import pcgmix_amd
from pcgmix_amd import augmentations as A, hostprep, synthetic

dev = torch.device('cuda:0')
B, C, T = 256, 4, 5000
x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=0)
data = torch.from_numpy(x).to(dev)
tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(dev)
fr = torch.from_numpy(frames)
N = 500
acc = {}
def tick(name, t0):
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
for it in range(N + 20):
    if it == 20:
        acc.clear(); torch.cuda.synchronize(); t_all = time.perf_counter()
    t = time.perf_counter(); lab = A.labels_from_ohe(tgt); tick('labels_d2h', t)
    t = time.perf_counter(); fnp = A._as_numpy_frames(fr); tick('frames_numpy', t)
    t = time.perf_counter(); plan = hostprep.make_plan('durratiomixup', lab, fnp, wav, it, B, C); tick('make_plan', t)
    t = time.perf_counter(); hostprep.validate_frames(fnp, T); tick('validate', t)
    t = time.perf_counter()
    with torch.cuda.device(dev):
        d, offs = A.upload_plan(plan, fnp, dev)
    tick('upload', t)
    t = time.perf_counter(); out = torch.empty_like(data); tick('alloc_out', t)
    t = time.perf_counter()
    base = d.data_ptr()
    A.launch_mix(data, out, base, base + offs['mix'], None, float(plan.lam32), None, None, 0, B, C, T)
    tick('launch', t)
torch.cuda.synchronize()
total = (time.perf_counter() - t_all) / N * 1e6
for k, v in acc.items():
    print(f"{k:14s} {v / N * 1e6:7.1f} us")
print(f"{'sum':14s} {sum(acc.values()) / N * 1e6:7.1f} us   wall/step {total:7.1f} us")
# pieces of make_plan
import random
t = time.perf_counter()
for s in range(N): random.Random(s).uniform(0, 1)
print('gate        ', (time.perf_counter() - t) / N * 1e6)
t = time.perf_counter()
for s in range(N): hostprep.shuffle_within_groups(labels, s)
print('shuffle     ', (time.perf_counter() - t) / N * 1e6)
t = time.perf_counter()
for s in range(N): np.random.seed(s); np.random.beta(1.0, 1.0)
print('seed+beta   ', (time.perf_counter() - t) / N * 1e6)
# the drop-in call itself
class Args: method = 'durratiomixup'; num_classes = 2
class SC:
    count = 0
sc = SC()
for name, extra in (('augment()', None), ('augment()+event pairs', 'ev')):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(N):
        if extra:
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
        out = A.augment(Args, data, tgt, fr, wav, sc, None, dev, '')
        if extra:
            e1.record()
        sc.count += 1
    torch.cuda.synchronize()
    print(f"{name:24s} {(time.perf_counter() - t0) / N * 1e6:7.1f} us/step")
