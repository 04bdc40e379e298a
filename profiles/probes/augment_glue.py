"""Probe: host time of one drop-in augment() call, by section (perf_counter, 2000 calls)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, cProfile, pstats
import pcgmix_amd
from pcgmix_amd import augmentations as A, synthetic
dev = torch.device('cuda:0')
B, C, T = 256, 4, 5000
x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=0)
data = torch.from_numpy(x).to(dev)
tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(dev)
fr = torch.from_numpy(frames)
class Args: method = 'durratiomixup'; num_classes = 2
class SC: count = 0
sc = SC()
for _ in range(50):
    A.augment(Args, data, tgt, fr, wav, sc, None, dev, ''); sc.count += 1
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    A.augment(Args, data, tgt, fr, wav, sc, None, dev, ''); sc.count += 1
torch.cuda.synchronize()
print(f"augment(): {(time.perf_counter() - t0) / N * 1e6:.1f} us/call")
pr = cProfile.Profile(); pr.enable()
for _ in range(N):
    A.augment(Args, data, tgt, fr, wav, sc, None, dev, ''); sc.count += 1
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime').print_stats(14)
