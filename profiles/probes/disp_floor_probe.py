"""Where does the displacement search's launch time go?  (a) the floor: a batch whose pairs all have
equal state lengths (every block exits at once); (b) staging only: every pair has ONE candidate
(gap 0 is 'no search', so gap 1: two candidates) on long segments; (c) the bench batch.  rocprofv3
kernel durations, not event pairs: run under `rocprofv3 --kernel-trace --stats`."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pcgmix_amd
from pcgmix_amd import saliency, synthetic
dev = torch.device("cuda:0")
B, T = 256, 5000
sal = torch.rand(B, T, device=dev)
mix = torch.from_numpy(np.random.RandomState(0).permutation(B).astype(np.int32)).to(dev)


def run(frames, label, reps=20):
    fr = torch.from_numpy(frames.astype(np.int32)).to(dev)
    ml = int(np.diff(frames, axis=1).max())
    f = lambda: saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, 0, B, T, max_len=ml)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:64s} {e0.elapsed_time(e1) / reps * 1e3:7.1f} us (events, incl. finalize)", flush=True)


same = np.tile(np.array([0, 300, 1100, 1400, 2800]), (B, 1))
run(same, "all pairs equal lengths: every block exits (launch floor)")
one = same.copy(); one[::2, 4] += 1                      # diastole 1400 vs 1401 on alternate samples
run(one, "diastole gap 1 (two candidates, mid 1400), other states equal")
small = same.copy(); small[::2, 4] += 64
run(small, "diastole gap 64 (one wave of candidates, mid 1400)")
full = same.copy(); full[::2, 4] += 255
run(full, "diastole gap 255 (four waves of candidates, mid 1400)")
allst = same.copy(); allst[::2, 1:] += np.array([40, 40 + 200, 40 + 200 + 40, 40 + 200 + 40 + 255])
run(allst, "all four states differ (gaps 40, 200, 40, 255)")
frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=2000, seed=0)
run(frames, "the bench batch")
