"""Probe: the displacement search for ONE heavy (sample, state) pair alone on the GPU — how long is
the serial chain of a single wave of candidates?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pcgmix_amd
from pcgmix_amd import saliency
dev = torch.device("cuda:0")
T = 5000
def run(n_long, n_short, own_longer, label):
    # states 0..2 equal in both samples (no search); state 3 (diastole) differs
    a = [0, 200, 600, 800, 800 + (n_long if own_longer else n_short)]
    b = [0, 200, 600, 800, 800 + (n_short if own_longer else n_long)]
    frames = np.array([a, b], dtype=np.int32)
    sal = torch.rand(2, T, device=dev)
    fr = torch.from_numpy(frames).to(dev)
    mix = torch.tensor([1, 0], dtype=torch.int32, device=dev)
    f = lambda: saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, 0, 2, T, max_len=n_long)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{label:48s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us", flush=True)
run(1398, 500, True, "own longer: gap 898, mid 500 (+head/tail) x2")
run(1398, 1300, True, "own longer: gap 98, mid 1300")
run(1398, 500, False, "partner longer: gap 898, mid 500, no head/tail")
run(700, 690, True, "gap 10, mid 690")
run(100, 90, True, "gap 10, mid 90")
