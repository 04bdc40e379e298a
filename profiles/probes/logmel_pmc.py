"""Probe: a few launches of the per-cycle log-mel kernel at bs=256 x 5000 for PMC collection."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
import pcgmix_amd  # noqa: F401
from pcgmix_amd import frontend, synthetic
dev = torch.device('cuda:0')
B, T = 256, 5000
x, frames, labels, wav = synthetic.make_batch(B, 1, T, sample_rate=2000, seed=0)
x1 = torch.from_numpy(x[:, 0, :].copy()).to(dev)
for _ in range(6):
    frontend.logmel(x1, frames)
torch.cuda.synchronize()
