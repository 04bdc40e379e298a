#!/usr/bin/env python3
"""Per-recording log-mel front end (pcgmix_logmel_recordings_f32): how the time splits between the
tile pass (STFT on the f64 matrix cores, un-referenced dB columns to the scratch image) and the
slice pass (cycle columns out of the scratch, referenced to the recording's maximum).  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel durations; prints sizes and the wall time of
the whole call (host plan + uploads + both passes)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pcgmix_amd import frontend

rng = np.random.default_rng(0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lengths = rng.integers(20 * 2000, 60 * 2000, R)
boundaries, seg_starts = [], []
for n in lengths:
    b = np.cumsum(rng.integers(250, 550, 400))
    b = b[b < n - 200]
    boundaries.append(b)
    seg_starts.append(list(range(0, len(b) - 4, 4)))
dev = torch.device("cuda:0")
y = torch.randn(int(lengths.sum()), device=dev)
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spec, fs, rec = frontend.logmel_recordings(y, lengths, boundaries, seg_starts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
n_fft, hop = frontend.stft_params(2000)
cols = int(sum(1 + n // hop for n in lengths))
print(f"{R} recordings, {int(lengths.sum())} samples ({4e-6 * lengths.sum():.1f} MB), {cols} STFT columns, "
      f"scratch {128 * cols * 4e-6:.1f} MB, {spec.shape[0]} cycles, images {spec.numel() * 4e-6:.1f} MB; "
      f"last call {dt * 1e3:.2f} ms wall")
