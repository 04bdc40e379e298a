"""Probe: displacement-scan kernel time vs batch size (is one block slow, or is it throughput?)."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import pcgmix_amd
from pcgmix_amd import saliency, synthetic
dev = torch.device('cuda:0')
T = 5000
for B in (1, 8, 64, 256, 1024):
    frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=2000, seed=0)
    rs = np.random.RandomState(0)
    sal = torch.from_numpy(rs.rand(B, T).astype(np.float32)).to(dev)
    fr = torch.from_numpy(frames.astype(np.int32)).to(dev)
    mix = torch.from_numpy(rs.permutation(B).astype(np.int32)).to(dev)
    gaps = np.abs(np.diff(frames, axis=1)[rs.permutation(B)] - np.diff(frames, axis=1))
    for _ in range(3):
        saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, 0, B, T)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, 0, B, T)
    e1.record(); torch.cuda.synchronize()
    print(B, 'us', e0.elapsed_time(e1) / 20 * 1e3, 'mean gap', gaps.mean(0))
