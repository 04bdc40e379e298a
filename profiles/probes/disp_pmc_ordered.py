"""Probe: the displacement scan at bs=256, pairs dispatched longest chain first (round 4)."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import pcgmix_amd
from pcgmix_amd import saliency, synthetic
dev = torch.device('cuda:0')
B, T = 256, 5000
frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=2000, seed=0)
rs = np.random.RandomState(0)
sal = torch.from_numpy(rs.rand(B, T).astype(np.float32)).to(dev)
fr = torch.from_numpy(frames.astype(np.int32)).to(dev)
mixh = rs.permutation(B)
mix = torch.from_numpy(mixh.astype(np.int32)).to(dev)
order = saliency.dispatch_order(frames, mixh)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for _ in range(6):
    saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, mode, B, T,
                                   max_len=int(np.diff(frames, axis=1).max()), order=order)
torch.cuda.synchronize()
