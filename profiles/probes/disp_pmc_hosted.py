"""Probe: a few launches of the PLANNED displacement scan (host copies of boundaries and partners
handed over: blocks with candidates only, longest chain first) at bs=256 for PMC collection."""
import sys, torch, numpy as np
sys.path.insert(0, '.')
import pcgmix_amd  # noqa: F401
from pcgmix_amd import saliency, synthetic
dev = torch.device('cuda:0')
B, T = 256, 5000
frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=2000, seed=0)
rs = np.random.RandomState(0)
sal = torch.from_numpy(rs.rand(B, T).astype(np.float32)).to(dev)
fr = torch.from_numpy(frames.astype(np.int32)).to(dev)
mix_np = rs.permutation(B).astype(np.int32)
mix = torch.from_numpy(mix_np).to(dev)
for _ in range(5):
    saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, 0, B, T,
                                   max_len=int(np.diff(frames, axis=1).max()), frames_host=frames, mix_host=mix_np)
torch.cuda.synchronize()
