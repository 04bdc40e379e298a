import sys, torch, numpy as np
sys.path.insert(0, '.')
import pcgmix_amd
from pcgmix_amd import frontend, synthetic
dev = torch.device('cuda:0')
x, frames, labels, wav = synthetic.make_batch(256, 1, 5000, sample_rate=2000, seed=0)
x1 = torch.from_numpy(x[:, 0]).to(dev)
for _ in range(20):
    frontend.logmel(x1, frames)
torch.cuda.synchronize()
