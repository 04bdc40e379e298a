import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device('cuda:0')
r = bench.train_steps_per_s("durmixmagwarp(0.2,4)", "Potes", 256, 4, 5000, 2000, dev, 1000, 50, lambda: None, 0)
print(os.environ.get("PCGMIX_FETCH_AHEAD", "default"), r["ms_per_step"] * 1e3, r["loss"])
