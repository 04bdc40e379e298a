import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda:0")
r0 = bench.train_steps_per_s("durmixmagwarp(0.2,4)", "Potes", 256, 4, 5000, 2000, dev, 300, 50, lambda: None, 0)
r = bench.cfg3_train(dev, 1000, 50, lambda: None, 0)
print("FETCH_AHEAD=%s  magwarp train %.1f us, then cfg3 train %.1f us" % (os.environ.get("PCGMIX_FETCH_AHEAD", "-"), r0["ms_per_step"]*1e3, r["ms_per_step"]*1e3), flush=True)
