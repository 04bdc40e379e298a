"""cfg3 as a training step (pipelined), us per step.   python profiles/probes/cfg3_train_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device("cuda:0")
r = bench.cfg3_train(dev, 1000, 50, lambda: None, 0)
print("PCGMIX_FETCH_AHEAD=%s cfg3 train step %.1f us  loss %.6f" % (os.environ.get("PCGMIX_FETCH_AHEAD", "-"), r["ms_per_step"] * 1e3, r["loss"]), flush=True)
