"""Probe: kernel mix of one saliency-guided step (BASELINE configs[2])."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
r = bench.cfg3_salopt(dev, steps=40, warmup=5)
print(r)
