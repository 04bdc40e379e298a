"""Probe: kernel mix of one saliency-guided step (BASELINE configs[2])."""
import sys, time, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
dev = torch.device('cuda:0')
r = bench.cfg3_salopt(dev, steps=40, warmup=5)
print(r)
