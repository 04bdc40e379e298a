"""Probe: kernel mix of one ResNet9-1D training step (bs=256, C=4, T=5000) through torch/MIOpen."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda:0')
r = bench.train_steps_per_s("durmixmagwarp(0.2,4)", "resnet9", 256, 4, 5000, 2000, dev, 10, 3, lambda: None, 0, use_graph=False)
print(r)
