"""Probe: kernel mix of the cfg4 step (log-mel -> 2D splice -> ResNet9-2D, bs=256)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(bench.cfg4_spectrogram(torch.device('cuda:0'), steps=6, warmup=2))
