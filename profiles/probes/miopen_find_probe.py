"""Probe: does MIOpen's find mode (torch.backends.cudnn.benchmark) speed up the ResNet9 steps?"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0')
for flag in (False, True):
    torch.backends.cudnn.benchmark = flag
    t0 = time.time()
    r1 = bench.train_steps_per_s("durmixmagwarp(0.2,4)", "resnet9", 256, 4, 5000, 2000, dev, 10, 3, lambda: None, 0)
    r2 = bench.cfg4_spectrogram(dev, steps=6, warmup=2)
    print('benchmark =', flag, 'resnet9-1d ms/step', round(r1['ms_per_step'], 2), 'resnet9-2d ms/step',
          round(r2['ms_per_step'], 2), 'wall', round(time.time() - t0, 1), flush=True)
