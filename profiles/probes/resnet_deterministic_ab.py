import sys, time, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    r = bench.train_steps_per_s("durmixmagwarp(0.2,4)", "resnet9", 256, 4, 5000, 2000, dev, 20, 5, lambda: None, 0)
    print("resnet9-1d deterministic", det, r["ms_per_step"], "ms  loss", r["loss"], flush=True)
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    r = bench.cfg4_spectrogram(dev)
    print("cfg4 resnet9-2d deterministic", det, r["ms_per_step"], "ms  loss", r["loss"], flush=True)
