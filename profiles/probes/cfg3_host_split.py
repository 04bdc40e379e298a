"""Probe: where a saliency-guided step (BASELINE configs[2]) spends its time — wall per step over
several repeats (run-to-run spread), GPU busy time per step from events, and the host profile."""
import cProfile, pstats, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pcgmix_amd import augmentations, models, saliency

dev = torch.device("cuda:0")
method = "(saloptenv)durmixmagwarp(0.2,4)"
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 7, dev)
torch.manual_seed(4)
saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=5000).to(dev))
for rep in range(5):
    dt, _ = bench.run_augment_steps(method, data, tgt, frames, wav, dev, 100, 10, lambda: None)
    print(f"rep {rep}: {1e6 * dt / 100:.1f} us/step", flush=True)
dt, _ = bench.run_augment_steps(method, data, tgt, frames, wav, dev, 100, 10, lambda: None, host_labels=labels)
print(f"host_labels: {1e6 * dt / 100:.1f} us/step", flush=True)
# GPU busy: events around 50 steps with a sync after each step (no overlap between steps)
args, sc = bench.Args(method), bench.StepCounter()
e = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
tot = 0.0
for i in range(50):
    torch.cuda.synchronize()
    e[0].record()
    augmentations.augment(args, data, tgt, frames, wav, sc, None, dev, "")
    e[1].record()
    torch.cuda.synchronize()
    tot += e[0].elapsed_time(e[1])
    sc.add()
print(f"GPU span per isolated step: {1e3 * tot / 50:.1f} us")
pr = cProfile.Profile()
pr.enable()
bench.run_augment_steps(method, data, tgt, frames, wav, dev, 200, 0, lambda: None)
pr.disable()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22)
print(st.getvalue())
