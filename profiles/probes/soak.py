"""Soak: many steps of the three hot loops (strict augment, saliency-guided augment, captured train
step); checks that device memory and host RSS stay flat and that results stay finite."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pcgmix_amd import augmentations, models, saliency

dev = torch.device("cuda:0")
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 7, dev)


def mem():
    return torch.cuda.memory_allocated(dev) >> 20, torch.cuda.memory_reserved(dev) >> 20, \
        resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10


def loop(name, n, fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    m0 = mem()
    t0 = time.perf_counter()
    for i in range(n):
        out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name:28s} {n} steps {1e6 * dt / n:7.1f} us/step  mem MiB (alloc, reserved, rss) {m0} -> {mem()}", flush=True)
    return out


sc = bench.StepCounter()
args = bench.Args("durratiomixup")
def aug():
    o = augmentations.augment(args, data, tgt, frames, wav, sc, None, dev, ""); sc.add(); return o[0]
y = loop("augment strict", 100_000, aug)
assert torch.isfinite(y).all()
args2 = bench.Args("durmixmagwarp(0.2,4)+0.5")
def aug2():
    o = augmentations.augment(args2, data, tgt, frames, wav, sc, None, dev, ""); sc.add(); return o[0]
y = loop("augment magwarp gate 0.5", 20_000, aug2)
assert torch.isfinite(y).all()
torch.manual_seed(4)
saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=5000).to(dev))
args3 = bench.Args("(saloptenv)durmixmagwarp(0.2,4)")
def aug3():
    o = augmentations.augment(args3, data, tgt, frames, wav, sc, None, dev, ""); sc.add(); return o[0]
y = loop("augment salopt", 10_000, aug3)
assert torch.isfinite(y).all()
saliency.set_saliency_model(None)
step, info = bench.build_train_step("durratiomixup", "Potes", 256, 4, 5000, 2000, dev, 40_000, 0)
loss = loop("captured train step", 30_000, step)
print("final loss", float(loss))
assert torch.isfinite(loss)
# round 3: BASELINE config 3 as a training step — pipelined (two captured slots, the augmentation
# of the next batch on a side stream), host labels, payload in the begin kernel's arguments
torch.manual_seed(5)
saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=5000).to(dev))
step3, info3 = bench.build_train_step("(saloptenv)durmixmagwarp(0.2,4)", "Potes", 256, 4, 5000, 2000, dev,
                                      20_000, 0)
assert info3["pipelined"]
loss = loop("pipelined cfg3 train step", 10_000, step3)
print("final loss", float(loss))
assert torch.isfinite(loss)
saliency.set_saliency_model(None)
print("armed steps | checked after a stream synchronisation | launched again:", bench.armed_step_stats(dev))
print("soak ok")
