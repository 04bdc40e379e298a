#!/usr/bin/env python3
"""Feasibility of pipelining the augmentation of batch k+1 with the training graph of batch k on a
second stream: N iterations of (a) the captured Potes train graph alone, (b) the saliency-guided
augment() alone, (c) both per iteration on the same stream (what a training step does today),
(d) the augmentation on a side stream while the graph replays on the main stream.
    python profiles/probes/overlap_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcgmix_amd import augmentations, models, saliency, train_model as tm  # noqa: E402

dev = torch.device("cuda", 0)
B, C, T = 256, 4, 5000
step_fn, info = bench.build_train_step("base", "Potes", B, C, T, 2000, dev, 100000, 0)   # graph replay only, no splice
_, data, tgt, frames, labels, wav = bench.make_device_batch(B, C, T, 2000, 7, dev)
torch.manual_seed(4)
saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=T).to(dev))
side = torch.cuda.Stream(dev)


def aug(method, sc):
    augmentations.augment(bench.Args(method), data, tgt, frames, wav, sc, None, dev, "", host_labels=labels)
    sc.add()


def timeit(fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for method in ("(saloptenv)durmixmagwarp(0.2,4)", "durratiomixup"):
    sc = bench.StepCounter()
    t_train = timeit(step_fn)
    t_aug = timeit(lambda: aug(method, sc))

    def serial():
        aug(method, sc)
        step_fn()

    def overlapped():
        with torch.cuda.stream(side):
            aug(method, sc)
        step_fn()
    t_ser = timeit(serial)
    t_ovl = timeit(overlapped)
    print(f"{method:34s} train graph {t_train:6.1f}  augment {t_aug:6.1f}  same stream {t_ser:6.1f}  "
          f"two streams {t_ovl:6.1f} us per iteration", flush=True)
