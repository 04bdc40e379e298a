"""Probe: the saliency-guided step in its two calling modes (labels read back / handed over) and
with bounded host run-ahead, to see which one leaves the GPU idle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from pcgmix_amd import augmentations, models, saliency

dev = torch.device("cuda:0")
method = "(saloptenv)durmixmagwarp(0.2,4)"
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 7, dev)
torch.manual_seed(4)
saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=5000).to(dev))


def loop(n, host_labels=None, sync_every=0):
    args, sc = bench.Args(method), bench.StepCounter()
    kw = {} if host_labels is None else {"host_labels": host_labels}
    for _ in range(10):
        augmentations.augment(args, data, tgt, frames, wav, sc, None, dev, "", **kw); sc.add()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        augmentations.augment(args, data, tgt, frames, wav, sc, None, dev, "", **kw); sc.add()
        if sync_every and i % sync_every == sync_every - 1:
            torch.cuda.synchronize()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e6 * (t2 - t0) / n, 1e6 * (t1 - t0) / n


for tag, kw in (("strict", {}), ("host_labels", {"host_labels": labels}), ("strict again", {}),
                ("host_labels sync/1", {"host_labels": labels, "sync_every": 1}),
                ("host_labels sync/2", {"host_labels": labels, "sync_every": 2}),
                ("host_labels sync/4", {"host_labels": labels, "sync_every": 4}),
                ("host_labels sync/16", {"host_labels": labels, "sync_every": 16}),
                ("host_labels", {"host_labels": labels})):
    for rep in range(2):
        total, host = loop(200, **kw)
        print(f"{tag:22s} rep {rep}: {total:7.1f} us/step (host enqueue {host:6.1f})", flush=True)
