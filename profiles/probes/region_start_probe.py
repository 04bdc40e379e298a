#!/usr/bin/env python3
"""Where does a 20-step timed region lose time against steady state?  (the
driver's `--steps 20 --warmup 5` read 39.8 us/step where 1000-step regions read 22.6.)

Repeats the bench's region — synchronize, then 20 drop-in augment() calls, synchronize — R times
and prints the host-side duration of every call per region (rows = regions, columns = calls),
plus the library's own phase timers for call 1, call 2 and the rest.  Run it under
`rocprofv3 --kernel-trace` to see the same regions from the GPU's side (kernel start/end).

    python profiles/probes/region_start_probe.py [--host-labels] [--idle-ms 0] [--regions 8]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pcgmix_amd  # noqa: E402,F401
from pcgmix_amd import _lib, augmentations, synthetic  # noqa: E402


class A:
    method = "durratiomixup"
    num_classes = 2


class SC:
    count = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--host-labels", action="store_true")
    ap.add_argument("--idle-ms", type=float, default=0.0, help="host sleep before each region")
    ap.add_argument("--regions", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warm", type=int, default=50, help="warm-up calls before the first region")
    ap.add_argument("--freeze", action="store_true", help="gc.collect() + gc.freeze() before the warm-up (bench.settle_heap)")
    ap.add_argument("--gc", choices=["on", "off", "collect0"], default="on",
                    help="inside the regions: collector on, disabled, or a gen-0 collection before each region")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    x, frames, labels, wav = synthetic.make_batch(256, 4, 5000, sample_rate=2000, seed=0)
    data = torch.from_numpy(x).to(dev)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(dev)
    fr = torch.from_numpy(frames)
    kw = {"host_labels": labels} if a.host_labels else {}
    args, sc = A(), SC()
    lib = _lib.load()
    import gc
    if a.freeze:
        gc.collect()
        gc.freeze()
    for _ in range(a.warm):
        augmentations.augment(args, data, tgt, fr, wav, sc, None, dev, "", **kw)
        sc.count += 1
    if a.gc == "off":
        gc.disable()
    torch.cuda.synchronize()
    ctx = augmentations.step_context(0)
    ph = (ctypes.c_double * 8)()
    names = ("label_launch", "slot", "pack+seed", "label_wait", "partners", "h2d", "launch", "event")
    rows, phases = [], {1: [], 2: [], 3: []}
    for r in range(a.regions):
        if a.idle_ms:
            time.sleep(a.idle_ms * 1e-3)
        if a.gc == "collect0":
            gc.collect(0)
        torch.cuda.synchronize()
        lib.pcgmix_ctx_phase_times(ctx, ph)
        st = [time.perf_counter()]
        for i in range(a.steps):
            augmentations.augment(args, data, tgt, fr, wav, sc, None, dev, "", **kw)
            sc.count += 1
            st.append(time.perf_counter())
            if i < 2 or i == a.steps - 1:
                n = lib.pcgmix_ctx_phase_times(ctx, ph)
                phases[min(i + 1, 3)].append([ph[k] / 1e3 for k in range(8)] if n else None)
        torch.cuda.synchronize()
        end = time.perf_counter()
        d = np.diff(st) * 1e6
        rows.append(list(d) + [(end - st[-1]) * 1e6, (end - st[0]) * 1e6 / a.steps])
    print(f"host_labels={a.host_labels} idle_ms={a.idle_ms} warm={a.warm} freeze={a.freeze} gc={a.gc} "
          f"gc.get_count()={gc.get_count()}: per-call host us (last two columns: drain, "
          f"region mean per step)")
    for r in rows:
        print(" ".join(f"{v:6.1f}" for v in r))
    print("median per column:", " ".join(f"{v:6.1f}" for v in np.median(np.asarray(rows), axis=0)))
    for k, tag in ((1, "call 1"), (2, "call 2"), (3, "last call (mean of calls 3..n)")):
        good = [p for p in phases[k] if p]
        if good:
            m = np.median(np.asarray(good), axis=0)
            print(f"library phases, {tag}: " + ", ".join(f"{n} {v:.1f}" for n, v in zip(names, m)))


if __name__ == "__main__":
    main()
