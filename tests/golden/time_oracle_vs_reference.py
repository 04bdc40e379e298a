#!/usr/bin/env python3
"""Time the CPU oracle next to the real reference on the bench workload (build container only).

SURVEY.md §8(d): the oracle is the CPU baseline bench.py times on the GPU box (the reference cannot
travel), so its value as a *timing* proxy has to be established where both can run: here.  Prints
ms per batch of each for the bench shapes; DESIGN.md §5 quotes the output.

    python tests/golden/time_oracle_vs_reference.py [--batch 256] [--reps 3]
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import pcgmix_amd  # noqa: E402,F401
from pcgmix_amd import synthetic  # noqa: E402
from _ref_import import import_reference  # noqa: E402
from make_golden import StepCounter, base_args  # noqa: E402
from oracle import pcgmix_oracle as oracle  # noqa: E402


def best(fn, reps):
    ts = []
    for r in range(reps):
        t0 = time.perf_counter()
        fn(r)
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    ref = import_reference()
    torch.set_num_threads(os.cpu_count())
    tmp = tempfile.mkdtemp()
    print(f"threads={torch.get_num_threads()}  batch={a.batch}")
    for method, C, T in (("durratiomixup", 4, 5000), ("durratiomixup", 1, 5000),
                         ("durmixmagwarp(0.2,4)", 4, 5000), ("durratiomixup", 4, 2500)):
        x, frames, labels, wav = synthetic.make_batch(a.batch, C, T, sample_rate=2000 if T == 5000 else 1000, seed=0)
        args = base_args(method, C, a.batch, tmp, sample_rate=2000 if T == 5000 else 1000)
        data = torch.from_numpy(x)
        ohe = torch.nn.functional.one_hot(torch.from_numpy(labels), 2)
        fr = torch.from_numpy(frames)

        def run_ref(r):
            return ref.augmentations.augment(args, data, ohe, fr, tuple(wav), StepCounter(r), None,
                                             torch.device("cpu"), tmp)

        def run_oracle(r):
            return oracle.augment(method, x, labels, frames, wav, r)

        yr = run_ref(0)[0].numpy()
        yo = oracle.augment(method, x, labels, frames, wav, 0)["y"]
        assert np.array_equal(yr, np.asarray(yo)), "oracle differs from the reference"
        tr, to = best(run_ref, a.reps), best(run_oracle, a.reps)
        print(f"{method:24s} ({a.batch},{C},{T})  reference {tr:8.2f} ms   oracle {to:8.2f} ms   "
              f"oracle/reference {to / tr:5.2f}")


if __name__ == "__main__":
    main()
