#!/usr/bin/env python3
"""Captured train step with one slot against the pipelined two-slot form (bench legs), us/step.
    python profiles/probes/pipeline_ab.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import torch, bench
dev = torch.device("cuda", 0)
for m in ("durratiomixup", "durmixmagwarp(0.2,4)"):
    r = bench.train_steps_per_s(m, "Potes", 256, 4, 5000, 2000, dev, 400, 20, lambda: None, 0)
    print(f"  {m:24s} Potes: {r['ms_per_step'] * 1e3:.1f} us ({r['steps_per_s']:.0f}/s)", flush=True)
t = bench.cfg3_train(dev, 300, 20, lambda: None, 0)
print(f"  cfg3 train step: {t['ms_per_step'] * 1e3:.1f} us ({t['steps_per_s']:.0f}/s), eager {t['eager_steps_per_s']:.0f}/s", flush=True)
r = bench.train_steps_per_s("durmixmagwarp(0.2,4)", "resnet9", 256, 4, 5000, 2000, dev, 20, 5, lambda: None, 0)
print(f"  ResNet9-1D magwarp: {r['ms_per_step']:.2f} ms ({r['steps_per_s']:.2f}/s)", flush=True)
'''
for tag, env in (("one slot", {"PCGMIX_BENCH_NO_PIPELINE": "1"}), ("pipelined", {})):
    print(f"--- {tag}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=900)
    print(r.stdout.rstrip(), flush=True)
    if r.returncode:
        print(r.stderr[-1500:], flush=True)
