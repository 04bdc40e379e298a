#!/usr/bin/env python3
"""Displacement search (salopt_disp_kernel), us per launch at BASELINE config 3's shapes, plus the
whole saliency-guided augment() step.  (Commit b1c49cb carried a second, leaf-parallel kernel
behind PCGMIX_DISP_LANE_PER_CANDIDATE; this script compared the two:
profiles/r3_disp_leaf_parallel_negative.txt.)
    python profiles/probes/disp_variants_time.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, bench
from pcgmix_amd import saliency, synthetic
dev = torch.device("cuda", 0)
for B, C, T, rate in ((256, 4, 5000, 2000), (256, 4, 2500, 1000), (32, 4, 2500, 1000)):
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=rate, seed=0)
    data = torch.from_numpy(x).to(dev)
    fr = torch.from_numpy(frames.astype(np.int32)).to(dev)
    sal = saliency.saliency_post(torch.randn_like(data), fr.data_ptr())
    mix = torch.from_numpy(np.random.RandomState(0).permutation(B).astype(np.int32)).to(dev)
    ml = int(np.diff(frames, axis=1).max())
    for mode, name in ((0, "env"), (1, "sum")):
        def f():
            saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, mode, B, T, max_len=ml)
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"  disp {name} ({B},{T}) max_len {ml}: {e0.elapsed_time(e1) * 10:8.1f} us (incl. finalize launch)", flush=True)
r = bench.cfg3_salopt(dev, steps=200, warmup=20, reps=3)
print(f"  cfg3 augment() step: {r['ms_per_step'] * 1e3:.1f} us  {r['ms_per_step_repeats']}", flush=True)
'''
for tag, env in (("lane per candidate", {}),):
    print(f"--- {tag} {env}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=900)
    print(r.stdout.rstrip(), flush=True)
    if r.returncode:
        print(r.stderr[-1500:], flush=True)
