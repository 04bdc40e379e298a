"""Captured Potes train step with the conv stack's gradient reduction folded into the optimiser launch
(default) against the separate reduction kernel (PCGMIX_NO_REDUCE_FOLD=1).  Child per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import torch, bench
dev = torch.device("cuda", 0)
for rep in range(3):
    r = bench.train_steps_per_s("durratiomixup", "Potes", 256, 4, 5000, 2000, dev, 1000, 50, lambda: None, 0)
    print(f"  captured train step: {r['ms_per_step'] * 1e3:.1f} us ({r['steps_per_s']:.0f} step/s)  loss {r['loss']:.6f}", flush=True)
'''
for tag, env in (("separate reduction kernel", {"PCGMIX_NO_REDUCE_FOLD": "1"}), ("reduction inside the optimiser launch", {})):
    print(f"--- {tag}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    print(r.stdout.rstrip(), flush=True)
    if r.returncode:
        print(r.stderr[-1500:], flush=True)
