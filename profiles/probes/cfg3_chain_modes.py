#!/usr/bin/env python3
"""BASELINE config 3: the frozen Potes saliency chain launched directly (default) against replayed
as a hipGraph (PCGMIX_SAL_CHAIN_GRAPH=1): augment() step and captured train step, us.
    python profiles/probes/cfg3_chain_modes.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import torch, bench
dev = torch.device("cuda", 0)
r = bench.cfg3_salopt(dev, steps=300, warmup=20, reps=3)
print(f"  cfg3 augment() step: {r['ms_per_step'] * 1e3:.1f} us  {[round(v * 1e3, 1) for v in r['ms_per_step_repeats']]}", flush=True)
t = bench.cfg3_train(dev, 300, 20, lambda: None, 0)
print(f"  cfg3 train step: captured {t['ms_per_step'] * 1e3:.1f} us ({t['steps_per_s']:.0f}/s), eager {t['eager_steps_per_s']:.0f}/s", flush=True)
'''
for tag, env in (("chain as hipGraph", {"PCGMIX_SAL_CHAIN_GRAPH": "1"}), ("chain launched directly", {})):
    print(f"--- {tag}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=900)
    print(r.stdout.rstrip(), flush=True)
    if r.returncode:
        print(r.stderr[-1500:], flush=True)
