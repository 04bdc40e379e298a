#!/usr/bin/env python3
"""Splice+warp kernel variants, back-to-back launch time (us) per shape.  The library reads its
tuning switches once per process, so every variant runs in a child:
  old      PCGMIX_NO_WARP_TQ=1   mix_warp_kernel<4,true,U>: per (channel, position) spline work
  tq       default               mix_warp_tq_kernel: per-position work shared by the channels
  tq cg2   PCGMIX_WARP_TQ_CG=2   same, two channels' loads in flight per lane instead of four
  tq ut2   PCGMIX_WARP_TQ_UT=2   same, two position quads per lane
    python profiles/probes/warp_variants_time.py
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import sys
sys.path.insert(0, sys.argv[1])
import torch, bench
dev = torch.device("cuda", 0)
for m, b, c, t in (("durmixmagwarp(0.2,4)", 256, 4, 5000), ("durmixmagwarp(0.2,4)", 256, 1, 5000),
                   ("durmixmagwarp(0.2,4)", 256, 4, 2500), ("durmixmagwarp(0.2,4)", 16384, 4, 5000),
                   ("durratiomixup", 16384, 4, 5000)):
    ms = bench.kernel_back_to_back_ms(m, b, c, t, 2000, dev, iters=50 if b > 1000 else 200)
    print(f"  {m:22s} ({b},{c},{t}) {ms * 1e3:9.2f} us", flush=True)
'''
for tag, env in (("old", {"PCGMIX_NO_WARP_TQ": "1"}), ("tq cg4", {"PCGMIX_WARP_TQ_CG": "4"}), ("tq cg2", {"PCGMIX_WARP_TQ_CG": "2"}),
                 ("tq cg1", {"PCGMIX_WARP_TQ_CG": "1"}), ("tq cg2 ut2", {"PCGMIX_WARP_TQ_CG": "2", "PCGMIX_WARP_TQ_UT": "2"})):
    print(f"--- {tag} {env}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=600)
    print(r.stdout.rstrip(), flush=True)
    if r.returncode:
        print(r.stderr[-800:], flush=True)
