"""Round 4: potes_bwd_fused_kernel against potes_bwd_pair_kernel (channel-pair packed multiply-adds,
no operand assembly, winner-only gw2).  Derived from potes_bwd_fused_time.py:
potes_bwd_kernel<true> (layer-1 recompute
spread over the block, selectors through LDS) against potes_bwd_fused_kernel (recompute on the
consuming lane), us per launch incl. the 212-column reduction, N = 1024 rows x 5000 samples, and
the captured train step with each.  The library reads its switch once per process: child per
variant.   python profiles/probes/potes_bwd_fused_time.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = r'''
import ctypes, sys, os
sys.path.insert(0, sys.argv[1])
import torch
import pcgmix_amd, bench
from pcgmix_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
N, T = 1024, 5000
P2 = lib.pcgmix_potes_out_len(T)
torch.manual_seed(0)
x = torch.randn(N, T, device=dev); g = torch.randn(N, 4, P2, device=dev)
w1 = torch.randn(8, 1, 5, device=dev) * 0.3; b1 = torch.randn(8, device=dev) * 0.1
w2 = torch.randn(4, 8, 5, device=dev) * 0.2; b2 = torch.randn(4, device=dev) * 0.1
h2 = torch.empty(N, 4, P2, device=dev)
m2 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=dev)
partial = torch.empty(4096, 212, device=dev); grads = torch.empty(212, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: t.data_ptr()
lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), None, N, T, None, 0, None, 0, st)
for blocks in (768, 1024, 1280):
    os.environ["PCGMIX_POTES_BWD_BLOCKS"] = str(blocks)
    f = lambda: lib.pcgmix_potes_stack_bwd_mask_f32(P(x), P(g), P(m2), P(w1), P(b1), P(w2), P(b2), P(partial), P(grads), N, T, st)
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    print(f"  bwd masks + reduce, {blocks} blocks: {e0.elapsed_time(e1) * 10:7.1f} us   grads checksum {float(grads.double().abs().sum()):.6f}", flush=True)
os.environ.pop("PCGMIX_POTES_BWD_BLOCKS")
r = bench.train_steps_per_s("durratiomixup", "Potes", 256, 4, 5000, 2000, dev, 400, 20, lambda: None, 0)
print(f"  captured train step: {r['ms_per_step'] * 1e3:.1f} us ({r['steps_per_s']:.0f} step/s)", flush=True)
'''
for tag, env in (("fused, position pairs (round 3)", {"PCGMIX_POTES_BWD_NO_PAIRS": "1"}), ("channel pairs (round 4)", {})):
    print(f"--- {tag}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=900)
    print(r.stdout.rstrip(), flush=True)
    if r.returncode:
        print(r.stderr[-1500:], flush=True)
