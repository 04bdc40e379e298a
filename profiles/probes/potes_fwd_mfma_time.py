"""Potes conv-stack forward: the VALU kernel (potes_fwd_kernel, packed FMAs) against the matrix-core
kernel (potes_fwd_mfma_kernel, v_mfma_f32_4x4x1_16b_f32), us per launch at N = 1024 rows x 5000
samples with and without the saved routing, maximum difference between the two, and the captured
train step with each.  The library reads its switch once per process: child per variant.
    python profiles/probes/potes_fwd_mfma_time.py"""
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
x = torch.randn(N, T, device=dev)
w1 = torch.randn(8, 1, 5, device=dev) * 0.3; b1 = torch.randn(8, device=dev) * 0.1
w2 = torch.randn(4, 8, 5, device=dev) * 0.2; b2 = torch.randn(4, device=dev) * 0.1
h2 = torch.empty(N, 4, P2, device=dev)
m2 = torch.zeros(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=dev)
s1 = torch.zeros(lib.pcgmix_potes_mask_bytes(N, T, 1), dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: t.data_ptr()
def timeit(tag, f):
    for _ in range(10): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    print(f"  {tag:34s} {e0.elapsed_time(e1) * 5:7.1f} us", flush=True)
for blocks in [None] + ([int(b) for b in os.environ.get("SWEEP", "").split(",") if b]):
    if blocks: os.environ["PCGMIX_POTES_FWD_BLOCKS"] = str(blocks)
    tag = f" [{blocks} blocks]" if blocks else ""
    timeit("forward" + tag, lambda: lib.pcgmix_potes_stack_fwd_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), N, T, st))
    timeit("forward + m2" + tag, lambda: lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), None, N, T, None, 0, None, 0, st))
    timeit("forward + m2 + s1" + tag, lambda: lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), P(s1), N, T, None, 0, None, 0, st))
os.environ.pop("PCGMIX_POTES_FWD_BLOCKS", None)
ref = torch.nn.functional.max_pool1d(torch.relu(torch.nn.functional.conv1d(
    torch.nn.functional.max_pool1d(torch.relu(torch.nn.functional.conv1d(x.double().unsqueeze(1), w1.double(), b1.double(), padding=1)), 2),
    w2.double(), b2.double(), padding=1)), 2)
print(f"  max |h2 - float64 torch| = {float((h2.double() - ref).abs().max()):.3g}; checksums h2 {float(h2.double().sum()):.6f} "
      f"m2 {int(m2.long().sum())} s1 {int(s1.long().sum())}", flush=True)
r = bench.train_steps_per_s("durratiomixup", "Potes", 256, 4, 5000, 2000, dev, 400, 20, lambda: None, 0)
print(f"  captured train step: {r['ms_per_step'] * 1e3:.1f} us ({r['steps_per_s']:.0f} step/s)", flush=True)
'''
for tag, env in (("VALU (rounds 1-2)", {"PCGMIX_POTES_FWD_VALU": "1"}), ("matrix cores", {"SWEEP": os.environ.get("SWEEP", "512,1024")})):
    print(f"--- {tag}", flush=True)
    r = subprocess.run([sys.executable, "-c", CODE, ROOT], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=900)
    print(r.stdout.rstrip(), flush=True)
    if r.returncode:
        print(r.stderr[-1500:], flush=True)
