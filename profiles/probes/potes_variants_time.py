"""Time the Potes conv-stack kernels alone at the bench shape (N=1024 rows, T=5000): plain and
mask-saving forward, recomputing and mask-based backward / input gradient, and the persistent
backward's block count (PCGMIX_POTES_BWD_BLOCKS)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pcgmix_amd
from pcgmix_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
N, T = 1024, 5000
P2 = lib.pcgmix_potes_out_len(T)
x = torch.randn(N, T, device=dev); g = torch.randn(N, 4, P2, device=dev)
w1 = torch.randn(8, 1, 5, device=dev) * 0.3; b1 = torch.randn(8, device=dev) * 0.1
w2 = torch.randn(4, 8, 5, device=dev) * 0.2; b2 = torch.randn(4, device=dev) * 0.1
h2 = torch.empty(N, 4, P2, device=dev); gx = torch.empty(N, T, device=dev)
m2 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=dev)
s1 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 1), dtype=torch.uint8, device=dev)
partial = torch.empty(4096, 212, device=dev); grads = torch.empty(212, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: t.data_ptr()


def timeit(name, fn, n=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:44s} {e0.elapsed_time(e1) / n * 1e3:8.1f} us", flush=True)


timeit("fwd", lambda: lib.pcgmix_potes_stack_fwd_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), N, T, st))
timeit("fwd + m2", lambda: lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), None, N, T, None, 0, None, 0, st))
timeit("fwd + m2 + s1", lambda: lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), P(s1), N, T, None, 0, None, 0, st))
timeit("input_grad (recompute)", lambda: lib.pcgmix_potes_stack_input_grad_f32(P(x), P(g), P(w1), P(b1), P(w2), P(b2), P(gx), N, T, st))
timeit("input_grad (masks)", lambda: lib.pcgmix_potes_stack_input_grad_mask_f32(P(g), P(m2), P(s1), P(w1), P(w2), P(gx), N, T, st))
for blocks in (512, 768, 1024, 1280, 1536, 2048, 3072):
    os.environ["PCGMIX_POTES_BWD_BLOCKS"] = str(blocks)
    timeit(f"bwd recompute, {blocks} blocks", lambda: lib.pcgmix_potes_stack_bwd_f32(P(x), P(g), P(w1), P(b1), P(w2), P(b2), P(partial), P(grads), N, T, st), 100)
    timeit(f"bwd masks,     {blocks} blocks", lambda: lib.pcgmix_potes_stack_bwd_mask_f32(P(x), P(g), P(m2), P(w1), P(b1), P(w2), P(b2), P(partial), P(grads), N, T, st), 100)
