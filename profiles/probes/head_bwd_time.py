"""Time the Potes head kernels at the bench shape (B=256, K=19968) with HIP events."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pcgmix_amd
from pcgmix_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
B, K, C = 256, 19968, 2
x = torch.randn(B, K, device=dev).relu_(); m1 = (torch.rand(B, K, device=dev) > 0.25)
m2 = (torch.rand(B, 20, device=dev) > 0.5).to(torch.uint8)
w1 = torch.randn(20, K, device=dev) / 100; b1 = torch.randn(20, device=dev)
w2 = torch.randn(C, 20, device=dev); b2 = torch.randn(C, device=dev)
ks = lib.pcgmix_skinny_linear_splits(B, K)
partial = torch.empty(ks, B, 20, device=dev); z = torch.empty(B, 20, device=dev); logits = torch.empty(B, C, device=dev)
dl = torch.randn(B, C, device=dev); dz = torch.empty(B, 20, device=dev)
dw2 = torch.empty(C, 20, device=dev); db2 = torch.empty(C, device=dev); db1 = torch.empty(20, device=dev)
dw1 = torch.empty(20, K, device=dev); dx = torch.empty(B, K, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
f = lambda: lib.pcgmix_potes_head_fwd_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), m2.data_ptr(), ctypes.c_float(2.0), w2.data_ptr(), b2.data_ptr(), partial.data_ptr(), z.data_ptr(), logits.data_ptr(), B, K, C, st)
g = lambda: lib.pcgmix_potes_head_bwd_f32(dl.data_ptr(), z.data_ptr(), m2.data_ptr(), ctypes.c_float(2.0), w2.data_ptr(), x.data_ptr(), m1.data_ptr(), ctypes.c_float(4 / 3), w1.data_ptr(), dz.data_ptr(), dw2.data_ptr(), db2.data_ptr(), db1.data_ptr(), dw1.data_ptr(), dx.data_ptr(), B, K, C, st)
g_nomask = lambda: lib.pcgmix_potes_head_bwd_f32(dl.data_ptr(), z.data_ptr(), m2.data_ptr(), ctypes.c_float(2.0), w2.data_ptr(), x.data_ptr(), None, ctypes.c_float(1.0), w1.data_ptr(), dz.data_ptr(), dw2.data_ptr(), db2.data_ptr(), db1.data_ptr(), dw1.data_ptr(), dx.data_ptr(), B, K, C, st)
g_nodx = lambda: lib.pcgmix_potes_head_bwd_f32(dl.data_ptr(), z.data_ptr(), m2.data_ptr(), ctypes.c_float(2.0), w2.data_ptr(), x.data_ptr(), None, ctypes.c_float(1.0), w1.data_ptr(), dz.data_ptr(), dw2.data_ptr(), db2.data_ptr(), db1.data_ptr(), dw1.data_ptr(), None, B, K, C, st)
for name, fn in (("head_fwd (partial+tail)", f), ("head_bwd (tail+head)", g), ("head_bwd, no dropout mask", g_nomask), ("head_bwd, no dx (dW1 only)", g_nodx)):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us")
