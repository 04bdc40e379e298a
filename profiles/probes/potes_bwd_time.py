"""Time pcgmix_potes_stack_{fwd,bwd,input_grad}_f32 alone at the bench shape (N=1024 rows, T=5000)."""
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
G = lib.pcgmix_potes_bwd_blocks(N, T)
partial = torch.empty(G, 212, device=dev); grads = torch.empty(212, device=dev)
h2 = torch.empty(N, 4, P2, device=dev); gx = torch.empty(N, T, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
fns = {
 "fwd": lambda: lib.pcgmix_potes_stack_fwd_f32(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), h2.data_ptr(), N, T, st),
 "bwd (weight grads + reduce)": lambda: lib.pcgmix_potes_stack_bwd_f32(x.data_ptr(), g.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), partial.data_ptr(), grads.data_ptr(), N, T, st),
 "input_grad": lambda: lib.pcgmix_potes_stack_input_grad_f32(x.data_ptr(), g.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), gx.data_ptr(), N, T, st),
}
for name, fn in fns.items():
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us   (G={G})")
