"""Input gradient of the Potes conv stack from saved routing (potes_input_grad_mask_kernel, the
saliency pass): us per launch at N = 1024 rows x 5000 samples for a few persistent-grid sizes, and a
checksum.   python profiles/probes/potes_ingrad_time.py [blocks ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
N, T = 1024, 5000
P2 = lib.pcgmix_potes_out_len(T)
torch.manual_seed(0)
x = torch.randn(N, T, device=dev); g = torch.randn(N, 4, P2, device=dev)
w1 = torch.randn(8, 1, 5, device=dev) * 0.3; b1 = torch.randn(8, device=dev) * 0.1
w2 = torch.randn(4, 8, 5, device=dev) * 0.2; b2 = torch.randn(4, device=dev) * 0.1
h2 = torch.empty(N, 4, P2, device=dev); gx = torch.empty(N, T, device=dev)
m2 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=dev)
s1 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 1), dtype=torch.uint8, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: t.data_ptr()
_lib.check(lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), P(s1), N, T, None, 0, None, 0, st), "fwd")
f = lambda: lib.pcgmix_potes_stack_input_grad_mask_f32(P(g), P(m2), P(s1), P(w1), P(w2), P(gx), N, T, st)
for blocks in [None] + [int(a) for a in sys.argv[1:]]:
    if blocks: os.environ["PCGMIX_POTES_INGRAD_BLOCKS"] = str(blocks)
    for _ in range(20): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): f()
    e1.record(); torch.cuda.synchronize()
    print(f"input gradient (masks){' [%d blocks]' % blocks if blocks else '':16s} {e0.elapsed_time(e1) * 5:7.1f} us   checksum {float(gx.double().abs().sum()):.6f}", flush=True)
