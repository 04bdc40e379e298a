"""Probe: a few launches of the Potes conv-stack forward with saved routing (N = 1024 rows of 5000
samples: bs 256) for SQ-counter collection (profiles/run_potes_fwd_pmc_r3.sh)."""
import ctypes, sys, torch
sys.path.insert(0, ".")
import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
N, T = 1024, 5000
P2 = lib.pcgmix_potes_out_len(T)
torch.manual_seed(0)
x = torch.randn(N, T, device=dev)
w1, b1 = torch.randn(8, 1, 5, device=dev) * 0.3, torch.randn(8, device=dev) * 0.1
w2, b2 = torch.randn(4, 8, 5, device=dev) * 0.2, torch.randn(4, device=dev) * 0.1
h2 = torch.empty(N, 4, P2, device=dev)
m2 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=dev)
P = lambda t: t.data_ptr()
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for _ in range(6):
    _lib.check(lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), None, N, T,
                                                   None, 0, None, 0, st), "fwd")
torch.cuda.synchronize()
