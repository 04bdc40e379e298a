"""Round 4: the Potes weight-gradient backward alone (no reduction launch) over row counts and block
counts — is its time per item, per launch, or per block?   python profiles/probes/potes_bwd_sweep.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib
lib = _lib.load(); dev = torch.device("cuda", 0)
T = 5000
P2 = lib.pcgmix_potes_out_len(T)
torch.manual_seed(0)
w1 = torch.randn(8, 1, 5, device=dev) * 0.3; b1 = torch.randn(8, device=dev) * 0.1
w2 = torch.randn(4, 8, 5, device=dev) * 0.2; b2 = torch.randn(4, device=dev) * 0.1
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t: t.data_ptr()
partial = torch.empty(8192, 212, device=dev)
print(f"variant: {'fused (round 3)' if os.environ.get('PCGMIX_POTES_BWD_NO_PAIRS') else 'channel pairs (round 4)'}")
for N in (128, 256, 512, 1024, 2048, 4096):
    x = torch.randn(N, T, device=dev); g = torch.randn(N, 4, P2, device=dev)
    h2 = torch.empty(N, 4, P2, device=dev)
    m2 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=dev)
    lib.pcgmix_potes_stack_fwd_save_f32(P(x), P(w1), P(b1), P(w2), P(b2), P(h2), P(m2), None, N, T, None, 0, None, 0, st)
    row = []
    for blocks in (256, 512, 768, 1024, 1536, 2048):
        os.environ["PCGMIX_POTES_BWD_BLOCKS"] = str(blocks)
        f = lambda: lib.pcgmix_potes_stack_bwd_mask_f32(P(x), P(g), P(m2), P(w1), P(b1), P(w2), P(b2), P(partial), None, N, T, st)
        for _ in range(5): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): f()
        e1.record(); torch.cuda.synchronize()
        row.append(f"{blocks}:{e0.elapsed_time(e1) * 20:6.1f}")
    print(f"N={N:5d} rows ({N * 10:6d} items)  us per launch by block count  " + "  ".join(row), flush=True)
