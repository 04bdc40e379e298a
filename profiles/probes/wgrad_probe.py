"""Probe: the dimreduc weight-gradient GEMM (20x256 . 256x19968) in its two orientations."""
import torch
dev = torch.device('cuda:0')
h = torch.randn(256, 19968, device=dev)
gz = torch.randn(256, 20, device=dev)
def t(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print('gz.t() @ h        ', t(lambda: gz.t().mm(h)))
print('(h.t() @ gz).t()  ', t(lambda: h.t().mm(gz).t()))
print('einsum bo,bj->oj  ', t(lambda: torch.einsum('bo,bj->oj', gz, h)))
gzp = torch.zeros(256, 32, device=dev); gzp[:, :20] = gz
print('padded 32 gz.t()@h', t(lambda: gzp.t().mm(h)))
W = torch.randn(20, 19968, device=dev)
b = torch.randn(20, device=dev)
print('dgrad gz @ W      ', t(lambda: gz.mm(W)))
print('fwd  h @ W.t() + b', t(lambda: torch.addmm(b, h, W.t())))
lin = torch.nn.Linear(19968, 20).to(dev)
hh = h.clone().requires_grad_(True)
def fb():
    z = lin(hh); z.backward(gz)
print('Linear fwd+bwd    ', t(fb))
def splitk(nc):
    kc = 19968 // nc
    Wt = W.view(20, nc, kc).permute(1, 2, 0).contiguous()          # (nc, kc, 20), prepared once
    def f():
        return torch.bmm(h.view(256, nc, kc).transpose(0, 1), Wt).sum(0) + b
    return f
for nc in (13, 26, 39, 78, 156):
    f = splitk(nc)
    ref = torch.addmm(b, h, W.t())
    print(f'split-K fwd nc={nc:4d}', t(f), float((f() - ref).abs().max()))
