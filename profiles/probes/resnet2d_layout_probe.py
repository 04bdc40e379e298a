"""Probe: ResNet9-2D fwd+bwd (bs=256, 1x128x128) with contiguous vs channels_last activations."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pcgmix_amd
from pcgmix_amd import models2d
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = models2d.ResNet9(2).to(dev).train()
x = torch.randn(256, 1, 128, 128, device=dev)
def run(name, f):
    for _ in range(2):
        m.zero_grad(set_to_none=True); f().sum().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        m.zero_grad(set_to_none=True); out = f(); out.sum().backward()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.perf_counter() - t0) / 4 * 1e3:7.2f} ms fwd+bwd", flush=True)
    return out.detach()
a = run("contiguous", lambda: m(x))
xc = x.contiguous(memory_format=torch.channels_last)
b = run("channels_last input", lambda: m(xc))
m = m.to(memory_format=torch.channels_last)
c = run("channels_last input + weights", lambda: m(xc))
print("max |diff|:", float((a - b).abs().max()), float((a - c).abs().max()), "scale", float(a.abs().max()))
