"""Probe: ResNet9-1D fwd+bwd through torch in three layouts — Conv1d NCL (as shipped), 4-D (B,C,1,L)
contiguous, 4-D channels_last — same parameters, to see what the NCHW<->NHWC transposes and the
BN/pool kernels cost in each."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
import pcgmix_amd
from pcgmix_amd import models
dev = torch.device('cuda:0')
torch.manual_seed(0)
T = int(os.environ.get("T", 5000)); B = 256
m = models.ResNet9(4, 2, linear=models.resnet9_flat_features(T)).to(dev).train()
x = torch.randn(B, 4, T, device=dev)

def block4d(seq, h, cl):
    conv, bn = seq[0], seq[1]
    h = F.conv2d(h, conv.weight.unsqueeze(2), conv.bias, padding=(0, 1))
    h = F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, True, bn.momentum, bn.eps)
    h = F.relu(h)
    if len(seq) > 3:
        h = F.max_pool2d(h, (1, 2))
    return h

def fwd4d(x, cl):
    h = x.unsqueeze(2)
    if cl:
        h = h.contiguous(memory_format=torch.channels_last)
    h = block4d(m.conv2, block4d(m.conv1, h, cl), cl)
    h = block4d(m.res1[1], block4d(m.res1[0], h, cl), cl) + h
    h = block4d(m.conv4, block4d(m.conv3, h, cl), cl)
    h = block4d(m.res2[1], block4d(m.res2[0], h, cl), cl) + h
    h = F.max_pool2d(h, (1, 4))
    return m.linear(h.squeeze(2).flatten(1))

def run(name, f):
    for _ in range(3):
        m.zero_grad(set_to_none=True); f().sum().backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        m.zero_grad(set_to_none=True); out = f(); out.sum().backward()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t0) / 5 * 1e3:7.2f} ms fwd+bwd", flush=True)
    return out.detach()
a = run("Conv1d (B,C,L)", lambda: m(x))
b = run("4-D contiguous", lambda: fwd4d(x, False))
c = run("4-D channels_last", lambda: fwd4d(x, True))
print("max |diff| vs Conv1d:", float((a - b).abs().max()), float((a - c).abs().max()), "scale", float(a.abs().max()))
