"""Round 4: cost of MIOpen's deterministic algorithm selection for the FROZEN saliency pass (forward +
backward-data only) of the ResNet9 models at bs 256; prints as it goes.
    python profiles/probes/sal_pass_deterministic_time.py"""
import sys, time, torch
sys.path.insert(0, '.')
import pcgmix_amd
from pcgmix_amd import models, models2d
dev = torch.device('cuda:0')
def run(name, net, x):
    net = net.to(dev).eval()
    for p in net.parameters(): p.requires_grad_(False)
    seed = torch.zeros(x.shape[0], 2, device=dev); seed[:, 0] = 1
    for det in (False, True):
        with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=det):
            ts = []
            for i in range(4):
                xx = x.detach().requires_grad_(True)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                with torch.enable_grad():
                    out = net(xx)
                    (g,) = torch.autograd.grad(out, xx, seed)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
                print(f"{name} deterministic={det} pass {i}: {ts[-1]:.1f} ms", flush=True)
torch.manual_seed(0)
run("resnet9-2d (256,1,128,128)", models2d.ResNet9(num_classes=2), torch.randn(256, 1, 128, 128, device=dev))
run("resnet9-1d (256,4,5000)", models.ResNet9(in_channels=4, num_classes=2, linear=79872), torch.randn(256, 4, 5000, device=dev))
