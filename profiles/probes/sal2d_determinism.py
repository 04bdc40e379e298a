import sys, os, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); sys.path.insert(0, 'tests/golden')
import pcgmix_amd
from pcgmix_amd import saliency, models2d
from conftest import golden_files, load_golden
from make_golden_salopt2d import SEED2D
dev = torch.device('cuda:0')
g = load_golden(golden_files("salopt2d_")[0])
torch.manual_seed(SEED2D)
net = models2d.ResNet9(num_classes=2).to(dev).eval()
for p in net.parameters(): p.requires_grad_(False)
data = torch.from_numpy(g["x"]).to(dev)
tgt = torch.nn.functional.one_hot(torch.from_numpy(g["labels"]), 2).to(dev)
class A: method="(saloptenv)durratiomixup"
for det in (False, True):
    maps = []
    with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=det):
        for i in range(4):
            sal = saliency.get_saliency_maps(A, dev, data, tgt, torch.from_numpy(g["frames"]), dim=2, model_sal=net)
            maps.append(sal.cpu().numpy())
    print("deterministic", det, "eps vs reference per run:", [float(np.abs(m - g["sal"]).max()) for m in maps],
          "run-to-run:", [float(np.abs(m - maps[0]).max()) for m in maps[1:]])
grad_ref = g["grad"]
from pcgmix_amd import saliency as S
gr = S.input_gradient(net, data, tgt).cpu().numpy()
print("raw grad rel err:", float(np.abs(gr - grad_ref).max() / np.abs(grad_ref).max()))
d = np.abs(gr - grad_ref)
b, c, f, w = np.unravel_index(np.argmax(d), d.shape)
print("max abs err at", (b, c, f, w), "f4 of that sample", g["frames"][b][4], "values", gr[b, c, f, w], grad_ref[b, c, f, w])
for bb in range(len(g["frames"])):
    f4 = int(g["frames"][bb][4])
    inside = np.abs(gr[bb, :, :, :f4] - grad_ref[bb, :, :, :f4]).max() / np.abs(grad_ref[bb]).max()
    outside = np.abs(gr[bb, :, :, f4:] - grad_ref[bb, :, :, f4:]).max() / np.abs(grad_ref[bb]).max() if f4 < 128 else 0.0
    print(f"sample {bb}: rel err inside cycle {inside:.2e}, in the zero padding {outside:.2e}")
with torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True):
    gd = S.input_gradient(net, data, tgt).cpu().numpy()
print("deterministic: raw grad rel err", float(np.abs(gd - grad_ref).max() / np.abs(grad_ref).max()))
