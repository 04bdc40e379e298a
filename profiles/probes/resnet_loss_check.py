"""Probe: are the large losses bench.py reports for its ResNet9 legs (tens after ~13 steps) the
network's true behaviour at the reference's hyper-parameters, or an artefact of the HIP path?

Runs the bench leg's exact setup (synthetic batch seed 100, torch seed 4, Adam + OneCycleLR with
max_lr 0.01 over 14 steps, clip 0.1, durmixmagwarp(0.2,4) for 1D / durratiomixup for 2D, bs from
argv) twice: through this package's execution path (channels_last, fused BN kernels, ClipAdam,
fp32) and through the plain torch modules in FLOAT64 (same initial weights, same augmented
batches), and prints both loss trajectories.

    python profiles/probes/resnet_loss_check.py 1d 256 > gpurun_out/resnet_loss_1d.json
"""
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pcgmix_amd import augmentations, augmentations2d, frontend, models, synthetic, train_model as tm  # noqa: E402

kind, B = sys.argv[1], int(sys.argv[2])
steps = 13
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
if kind == "1d":
    C, T, method = 4, 5000, "durmixmagwarp(0.2,4)"
    args = bench.TrainArgs(method, "resnet9", B, C, T, steps + 1)
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=100)
    data = torch.from_numpy(x).to(dev)
    fr = torch.from_numpy(frames)
else:
    method = "durratiomixup"
    args = bench.TrainArgs(method, "resnet9", B, 1, 5000, steps + 1)
    args.dataset = "PhysioNet(spec128)"
    x, frames, labels, wav = synthetic.make_batch(B, 1, 5000, sample_rate=2000, seed=3)
    data, fspec = frontend.logmel(torch.from_numpy(x).to(dev), frames)
    fr = torch.from_numpy(fspec)
target = torch.from_numpy(labels)
tgt = torch.nn.functional.one_hot(target, 2).to(dev)

torch.manual_seed(4)
net = tm.build_model(args).to(dev).train()
ref = copy.deepcopy(net).double()
ref.nhwc = False                       # plain Conv/BatchNorm/ReLU/MaxPool modules
models.FUSED_BN = True
opt, sched = tm.make_optimizer(args, net)
ropt = torch.optim.Adam([p for p in ref.parameters() if p.requires_grad], lr=args.lr_max,
                        weight_decay=args.weight_decay)
rsched = torch.optim.lr_scheduler.OneCycleLR(ropt, max_lr=args.lr_max, total_steps=args.num_steps)
aug = augmentations2d if kind == "2d" else augmentations
out = {"kind": kind, "batch": B, "method": method, "hip_fp32": [], "torch_fp64": [], "lr": [],
       "logit_abs_max_fp64": []}


class SC:
    count = 0


for step in range(steps):
    SC.count = step
    y, _, _, _ = aug.augment(args, data, tgt, fr, wav, SC, None, dev, "", host_labels=labels)
    out["lr"].append(opt.param_groups[0]["lr"])
    # HIP path
    lo = net(y, depth=0, pass_part="second")
    loss = tm.CELoss(2)(lo, tgt)
    loss.backward()
    opt.step(); opt.zero_grad(set_to_none=True); sched.step()
    # float64 reference on the same augmented batch
    lo64 = ref(y.double(), depth=0, pass_part="second")
    l64 = -(torch.log_softmax(lo64, 1) * tgt).sum(1).mean()
    l64.backward()
    torch.nn.utils.clip_grad_value_([p for p in ref.parameters() if p.grad is not None], args.grad_clip)
    ropt.step(); ropt.zero_grad(set_to_none=True); rsched.step()
    out["hip_fp32"].append(float(loss))
    out["torch_fp64"].append(float(l64))
    out["logit_abs_max_fp64"].append(float(lo64.abs().max()))
    print(f"step {step:2d} lr {out['lr'][-1]:.5f} loss hip {float(loss):10.4f}  fp64 {float(l64):10.4f}",
          file=sys.stderr, flush=True)
print(json.dumps(out))
