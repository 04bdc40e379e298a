#!/usr/bin/env python3
"""Round-3 train-step goldens, recorded by RUNNING the upstream reference's ``train_epoch``
(train_model.py:490-589) on CPU in the build container (same import recipe as
make_golden_train.py; only data is written, no reference source, bytecode or stub).

    python tests/golden/make_golden_train_r3.py        (needs /root/reference)

  train_salopt_ref.npz   BASELINE config 3's method, ``(saloptenv)durmixmagwarp(0.2,4)``, training
      the Potes 1D-CNN (seed 7, dropout p = 0) for 5 steps of 8 cycles.  The frozen saliency model
      is the 'base' run's ``model.pth`` as saliency.py:26-51 loads it (the weights of
      potes_state_seed1234.npz, written with DataParallel's 'module.' prefix into a temporary
      EXPERIMENTS directory).  Per step: loss, lr, partner indices, lambda, the displacement the
      reference chose per (sample, state) (wrappers around optimal_displacement_max_envelope),
      the reference's saliency maps (for the near-tie rule of tests/test_saliency_gpu.py); the
      trained parameters after step 5.
  train_resnet_ref.npz   the reference's ResNet9 in TRAIN mode (batch statistics, running-stat
      updates, backward): ``models.ResNet9(4, 2)`` at (8,4,2500) with ``durmixmagwarp(0.2,4)`` and
      ``models2d.ResNet9(2)`` at (4,1,128,128) with 2D ``durratiomixup``, 3 steps each, Adam +
      OneCycleLR + clip 0.1 as train_model.py:404-410.  Per-step losses and lrs, every BatchNorm
      buffer after step 3 in full, every parameter as a digest (float64 sum, sum of squares, 512
      strided elements: the networks have 2.3 M / 6.6 M parameters).
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import pcgmix_amd  # noqa: E402,F401
from _ref_import import import_reference  # noqa: E402

import train_cases as TC  # noqa: E402


def run_epoch(T, args, model, batches, exp_dir=""):
    opt = torch.optim.Adam(model.parameters(), lr=args.lr_max, weight_decay=args.weight_decay)   # :405
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr_max, total_steps=args.num_steps)  # :410
    ce = T.CELoss(2)
    losses = []

    def criterion(logits_, target_, index_, epoch_, mode_):
        loss_ = ce(logits_, target_)
        losses.append(float(loss_.item()))
        return loss_
    sc = T.step_counter_class()
    mean_loss, acc, lrs = T.train_epoch(args, model, batches, torch.device("cpu"), opt, sched, criterion,
                                        1, sc, None, exp_dir)
    assert sc.count == len(batches) == len(losses)
    return np.asarray(losses, dtype=np.float64), np.asarray(lrs, dtype=np.float64), mean_loss, acc


def salopt_trajectory(ref):
    T, A = ref.train_model, ref.augmentations
    tmp = tempfile.mkdtemp(prefix="pcgmix_golden_salopt_")
    args = TC.salopt_traj_args(tmp)
    # the frozen 'base' model the saliency maps come from
    sd = np.load(os.path.join(HERE, "potes_state_seed1234.npz"))
    base = TC.salopt_traj_args(tmp)
    base.method = "base"
    exp = ref.utils.experiment_dir(base)
    os.makedirs(exp, exist_ok=True)
    torch.save({"module." + k: torch.from_numpy(sd[k]) for k in sd.files}, os.path.join(exp, "model.pth"))
    torch.manual_seed(7)
    model = ref.models.CNN_potes_TS(num_channels=4, num_classes=2, dataset="PhysioNet")
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    rec = {"disp_calls": [], "sal": [], "lam": [], "mix": []}
    patched = []

    def patch(obj, name, new):
        patched.append((obj, name, getattr(obj, name)))
        setattr(obj, name, new)
    orig_env = A.optimal_displacement_max_envelope

    def env(s1, s2, lam):
        d = orig_env(s1, s2, lam)
        rec["disp_calls"][-1].append(int(d))
        return d
    orig_sal = ref.saliency.get_saliency_maps

    def get_sal(*a, **k):
        out = orig_sal(*a, **k)
        rec["sal"].append(np.array(out, dtype=np.float32, copy=True))
        rec["disp_calls"].append([])
        return out
    orig_lam = A.get_lambda

    def get_lambda(*a, **k):
        v = float(orig_lam(*a, **k))
        rec["lam"].append(v)
        return v
    orig_mix = A.get_same_label_mix_indices

    def get_mix(*a, **k):
        v = orig_mix(*a, **k)
        rec["mix"].append(np.asarray(v, dtype=np.int64).copy())
        return v
    patch(A, "optimal_displacement_max_envelope", env)
    patch(ref.saliency, "get_saliency_maps", get_sal)
    patch(A, "get_lambda", get_lambda)
    patch(A, "get_same_label_mix_indices", get_mix)
    batches = TC.salopt_traj_batches()
    try:
        losses, lrs, mean_loss, acc = run_epoch(T, args, model, batches, tmp)
    finally:
        for obj, name, old in reversed(patched):
            setattr(obj, name, old)
    n = TC.SALOPT_STEPS
    assert len(rec["sal"]) == n and len(rec["lam"]) == n and len(rec["mix"]) == n
    disp = np.zeros((n, TC.SALOPT_B, 4), dtype=np.int64)
    for s, b in enumerate(batches):
        frames = b[2].numpy()
        it = iter(rec["disp_calls"][s])
        for i in range(TC.SALOPT_B):
            f1, f2 = frames[i], frames[rec["mix"][s][i]]
            for k in range(4):
                if (f1[k + 1] - f1[k]) != (f2[k + 1] - f2[k]):
                    disp[s, i, k] = next(it)
        assert next(it, None) is None
    out = {"losses": losses, "lrs": lrs, "mean_loss": np.float64(mean_loss), "acc": np.float64(acc),
           "lam": np.asarray(rec["lam"]), "mix": np.stack(rec["mix"]), "disp": disp,
           "sal": np.stack(rec["sal"])}
    for k, v in model.state_dict().items():
        if not k.startswith(("cnn2", "cnn3", "cnn4")):
            out["final." + k] = v.numpy().copy()
    path = os.path.join(HERE, "train_salopt_ref.npz")
    np.savez_compressed(path, **out)
    print(f"train_salopt_ref.npz {os.path.getsize(path) / 1024:.1f} KiB  losses {losses}")


def resnet_trajectories(ref):
    T = ref.train_model
    out = {}
    for tag, args, batches, build in (
            ("r1d", TC.resnet1d_args(), TC.resnet1d_batches(),
             lambda: ref.models.ResNet9(in_channels=4, num_classes=2)),
            ("r2d", TC.resnet2d_args(), TC.resnet2d_batches(),
             lambda: ref.models2d.ResNet9(num_classes=2))):
        torch.manual_seed(7)
        model = build()
        init = {k: v.clone() for k, v in model.state_dict().items()}
        losses, lrs, mean_loss, acc = run_epoch(T, args, model, batches)
        out[f"{tag}_losses"], out[f"{tag}_lrs"] = losses, lrs
        out[f"{tag}_mean_loss"], out[f"{tag}_acc"] = np.float64(mean_loss), np.float64(acc)
        for k, v in model.state_dict().items():
            if "running_" in k or "num_batches" in k:
                out[f"{tag}_buf.{k}"] = v.numpy().copy()
            else:
                out[f"{tag}_par.{k}"] = TC.tensor_digest(v.numpy())
                out[f"{tag}_ini.{k}"] = TC.tensor_digest(init[k].numpy())[:2]
        print(tag, "losses", losses, "lrs", lrs)
    path = os.path.join(HERE, "train_resnet_ref.npz")
    np.savez_compressed(path, **out)
    print(f"train_resnet_ref.npz {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    ref = import_reference(extra=("train_model",))
    which = sys.argv[1:] or ["salopt", "resnet"]
    if "salopt" in which:
        salopt_trajectory(ref)
    if "resnet" in which:
        resnet_trajectories(ref)


if __name__ == "__main__":
    main()
