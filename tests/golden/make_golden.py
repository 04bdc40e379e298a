#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the upstream reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's own ``augmentations.augment`` / ``augmentations2d.augment``
/ ``saliency.get_saliency_maps`` / ``models`` (see _ref_import.py for how the unused
third-party imports are neutralised), feeds them synthetic heart-cycle frames and
stores inputs + outputs as ``.npz``.  Only data is written: no reference source,
bytecode or stub is copied into the repo.

Recorded per case
  x, frames, labels, wav      inputs (synthetic, pcgmix_amd.synthetic.make_batch)
  method, step                args.method, step_counter.count
  fired                       1 if the probability gate let the method run
  same_object                 1 if augment() returned the very input tensor
  y                           augment() output data
  mix                         augment() mix_indices ([] -> empty array)
  target_out                  augment() target_ohe output
  lam                         float64 returned by the reference's get_lambda
  knots                       the np.random.normal draw inside magnitude_warp
  (salopt) grad, sal, disp    raw input gradient, reference saliency maps, and the
                              displacement each optimal_displacement_* call returned
"""
import argparse
import importlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import pcgmix_amd  # noqa: E402  (root shim for the hyphenated package)
from pcgmix_amd import synthetic  # noqa: E402
from _ref_import import import_reference  # noqa: E402


class StepCounter:
    def __init__(self, count):
        self.count = count


def base_args(method, num_channels, batch_size, experiments_dir, model="Potes",
              sample_rate=1000):
    return argparse.Namespace(
        dataset="PhysioNet", model=model, method=method, num_epochs=50,
        batch_size=batch_size, n_fraction=1.0, op="adam", use_sched=True, lr_max=0.01,
        train_balance=True, num_channels=num_channels, grad_clip=0.1, seed_data=1100001,
        valid=False, seed=4, EXPERIMENTS=experiments_dir, num_classes=2,
        sample_rate=sample_rate, depth=0, latent_space=False, classical_space=False)


def run_case(ref, mod, x, frames, labels, wav, method, step, experiments_dir,
             num_classes=2, record_salopt=False):
    """Call the reference's augment() once and capture everything of interest."""
    aug = mod
    rec = {"lam": np.nan, "knots": np.zeros((0,)), "disp_calls": [], "sal": None, "grad": None}

    data = torch.from_numpy(x.copy())
    target_ohe = torch.nn.functional.one_hot(torch.from_numpy(labels), num_classes)
    fr = torch.from_numpy(frames.copy())
    args = base_args(method, x.shape[1], x.shape[0], experiments_dir)

    # --- instrumentation (wrappers around the reference's own functions) ---
    orig_get_lambda = aug.get_lambda
    orig_normal = np.random.normal

    def get_lambda(*a, **k):
        rec["lam"] = float(orig_get_lambda(*a, **k))
        return rec["lam"]

    def normal(*a, **k):
        out = orig_normal(*a, **k)
        rec["knots"] = np.array(out, dtype=np.float64, copy=True)
        return out

    patched = []

    def patch(obj, name, new):
        patched.append((obj, name, getattr(obj, name)))
        setattr(obj, name, new)

    patch(aug, "get_lambda", get_lambda)
    patch(np.random, "normal", normal)
    if record_salopt:
        for fname in ("optimal_displacement_max_envelope", "optimal_displacement_max_sum"):
            orig = getattr(aug, fname)

            def wrapped(s1, s2, lam, _orig=orig):
                d = _orig(s1, s2, lam)
                rec["disp_calls"].append(int(d))
                return d
            patch(aug, fname, wrapped)
        orig_sal = ref.saliency.get_saliency_maps

        def get_sal(args_, device_, data_, target_, frames_, **kw):
            out = orig_sal(args_, device_, data_, target_, frames_, **kw)
            rec["sal"] = np.array(out, copy=True)
            rec["grad"] = data_.grad.detach().clone().numpy()
            return out
        patch(ref.saliency, "get_saliency_maps", get_sal)
    try:
        out = aug.augment(args, data, target_ohe, fr, wav, StepCounter(step), None,
                          torch.device("cpu"), experiments_dir)
    finally:
        for obj, name, old in reversed(patched):
            setattr(obj, name, old)
    y, t_out, mix, cut = out
    assert cut is None
    case = {
        "x": x, "frames": frames, "labels": labels, "wav": np.array(wav),
        "method": np.array(method), "step": np.int64(step),
        "fired": np.int64(len(mix) > 0),
        "same_object": np.int64(y is data),
        "y": y.detach().numpy().copy(),
        "mix": np.asarray(mix, dtype=np.int64),
        "target_out": t_out.detach().numpy().copy(),
        "lam": np.float64(rec["lam"]),
        "knots": rec["knots"],
    }
    if record_salopt:
        # replay the call order of mixup_keepdur_multidim_tensors_salopt to place the
        # recorded displacements: one call per (sample, state) with unequal lengths
        disp = np.zeros((x.shape[0], 4), dtype=np.int64)
        it = iter(rec["disp_calls"])
        for i in range(x.shape[0]):
            f1, f2 = frames[i], frames[case["mix"][i]]
            for k in range(4):
                if (f1[k + 1] - f1[k]) != (f2[k + 1] - f2[k]):
                    disp[i, k] = next(it)
        assert next(it, None) is None
        case.update(disp=disp, sal=rec["sal"].astype(np.float32), grad=rec["grad"])
    return case


def save(name, case):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **case)
    print(f"{name:42s} {os.path.getsize(path) / 1024:8.1f} KiB  fired={int(case.get('fired', -1))}")


def main():
    ref = importlib.import_module("_ref_import").import_reference()
    print("mocked third-party modules:", ref.mocked)
    tmp = tempfile.mkdtemp(prefix="pcgmix_golden_")

    # ---------------- 1D: reference-native and benchmark shapes ----------------
    xa, fa, la, wa = synthetic.make_batch(8, 4, 2500, sample_rate=1000, seed=11)
    xb, fb, lb, wb = synthetic.make_batch(8, 1, 5000, sample_rate=2000, seed=12)
    for tag, (x, f, l, w), cases in (
        ("a8x4x2500", (xa, fa, la, wa), [("durratiomixup", 0), ("durratiomixup", 7),
                                           ("durmixmagwarp(0.2,4)", 3)]),
        ("b8x1x5000", (xb, fb, lb, wb), [("durratiomixup", 1), ("durmixmagwarp(0.2,4)", 5)]),
    ):
        for j, (method, step) in enumerate(cases):
            save(f"mix1d_{tag}_{j}", run_case(ref, ref.augmentations, x, f, l, w, method, step, tmp))

    # ---------------- 1D: small shape, many variants ----------------
    xs, fs_, ls, ws = synthetic.make_batch(12, 2, 640, seed=13, rate_scale=0.4)
    fs_[1] = fs_[0]
    fs_[2] = fs_[0]              # identical cycles -> equal-length states (zero-gap paths)
    fs_[5, 1:] = fs_[4, 1:] + np.array([3, 3, -2, -2])
    ls[:] = [0, 0, 0, 1, 1, 0, 1, 0, 1, 1, 0, 0]
    xs[np.broadcast_to(np.arange(640)[None, None, :] >= fs_[:, 4][:, None, None], xs.shape)] = 0
    variants = [
        ("base", 0), ("durratiomixup", 2), ("durratiomixup", 1234),
        ("durratiomixup+0.5", 0), ("durratiomixup+0.5", 1), ("durratiomixup+0.5", 2),
        ("durratiomixup+0.5", 3), ("durmixmagwarp(0.2,4)+0.5", 4), ("durmixmagwarp(0.2,4)+0.5", 5),
        ("(rand)durratiomixup", 3), ("(rand)durratiomixup", 9), ("(rand)durmixmagwarp(0.2,4)", 500),
        ("(alpha=0.5)durratiomixup", 2), ("(alpha=3)durmixmagwarp(0.1,6)", 4),
        ("(alpha=0.05)durmixmagwarp(0.2,4)", 77), ("durmixmagwarp(0.3,2)", 8),
        ("durmixmagwarp", 6), ("(mixAll)durratiomixup", 2), ("(mixAll)durmixmagwarp(0.2,4)", 3),
        ("(samePCG)durratiomixup", 4), ("(sameDataset)durmixmagwarp(0.2,4)", 6),
    ]
    for j, (method, step) in enumerate(variants):
        save(f"mix1d_s12x2x640_{j:02d}", run_case(ref, ref.augmentations, xs, fs_, ls, ws, method, step, tmp))

    # single-class batch and a class with one member (sample mixes with itself)
    l1 = np.zeros(12, dtype=np.int64)
    save("mix1d_s12x2x640_oneclass", run_case(ref, ref.augmentations, xs, fs_, l1, ws, "durratiomixup", 5, tmp))
    l2 = np.zeros(12, dtype=np.int64); l2[7] = 1
    save("mix1d_s12x2x640_singleton", run_case(ref, ref.augmentations, xs, fs_, l2, ws, "durmixmagwarp(0.2,4)", 5, tmp))

    # ---------------- saliency-guided variants (random-init Potes "baseline") ----------------
    torch.manual_seed(1234)
    potes = ref.models.CNN_potes_TS(num_channels=4, num_classes=2, dataset="PhysioNet")
    sd = {"module." + k: v for k, v in potes.state_dict().items()}
    args0 = base_args("base", 4, 8, tmp)
    exp_dir = ref.utils.experiment_dir(args0)
    os.makedirs(exp_dir, exist_ok=True)
    torch.save(sd, os.path.join(exp_dir, "model.pth"))
    np.savez_compressed(os.path.join(HERE, "potes_state_seed1234.npz"),
                        **{k: v.numpy() for k, v in potes.state_dict().items()})
    for j, (method, step) in enumerate([("(saloptenv)durratiomixup", 2),
                                        ("(saloptsum)durmixmagwarp(0.2,4)", 4),
                                        ("(saloptenv)durmixmagwarp(0.2,4)", 9)]):
        save(f"salopt_a8x4x2500_{j}", run_case(ref, ref.augmentations, xa, fa, la, wa, method, step, tmp,
                                               record_salopt=True))

    # ---------------- 2D durratiomixup on (B,1,128,128) ----------------
    rs = np.random.RandomState(21)
    f2d = synthetic.spec_frames(fb, 148, 5000)
    x2d = rs.standard_normal((8, 1, 128, 128)).astype(np.float32)
    x2d[np.broadcast_to(np.arange(128)[None, None, None, :] >= f2d[:, 4][:, None, None, None], x2d.shape)] = 0
    for j, (method, step) in enumerate([("durratiomixup", 0), ("durratiomixup+0.5", 3), ("durratiomixup", 11)]):
        save(f"mix2d_8x1x128x128_{j}", run_case(ref, ref.augmentations2d, x2d, f2d, lb, wb, method, step, tmp))
    # mask variants fused after the 2D splice (small images keep fixtures small)
    rs = np.random.RandomState(22)
    xm2 = rs.standard_normal((6, 1, 32, 32)).astype(np.float32)
    fm2 = np.array([[0, 3, 9, 12, 25], [0, 4, 11, 15, 30], [0, 2, 8, 10, 20], [0, 5, 12, 16, 32],
                    [0, 3, 10, 13, 27], [0, 4, 9, 12, 22]], dtype=np.int64)
    lm2 = np.array([0, 1, 0, 1, 0, 0], dtype=np.int64)
    wm2 = tuple("abcdef")
    for j, (method, step) in enumerate([("durmixcutout", 1), ("durmixcutout(0.5,0.6)", 2),
                                        ("durmixcutout(0.9,0.9)+0.9", 5), ("durmixtimemask", 3),
                                        ("durmixtimemask(0.7)", 4), ("durmixfreqmask", 6),
                                        ("durmixfreqmask(0.8)", 7), ("durmixfreqmask(0.8)+0.5", 8)]):
        save(f"mask2d_6x1x32x32_{j}", run_case(ref, ref.augmentations2d, xm2, fm2, lm2, wm2, method, step, tmp))

    # ---------------- models: seed-initialised forward logits + soft-target CE ----------------
    out = {}
    xm = torch.from_numpy(xa)
    torch.manual_seed(7)
    m = ref.models.CNN_potes_TS(num_channels=4, num_classes=2, dataset="PhysioNet").eval()
    out["potes_logits"] = m(xm, depth=0, pass_part="second").detach().numpy()
    out["potes_nparams"] = np.int64(sum(p.numel() for p in m.parameters()))
    torch.manual_seed(7)
    m = ref.models.ResNet9(in_channels=4, num_classes=2).eval()
    out["resnet1d_logits"] = m(xm, depth=0, pass_part="second").detach().numpy()
    out["resnet1d_nparams"] = np.int64(sum(p.numel() for p in m.parameters()))
    torch.manual_seed(7)
    m = ref.models2d.ResNet9(num_classes=2).eval()
    out["resnet2d_logits"] = m(torch.from_numpy(x2d[:4]), depth=0, pass_part="second").detach().numpy()
    out["resnet2d_nparams"] = np.int64(sum(p.numel() for p in m.parameters()))
    # CELoss, train_epoch and test_data_accuracy (train_model.py) are recorded by
    # make_golden_train.py -> train_ref.npz.
    np.savez_compressed(os.path.join(HERE, "models_seed7.npz"), x1d=xa, x2d=x2d[:4], **out)
    print("models_seed7.npz written")


if __name__ == "__main__":
    main()
