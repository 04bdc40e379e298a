#!/usr/bin/env python3
"""Golden vectors for saliency-guided mixing of spectrograms (the '(salopt' lines
augmentations2d.py:416-423 with saliency.get_saliency_maps(dim=2), saliency.py:93-113), recorded by
RUNNING the reference's ``augmentations2d.augment`` in the build container.

    python tests/golden/make_golden_salopt2d.py        (needs /root/reference)

The frozen saliency model is the reference's ``models2d.ResNet9`` initialised with
``torch.manual_seed(SEED2D)`` (6.6 M parameters: not stored — this package's models2d.ResNet9 draws
the same weights from the same seed, tests/test_oracle_golden.py::test_model_goldens) and written
as 'model.pth' where ``utils.experiment_dir`` expects the 'base' run.

salopt2d_6x1x128x128_{j}.npz: x, frames, labels, wav, method, step, fired, y, mix, target_out, lam,
sal (the maps ``get_saliency_maps`` returned), disp (what each optimal_displacement_* call
returned, placed per (sample, state)).
"""
import importlib
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
from pcgmix_amd import synthetic  # noqa: E402

SEED2D = 4321
B2D = 6


def inputs():
    """Six 128x128 log-mel-like images with cycle boundaries in columns (zero behind the cycle)."""
    _, fb, lb, wb = synthetic.make_batch(B2D, 1, 5000, sample_rate=2000, seed=31)
    f2d = synthetic.spec_frames(fb, 148, 5000)
    rs = np.random.RandomState(32)
    x = rs.standard_normal((B2D, 1, 128, 128)).astype(np.float32)
    x[np.broadcast_to(np.arange(128)[None, None, None, :] >= f2d[:, 4][:, None, None, None], x.shape)] = 0
    lb = np.array([0, 1, 0, 1, 0, 0], dtype=np.int64)        # both classes, groups of 4 and 2
    return x, f2d, lb, wb


def main():
    mg = importlib.import_module("make_golden")
    ref = importlib.import_module("_ref_import").import_reference()
    tmp = tempfile.mkdtemp(prefix="pcgmix_golden2d_")
    x, f2d, labels, wav = inputs()
    a = mg.base_args("base", 1, B2D, tmp, model="resnet9")
    a.dataset = "PhysioNet(spec128)"
    exp = ref.utils.experiment_dir(a)
    os.makedirs(exp, exist_ok=True)
    torch.manual_seed(SEED2D)
    net = ref.models2d.ResNet9(num_classes=2)
    torch.save({"module." + k: v for k, v in net.state_dict().items()}, os.path.join(exp, "model.pth"))

    orig_base_args = mg.base_args

    def base_args_2d(method, num_channels, batch_size, experiments_dir, **kw):
        b = orig_base_args(method, num_channels, batch_size, experiments_dir, model="resnet9")
        b.dataset = "PhysioNet(spec128)"
        return b
    mg.base_args = base_args_2d
    for j, (method, step) in enumerate([("(saloptenv)durratiomixup", 7), ("(saloptsum)durratiomixup", 5),
                                        ("(saloptenv)durratiomixup+0.5", 9)]):
        case = mg.run_case(ref, ref.augmentations2d, x, f2d, labels, wav, method, step, tmp,
                           record_salopt=True)
        if j:
            case.pop("grad", None)                   # the raw input gradient once (pins the post-
        case.pop("knots", None)                      # processing restatement), the maps every time
        mg.save(f"salopt2d_{B2D}x1x128x128_{j}", case)


if __name__ == "__main__":
    main()
