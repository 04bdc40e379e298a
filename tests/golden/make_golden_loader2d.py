#!/usr/bin/env python3
"""Golden vectors for the spectrogram dataset selection: runs the reference's
``dataloader_physionet2d.physionet_dataset`` (dataloader_physionet2d.py:9-116) on a synthetic
dataset dictionary in its layout (one image per heart cycle) and records which cycles it keeps.

    python tests/golden/make_golden_loader2d.py        (build container only)
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def synthetic_dataset2d(seed=0, n_rec=60, F=6, W=6):
    rs = np.random.RandomState(seed)
    out = {}
    for split, n in (("train", n_rec), ("test", n_rec // 3)):
        d = {"data": [], "label": [], "frames": [], "wav": [], "sig_qual": []}
        for r in range(n):
            wav = f"{'abcdef'[rs.randint(0, 6)]}{r:04d}"
            label = int(rs.rand() < 0.4)
            qual = int(rs.rand() < 0.85)
            for _ in range(rs.randint(2, 5)):
                d["data"].append(rs.standard_normal((F, W)).astype(np.float32))
                d["label"].append(label)
                d["frames"].append(np.array([0, 1, 2, 3, 4], dtype=np.int64))
                d["wav"].append(wav)
                d["sig_qual"].append(qual)
        out[split] = d
    return out


CONFIGS2D = [dict(n_fraction=1.0, valid=False, seed=4, seed_data=1100001),
             dict(n_fraction=0.5, valid=False, seed=4, seed_data=1100001),
             dict(n_fraction=0.2, valid=False, seed=4, seed_data=7),
             dict(n_fraction=1.0, valid=True, seed=2, seed_data=1100001),
             dict(n_fraction=0.6, valid=True, seed=5, seed_data=11)]


RUN2D_ARGS = dict(dataset="PhysioNet(spec128)", seed_data=1100001, n_fraction=1.0, batch_size=8,
                  num_classes=2, num_channels=1, seed=4, method="durratiomixup", valid=False)
RUN2D_EPOCHS = 2


def record_run2d(dl, ds):
    """The batches of the reference's ``physionet_dataloader(args, ds).run('train', 4)``
    (dataloader_physionet2d.py:137-157) over two epochs seeded as train_model.py:497 does."""
    import argparse
    import torch
    a = argparse.Namespace(**RUN2D_ARGS)
    loader, labels = dl.physionet_dataloader(a, ds).run("train", 4)
    out = {"run_labels": np.asarray(labels), "run_len": np.int64(len(loader))}
    count = 0
    for e in range(RUN2D_EPOCHS):
        torch.manual_seed(a.seed * 635410 + count)
        idx, tgt, wav, data = [], [], [], []
        for d, t, _f, w, _q, i in loader:
            idx.append(i.numpy()); tgt.append(t.numpy()); wav.append(np.array(w)); data.append(d.numpy())
            count += 1
        out[f"run_e{e}_idx"] = np.stack(idx)
        out[f"run_e{e}_target"] = np.stack(tgt)
        out[f"run_e{e}_wav"] = np.stack(wav)
        out[f"run_e{e}_data"] = np.stack(data).astype(np.float32)
    return out


def main():
    importlib.import_module("_ref_import").import_reference()
    dl = importlib.import_module("dataloader_physionet2d")        # the reference's module
    ds = synthetic_dataset2d()
    out = {}
    for i, cfg in enumerate(CONFIGS2D):
        d = dl.physionet_dataset(dataset=ds, dataset_name="PhysioNet(spec128)", seed_data=cfg["seed_data"],
                                 num_classes=2, n_fraction=cfg["n_fraction"], mode="train",
                                 seed=cfg["seed"], method="base", valid=cfg["valid"])
        key = f"cfg{i}"
        out[key + "_train_wav"] = np.array(d.train_wav)
        out[key + "_train_label"] = np.array(d.train_label)
        out[key + "_train_data_sum"] = np.array(d.train_data).reshape(len(d.train_data), -1).sum(1)
        if cfg["valid"]:
            out[key + "_valid_wav"] = np.array(d.test_wav)
            out[key + "_valid_label"] = np.array(d.test_label)
    t = dl.physionet_dataset(dataset=ds, dataset_name="PhysioNet(spec128)", seed_data=1, num_classes=2,
                             n_fraction=None, mode="test", seed=None, method="base", valid=None)
    out["test_wav"] = np.array(t.test_wav)
    item = t[3]
    out["test_item3_shape"] = np.array(item[0].shape)
    out.update(record_run2d(dl, ds))
    path = os.path.join(HERE, "loader2d_selection.npz")
    np.savez_compressed(path, **out)
    print("loader2d_selection.npz:", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
