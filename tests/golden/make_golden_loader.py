#!/usr/bin/env python3
"""Golden vectors for the dataset selection logic: runs the reference's
``dataloader_physionet.physionet_dataset`` (dataloader_physionet.py:9-149) on a synthetic dataset
dictionary in the reference's own layout and records which heart cycles it keeps.

    python tests/golden/make_golden_loader.py        (build container only)
"""
import argparse
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)


def synthetic_dataset(seed=0, n_rec=72, T=48):
    """{'train'|'test': {'data': {band: [arrays]}, 'label', 'frames', 'wav', 'sig_qual'}} as
    utils.file2dict returns it (utils.py:181-186, databuilder.ipynb cell 25)."""
    rs = np.random.RandomState(seed)
    bands = ["25-45", "45-80", "80-200", "200-400", "25-400"]
    out = {}
    for split, n in (("train", n_rec), ("test", n_rec // 3)):
        d = {"data": {b: [] for b in bands}, "label": [], "frames": [], "wav": [], "sig_qual": []}
        for r in range(n):
            letter = "abcdef"[rs.randint(0, 6)]
            wav = f"{letter}{r:04d}"
            label = int(rs.rand() < 0.35)
            qual = int(rs.rand() < 0.85)
            for _ in range(rs.randint(2, 6)):
                lens = rs.randint(2, 9, size=4)
                for b in bands:
                    d["data"][b].append(rs.standard_normal(T).astype(np.float32))
                d["label"].append(label)
                d["frames"].append(np.concatenate([[0], np.cumsum(lens)]).astype(np.int64))
                d["wav"].append(wav)
                d["sig_qual"].append(qual)
        out[split] = d
    return out


CONFIGS = [dict(train_balance=True, n_fraction=1.0, valid=False, seed=4, seed_data=1100001),
           dict(train_balance=False, n_fraction=1.0, valid=False, seed=4, seed_data=1100001),
           dict(train_balance=True, n_fraction=0.5, valid=False, seed=4, seed_data=1100001),
           dict(train_balance=True, n_fraction=0.1, valid=False, seed=4, seed_data=7),
           dict(train_balance=False, n_fraction=0.3, valid=False, seed=4, seed_data=3),
           dict(train_balance=True, n_fraction=1.0, valid=True, seed=1, seed_data=1100001),
           dict(train_balance=True, n_fraction=1.0, valid=True, seed=5, seed_data=1100001),
           dict(train_balance=True, n_fraction=0.5, valid=True, seed=3, seed_data=11)]


RUN_ARGS = dict(dataset="PhysioNet", seed_data=1100001, n_fraction=1.0, batch_size=16, num_classes=2,
                sample_rate=1000, num_channels=4, seed=4, train_balance=True, method="durratiomixup",
                valid=False, classical_space=False)
RUN_EPOCHS = 2


def record_run(dl, ds):
    """What the reference's ``physionet_dataloader(args, ds).run('train', 4)`` (a real
    ``DataLoader(shuffle=True, drop_last=True)``, dataloader_physionet.py:204-229) yields over two
    epochs seeded as ``train_epoch`` seeds them (train_model.py:497): per batch the dataset indices,
    labels, frames, recording ids and the cycles themselves.  ``torch_audiomentations`` is not
    installed; the loader's only use of it is ``Compose([Identity()])`` (:191-194), served here as
    the identity function it is."""
    import torch
    dl.torch_audiomentations.Compose = lambda transforms: (lambda x, sample_rate=None: x)
    a = argparse.Namespace(**RUN_ARGS)
    loader, labels = dl.physionet_dataloader(a, ds).run("train", 4)
    out = {"run_labels": np.asarray(labels), "run_len": np.int64(len(loader))}
    count = 0
    for e in range(RUN_EPOCHS):
        torch.manual_seed(a.seed * 635410 + count)                    # train_model.py:497
        idx, tgt, fr, wav, data = [], [], [], [], []
        for d, t, f, w, _q, i in loader:
            idx.append(i.numpy()); tgt.append(t.numpy()); fr.append(f.numpy())
            wav.append(np.array(w)); data.append(d.numpy())
            count += 1
        out[f"run_e{e}_idx"] = np.stack(idx)
        out[f"run_e{e}_target"] = np.stack(tgt)
        out[f"run_e{e}_frames"] = np.stack(fr)
        out[f"run_e{e}_wav"] = np.stack(wav)
        out[f"run_e{e}_data"] = np.stack(data).astype(np.float32)
    test = dl.physionet_dataloader(a, ds).run("test", None)
    out["run_test_wav"] = np.concatenate([np.array(b[3]) for b in test])
    out["run_test_data_sum"] = np.concatenate([b[0].numpy().reshape(len(b[1]), -1).sum(1) for b in test])
    return out


def main():
    ref = importlib.import_module("_ref_import").import_reference()
    dl = importlib.import_module("dataloader_physionet")        # the reference's module
    ds = synthetic_dataset()
    out = {}
    for i, cfg in enumerate(CONFIGS):
        for ch in (1, 4):
            a = argparse.Namespace(**cfg)
            d = dl.physionet_dataset(arguments=a, dataset=ds, dataset_name="PhysioNet",
                                     seed_data=cfg["seed_data"], num_classes=2,
                                     n_fraction=cfg["n_fraction"], mode="train", transform=None,
                                     sample_rate=1000, num_channels=ch, seed=cfg["seed"],
                                     train_balance=cfg["train_balance"], method="base",
                                     valid=cfg["valid"], classical_space=False)
            key = f"cfg{i}_ch{ch}"
            out[key + "_train_wav"] = np.array(d.train_wav)
            out[key + "_train_label"] = np.array(d.train_label)
            out[key + "_train_frames"] = np.array(d.train_frames)
            out[key + "_train_data_sum"] = np.array(d.train_data).reshape(len(d.train_data), -1).sum(1)
            out[key + "_train_data_shape"] = np.array(np.array(d.train_data).shape)
            if cfg["valid"]:
                out[key + "_valid_wav"] = np.array(d.test_wav)
                out[key + "_valid_label"] = np.array(d.test_label)
    t = dl.physionet_dataset(arguments=argparse.Namespace(), dataset=ds, dataset_name="PhysioNet",
                             seed_data=1, num_classes=2, n_fraction=1.0, mode="test", transform=None,
                             sample_rate=1000, num_channels=4, seed=4, train_balance=True,
                             method="base", valid=False, classical_space=False)
    out["test_wav"] = np.array(t.test_wav)
    out["test_data_shape"] = np.array(np.array(t.test_data).shape)
    out.update(record_run(dl, ds))
    np.savez_compressed(os.path.join(HERE, "loader_selection.npz"), **out)
    print("loader_selection.npz:", os.path.getsize(os.path.join(HERE, "loader_selection.npz")) // 1024, "KiB",
          {k: v.shape for k, v in list(out.items())[:4]})


if __name__ == "__main__":
    main()
