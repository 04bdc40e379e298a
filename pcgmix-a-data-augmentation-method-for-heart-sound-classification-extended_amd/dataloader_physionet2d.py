"""Dataset selection and GPU-resident loader for the spectrogram path (BASELINE config 4).

Mirrors the reference's ``dataloader_physionet2d.py``: the dataset dictionary holds one
(n_mels, frames) image per heart cycle (``dataset[split]['data']`` is a list of arrays — what
``databuilder.ipynb`` cell 6 writes, here what ``frontend.logmel*`` computes) instead of band
dictionaries, and the selection is the time-series one without its first two steps:

* no signal-quality filter and no class balancing (dataloader_physionet2d.py:27-42 has neither);
* ``n_fraction`` subsetting (``random.Random(seed_data).shuffle`` per label, :43-61) and the
  5-fold cross-validation split (:62-98) exactly as in the time-series loader;
* a batch is ``(spectrogram (B,1,F,W), target, frames, wav, sig_qual, index)`` (:101-116), the
  training loader shuffles with ``drop_last=True`` (:147-152), evaluation uses batch 256
  (:166-169, :186-189).

Selection is checked against the reference's own output (tests/golden/loader2d_selection.npz).
"""
from __future__ import annotations

import random

import numpy as np
import torch

from .dataloader_physionet import ResidentLoader


class physionet_dataset:
    """dataloader_physionet2d.py:9-116 (reference signature)."""

    def __init__(self, dataset, dataset_name, seed_data, num_classes, n_fraction, mode, seed, method,
                 valid):
        self.mode = mode
        if mode == "test":
            t = dataset["test"]
            self.test_data = np.array(t["data"])
            self.test_label = np.array(t["label"])
            self.test_frames = np.array(t["frames"])
            self.test_wav = np.array(t["wav"])
            self.test_sig_qual = np.array(t["sig_qual"])
            return
        if mode not in ("train", "valid"):
            raise ValueError(mode)
        t = dataset["train"]
        data, label, frames = np.array(t["data"]), np.array(t["label"]), np.array(t["frames"])
        wav, qual = np.array(t["wav"]), np.array(t["sig_qual"])

        def keep(idx):
            nonlocal data, label, frames, wav, qual
            data, label, frames, wav, qual = data[idx], label[idx], frames[idx], wav[idx], qual[idx]

        letters = {"a": 0, "b": 1, "c": 2, "d": 3, "e": 4, "f": 5}
        groups = [[] for _ in range(6 * num_classes)]
        seen = set()
        for w, l in zip(wav, label):
            if w not in seen:
                seen.add(w)
                groups[letters[w[0]] + 6 * int(l)].append(w)
        if n_fraction < 1.0:                                        # :43-61
            per_label = []
            for half in (groups[:6], groups[6:]):
                flat = sorted(w for g in half for w in g)
                random.Random(seed_data).shuffle(flat)
                per_label.append(flat)
            n_take = int(np.ceil(n_fraction * len(set(wav)) / 2))
            chosen = set(per_label[0][:n_take]) | set(per_label[1][:n_take])
            keep([i for i, w in enumerate(wav) if w in chosen])
        if valid is True:                                           # :62-98 5-fold CV
            k_folds = 5
            if seed not in range(1, k_folds + 1):
                raise Exception(f"Parameter 'self.seed' (was set to {seed}) must be in "
                                f"{list(range(1, k_folds + 1))} (we are applying {k_folds}-fold-CV)!")
            by_label = ([], [])
            seen = set()
            for w, l in zip(wav, label):
                if w not in seen:
                    seen.add(w)
                    if l in (0, 1):
                        by_label[int(l)].append(w)
            part0 = [by_label[0][i::k_folds] for i in range(k_folds)]
            part1 = [by_label[1][i::k_folds] for i in range(k_folds)]
            folds = [part0[i] + part1[k_folds - i - 1] for i in range(k_folds)]
            held = set(folds[seed - 1])
            iv = [i for i, w in enumerate(wav) if w in held]
            self.test_data, self.test_label, self.test_frames = data[iv], label[iv], frames[iv]
            self.test_wav, self.test_sig_qual = wav[iv], qual[iv]
            rest = {w for f in folds for w in f if w not in held}
            keep([i for i, w in enumerate(wav) if w in rest])
        self.train_data, self.train_label, self.train_frames = data, label, frames
        self.train_wav, self.train_sig_qual = wav, qual

    def arrays(self):
        p = "train" if self.mode == "train" else "test"
        return tuple(getattr(self, f"{p}_{k}") for k in ("data", "label", "frames", "wav", "sig_qual"))

    def __len__(self):
        return len(self.arrays()[0])


class physionet_dataloader:
    """dataloader_physionet2d.py:124-193: ``run('train', seed)`` -> (loader, labels);
    ``run('test'|'valid', None)`` -> loader.  ``args.device`` (optional): where the images stay
    resident ((N,1,128,128) float32: 64 KB per cycle)."""

    def __init__(self, args, dataset):
        self.args, self.dataset = args, dataset

    def _loader(self, ds, batch_size, shuffle, drop_last):
        data, label, frames, wav, qual = ds.arrays()
        data = np.asarray(data, dtype=np.float32)
        if data.ndim != 3:
            raise ValueError("spectrogram dataset: 'data' must be a list of equally sized 2-D images")
        return ResidentLoader(data[:, None, :, :], label, frames, wav, qual, batch_size, shuffle, drop_last,
                              device=getattr(self.args, "device", None))

    def run(self, mode, transform_seed):
        a = self.args
        ds = physionet_dataset(dataset=self.dataset, dataset_name=a.dataset, seed_data=a.seed_data,
                               num_classes=a.num_classes,
                               n_fraction=a.n_fraction if mode != "test" else None, mode=mode,
                               seed=a.seed if mode != "test" else None, method=a.method,
                               valid=(a.valid if mode == "train" else (True if mode == "valid" else None)))
        if mode == "train":
            random.seed(transform_seed)                    # :145-146
            torch.manual_seed(transform_seed)
            return self._loader(ds, a.batch_size, True, True), np.asarray(ds.train_label)
        if mode in ("test", "valid"):
            return self._loader(ds, 256, False, False)
        raise ValueError(mode)
