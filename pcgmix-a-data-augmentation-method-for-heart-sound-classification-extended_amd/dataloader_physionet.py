"""Dataset selection and a GPU-resident loader for the time-series path.

Mirrors the reference's ``dataloader_physionet.py``:

* ``physionet_dataset`` — which heart cycles a run trains / validates / tests on
  (dataloader_physionet.py:9-149): drop noisy recordings (sig_qual == 0), split recordings into
  the 12 (sub-dataset a-f x class) groups, optional class balancing with
  ``random.Random(18).sample`` per group, optional ``n_fraction`` subsetting with
  ``random.Random(seed_data).shuffle``, optional 5-fold cross-validation split.  Pure host
  index logic, reproduced decision for decision and checked against the reference's own output
  (tests/golden/loader_selection.npz).
* ``physionet_dataloader(args, dataset).run(mode, transform_seed)`` — same call as the reference
  (:174-276).  What comes back is not a ``torch.utils.data.DataLoader`` over host memory but a
  ``ResidentLoader``: the selected cycles live in HBM ((N, C, T) float32 — the whole PhysioNet
  set is a few hundred MB against 288 GB), every batch is one device gather, and the batch order
  is exactly the one ``DataLoader(shuffle=True, drop_last=True)`` would draw from torch's global
  generator (dataloader_physionet.py:224-227), so ``train_epoch``'s per-epoch ``manual_seed``
  (train_model.py:497) fixes the same shuffles.
* ``file2dict`` / ``dict2file`` — the reference's dataset container (zlib-compressed pickle of
  ``{'train'|'test': {'data': {band: [arrays]}, 'label', 'frames', 'wav', 'sig_qual'}}``,
  utils.py:172-186), read through a restricted unpickler (plain containers and numpy arrays only).
"""
from __future__ import annotations

import io
import pickle
import random
import zlib
from typing import Optional

import numpy as np
import torch

BANDS_4 = ("25-45", "45-80", "80-200", "200-400")


def dict2file(dataset: dict, path: str) -> None:
    """utils.py:172-179."""
    buf = io.BytesIO()
    pickle.dump(dataset, buf)
    with open(path, "wb") as fd:
        fd.write(zlib.compress(buf.getbuffer()))


class _DatasetUnpickler(pickle.Unpickler):
    """The dataset container holds dicts, lists, strings, ints and numpy arrays
    (databuilder.ipynb cell 25) — nothing else is allowed to be constructed: an unpickler that
    resolves only numpy's array/dtype/scalar reconstructors refuses every other global, so a
    tampered file cannot run code through ``__reduce__``."""
    _ALLOWED = {
        ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
        ("numpy", "ndarray"), ("numpy", "dtype"),
        ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
        ("builtins", "list"), ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "set"),
        ("collections", "OrderedDict"),
    }

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"dataset container may not reference {module}.{name}")


def file2dict(path: str) -> dict:
    """utils.py:181-186: zlib-compressed pickle -> dataset dictionary, through an unpickler
    restricted to the types the container holds (object-dtype arrays are refused by numpy's own
    reconstruction only if they carry foreign classes — those globals are not on the list)."""
    with open(path, "rb") as fd:
        return _DatasetUnpickler(io.BytesIO(zlib.decompress(fd.read()))).load()


def _stack_bands(split: dict, num_channels: int) -> np.ndarray:
    if num_channels == 1:
        return np.array(split["data"]["25-400"])
    if num_channels == 4:
        return np.array([np.vstack(rows) for rows in zip(*(split["data"][b] for b in BANDS_4))])
    raise ValueError("num_channels must be 1 or 4")


class physionet_dataset:
    """Selection of dataloader_physionet.py:27-149; attributes carry the reference's names
    (``train_data``, ``train_label``, ``train_frames``, ``train_wav``, ``train_sig_qual`` and the
    ``test_*`` counterparts)."""

    def __init__(self, arguments, dataset, dataset_name, seed_data, num_classes, n_fraction, mode,
                 transform, sample_rate, num_channels, seed, train_balance, method, valid,
                 classical_space=False):
        self.mode, self.num_channels = mode, num_channels
        if classical_space:
            raise NotImplementedError("classical_space is outside the PCGmix hot path")
        if mode == "test":
            t = dataset["test"]
            self.test_data = _stack_bands(t, num_channels)
            self.test_label = np.array(t["label"])
            self.test_frames = np.array(t["frames"])
            self.test_wav = np.array(t["wav"])
            self.test_sig_qual = np.array(t["sig_qual"])
            return
        if mode not in ("train", "valid"):
            raise ValueError(mode)
        t = dataset["train"]
        data = _stack_bands(t, num_channels)
        label, frames = np.array(t["label"]), np.array(t["frames"])
        wav, qual = np.array(t["wav"]), np.array(t["sig_qual"])

        def keep(idx):
            nonlocal data, label, frames, wav, qual
            data, label, frames, wav, qual = data[idx], label[idx], frames[idx], wav[idx], qual[idx]

        keep(np.nonzero(qual)[0])                                   # :60-65 noisy recordings out
        # 12 groups: sub-dataset letter (a-f) x class, recordings in order of first appearance
        letters = {"a": 0, "b": 1, "c": 2, "d": 3, "e": 4, "f": 5}
        groups = [[] for _ in range(6 * num_classes)]
        seen = set()
        for w, l in zip(wav, label):
            if w not in seen:
                seen.add(w)
                groups[letters[w[0]] + 6 * int(l)].append(w)
        if train_balance:                                           # :76-91 (two classes)
            caps = [min(len(groups[i]), len(groups[i + 6])) for i in range(6)] * 2
            tbal_seed = getattr(arguments, "true_seed", 18)
            groups = [random.Random(tbal_seed).sample(g, c) for g, c in zip(groups, caps)]
            chosen = {w for g in groups for w in g}
            keep([i for i, w in enumerate(wav) if w in chosen])
        if n_fraction < 1.0:                                        # :92-110
            per_label = []
            for half in (groups[:6], groups[6:]):
                flat = sorted(w for g in half for w in g)
                random.Random(seed_data).shuffle(flat)
                per_label.append(flat)
            n_take = int(np.ceil(n_fraction * len(set(wav)) / 2))
            chosen = set(per_label[0][:n_take]) | set(per_label[1][:n_take])
            keep([i for i, w in enumerate(wav) if w in chosen])
        if valid is True:                                           # :111-149 5-fold CV
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
        """(data, label, frames, wav, sig_qual) of this mode's split."""
        p = "train" if self.mode == "train" else "test"
        return tuple(getattr(self, f"{p}_{k}") for k in ("data", "label", "frames", "wav", "sig_qual"))

    def __len__(self):
        return len(self.arrays()[0])


class ResidentLoader:
    """Batches of (data, target, frames, wav, sig_qual, index) like the reference's DataLoader
    yields them (dataloader_physionet.py:151-172), with ``data`` gathered on ``device`` from a
    tensor that stays resident there; the small per-sample fields stay on the host (they feed the
    host prologue of ``augment``).  ``shuffle=True`` draws the permutation exactly as
    ``RandomSampler`` does (a fresh generator seeded from torch's global RNG, then ``randperm``)."""

    def __init__(self, data, label, frames, wav, sig_qual, batch_size: int, shuffle: bool,
                 drop_last: bool, device: Optional[torch.device] = None):
        data = np.asarray(data, dtype=np.float32)
        if data.ndim == 2:                                # one channel: (N,T) -> (N,1,T), :155-156
            data = data[:, None, :]
        self.data = torch.from_numpy(np.ascontiguousarray(data))
        if device is not None:
            self.data = self.data.to(device)
        self.label = torch.from_numpy(np.asarray(label, dtype=np.int64))
        self.frames = torch.from_numpy(np.asarray(frames, dtype=np.int64))
        self.wav = [str(w) for w in wav]
        self.sig_qual = torch.from_numpy(np.asarray(sig_qual, dtype=np.int64))
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.dataset = range(len(self.label))             # len(loader.dataset), train_model.py:390

    def __len__(self):
        n = len(self.label)
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        n = len(self.label)
        if self.shuffle:
            # DataLoader.__iter__ first draws its worker base seed from the global generator
            # (_BaseDataLoaderIter.__init__), then RandomSampler.__iter__ draws the seed of the
            # permutation's private generator: two draws, in this order
            torch.empty((), dtype=torch.int64).random_()
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
            gen = torch.Generator()
            gen.manual_seed(seed)
            order = torch.randperm(n, generator=gen)
        else:
            order = torch.arange(n)
        for i in range(len(self)):
            idx = order[i * self.batch_size:(i + 1) * self.batch_size]
            dev_idx = idx.to(self.data.device, non_blocking=True)
            yield (self.data.index_select(0, dev_idx), self.label[idx], self.frames[idx],
                   tuple(self.wav[j] for j in idx.tolist()), self.sig_qual[idx], idx)


class physionet_dataloader:
    """dataloader_physionet.py:174-276: ``run('train', seed)`` -> (loader, labels);
    ``run('test'|'valid', None)`` -> loader.  ``args.device`` (optional) selects where the
    dataset is kept resident; evaluation uses batch 1000 without shuffling (:245-276)."""

    def __init__(self, args, dataset):
        self.args, self.dataset = args, dataset

    def _select(self, mode):
        a = self.args
        return physionet_dataset(arguments=a, dataset=self.dataset, dataset_name=a.dataset,
                                 seed_data=a.seed_data, num_classes=a.num_classes,
                                 n_fraction=a.n_fraction, mode=mode, transform=None,
                                 sample_rate=a.sample_rate, num_channels=a.num_channels, seed=a.seed,
                                 train_balance=a.train_balance, method=a.method, valid=a.valid,
                                 classical_space=getattr(a, "classical_space", False))

    def run(self, mode, transform_seed):
        device = getattr(self.args, "device", None)
        ds = self._select(mode)
        if mode == "train":
            random.seed(transform_seed)                    # :221-222
            torch.manual_seed(transform_seed)
            loader = ResidentLoader(*ds.arrays(), batch_size=self.args.batch_size, shuffle=True,
                                    drop_last=True, device=device)
            return loader, np.asarray(ds.train_label)
        if mode in ("test", "valid"):
            return ResidentLoader(*ds.arrays(), batch_size=1000, shuffle=False, drop_last=False,
                                  device=device)
        raise ValueError(mode)
