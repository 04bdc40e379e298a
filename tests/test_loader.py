"""Dataset selection against the reference's own output, the dataset container round trip, and
the resident loader's batch order against a real torch DataLoader."""
import argparse
import os
import sys

import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import dataloader_physionet as dl
from conftest import GOLDEN

sys.path.insert(0, GOLDEN)
from make_golden_loader import CONFIGS, synthetic_dataset  # noqa: E402  (data generator only)


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(GOLDEN, "loader_selection.npz"))


@pytest.mark.parametrize("i", range(len(CONFIGS)))
@pytest.mark.parametrize("ch", [1, 4])
def test_selection_matches_reference(i, ch, golden):
    cfg = CONFIGS[i]
    ds = synthetic_dataset()
    d = dl.physionet_dataset(arguments=argparse.Namespace(**cfg), dataset=ds, dataset_name="PhysioNet",
                             seed_data=cfg["seed_data"], num_classes=2, n_fraction=cfg["n_fraction"],
                             mode="train", transform=None, sample_rate=1000, num_channels=ch,
                             seed=cfg["seed"], train_balance=cfg["train_balance"], method="base",
                             valid=cfg["valid"])
    k = f"cfg{i}_ch{ch}"
    assert np.array_equal(d.train_wav, golden[k + "_train_wav"])           # same cycles, same order
    assert np.array_equal(d.train_label, golden[k + "_train_label"])
    assert np.array_equal(d.train_frames, golden[k + "_train_frames"])
    assert list(d.train_data.shape) == golden[k + "_train_data_shape"].tolist()
    assert np.allclose(d.train_data.reshape(len(d.train_data), -1).sum(1), golden[k + "_train_data_sum"])
    if cfg["valid"]:
        assert np.array_equal(d.test_wav, golden[k + "_valid_wav"])
        assert np.array_equal(d.test_label, golden[k + "_valid_label"])


def test_test_split_and_bad_fold(golden):
    ds = synthetic_dataset()
    t = dl.physionet_dataset(argparse.Namespace(), ds, "PhysioNet", 1, 2, 1.0, "test", None, 1000, 4,
                             4, True, "base", False)
    assert np.array_equal(t.test_wav, golden["test_wav"])
    assert list(t.test_data.shape) == golden["test_data_shape"].tolist()
    with pytest.raises(Exception):
        dl.physionet_dataset(argparse.Namespace(), ds, "PhysioNet", 1, 2, 1.0, "train", None, 1000, 4,
                             7, True, "base", True)


def test_container_round_trip(tmp_path):
    ds = synthetic_dataset(seed=3, n_rec=6)
    path = str(tmp_path / "set.dat")
    dl.dict2file(ds, path)
    back = dl.file2dict(path)
    assert back["train"]["wav"] == ds["train"]["wav"]
    assert np.array_equal(back["test"]["data"]["25-400"][2], ds["test"]["data"]["25-400"][2])


def test_resident_loader_order_equals_torch_dataloader():
    """Same global seed -> the same shuffled batches as DataLoader(shuffle=True, drop_last=True)."""
    from torch.utils.data import DataLoader, TensorDataset
    n, bs = 103, 8
    x = np.arange(n * 2 * 5, dtype=np.float32).reshape(n, 2, 5)
    lab = np.arange(n) % 2
    fr = np.tile(np.array([0, 1, 2, 3, 4]), (n, 1))
    loader = dl.ResidentLoader(x, lab, fr, [f"w{i}" for i in range(n)], np.ones(n), bs, True, True)
    ref = DataLoader(TensorDataset(torch.arange(n)), batch_size=bs, shuffle=True, drop_last=True)
    for epoch in range(3):
        torch.manual_seed(4 * 635410 + epoch * 12)          # train_model.py:497
        want = [b[0].tolist() for b in ref]
        torch.manual_seed(4 * 635410 + epoch * 12)
        got = [b[5].tolist() for b in loader]
        assert got == want and len(got) == n // bs
    batch = next(iter(loader))
    assert batch[0].shape == (bs, 2, 5) and batch[2].shape == (bs, 5) and len(batch[3]) == bs
    assert torch.equal(batch[0], torch.from_numpy(x)[batch[5]])


def test_dataloader_run_interface():
    ds = synthetic_dataset()
    a = argparse.Namespace(dataset="PhysioNet", seed_data=1100001, n_fraction=1.0, batch_size=16,
                           num_classes=2, sample_rate=1000, num_channels=4, seed=4,
                           train_balance=True, method="durratiomixup", valid=False)
    loader, labels = dl.physionet_dataloader(a, ds).run("train", 4)
    assert len(labels) == len(loader.dataset) and len(loader) == len(labels) // 16
    data, target, frames, wav, qual, idx = next(iter(loader))
    assert data.shape[1:] == (4, 48) and data.dtype == torch.float32 and frames.dtype == torch.int64
    test = dl.physionet_dataloader(a, ds).run("test", None)
    assert sum(len(b[1]) for b in test) == len(test.dataset)


# ---- spectrogram datasets (dataloader_physionet2d.py) -------------------------------------------
from make_golden_loader2d import CONFIGS2D, synthetic_dataset2d  # noqa: E402  (data generator only)


@pytest.mark.parametrize("i", range(len(CONFIGS2D)))
def test_selection2d_matches_reference(i):
    from pcgmix_amd import dataloader_physionet2d as dl2
    g = np.load(os.path.join(GOLDEN, "loader2d_selection.npz"))
    cfg = CONFIGS2D[i]
    d = dl2.physionet_dataset(dataset=synthetic_dataset2d(), dataset_name="PhysioNet(spec128)",
                              seed_data=cfg["seed_data"], num_classes=2, n_fraction=cfg["n_fraction"],
                              mode="train", seed=cfg["seed"], method="base", valid=cfg["valid"])
    k = f"cfg{i}"
    assert np.array_equal(d.train_wav, g[k + "_train_wav"])
    assert np.array_equal(d.train_label, g[k + "_train_label"])
    assert np.allclose(d.train_data.reshape(len(d.train_data), -1).sum(1), g[k + "_train_data_sum"])
    if cfg["valid"]:
        assert np.array_equal(d.test_wav, g[k + "_valid_wav"])
        assert np.array_equal(d.test_label, g[k + "_valid_label"])


def test_dataloader2d_run_interface():
    from pcgmix_amd import dataloader_physionet2d as dl2
    g = np.load(os.path.join(GOLDEN, "loader2d_selection.npz"))
    ds = synthetic_dataset2d()
    a = argparse.Namespace(dataset="PhysioNet(spec128)", seed_data=1100001, n_fraction=1.0, batch_size=8,
                           num_classes=2, num_channels=1, seed=4, method="durratiomixup", valid=False)
    loader, labels = dl2.physionet_dataloader(a, ds).run("train", 4)
    assert len(labels) == len(loader.dataset) and len(loader) == len(labels) // 8
    data, target, frames, wav, qual, idx = next(iter(loader))
    assert data.shape == (8, 1, 6, 6) and data.dtype == torch.float32        # channel dim added (:104)
    test = dl2.physionet_dataloader(a, ds).run("test", None)
    assert [w for b in test for w in b[3]] == g["test_wav"].tolist()
    assert list(next(iter(test))[0].shape[1:]) == g["test_item3_shape"].tolist()


# ---- the loader's batches against the reference's DataLoader (round 4) --------------------------
from make_golden_loader import RUN_ARGS, RUN_EPOCHS  # noqa: E402


def _check_run_against_reference(golden, device):
    """``physionet_dataloader(args, ds).run('train', 4)`` for two epochs, seeded as train_epoch
    seeds them (train_model.py:497), against what the reference's own
    ``DataLoader(shuffle=True, drop_last=True)`` yielded (dataloader_physionet.py:204-229, recorded
    by make_golden_loader.record_run): same indices, labels, frames, recording ids and cycles,
    batch by batch; the gathered data lives on ``device``."""
    a = argparse.Namespace(**RUN_ARGS)
    if device is not None:
        a.device = device
    loader, labels = dl.physionet_dataloader(a, synthetic_dataset()).run("train", 4)
    assert np.array_equal(labels, golden["run_labels"]) and len(loader) == int(golden["run_len"])
    count = 0
    for e in range(RUN_EPOCHS):
        torch.manual_seed(a.seed * 635410 + count)
        n = 0
        for b, (data, target, frames, wav, _q, idx) in enumerate(loader):
            if device is not None:
                assert data.device.type == device.type and data.is_contiguous()
            assert target.dtype == torch.int64 and frames.dtype == torch.int64
            assert np.array_equal(idx.numpy(), golden[f"run_e{e}_idx"][b])
            assert np.array_equal(target.numpy(), golden[f"run_e{e}_target"][b])
            assert np.array_equal(frames.numpy(), golden[f"run_e{e}_frames"][b])
            assert list(wav) == golden[f"run_e{e}_wav"][b].tolist()
            assert np.array_equal(data.cpu().numpy(), golden[f"run_e{e}_data"][b])     # bit for bit
            count += 1
            n += 1
        assert n == int(golden["run_len"])
    test = dl.physionet_dataloader(a, synthetic_dataset()).run("test", None)
    assert [w for b in test for w in b[3]] == golden["run_test_wav"].tolist()
    got = np.concatenate([b[0].cpu().numpy().reshape(len(b[1]), -1).sum(1) for b in test])
    assert np.allclose(got, golden["run_test_data_sum"], rtol=0, atol=1e-4)


def test_run_batches_match_reference_dataloader_cpu(golden):
    _check_run_against_reference(golden, None)


@pytest.mark.gpu
def test_run_batches_match_reference_dataloader_on_device(golden, device):
    """§8 f2 on the GPU: the device-resident gather yields the reference DataLoader's batches."""
    _check_run_against_reference(golden, device)


from make_golden_loader2d import RUN2D_ARGS, RUN2D_EPOCHS  # noqa: E402


def _check_run2d_against_reference(device):
    from pcgmix_amd import dataloader_physionet2d as dl2
    g = np.load(os.path.join(GOLDEN, "loader2d_selection.npz"))
    a = argparse.Namespace(**RUN2D_ARGS)
    if device is not None:
        a.device = device
    loader, labels = dl2.physionet_dataloader(a, synthetic_dataset2d()).run("train", 4)
    assert np.array_equal(labels, g["run_labels"]) and len(loader) == int(g["run_len"])
    count = 0
    for e in range(RUN2D_EPOCHS):
        torch.manual_seed(a.seed * 635410 + count)
        for b, (data, target, _f, wav, _q, idx) in enumerate(loader):
            if device is not None:
                assert data.device.type == device.type
            assert np.array_equal(idx.numpy(), g[f"run_e{e}_idx"][b])
            assert np.array_equal(target.numpy(), g[f"run_e{e}_target"][b])
            assert list(wav) == g[f"run_e{e}_wav"][b].tolist()
            assert np.array_equal(data.cpu().numpy(), g[f"run_e{e}_data"][b])
            count += 1
    assert count == RUN2D_EPOCHS * int(g["run_len"])


def test_run2d_batches_match_reference_dataloader_cpu():
    _check_run2d_against_reference(None)


@pytest.mark.gpu
def test_run2d_batches_match_reference_dataloader_on_device(device):
    _check_run2d_against_reference(device)


def test_container_refuses_foreign_globals(tmp_path):
    """file2dict reads the reference's container (utils.py:181-186) through a restricted
    unpickler: numpy arrays and plain containers load, anything that would import and call a
    foreign global is refused before it runs."""
    import io
    import pickle
    import zlib

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned > /dev/null",))
    path = str(tmp_path / "evil.dat")
    buf = io.BytesIO()
    pickle.dump({"train": Evil()}, buf)
    with open(path, "wb") as fd:
        fd.write(zlib.compress(buf.getbuffer()))
    with pytest.raises(pickle.UnpicklingError):
        dl.file2dict(path)
    ds = synthetic_dataset(seed=1, n_rec=4)
    ds["train"]["extra"] = {"scalar": np.float32(1.5), "ints": np.arange(3), "t": (1, 2)}
    good = str(tmp_path / "good.dat")
    dl.dict2file(ds, good)
    back = dl.file2dict(good)
    assert back["train"]["extra"]["scalar"] == np.float32(1.5) and back["train"]["extra"]["t"] == (1, 2)
    assert np.array_equal(back["train"]["frames"][0], ds["train"]["frames"][0])
