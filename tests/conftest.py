import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def load_golden(path):
    g = np.load(path)
    d = {k: g[k] for k in g.files}
    d["method"] = str(d["method"])
    d["step"] = int(d["step"])
    d["wav"] = tuple(str(w) for w in d["wav"])
    return d


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class Args:
    """Minimal stand-in for the reference's argparse.Namespace (experiments_timeseries.ipynb
    cell 4:1-28)."""

    def __init__(self, method, **kw):
        self.method = method
        self.num_classes = 2
        self.batch_size = 64
        self.sample_rate = 1000
        self.model = "Potes"
        self.dataset = "PhysioNet"
        self.num_channels = 4
        self.__dict__.update(kw)


class StepCounter:
    def __init__(self, count=0):
        self.count = count

    def add(self):
        self.count += 1


def learnable_dataset(n_rec=48, T=2500, seed=0):
    """A dataset dictionary in the reference's container layout whose classes are separable:
    class-1 cycles carry a louder 80-200 Hz band."""
    import pcgmix_amd  # noqa: F401
    from pcgmix_amd import synthetic
    rs = np.random.RandomState(seed)
    bands = ["25-45", "45-80", "80-200", "200-400", "25-400"]
    out = {}
    for split, n in (("train", n_rec), ("test", n_rec // 2)):
        d = {"data": {b: [] for b in bands}, "label": [], "frames": [], "wav": [], "sig_qual": []}
        for r in range(n):
            wav, label = f"{'abcdef'[r % 6]}{r:04d}", (r // 6) % 2
            for _ in range(4):
                fr = synthetic.make_frames(1, 1.0, rs)[0]
                for b in bands:
                    sig = rs.standard_normal(T).astype(np.float32)
                    if b == "80-200" and label:
                        sig *= 3.0
                    sig[fr[4]:] = 0
                    d["data"][b].append(sig)
                d["label"].append(label); d["frames"].append(fr); d["wav"].append(wav); d["sig_qual"].append(1)
        out[split] = d
    return out
