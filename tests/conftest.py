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
