"""Import the upstream reference's hot-path modules in THIS container only.

Test infrastructure for generating golden vectors (tests/golden/make_golden.py).
Nothing here is shipped to, or used on, the GPU box: /root/reference does not
exist there.  The reference is imported as-is from /root/reference; packages it
imports at module level but never touches on the PCGmix path (tkinter,
torchvision, tsp_solver, audiomentations, python_tsp, tsai, librosa, ... and the
non-existent ``results_new``) are served as inert MagicMock modules, and the
module-level ``pd.read_csv`` of an absolute lab path (augmentations.py:25-28)
is answered with an empty frame.  No reference source is copied.
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys
import types
from unittest import mock

REF = "/root/reference"


class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []          # behave as a package so submodule imports resolve
        m.__spec__ = spec
        m.__loader__ = self
        m.__all__ = []           # ``from tsai.models.X import *`` must import nothing
        return m

    def exec_module(self, module):
        pass


class _MockFinder(importlib.abc.MetaPathFinder):
    """Serve MagicMock modules for top-level names that are NOT installed."""

    def __init__(self, allowed):
        self.allowed = set(allowed)   # top-level names the reference itself imports
        self.served = set()
        self._loader = _MockLoader()

    def find_spec(self, fullname, path, target=None):
        top = fullname.split(".")[0]
        if top in self.served:
            return importlib.machinery.ModuleSpec(fullname, self._loader, is_package=True)
        if "." in fullname or top not in self.allowed:
            return None
        # only serve a mock when no real finder can find it
        for finder in sys.meta_path:
            if finder is self:
                continue
            try:
                spec = finder.find_spec(fullname, path, target)
            except Exception:
                spec = None
            if spec is not None:
                return None
        self.served.add(top)
        return importlib.machinery.ModuleSpec(fullname, self._loader, is_package=True)


def _reference_import_names():
    """Top-level module names imported anywhere in the reference's .py files."""
    import ast
    import glob
    names = set()
    for fn in glob.glob(REF + "/*.py"):
        tree = ast.parse(open(fn, encoding="utf-8").read())
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                names.update(a.name.split(".")[0] for a in node.names)
            elif isinstance(node, ast.ImportFrom) and node.module and node.level == 0:
                names.add(node.module.split(".")[0])
    return names


def import_reference(extra=()):
    """Returns a namespace with the reference modules used on the hot path (plus ``extra``
    module names, e.g. ``("train_model",)`` — it imports too: its only CUDA-only statement is
    inside ``SELCLoss.__init__``, train_model.py:60, which the golden scripts never call)."""
    import pandas as pd

    sys.dont_write_bytecode = True
    finder = _MockFinder(_reference_import_names())
    sys.meta_path.append(finder)          # last: real packages always win
    sys.path.insert(0, REF)
    real_read_csv = pd.read_csv
    pd.read_csv = lambda *a, **k: pd.DataFrame({"wav": [], "diagnosis": []})
    try:
        mods = {}
        for name in ("utils", "models", "models2d", "saliency", "augmentations",
                     "augmentations2d") + tuple(extra):
            mods[name] = importlib.import_module(name)
    finally:
        pd.read_csv = real_read_csv
    ns = types.SimpleNamespace(**mods)
    ns.mocked = sorted(finder.served)
    return ns


if __name__ == "__main__":
    ns = import_reference()
    print("mocked:", ns.mocked)
    print("augment:", ns.augmentations.augment)
