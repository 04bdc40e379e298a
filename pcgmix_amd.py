"""Import alias: the package directory name is fixed by the build contract and is not
a Python identifier, so ``import pcgmix_amd`` resolves to it (one module object, no
second copy: sub-modules are aliased in ``sys.modules`` too)."""
import importlib
import os
import sys

_REAL = "pcgmix-a-data-augmentation-method-for-heart-sound-classification-extended_amd"
_here = os.path.dirname(os.path.abspath(__file__))
if _here not in sys.path:
    sys.path.insert(0, _here)
_pkg = importlib.import_module(_REAL)
for _name, _mod in list(sys.modules.items()):
    if _name == _REAL or _name.startswith(_REAL + "."):
        sys.modules["pcgmix_amd" + _name[len(_REAL):]] = _mod
sys.modules[__name__] = _pkg
