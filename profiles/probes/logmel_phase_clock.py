"""Round 4: where one block of the per-cycle log-mel kernel spends its time.  Builds a PROBE copy of
the library with -DPCGMIX_PHASE_CLOCK (block 7 leaves wall_clock64, 100 MHz, at its phase boundaries)
into build_probe/ and runs the bs-256 x 5000 workload through it.
    python profiles/probes/logmel_phase_clock.py        (on the GPU box, from the repo root)"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
PKG = glob.glob(os.path.join(ROOT, "pcgmix-*_amd"))[0]
out = os.path.join(ROOT, "build_probe", "libpcgmix_phase_clock.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-shared", "-std=c++17", "--offload-arch=gfx950",
                "-ffp-contract=off", "-DPCGMIX_PHASE_CLOCK", "-I" + os.path.join(ROOT, "include"), "-o", out]
               + srcs, check=True)
import numpy as np
import torch
import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib
_lib.LIB_PATH = out                      # the probe build instead of the product library
from pcgmix_amd import frontend, synthetic
dev = torch.device("cuda:0")
B, T = 256, 5000
x, frames, labels, wav = synthetic.make_batch(B, 1, T, sample_rate=2000, seed=0)
x1 = torch.from_numpy(x[:, 0, :].copy()).to(dev)
lib = _lib.load()
raw = ctypes.CDLL(out)
print("# wall_clock64 (100 MHz) around the phases of block 7, logmel_kernel<false, 17>, (256, 5000):")
print("#   staging | STFT (f64 matrix + leftover bins on the VALU) | mel + dB + max | reference, normalise, store")
for it in range(6):
    frontend.logmel(x1, frames)
    torch.cuda.synchronize()
    t = (ctypes.c_longlong * 5)()
    raw.pcgmix_logmel_phase_clock(t)
    t = np.array(list(t), dtype=np.int64)
    print("us:", ((t[1:] - t[:-1]) / 100.0).round(2), "block total", (t[4] - t[0]) / 100.0)
