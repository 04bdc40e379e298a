"""The armed plain step as a timeline (probe build, -DPCGMIX_PHASE_CLOCK): 100 MHz clock at block (0,0)
entry / labels flagged, and per sample: relay entry / relay saw the host's record / block (1,b) saw the
relayed record / block (1,b) done.  Two launches are kept (parity of the sequence number), so the gap
between one kernel's last block and the next kernel's entry is visible.
python profiles/probes/armed_phase_clock.py   (GPU box, repo root)"""
import ctypes, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
_lib.LIB_PATH = out
from pcgmix_amd import augmentations, hostprep, synthetic
from conftest import Args, StepCounter
print(hostprep.bind_host_threads(0), flush=True)
dev = torch.device("cuda:0")
raw = ctypes.CDLL(out)
for (B, C, T) in [(256, 4, 5000), (256, 1, 5000)]:
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=0)
    data = torch.from_numpy(x).to(dev)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(dev)
    fr = torch.from_numpy(frames)
    args, sc = Args(os.environ.get("PCGMIX_PROBE_METHOD", "durratiomixup")), StepCounter(0)
    for rep in range(3):
        n = 400 + rep            # both parities end up "last"
        for k in range(n):
            sc.count = k
            augmentations.augment(args, data, tgt, fr, wav, sc, None, dev, "")
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * (2 * 257 * 4))()
        assert raw.pcgmix_armed_phase_clock(buf) == 0
        t = np.frombuffer(buf, dtype=np.int64).reshape(2, 257, 4) / 100.0
        last = int(np.argmax(t[:, 256, 0])); prev = 1 - last
        L, P = t[last], t[prev]
        e0 = L[256, 0]
        print((B, C, T), "kernel entry -> labels flagged %.2f | flagged -> relays saw records: median %.2f min %.2f max %.2f | "
              "relay -> block 1 saw: median %.2f max %.2f | body (block 1): median %.2f max %.2f | entry -> last block done %.2f | "
              "relay entry after kernel entry: median %.2f max %.2f | previous kernel's last block done -> this entry %.2f | "
              "period (entry to entry) %.2f" %
              (L[256, 1] - e0, np.median(L[:B, 1]) - L[256, 1], L[:B, 1].min() - L[256, 1], L[:B, 1].max() - L[256, 1],
               np.median(L[:B, 2] - L[:B, 1]), (L[:B, 2] - L[:B, 1]).max(), np.median(L[:B, 3] - L[:B, 2]),
               (L[:B, 3] - L[:B, 2]).max(), L[:B, 3].max() - e0, np.median(L[:B, 0]) - e0, L[:B, 0].max() - e0,
               e0 - P[:B, 3].max(), e0 - P[256, 0]), flush=True)
