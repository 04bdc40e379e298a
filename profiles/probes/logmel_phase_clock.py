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
                "-ffp-contract=off", "-DPCGMIX_PHASE_CLOCK"] + os.environ.get("PCGMIX_PROBE_DEFINES", "").split()
               + ["-I" + os.path.join(ROOT, "include"), "-o", out]
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
print("# wall_clock64 (100 MHz) around the phases of every block, logmel_kernel<false, 9>, (256, 5000):")
print("#   staging | STFT (f64 matrix + leftover bins on the VALU) | mel + dB + max | reference, normalise, store")
for it in range(6):
    frontend.logmel(x1, frames)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (B * 8))()
    assert raw.pcgmix_logmel_phase_clock(buf, B) == 0
    t = np.frombuffer(buf, dtype=np.int64).reshape(B, 8)
    us = (t[:, :5] - t[:, 0].min()) / 100.0
    ph = us[:, 1:] - us[:, :-1]
    print("median block, us:", np.median(ph, axis=0).round(2), "block total median %.2f max %.2f" %
          (np.median(us[:, 4] - us[:, 0]), (us[:, 4] - us[:, 0]).max()))
    print("   slowest block's phases:", ph[np.argmax(us[:, 4] - us[:, 0])].round(2),
          "| entry: median %.2f p90 %.2f max %.2f | exit: median %.2f max %.2f (= launch span)" %
          (np.median(us[:, 0]), np.percentile(us[:, 0], 90), us[:, 0].max(), np.median(us[:, 4]), us[:, 4].max()))
hw = t[:, 5] & 0xffffffff
xcc = (t[:, 5] >> 32) & 0xf
cu = (xcc << 12) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
u, cnt = np.unique(cu, return_counts=True)
print("blocks per CU: %d CUs used, max %d blocks on one CU; blocks per XCC:" % (len(u), cnt.max()), np.bincount(xcc.astype(int)))
late = np.argsort(us[:, 0])[::-1][:8]
print("latest entries: block, entry us, CU id:", [(int(b), float(us[b, 0].round(1)), hex(int(cu[b]))) for b in late])

wbuf = (ctypes.c_longlong * 64)()
assert raw.pcgmix_logmel_wave_clock(wbuf) == 0
w = np.frombuffer(wbuf, dtype=np.int64).reshape(16, 4)
t7 = t[7]
print("block 7, per wave (us after the staging barrier): VALU rows done | first unit done | second unit done")
for i in range(16):
    r = [(w[i, j] - t7[1]) / 100.0 if w[i, j] >= t7[1] else float("nan") for j in range(4)]
    print("  wave %2d: %6.2f %6.2f %6.2f   (VALU rows started %.2f)" % (i, r[0], r[1], r[2], r[3]))
print("  matrix phase of block 7: %.2f us" % ((t7[2] - t7[1]) / 100.0))
