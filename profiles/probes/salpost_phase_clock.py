"""Round 4: saliency_post_kernel<101> as a timeline of its blocks (probe build, -DPCGMIX_PHASE_CLOCK):
wall_clock64 at entry / row staged (|grad| summed over the channels) / smoothed / minimum known /
maximum known / stored.   python profiles/probes/salpost_phase_clock.py   (GPU box, repo root)"""
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
_lib.LIB_PATH = out
from pcgmix_amd import saliency, synthetic
dev = torch.device("cuda:0")
B, C, T = 256, 4, 5000
x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=0)
grad = torch.randn(B, C, T, device=dev)
fr = torch.from_numpy(frames.astype(np.int32)).to(dev)
lib = _lib.load()
raw = ctypes.CDLL(out)
for it in range(5):
    saliency.saliency_post(grad, fr.data_ptr())
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * (B * 8))()
    assert raw.pcgmix_salpost_phase_clock(buf, B) == 0
    t = np.frombuffer(buf, dtype=np.int64).reshape(B, 8)
    us = (t[:, :6] - t[:, 0].min()) / 100.0
    ph = us[:, 1:] - us[:, :-1]
    print("median block, us: staged | smoothed | min | subtract + max | divide + store:", np.median(ph, axis=0).round(2),
          "block total median %.2f max %.2f; entry p90 %.2f max %.2f; span %.2f" %
          (np.median(us[:, 5] - us[:, 0]), (us[:, 5] - us[:, 0]).max(), np.percentile(us[:, 0], 90), us[:, 0].max(), us[:, 5].max()))
