import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.')
import pcgmix_amd
from pcgmix_amd import frontend, synthetic, _lib
dev = torch.device('cuda:0')
B, T = 256, 5000
x, frames, labels, wav = synthetic.make_batch(B, 1, T, sample_rate=2000, seed=0)
x1 = torch.from_numpy(x[:, 0, :].copy()).to(dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
for it in range(6):
    frontend.logmel(x1, frames)
    torch.cuda.synchronize()
    out = (ctypes.c_longlong * 5)()
    lib.pcgmix_logmel_debug(out)
    t = np.array(list(out), dtype=np.int64)
    print("phases (wall_clock64 ticks @100MHz -> us):", ((t[1:] - t[:-1]) / 100.0).round(2), "total", (t[4]-t[0])/100.0)
