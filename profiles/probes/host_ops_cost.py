"""Probe: cost of the individual host operations on the strict-signature augment() path (us each)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from pcgmix_amd import augmentations as A, hostprep, _lib

dev = torch.device("cuda:0")
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 0, dev)
lib = _lib.load()


def t(name, f, n=20000):
    for _ in range(200): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    print(f"{name:44s} {(time.perf_counter() - t0) / n * 1e6:7.2f} us", flush=True)


i = [0]
def sb():
    i[0] += 1
    np.random.seed(i[0]); return np.random.beta(1.0, 1.0)
t("data.shape unpack", lambda: data.shape)
t("data.device.index", lambda: data.device.index)
t("target_ohe.detach()", lambda: tgt.detach())
t("4 tensor predicates", lambda: (tgt.is_cuda, tgt.dtype == torch.int64, tgt.dim() == 2, tgt.is_contiguous()))
t("_check_data", lambda: A._check_data(data, 3))
t("data_ptr()", lambda: data.data_ptr())
t("_frames_ptr", lambda: A._frames_ptr(frames, 256))
t("np.random.seed + beta", sb)
t("torch.empty_like(data)", lambda: torch.empty_like(data), 5000)
t("np.empty(256, int64)", lambda: np.empty(256, dtype=np.int64))
t("_get_raw_stream", lambda: A._get_raw_stream(0))
t("_lib.load()", lambda: _lib.load())
t("ctypes call, 0 args (abi_version)", lambda: lib.pcgmix_abi_version())
t("ctypes call, 2 args (py_uniform01: MT init)", lambda: lib.pcgmix_py_uniform01(12345))
t("c_float(x)", lambda: ctypes.c_float(0.5))
args, sc = bench.Args("durratiomixup"), bench.StepCounter()
def full():
    A.augment(args, data, tgt, frames, wav, sc, None, dev, "", host_labels=labels); sc.add()
t("augment(host_labels) [host+launch]", full, 3000)
torch.cuda.synchronize()
def strict():
    A.augment(args, data, tgt, frames, wav, sc, None, dev, ""); sc.add()
t("augment(strict)", strict, 3000)
torch.cuda.synchronize()
import numpy as _np
out8 = _np.zeros(8)
for mode, fn in (("host_labels", full), ("strict", strict)):
    lib.pcgmix_ctx_phase_times(A._CTX[0], out8.ctypes.data)
    for _ in range(3000): fn()
    torch.cuda.synchronize()
    n = lib.pcgmix_ctx_phase_times(A._CTX[0], out8.ctypes.data)
    names = ["label kernel launch", "slot reserve", "pack + seed", "label wait", "group + permutation",
             "H2D enqueue", "kernel launch", "event record"]
    print(mode, n, "calls; library phases (us):", {k: round(v / 1e3, 2) for k, v in zip(names, out8)},
          "sum", round(out8.sum() / 1e3, 2))
