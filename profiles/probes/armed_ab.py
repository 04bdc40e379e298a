"""Strict-signature step (256,4,5000) `durratiomixup`, 1000-step regions, armed kernel against the
two-launch path (run twice: PCGMIX_NO_ARMED unset / =1).   python profiles/probes/armed_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib
if os.environ.get("PCGMIX_PROBE_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["PCGMIX_PROBE_LIB"])
from pcgmix_amd import augmentations, hostprep, synthetic
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from conftest import Args, StepCounter
dev = torch.device("cuda", 0)
print(hostprep.bind_host_threads(0), "| PCGMIX_NO_ARMED =", os.environ.get("PCGMIX_NO_ARMED"), "| library:",
      os.environ.get("PCGMIX_PROBE_LIB") or "product", flush=True)
for (B, C, T) in [(256, 4, 5000), (256, 1, 5000)]:
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=0)
    data = torch.from_numpy(x).to(dev)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(dev)
    fr = torch.from_numpy(frames)
    args, sc = Args(os.environ.get("PCGMIX_PROBE_METHOD", "durratiomixup")), StepCounter(0)
    def region(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(n):
            sc.count = k
            augmentations.augment(args, data, tgt, fr, wav, sc, None, dev, "")
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
    region(200)
    import ctypes, numpy as np
    lib = _lib.load()
    mix = np.random.RandomState(0).permutation(B).astype(np.int16); fr16 = frames.astype(np.int16)
    out = torch.empty_like(data); st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    def karg(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            lib.pcgmix_mix_karg_f32(data.data_ptr(), out.data_ptr(), fr16.ctypes.data, mix.ctypes.data,
                                    ctypes.c_float(0.3), B, C, T, st)
        e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
    karg(50)
    print("   karg kernel back to back: %.2f %.2f us" % (karg(300), karg(300)), flush=True)
    print((B, C, T), " ".join("%.2f" % region(1000) for _ in range(5)), "us/step;  20-step regions:",
          " ".join("%.2f" % region(20) for _ in range(5)), flush=True)
    import ctypes
    from pcgmix_amd import _lib
    lib = _lib.load(); ctx = augmentations.step_context(0)
    out = (ctypes.c_double * 8)()
    lib.pcgmix_ctx_phase_times(ctx, out)
    t = region(1000)
    n = lib.pcgmix_ctx_phase_times(ctx, out)
    print("   phases (ns/call over %d calls, step %.2f us):" % (n, t), " ".join("%.0f" % v for v in out), " sum %.0f" % sum(out), flush=True)
