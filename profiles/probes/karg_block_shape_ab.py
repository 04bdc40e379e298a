"""Round 4: the kernarg splice at (256,4,5000) — one 1024-thread block per sample (default) against the
chunked grid of 256-thread blocks (PCGMIX_MIX_UNROLL=2), back-to-back launches and the strict step.
    python profiles/probes/karg_block_shape_ab.py          (GPU box, repo root; run once per setting)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if os.environ.get("PCGMIX_PROBE_LIB"):
    import pcgmix_amd  # noqa: F401
    from pcgmix_amd import _lib
    _lib.LIB_PATH = os.path.abspath(os.environ["PCGMIX_PROBE_LIB"])
import bench
dev = torch.device("cuda:0")
info = {}
bench.settle_clocks(dev)
ms = bench.kernel_back_to_back_ms("durratiomixup", 256, 4, 5000, 2000, dev, iters=400, info=info)
_, data, tgt, frames, labels, wav = bench.make_device_batch(256, 4, 5000, 2000, 0, dev)
dt, _ = bench.run_augment_steps("durratiomixup", data, tgt, frames, wav, dev, 2000, 50, lambda: None)
print((os.environ.get("PCGMIX_PROBE_LIB") or "product library") + ": PCGMIX_MIX_UNROLL=%s  kernel %.2f us   strict augment() step %.2f us" %
      (os.environ.get("PCGMIX_MIX_UNROLL", "-"), ms * 1e3, dt / 2000 * 1e6), flush=True)
