#!/usr/bin/env python3
"""Launch the splice kernel N times on ONE workload, for rocprofv3 (kernel trace or one --pmc
counter per pass: FETCH_SIZE costs 3 of the 4 TCC slots, WRITE_SIZE 2, so they cannot share a
pass — MI355X_MICROARCH.md 'rocprofv3 PMC slots').

    rocprofv3 --kernel-trace --stats -d OUT -- python3 profiles/mix_pmc_probe.py splice 16384
    rocprofv3 --pmc FETCH_SIZE       -d OUT -- python3 profiles/mix_pmc_probe.py splice 16384
    rocprofv3 --pmc WRITE_SIZE       -d OUT -- python3 profiles/mix_pmc_probe.py splice 16384

modes   splice   durratiomixup                       (partner read inside blended ranges)
        warp     durmixmagwarp(0.2,4)                (same traffic + spline in registers)
        copy     durratiomixup on zero-length states (own read + write only: 8*C*T*B bytes, known
                 exactly — the calibration point for the FETCH_SIZE x2 correction)
        karg     durratiomixup through pcgmix_mix_karg_f32 (index block in the kernel arguments:
                 the instantiation the drop-in step launches for B <= 256)
Writes <mode>_<B>.json next to the traces with the exact byte counts of the batch it ran.
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pcgmix_amd  # noqa: E402,F401
from pcgmix_amd import augmentations, hostprep, synthetic  # noqa: E402
import bench  # noqa: E402

mode, B = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
out_dir = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out")
C, T, rate = 4, 5000, 2000
device = torch.device("cuda:0")
torch.cuda.set_device(device)
method = "durmixmagwarp(0.2,4)" if mode == "warp" else "durratiomixup"
frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=rate, seed=0)
if mode == "copy":
    frames = np.zeros_like(frames)
data = torch.randn(B, C, T, device=device)
plan = hostprep.make_plan(method, labels, frames, wav, 1, B, C)
dev, offs = augmentations.upload_plan(plan, frames, device)
base = dev.data_ptr()
out = torch.empty_like(data)
knots_ptr = op_ptr = None
if plan.knots is not None:
    op = augmentations.spline_operator(device, T, plan.n_knots)
    knots_ptr, op_ptr = base + offs["knots"], op.data_ptr()
if mode == "karg":
    import ctypes
    from pcgmix_amd import _lib
    fr16, mx16 = frames.astype(np.int16), plan.mix.astype(np.int16)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    for _ in range(iters):
        _lib.check(_lib.load().pcgmix_mix_karg_f32(data.data_ptr(), out.data_ptr(), fr16.ctypes.data,
                                                   mx16.ctypes.data, ctypes.c_float(float(plan.lam32)),
                                                   B, C, T, st), "pcgmix_mix_karg_f32")
else:
    for _ in range(iters):
        augmentations.launch_mix(data, out, base + offs["frames"], base + offs["mix"], None,
                                 float(plan.lam32), knots_ptr, op_ptr, plan.n_knots, B, C, T)
torch.cuda.synchronize()
info = {"mode": mode, "method": method, "B": B, "C": C, "T": T, "iters": iters,
        "kernel": bench.mix_kernel_name(B, C, T, bench._n_knots(plan)) if mode == "karg" else
        bench.mix_warp_kernel_name(B, C, T, bench._n_knots(plan)),
        "exact_bytes": bench.exact_mix_bytes(frames, plan.mix, C, T),
        "contract_12CT_bytes": 12.0 * B * C * T, "own_plus_write_bytes": 8.0 * B * C * T}
os.makedirs(out_dir, exist_ok=True)
json.dump(info, open(os.path.join(out_dir, f"mixprobe_{mode}_{B}.json"), "w"), indent=1)
print(info)
