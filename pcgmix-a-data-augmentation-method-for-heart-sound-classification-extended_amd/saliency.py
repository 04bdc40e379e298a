"""Saliency maps for saliency-guided PCGmix, on device.

Mirrors ``saliency.get_saliency_maps(args, device, data, target_ohe, frames, dim=1,
gauss_k_n=101)`` of the reference (saliency.py:20-116): one forward + backward of a FROZEN
model gives |d score(true class) / d input|; everything after the backward pass
(saliency.py:63-91: zero tail, channel sum, 101-tap Gaussian, per-row min/max) is one HIP
kernel, ``pcgmix_saliency_post_f32``.  Differences from the reference, all at the boundary:

* the maps are returned as a float32 DEVICE tensor (B, T), not a numpy array — their only
  consumer is the displacement search, which also runs on the device;
* the frozen model is loaded once and cached, not re-read from disk every step
  (saliency.py:50); ``set_saliency_model`` lets a training loop hand over a model directly
  ("saliency model = the model being trained, frozen copy");
* the reference's '-1'/'-2' method suffixes (saliency.py:29-34) depend on a module that is not
  in its repository (``results_new``) and are refused.
"""
from __future__ import annotations

import copy
import ctypes
import os
from typing import Optional

import numpy as np
import torch

from . import _lib

_INJECTED: Optional[torch.nn.Module] = None
_LOADED: dict = {}          # checkpoint path -> (mtime, model)
_GRAPHS: dict = {}          # (id(model), shape, classes, k) -> _SaliencyGraph
USE_GRAPHS = True           # replay the frozen model's fwd+bwd+post-processing as one hipGraph


def set_saliency_model(model: Optional[torch.nn.Module], freeze_copy: bool = True) -> None:
    """Use ``model`` (deep-copied and put in eval mode unless ``freeze_copy`` is False) for all
    following saliency computations; ``None`` restores checkpoint loading."""
    global _INJECTED
    _GRAPHS.clear()
    if model is None:
        _INJECTED = None
        return
    m = copy.deepcopy(model) if freeze_copy else model
    m = m.module if isinstance(m, (torch.nn.DataParallel,
                                   torch.nn.parallel.DistributedDataParallel)) else m
    for p in m.parameters():
        p.requires_grad_(False)
    _INJECTED = m.eval()


def experiment_dir(args) -> str:
    """Directory name of a run, reference utils.py:34-53."""
    return os.path.join(args.EXPERIMENTS,
                        "{0}_{1}_{2}_epochs={3}_bs={4}_nfrac={5}_op={6}_sched={7}_lrmax={8}_tbal={9}"
                        "_chs={10}_gc={11}_seed(data)={12}_valid={13}_seed={14}".format(
                            args.dataset, args.model, args.method, args.num_epochs, args.batch_size,
                            args.n_fraction, args.op, args.use_sched, args.lr_max,
                            args.train_balance, args.num_channels, args.grad_clip, args.seed_data,
                            args.valid, args.seed))


def _build_model(args, dim: int) -> torch.nn.Module:
    from . import models, models2d
    if dim == 1:
        if args.model == "resnet9":
            return models.ResNet9(in_channels=args.num_channels, num_classes=args.num_classes)
        if args.model == "Potes":
            return models.CNN_potes_TS(num_channels=args.num_channels, num_classes=args.num_classes,
                                       dataset=args.dataset)
    elif dim == 2 and args.model == "resnet9":
        return models2d.ResNet9(num_classes=2)
    raise ValueError(f"no saliency model for model={args.model!r}, dim={dim}")


def _baseline_model(args, device, dim: int) -> torch.nn.Module:
    """The un-augmented ('base') run's checkpoint, as saliency.py:26-51 selects it."""
    if "-1" in args.method or "-2" in args.method:
        raise NotImplementedError("'-1'/'-2' saliency-model selectors need the reference's "
                                  "missing results_new module")
    a = copy.copy(args)
    a.method = "base"
    path = os.path.join(experiment_dir(a), "model.pth")
    mtime = os.path.getmtime(path)            # FileNotFoundError if the baseline run is absent
    hit = _LOADED.get((path, str(device)))
    if hit is not None and hit[0] == mtime:
        return hit[1]
    model = _build_model(args, dim)
    state = torch.load(path, map_location="cpu", weights_only=True)
    # the reference saves a DataParallel-wrapped model: keys carry a 'module.' prefix
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}
    model.load_state_dict(state)
    model = model.to(device).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    _LOADED[(path, str(device))] = (mtime, model)
    return model


def input_gradient(model: torch.nn.Module, data: torch.Tensor, target_ohe: torch.Tensor):
    """d score[true class] / d input (saliency.py:52-61); model is in eval mode and frozen."""
    target = target_ohe.max(1, keepdim=True)[1]            # first maximum, as the reference
    x = data.detach().requires_grad_(True)                 # shares storage; nothing writes to it
    with torch.enable_grad():
        out = model(x)
        # d(out[b, target_b])/dx summed over b == backward of `out` seeded with one-hot(target):
        # spares the gather, its backward scatter and the ones() of the reference formulation
        seed = torch.zeros_like(out).scatter_(1, target, 1.0)
        (grad,) = torch.autograd.grad(out, x, seed)
    return grad.contiguous()


def saliency_post(grad: torch.Tensor, frames_dev_ptr: int, gauss_k_n: int = 101) -> torch.Tensor:
    """|grad| (B,C,T) -> normalised saliency (B,T) with one kernel launch."""
    B, C, T = grad.shape
    sal = torch.empty((B, T), dtype=torch.float32, device=grad.device)
    sigma = (12 / 101) * gauss_k_n                              # saliency.py:24
    lib = _lib.load()
    stream = torch.cuda.current_stream(grad.device).cuda_stream
    _lib.check(lib.pcgmix_saliency_post_f32(grad.data_ptr(), frames_dev_ptr, sal.data_ptr(),
                                            gauss_k_n, ctypes.c_double(sigma), B, C, T,
                                            ctypes.c_void_p(stream)), "pcgmix_saliency_post_f32")
    return sal


class _SaliencyGraph:
    """Forward + input-gradient + post-processing of a FROZEN model for one batch shape, captured
    once in a hipGraph and replayed: the eager chain is ~40 small launches driven by Python
    autograd (~0.4 ms of host time per step at bs=256) for ~0.2 ms of GPU work."""

    def __init__(self, model, shape, num_classes, device, gauss_k_n):
        B, C, T = shape
        self.model, self.k = model, gauss_k_n
        self.x = torch.zeros(B, C, T, device=device)
        self.t = torch.zeros(B, num_classes, dtype=torch.int64, device=device)
        self.t[:, 0] = 1
        self.fr = torch.zeros(B, 5, dtype=torch.int32, device=device)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(3):
                self._run()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of other threads (e.g. RCCL's watchdog) must not abort the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.sal = self._run()

    def _run(self):
        return saliency_post(input_gradient(self.model, self.x, self.t), self.fr.data_ptr(), self.k)

    def __call__(self, data, target_ohe, frames_dev):
        self.x.copy_(data, non_blocking=True)
        self.t.copy_(target_ohe, non_blocking=True)
        self.fr.copy_(frames_dev, non_blocking=True)
        self.graph.replay()
        return self.sal.clone()


def get_saliency_maps(args, device, data, target_ohe, frames, dim=1, gauss_k_n=101,
                      model_sal: Optional[torch.nn.Module] = None) -> torch.Tensor:
    """Reference signature plus an optional explicit model.  Returns a (B, T) float32 tensor on
    ``data``'s device."""
    if dim != 1:
        raise NotImplementedError("spectrogram (dim=2) saliency is out of scope")
    if not data.is_cuda:
        raise ValueError("data must live on a HIP device")
    model = model_sal or _INJECTED or _baseline_model(args, data.device, dim)
    frames_np = frames.detach().cpu().numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames)
    with torch.cuda.device(data.device):
        from .augmentations import upload_array
        fr = upload_array(frames_np.astype(np.int32), data.device)
        if USE_GRAPHS and target_ohe.dtype == torch.int64 and not torch.cuda.is_current_stream_capturing():
            key = (id(model), tuple(data.shape), int(target_ohe.shape[1]), gauss_k_n, str(data.device))
            g = _GRAPHS.get(key)
            if g is None:
                if len(_GRAPHS) >= 8:
                    _GRAPHS.clear()
                g = _GRAPHS[key] = _SaliencyGraph(model, tuple(data.shape), int(target_ohe.shape[1]),
                                                  data.device, gauss_k_n)
            return g(data, target_ohe, fr)
        grad = input_gradient(model, data, target_ohe)
        return saliency_post(grad, fr.data_ptr(), gauss_k_n)


def optimal_displacements(saliency_maps: torch.Tensor, frames_dev_ptr: int, mix_dev_ptr: int,
                          lam: float, mode: int, B: int, T: int, max_len: int = 0) -> torch.Tensor:
    """Displacement of the shorter state inside the longer one for every (sample, state):
    int32 (B,4) on device (augmentations.py:60-128 via pcgmix_salopt_disp_f32).  ``max_len``: the
    longest heart state of the batch in samples (from the host copy of ``frames``); 0 = unknown."""
    if saliency_maps.shape != (B, T) or saliency_maps.dtype != torch.float32 \
            or not saliency_maps.is_contiguous() or not saliency_maps.is_cuda:
        raise ValueError("saliency maps must be a contiguous float32 (B, T) device tensor")
    disp = torch.empty((B, 4), dtype=torch.int32, device=saliency_maps.device)
    lib = _lib.load()
    ws = torch.empty(max(1, lib.pcgmix_salopt_workspace_bytes(B) // 8), dtype=torch.int64,
                     device=saliency_maps.device)
    stream = torch.cuda.current_stream(saliency_maps.device).cuda_stream
    _lib.check(lib.pcgmix_salopt_disp_f32(saliency_maps.data_ptr(), frames_dev_ptr, mix_dev_ptr,
                                          ctypes.c_float(lam), mode, disp.data_ptr(), ws.data_ptr(),
                                          int(max_len), B, T, ctypes.c_void_p(stream)),
               "pcgmix_salopt_disp_f32")
    return disp
