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

import contextlib
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
DETERMINISTIC_FROZEN_PASS: Optional[bool] = None   # None: PCGMIX_SALIENCY_DETERMINISTIC decides (default off)
CHAIN_EAGER = os.environ.get("PCGMIX_SAL_CHAIN_GRAPH") is None   # see _SaliencyGraph.__init__


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


def _potes_direct(model: torch.nn.Module, data: torch.Tensor):
    """The CNN_potes whose input gradient can be written as a fixed chain of HIP launches
    (eval mode, fused conv stack and head, four band channels, saved-routing kernels), or None."""
    from . import models
    m = model.module if isinstance(model, (torch.nn.DataParallel,
                                           torch.nn.parallel.DistributedDataParallel)) else model
    if (isinstance(m, models.CNN_potes) and not m.training and data.dim() == 3 and data.shape[1] == 4
            and data.shape[0] > 0 and data.is_contiguous() and models.PotesStackFunction.use_masks
            and m._fused_head(data)):
        return m
    return None


class _PotesChain:
    """d (sum_b seed_b . logits_b) / d input of a frozen CNN_potes without autograd, for one batch
    shape: conv stack forward saving its ReLU / max-pool routing (``forward``) -> head (split-K
    product, dz = (z > 0) * (seed W2), dx = dz W1: ``pcgmix_potes_head_saliency_f32``) -> input
    gradient from the saved routing (``backward``).  Six launches; the logits are never formed
    (models.py:444-465 forward, autograd backward).  The buffers between the stages are owned by
    the object, so the two halves can be enqueued separately: the forward is the only kernel that
    reads the batch — a captured saliency pass launches it eagerly on the caller's tensor and
    captures the rest, instead of copying 20 MB into a static input first."""

    def __init__(self, m, shape, device):
        B, C, T = shape
        lib = _lib.load()
        self.m, self.shape = m, (B, C, T)
        self.N, self.P2, self.K = B * 4, lib.pcgmix_potes_out_len(T), m.dimreduc.in_features
        self.ncls = m.linear.weight.shape[0]
        if self.K != 16 * self.P2:
            raise ValueError("model head does not match the input shape")
        f32 = dict(dtype=torch.float32, device=device)
        u8 = dict(dtype=torch.uint8, device=device)
        self.h2 = torch.empty((self.N, 4, self.P2), **f32)
        self.m2 = torch.empty(lib.pcgmix_potes_mask_bytes(self.N, T, 2), **u8)
        self.s1 = torch.empty(lib.pcgmix_potes_mask_bytes(self.N, T, 1), **u8)
        self.partial = torch.empty((lib.pcgmix_skinny_linear_splits(B, self.K), B, 20), **f32)
        self.dz, self.gfeat = torch.empty((B, 20), **f32), torch.empty((B, self.K), **f32)
        self._pass_args = None      # pointer block of pass_into(), built on first use

    def pass_into(self, data: torch.Tensor, seed: torch.Tensor, frames_dev_ptr: int, gauss_k_n: int,
                  sal: torch.Tensor, gx: torch.Tensor) -> None:
        """forward + backward + post-processing as ONE library call (pcgmix_potes_saliency_pass_f32)
        into ``sal`` (B,T); ``gx`` (B,4,T) receives the input gradient.  The model is frozen
        (saliency.py:26-51): its weight pointers are looked up once."""
        B, C, T = self.shape
        if tuple(data.shape) != self.shape or data.dtype != torch.float32 or not data.is_contiguous():
            raise ValueError("batch does not match the saliency chain's shape")
        if self._pass_args is None:
            m = self.m
            w1, b1, w2, b2 = self._weights()
            W1, W2 = m.dimreduc.weight.detach().contiguous(), m.linear.weight.detach().contiguous()
            bh = m.dimreduc.bias.detach().contiguous() if m.dimreduc.bias is not None else None
            self._keep = (w1, b1, w2, b2, W1, W2, bh)          # the storages behind the pointers
            self._pass_args = (w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                               self.h2.data_ptr(), self.m2.data_ptr(), self.s1.data_ptr(), W1.data_ptr(),
                               bh.data_ptr() if bh is not None else None, W2.data_ptr())
        a = self._pass_args
        stream = ctypes.c_void_p(torch.cuda.current_stream(data.device).cuda_stream)
        _lib.check(_lib.load().pcgmix_potes_saliency_pass_f32(
            data.data_ptr(), a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], seed.data_ptr(),
            self.partial.data_ptr(), self.dz.data_ptr(), self.gfeat.data_ptr(), gx.data_ptr(),
            frames_dev_ptr, sal.data_ptr(), gauss_k_n, ctypes.c_double((12 / 101) * gauss_k_n), B, T,
            self.K, self.ncls, stream), "pcgmix_potes_saliency_pass_f32")

    def _weights(self):
        m = self.m
        c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
        return tuple(t.detach().contiguous() for t in (c1.weight, c1.bias, c2.weight, c2.bias))

    def forward(self, data: torch.Tensor) -> None:
        B, C, T = self.shape
        if tuple(data.shape) != self.shape or data.dtype != torch.float32 or not data.is_contiguous():
            raise ValueError("batch does not match the saliency chain's shape")
        w1, b1, w2, b2 = self._weights()
        stream = ctypes.c_void_p(torch.cuda.current_stream(data.device).cuda_stream)
        _lib.check(_lib.load().pcgmix_potes_stack_fwd_save_f32(
            data.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
            self.h2.data_ptr(), self.m2.data_ptr(), self.s1.data_ptr(), self.N, T, None, 0, None, 0,
            stream), "pcgmix_potes_stack_fwd_save_f32")

    def backward(self, seed: torch.Tensor) -> torch.Tensor:
        B, C, T = self.shape
        m, lib = self.m, _lib.load()
        if seed.shape != (B, self.ncls) or seed.dtype != torch.float32 or not seed.is_contiguous():
            raise ValueError("saliency seed does not match the model head")
        w1, _b1, w2, _b2 = self._weights()
        W1, W2 = m.dimreduc.weight.detach().contiguous(), m.linear.weight.detach().contiguous()
        bh = m.dimreduc.bias.detach() if m.dimreduc.bias is not None else None
        gx = torch.empty((B, C, T), dtype=torch.float32, device=seed.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(seed.device).cuda_stream)
        _lib.check(lib.pcgmix_potes_head_saliency_f32(
            self.h2.data_ptr(), W1.data_ptr(), bh.data_ptr() if bh is not None else None, W2.data_ptr(),
            seed.data_ptr(), self.partial.data_ptr(), self.dz.data_ptr(), self.gfeat.data_ptr(), B,
            self.K, self.ncls, stream), "pcgmix_potes_head_saliency_f32")
        _lib.check(lib.pcgmix_potes_stack_input_grad_mask_f32(
            self.gfeat.data_ptr(), self.m2.data_ptr(), self.s1.data_ptr(), w1.data_ptr(), w2.data_ptr(),
            gx.data_ptr(), self.N, T, stream), "pcgmix_potes_stack_input_grad_mask_f32")
        return gx


def _potes_input_gradient(m, data: torch.Tensor, seed: torch.Tensor) -> torch.Tensor:
    chain = _PotesChain(m, tuple(data.shape), data.device)
    chain.forward(data.detach())
    return chain.backward(seed)


def input_gradient_seeded(model: torch.nn.Module, data: torch.Tensor, seed: torch.Tensor):
    """Backward of the model's output seeded with ``seed`` (B, classes), w.r.t. the input."""
    m = _potes_direct(model, data)
    if m is not None:
        return _potes_input_gradient(m, data, seed)
    x = data.detach().requires_grad_(True)                 # shares storage; nothing writes to it
    # MIOpen's algorithm choice for this frozen pass (ResNet9 saliency models).  Default selection,
    # as the reference runs it (it never asks cuDNN for determinism): measured on the reference's
    # recorded ResNet9-2D gradient (profiles/r4_sal2d_determinism.txt) it is off by O(1) of the
    # gradient's scale wherever the input is flat (the zero padding behind a cycle: ties in ReLU /
    # max-pool routing, accumulation with atomics), 1e-6 .. 1e-3 inside the cycle, and moves from run
    # to run -> maps 3e-5 .. 2e-3 from the reference's CPU maps.  The deterministic selection is
    # within 1e-6 everywhere and bit-reproducible (maps 4e-7) but 40x (1D) / 77x (2D) slower at
    # bs 256 (1427 vs 36 ms, 5970 vs 78 ms), so it is opt-in: PCGMIX_SALIENCY_DETERMINISTIC=1 or
    # saliency.DETERMINISTIC_FROZEN_PASS = True (parity runs, the golden tests).
    det = DETERMINISTIC_FROZEN_PASS
    if det is None:
        det = os.environ.get("PCGMIX_SALIENCY_DETERMINISTIC", "0") not in ("", "0")
    algo = (torch.backends.cudnn.flags(enabled=True, benchmark=False, deterministic=True)
            if det else contextlib.nullcontext())
    with torch.enable_grad(), algo:
        out = model(x)
        (grad,) = torch.autograd.grad(out, x, seed)
    return grad.contiguous()


def class_seed(target_ohe: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """one_hot(first maximum of each row) as float32 — d(out[b, target_b])/d out.  Backward of
    `out` seeded with it == the reference's gather + sum + backward (saliency.py:52-61) without
    the gather, its backward scatter and the ones()."""
    target = target_ohe.max(1, keepdim=True)[1]            # first maximum, as the reference
    if out is None:
        out = torch.zeros(target_ohe.shape, dtype=torch.float32, device=target_ohe.device)
    else:
        out.zero_()
    return out.scatter_(1, target, 1.0)


def input_gradient(model: torch.nn.Module, data: torch.Tensor, target_ohe: torch.Tensor):
    """d score[true class] / d input (saliency.py:52-61); model is in eval mode and frozen."""
    return input_gradient_seeded(model, data, class_seed(target_ohe))


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


def saliency_post2d(grad: torch.Tensor, frames_dev_ptr: int, gauss_k_n: int = 11,
                    sigma: float = 1.0) -> torch.Tensor:
    """The spectrogram branch (saliency.py:93-113): |grad| (B,1,F,W) or (B,F,W) -> (B,W) maps,
    normalised over each cycle's own columns, with one kernel launch.  11 taps, sigma 1 are the
    reference's constants (:101)."""
    if grad.dim() == 4:
        if grad.shape[1] != 1:
            raise ValueError("spectrogram saliency expects one image channel (models2d.ResNet9)")
        grad = grad[:, 0]
    grad = grad.contiguous()
    B, Fq, W = grad.shape
    sal = torch.empty((B, W), dtype=torch.float32, device=grad.device)
    stream = torch.cuda.current_stream(grad.device).cuda_stream
    _lib.check(_lib.load().pcgmix_saliency_post2d_f32(
        grad.data_ptr(), frames_dev_ptr, sal.data_ptr(), gauss_k_n, ctypes.c_double(sigma), B, Fq, W,
        ctypes.c_void_p(stream)), "pcgmix_saliency_post2d_f32")
    return sal


class _SaliencyGraph:
    """Forward + input-gradient + post-processing of a FROZEN model for one batch shape, captured
    once in a hipGraph and replayed: the eager chain is ~40 small launches driven by Python
    autograd (~0.4 ms of host time per step at bs=256) for ~0.2 ms of GPU work.

    Static inputs: ``x`` (the batch), ``seed`` (float one-hot of the labels — the label read-back
    kernel of the step context writes it as a by-product, ``pcgmix_ctx_labels_begin``), ``fr``
    (int32 boundaries, uploaded straight into place)."""

    def __init__(self, model, shape, num_classes, device, gauss_k_n):
        B, C, T = shape
        self.model, self.k = model, gauss_k_n
        self.x = torch.zeros(B, C, T, device=device)
        self.seed = torch.zeros(B, num_classes, device=device)
        self.seed[:, 0] = 1
        self.fr = torch.zeros(B, 5, dtype=torch.int32, device=device)
        # A frozen fused CNN_potes: only the conv stack's forward reads the batch — it is launched
        # eagerly on the caller's tensor and everything behind it is captured; other models get
        # the batch copied into the static input and the whole pass captured.
        m = _potes_direct(model, self.x)
        self.chain = _PotesChain(m, (B, C, T), device) if m is not None else None
        # The direct Potes chain is five plain launches with no autograd in between: replaying
        # them as a hipGraph saves ~20 us of host time per step but costs a stream -> graph ->
        # stream hand-over on the GPU (~9 us of idle queue behind the graph's last kernel,
        # profiles/r3_cfg3_step_timeline.txt) — and the step is GPU-bound.  CHAIN_EAGER launches
        # them directly instead (PCGMIX_SAL_CHAIN_GRAPH=1 restores the captured form).
        self.eager = self.chain is not None and CHAIN_EAGER
        if self.eager:
            self.graph = None
            # one library call per pass into buffers owned here (valid until the next pass)
            self.sal = torch.empty((B, T), dtype=torch.float32, device=device)
            self.gx = torch.empty((B, C, T), dtype=torch.float32, device=device)
            self.chain.pass_into(self.x, self.seed, self.fr.data_ptr(), self.k, self.sal, self.gx)
            return
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            if self.chain is not None:
                self.chain.forward(self.x)
            for _ in range(3):
                self._run()
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of other threads (e.g. RCCL's watchdog) must not abort the capture
        with _lib.capture_without_gc(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.sal = self._run()

    def _run(self):
        grad = self.chain.backward(self.seed) if self.chain is not None \
            else input_gradient_seeded(self.model, self.x, self.seed)
        return saliency_post(grad, self.fr.data_ptr(), self.k)

    def _enqueue(self, data):
        if self.eager:
            self.chain.pass_into(data.detach(), self.seed, self.fr.data_ptr(), self.k, self.sal, self.gx)
            return
        if self.chain is not None:
            self.chain.forward(data.detach())
        else:
            self.x.copy_(data, non_blocking=True)
        self.graph.replay()

    def replay(self, data):
        """``seed`` and ``fr`` are in place (``pcgmix_ctx_salopt_begin`` on this stream): run the
        pass on ``data``; returns the graph's static output, valid until the next replay."""
        self._enqueue(data)
        return self.sal

    def run(self, data, frames_np, target_ohe=None, keep: bool = True):
        """``target_ohe`` None: ``seed`` has been written by the caller (on this stream).
        ``keep`` False: the returned maps are the graph's static output, valid until the next
        run — for a caller that consumes them right away."""
        from .augmentations import upload_into
        if target_ohe is not None:
            class_seed(target_ohe, self.seed)
        upload_into(self.fr, frames_np.astype(np.int32))
        self._enqueue(data)
        return self.sal.clone() if keep else self.sal


def step_graph(args, data, num_classes: int, dim: int = 1, gauss_k_n: int = 101,
               model_sal: Optional[torch.nn.Module] = None) -> Optional[_SaliencyGraph]:
    """The captured saliency pass for this model and batch shape (captured on first use), or None
    when graphs are off or the stream is capturing."""
    if dim != 1:
        return None          # spectrograms: the pass is 25 ms of MIOpen convolutions, run eagerly
    if not data.is_cuda:
        raise ValueError("data must live on a HIP device")
    if not USE_GRAPHS or torch.cuda.is_current_stream_capturing():
        return None
    model = model_sal or _INJECTED or _baseline_model(args, data.device, dim)
    key = (id(model), tuple(data.shape), int(num_classes), gauss_k_n, str(data.device))
    g = _GRAPHS.get(key)
    if g is None:
        if len(_GRAPHS) >= 8:
            _GRAPHS.clear()
        with torch.cuda.device(data.device):
            g = _GRAPHS[key] = _SaliencyGraph(model, tuple(data.shape), int(num_classes), data.device,
                                              gauss_k_n)
    return g


def get_saliency_maps(args, device, data, target_ohe, frames, dim=1, gauss_k_n=101,
                      model_sal: Optional[torch.nn.Module] = None) -> torch.Tensor:
    """Reference signature plus an optional explicit model.  Returns a (B, T) float32 tensor on
    ``data``'s device; ``dim=2`` (saliency.py:93-113): ``data`` is (B, 1, F, W), the maps are
    (B, W), the model is the ResNet9-2D 'base' checkpoint (``gauss_k_n`` is not used on that
    branch: the reference hard-codes 11 taps, sigma 1)."""
    if dim not in (1, 2):
        raise ValueError("Set dimension to either 1 or 2")          # saliency.py:45
    if not data.is_cuda:
        raise ValueError("data must live on a HIP device")
    frames_np = frames.detach().cpu().numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames)
    if dim == 2:
        with torch.cuda.device(data.device):
            from .augmentations import upload_array
            model = model_sal or _INJECTED or _baseline_model(args, data.device, 2)
            fr = upload_array(frames_np.astype(np.int32), data.device)
            grad = input_gradient(model, data, target_ohe)
            return saliency_post2d(grad, fr.data_ptr())
    with torch.cuda.device(data.device):
        g = step_graph(args, data, target_ohe.shape[1], dim, gauss_k_n, model_sal) \
            if target_ohe.dtype == torch.int64 else None
        if g is not None:
            return g.run(data, frames_np, target_ohe)
        from .augmentations import upload_array
        model = model_sal or _INJECTED or _baseline_model(args, data.device, dim)
        fr = upload_array(frames_np.astype(np.int32), data.device)
        grad = input_gradient(model, data, target_ohe)
        return saliency_post(grad, fr.data_ptr(), gauss_k_n)


def optimal_displacements(saliency_maps: torch.Tensor, frames_dev_ptr: int, mix_dev_ptr: int,
                          lam: float, mode: int, B: int, T: int, max_len: int = 0,
                          frames_host: Optional[np.ndarray] = None,
                          mix_host: Optional[np.ndarray] = None) -> torch.Tensor:
    """Displacement of the shorter state inside the longer one for every (sample, state):
    int32 (B,4) on device (augmentations.py:60-128 via pcgmix_salopt_disp_f32).  ``max_len``: the
    longest heart state of the batch in samples (from the host copy of ``frames``); 0 = unknown.
    ``frames_host`` (B,5) / ``mix_host`` (B): the host copies of the two index arrays (the reference
    holds them as CPU arrays anyway) — with both the launch is planned on the host
    (pcgmix_salopt_disp_hosted_f32: only blocks with candidates, longest chain first)."""
    if saliency_maps.shape != (B, T) or saliency_maps.dtype != torch.float32 \
            or not saliency_maps.is_contiguous() or not saliency_maps.is_cuda:
        raise ValueError("saliency maps must be a contiguous float32 (B, T) device tensor")
    disp = torch.empty((B, 4), dtype=torch.int32, device=saliency_maps.device)
    lib = _lib.load()
    ws = torch.empty(max(1, lib.pcgmix_salopt_workspace_bytes(B) // 8), dtype=torch.int64,
                     device=saliency_maps.device)
    stream = torch.cuda.current_stream(saliency_maps.device).cuda_stream
    if frames_host is not None and mix_host is not None:
        fh = np.ascontiguousarray(frames_host, dtype=np.int32)
        mh = np.ascontiguousarray(mix_host, dtype=np.int32)
        if fh.shape != (B, 5) or mh.shape != (B,):
            raise ValueError("frames_host must be (B,5) and mix_host (B,)")
        if max_len <= 0:
            max_len = int(np.diff(fh, axis=1).max())
        _lib.check(lib.pcgmix_salopt_disp_hosted_f32(
            saliency_maps.data_ptr(), frames_dev_ptr, mix_dev_ptr, ctypes.c_float(lam), mode,
            disp.data_ptr(), ws.data_ptr(), int(max_len), B, T, ctypes.c_void_p(stream),
            fh.ctypes.data, mh.ctypes.data), "pcgmix_salopt_disp_hosted_f32")
        return disp
    _lib.check(lib.pcgmix_salopt_disp_f32(saliency_maps.data_ptr(), frames_dev_ptr, mix_dev_ptr,
                                          ctypes.c_float(lam), mode, disp.data_ptr(), ws.data_ptr(),
                                          int(max_len), B, T, ctypes.c_void_p(stream)),
               "pcgmix_salopt_disp_f32")
    return disp
