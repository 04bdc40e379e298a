"""Drop-in replacement for the PCGmix branches of the reference's ``augmentations.augment``.

Same name, same positional signature, same return tuple as augmentations.py:698 (called once
per batch from train_model.py:507):

    data, target_ohe, mix_indices, cut = augment(args, data, target_ohe, frames, wav,
                                                 step_counter, model, device, RESULTS_ARGS)

Implemented methods (the PCGmix hot path): ``durratiomixup`` (augmentations.py:931-981) and
``durmixmagwarp(sigma,knot)`` (augmentations.py:864-929) with the selectors ``(rand)``,
``(alpha=a)``, ``(samePCG)``, ``(sameDataset)``, ``(mixAll)``, ``(saloptenv…)``,
``(saloptsum…)`` and the ``+p`` probability gate.  The host part (RNG, partner indices) is in
``hostprep``; the O(B*C*T) part is ONE launch of ``pcgmix_mix_warp_f32`` (HIP, gfx950) on the
current torch stream, with no host synchronisation after the labels have been read.

The split form ``make_plan`` / ``apply_plan`` lets a training loop that already holds the
labels on the host prepare step n+1 while the GPU still runs step n.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np
import torch

from . import _lib, hostprep
from .hostprep import MixPlan

_OP_CACHE: dict = {}      # (device index, T, n_knots) -> device tensor with the spline operator
_RECIPES: dict = {}       # method string -> plain recipe | None (general plan path) | False (passthrough)


def _raw_stream(device: torch.device) -> int:
    """hipStream_t of torch's current stream on ``device`` (the cheap private getter when this
    torch build has it, the public Stream object otherwise)."""
    idx = device.index if device.index is not None else torch.cuda.current_device()
    try:
        return torch._C._cuda_getCurrentRawStream(idx)
    except AttributeError:      # pragma: no cover
        return torch.cuda.current_stream(device).cuda_stream


def _as_numpy_frames(frames) -> np.ndarray:
    if isinstance(frames, torch.Tensor):
        frames = frames.detach().cpu().numpy()
    return np.ascontiguousarray(frames, dtype=np.int64)


def _check_data(data: torch.Tensor, ndim: int) -> None:
    if not isinstance(data, torch.Tensor) or data.dim() != ndim:
        raise ValueError(f"data must be a {ndim}-D tensor")
    if data.dtype != torch.float32:
        raise ValueError(f"data must be float32, got {data.dtype}")
    if not data.is_contiguous():
        raise ValueError("data must be contiguous")
    if not data.is_cuda:
        raise ValueError("data must live on a HIP device: the PCGmix kernels have no CPU path")


def spline_operator(device: torch.device, sig_len: int, n_knots: int) -> torch.Tensor:
    """Constant knots->coefficients operator of the warp spline, resident on ``device``."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           sig_len, n_knots)
    op = _OP_CACHE.get(key)
    if op is None:
        lib = _lib.load()
        host = np.empty(lib.pcgmix_spline_operator_size(n_knots), dtype=np.float64)
        _lib.check(lib.pcgmix_spline_operator_f64(sig_len, n_knots, host.ctypes.data),
                   "pcgmix_spline_operator_f64")
        op = torch.from_numpy(host).to(device)
        _OP_CACHE[key] = op
    return op


class _StagingRing:
    """Pinned host staging for the per-step index upload.

    ``tensor.pin_memory()`` per call costs ~1 ms on this stack (a fresh hipHostMalloc whenever
    the previous block's copy has not retired yet), so a small ring of pinned buffers is kept
    per device instead; a slot is reused only after the event recorded behind its last copy has
    completed, which keeps the async H2D copy race-free without ever blocking in steady state."""

    SLOTS = 8

    def __init__(self):
        self.bufs = [None] * self.SLOTS
        self.events = [None] * self.SLOTS
        self.next = 0

    def stage(self, nbytes: int):
        i = self.next
        self.next = (i + 1) % self.SLOTS
        if self.events[i] is not None:
            self.events[i].synchronize()
        buf = self.bufs[i]
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(4096, 1 << (nbytes - 1).bit_length()), dtype=torch.uint8,
                              pin_memory=True)
            self.bufs[i] = buf
        return i, buf

    def sent(self, i: int, stream):
        ev = self.events[i]
        if ev is None:
            ev = self.events[i] = torch.cuda.Event()
        ev.record(stream)


_RINGS: dict = {}
_BLIT_LIMIT = 16384       # hipMemcpyAsync uses a blit kernel up to here, the SDMA engine above


def _h2d(dev: torch.Tensor, pinned: torch.Tensor, nbytes: int) -> None:
    """Async copy of the first ``nbytes`` of a pinned staging buffer into the uint8 device tensor
    ``dev`` on the current stream.  Above 16 KB the copy is one launch of the library's fetch
    kernel (``pcgmix_fetch_h2d``) instead of an SDMA transfer that stalls the stream for ~25 us;
    both buffers then have to hold ``nbytes`` rounded up to 16."""
    if nbytes <= _BLIT_LIMIT:
        dev[:nbytes].copy_(pinned[:nbytes], non_blocking=True)
        return
    _lib.check(_lib.load().pcgmix_fetch_h2d(pinned.data_ptr(), dev.data_ptr(), nbytes,
                                            ctypes.c_void_p(_raw_stream(dev.device))),
               "pcgmix_fetch_h2d")


def upload_array(arr: np.ndarray, device: torch.device) -> torch.Tensor:
    """Async H2D copy of a small host array through the pinned staging ring (never synchronises
    the stream, unlike ``torch.from_numpy(a).to(device)`` from pageable memory).  Must be called
    with ``device`` current.  Returns a device tensor of arr's dtype and shape."""
    arr = np.ascontiguousarray(arr)
    nbytes = arr.nbytes
    ring = _RINGS.setdefault(device.index, _StagingRing())
    slot, pinned = ring.stage(max(nbytes, 1))
    pinned.numpy()[:nbytes] = arr.reshape(-1).view(np.uint8)
    dev = torch.empty((max(nbytes, 1) + 15) // 16 * 16, dtype=torch.uint8, device=device)
    _h2d(dev, pinned, max(nbytes, 1))
    ring.sent(slot, torch.cuda.current_stream(device))
    return dev[:nbytes].view(torch.from_numpy(arr[:0].reshape(-1)).dtype).view(arr.shape)


def upload_into(dst: torch.Tensor, arr: np.ndarray) -> None:
    """Async H2D copy of a small host array straight into the device tensor ``dst`` (same byte
    size) through the pinned staging ring.  Must be called with ``dst``'s device current."""
    arr = np.ascontiguousarray(arr)
    nbytes = arr.nbytes
    if nbytes != dst.numel() * dst.element_size() or not dst.is_contiguous():
        raise ValueError("upload_into: size mismatch")
    ring = _RINGS.setdefault(dst.device.index, _StagingRing())
    slot, pinned = ring.stage(max(nbytes, 1))
    pinned.numpy()[:nbytes] = arr.reshape(-1).view(np.uint8)
    dst.view(torch.uint8).view(-1).copy_(pinned[:nbytes], non_blocking=True)
    ring.sent(slot, torch.cuda.current_stream(dst.device))


_PACK_ERRORS = {1: "frames must be non-decreasing and non-negative",
                2: "heart cycle ends beyond the signal length",
                3: "partner index out of range"}


def upload_plan(plan: MixPlan, frames: np.ndarray, device: torch.device, sig_len: int = 2**31 - 1):
    """Validate the boundaries and send every small per-step array in ONE async H2D copy:
    int32 frames (B,5) | mix (B) | offsets (B,4, optional) | zero rectangles (B,4, optional) |
    float64 knots (optional).  Packing + validation run in the library
    (pcgmix_pack_plan_i32) straight into pinned staging.  Returns (device buffer, byte offsets).
    Must be called with ``device`` current.  Raises ValueError on malformed frames."""
    B = frames.shape[0]
    n_off = B * 4 if plan.rand_off is not None else 0
    n_rect = B * 4 if plan.zero_rect is not None else 0
    n_int = B * 5 + B + n_off + n_rect
    n_int_pad = (n_int + 1) & ~1                      # keep the float64 block 8-byte aligned
    n_kn = plan.knots.size if plan.knots is not None else 0
    nbytes = n_int_pad * 4 + n_kn * 8
    ring = _RINGS.setdefault(device.index, _StagingRing())
    slot, pinned = ring.stage(nbytes)
    lib = _lib.load()
    err = lib.pcgmix_pack_plan_i32(
        frames.ctypes.data, plan.mix.ctypes.data,
        plan.rand_off.ctypes.data if n_off else None,
        plan.zero_rect.ctypes.data if n_rect else None, B, int(sig_len), pinned.data_ptr())
    if err:
        raise ValueError(_PACK_ERRORS.get(err, f"pcgmix_pack_plan_i32 error {err}"))
    if n_kn:
        pinned.numpy()[n_int_pad * 4:nbytes].view(np.float64)[:] = plan.knots.reshape(-1)
    dev = torch.empty((nbytes + 15) // 16 * 16, dtype=torch.uint8, device=device)
    _h2d(dev, pinned, nbytes)
    ring.sent(slot, torch.cuda.current_stream(device))
    offs = {"frames": 0, "mix": B * 5 * 4, "off": B * 6 * 4 if n_off else None,
            "rect": (B * 6 + n_off) * 4 if n_rect else None,
            "knots": n_int_pad * 4 if n_kn else None}
    return dev, offs


def launch_mix(data: torch.Tensor, out: torch.Tensor, frames_ptr: int, mix_ptr: int,
               off_ptr: Optional[int], lam: float, knots_ptr: Optional[int],
               op_ptr: Optional[int], n_knots: int, B: int, C: int, T: int,
               rect_ptr: Optional[int] = None) -> None:
    lib = _lib.load()
    stream = _raw_stream(data.device)
    err = lib.pcgmix_mix_warp_f32(data.data_ptr(), out.data_ptr(), frames_ptr, mix_ptr, off_ptr,
                                  ctypes.c_float(lam), knots_ptr, op_ptr, n_knots, rect_ptr,
                                  B, C, T, ctypes.c_void_p(stream))
    _lib.check(err, "pcgmix_mix_warp_f32")


def apply_plan(plan: MixPlan, data: torch.Tensor, frames: np.ndarray,
               saliency_maps: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Run the device part of a fired plan; returns the new (B,C,T) tensor (``out`` if given:
    a static buffer a captured hipGraph reads from)."""
    B, C, T = data.shape
    device = data.device
    if B == 0:                        # nothing to launch (and an empty tensor has no storage)
        return torch.empty_like(data) if out is None else out
    with torch.cuda.device(device):
        dev, offs = upload_plan(plan, frames, device, T)
        base = dev.data_ptr()
        frames_ptr, mix_ptr = base + offs["frames"], base + offs["mix"]
        off_ptr = base + offs["off"] if offs["off"] is not None else None
        keep = [dev]
        if plan.salopt_mode is not None and saliency_maps is None:
            raise ValueError("saliency-guided mixing needs saliency maps")
        knots_ptr = op_ptr = None
        if plan.knots is not None:
            op = spline_operator(device, T, plan.n_knots)
            keep.append(op)
            knots_ptr, op_ptr = base + offs["knots"], op.data_ptr()
        if out is None:
            out = torch.empty_like(data)
        elif out.shape != data.shape or out.dtype != data.dtype or not out.is_contiguous() \
                or out.data_ptr() == data.data_ptr():
            raise ValueError("out must be a distinct contiguous tensor shaped like data")
        rect_ptr = base + offs["rect"] if offs["rect"] is not None else None
        if plan.salopt_mode is not None:
            # displacement search + splice in one call: the splice kernel reduces the search's
            # per-block results itself (pcgmix_salopt_mix_warp_f32)
            sal = saliency_maps
            if sal.shape != (B, T) or sal.dtype != torch.float32 or not sal.is_contiguous() \
                    or sal.device != device:
                raise ValueError("saliency maps must be a contiguous float32 (B, T) device tensor")
            lib = _lib.load()
            ws = torch.empty(max(1, lib.pcgmix_salopt_workspace_bytes(B) // 8), dtype=torch.int64,
                             device=device)
            keep.append(ws)
            _lib.check(lib.pcgmix_salopt_mix_warp_f32(
                data.data_ptr(), out.data_ptr(), sal.data_ptr(), frames_ptr, mix_ptr,
                ctypes.c_float(float(plan.lam32)), plan.salopt_mode, knots_ptr, op_ptr, plan.n_knots,
                ws.data_ptr(), int(np.diff(frames, axis=1).max()), None, B, C, T,
                ctypes.c_void_p(_raw_stream(device))), "pcgmix_salopt_mix_warp_f32")
        else:
            launch_mix(data, out, frames_ptr, mix_ptr, off_ptr, float(plan.lam32), knots_ptr, op_ptr,
                       plan.n_knots, B, C, T, rect_ptr)
        # the small buffers are only read by work already enqueued on this stream; torch's
        # caching allocator reuses them in stream order, so dropping the references is safe
        del keep
    return out


_SPLICE_ERRORS = {-1: _PACK_ERRORS[1], -2: _PACK_ERRORS[2], -3: _PACK_ERRORS[3]}

_CTX: dict = {}           # device index -> pcgmix_ctx* (per-device step context, never freed)
_c_float = ctypes.c_float
_get_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


_NOT_ARMED = -3          # PCGMIX_NOT_ARMED (include/pcgmix_hip.h)
_I64_ARRAYS = {}


def _fresh_out_ok(out: torch.Tensor, data: torch.Tensor) -> bool:
    return out.shape == data.shape and out.dtype == data.dtype and out.is_contiguous() \
        and out.data_ptr() != data.data_ptr()



def _index_out(n: int):
    """(ctypes buffer, ndarray over it) for ``n`` int64 the library fills: a fresh ctypes array handed to
    the call as it is and wrapped by numpy costs 0.5 us, ``np.empty`` + ``.ctypes.data`` 1.45 us (the
    ``.ctypes`` helper object) — on the host chain of a 16 us step."""
    t = _I64_ARRAYS.get(n)
    if t is None:
        t = _I64_ARRAYS[n] = ctypes.c_int64 * n
    buf = t()
    return buf, np.ndarray((n,), np.int64, buf)


def step_context(index: int) -> int:
    """The library's per-device step context (pinned staging ring + device twins + label
    read-back memory + spline operators), created on first use."""
    ctx = _CTX.get(index)
    if ctx is None:
        lib = _lib.load()
        h = ctypes.c_void_p()
        _lib.check(lib.pcgmix_ctx_create(index, ctypes.byref(h)), "pcgmix_ctx_create")
        ctx = _CTX[index] = h.value
    return ctx


def _frames_ptr(frames, B: int):
    """(address, keep-alive) of the (B,5) int64 boundaries on the host.  The reference's loader
    yields a contiguous CPU int64 tensor (dataloader_physionet.py:151-172): its storage is used as
    it is; anything else goes through numpy."""
    if isinstance(frames, torch.Tensor) and frames.dtype == torch.int64 and not frames.is_cuda \
            and frames.is_contiguous():
        if frames.shape != (B, 5):
            raise ValueError("labels/frames do not match the batch size")
        return frames.data_ptr(), frames
    arr = _as_numpy_frames(frames)
    if arr.shape != (B, 5):
        raise ValueError("labels/frames do not match the batch size")
    return arr.ctypes.data, arr


def splice_plain(recipe, data: torch.Tensor, labels, frames, step: int,
                 out: Optional[torch.Tensor] = None, target_ohe: Optional[torch.Tensor] = None):
    """One fired step of a plain splice (``hostprep.plain_recipe``) through
    ``pcgmix_augment_plain_f32``: label read-back, partner draw, validation, packing, the single
    H2D copy and the launch happen inside the library, on staging the library owns; Python only
    draws lambda (and the warp knots) from numpy's global stream exactly where the reference
    does (augmentations.py:661-663, 677) and allocates the output.  ``data`` is (B, C, T) — the
    2D path passes (B, F, W).  ``labels`` = host class labels, or None with ``target_ohe`` = the
    device one-hot int64 matrix.  Returns (out, mix_indices)."""
    _name, _p, alpha, sigma, n_knots = recipe
    B, C, T = data.shape
    idx = data.device.index
    ohe_ptr, n_cls, lab_ptr = None, 0, None
    if labels is None:
        ohe = target_ohe                          # (only its metadata and address are read)
        if ohe.is_cuda and ohe.dtype == torch.int64 and ohe.dim() == 2 and ohe.is_contiguous():
            if ohe.shape[0] != B:
                raise ValueError("labels/frames do not match the batch size")
            ohe_ptr, n_cls = ohe.data_ptr(), ohe.shape[1]
        else:
            labels = labels_from_ohe(target_ohe)
    if ohe_ptr is None:
        labels = np.ascontiguousarray(np.asarray(labels).reshape(-1), dtype=np.int64)
        if labels.shape[0] != B:
            raise ValueError("labels/frames do not match the batch size")
        lab_ptr = labels.ctypes.data
    fr_ptr, fr_keep = _frames_ptr(frames, B)
    stream = _get_raw_stream(idx) if _get_raw_stream is not None \
        else torch.cuda.current_stream(data.device).cuda_stream
    if ohe_ptr is not None and n_knots == 0 and (out is None or _fresh_out_ok(out, data)):
        # Strict signature, no warp: the ARMED kernel is launched FIRST (it needs neither lambda nor the
        # partners: both reach it in its records), lambda is drawn from numpy's stream while the GPU
        # gets to the kernel and the labels come back, then the second call delivers the records.
        # (The reference draws the partners from `random` and lambda from numpy: independent streams,
        # the order between them is free.)
        lib = _lib.load()
        ctx = _CTX.get(idx) or step_context(idx)
        if out is None:
            out = torch.empty_like(data)
        err = lib.pcgmix_augment_plain_begin(ctx, data.data_ptr(), out.data_ptr(), ohe_ptr, n_cls, B, C, T, stream)
        if err == 0:
            lam, _ = hostprep.draw_lambda_knots(step, alpha, sigma, 0)
            mix_buf, mix = _index_out(B)
            err = lib.pcgmix_augment_plain_finish(ctx, fr_ptr, step, _c_float(lam), mix_buf)
            if err:
                if err < 0:
                    raise ValueError(_SPLICE_ERRORS.get(err, f"pcgmix_augment_plain_finish error {err}"))
                _lib.check(err, "pcgmix_augment_plain_finish")
            return out, mix
        if err != _NOT_ARMED:
            _lib.check(err, "pcgmix_augment_plain_begin")
    # numpy's global stream, as the reference: seed -> beta -> normal (c_float rounds lam like
    # np.float32, :903); the block was usually drawn ahead by the library (hostprep)
    lam, knots = hostprep.draw_lambda_knots(step, alpha, sigma, B * n_knots * C)
    knots_ptr = knots.ctypes.data if isinstance(knots, np.ndarray) else knots
    if out is None:
        out = torch.empty_like(data)
    elif out.shape != data.shape or out.dtype != data.dtype or not out.is_contiguous() \
            or out.data_ptr() == data.data_ptr():
        raise ValueError("out must be a distinct contiguous tensor shaped like data")
    mix_buf, mix = _index_out(B)
    err = _lib.load().pcgmix_augment_plain_f32(
        _CTX.get(idx) or step_context(idx), data.data_ptr(), out.data_ptr(), ohe_ptr, n_cls, lab_ptr,
        fr_ptr, step, _c_float(lam), knots_ptr, n_knots, mix_buf, B, C, T, stream)
    if err:
        if err < 0:
            raise ValueError(_SPLICE_ERRORS.get(err, f"pcgmix_augment_plain_f32 error {err}"))
        _lib.check(err, "pcgmix_augment_plain_f32")
    return out, mix


def _salopt_step(srec, g, data: torch.Tensor, ohe: Optional[torch.Tensor], labels, frames, step: int,
                 out: Optional[torch.Tensor] = None):
    """One fired saliency-guided step with same-label partners: two library calls around the
    captured saliency pass.  ``pcgmix_ctx_salopt_begin``: label arg-max kernel (labels to the
    host, gradient seed to the graph) + boundaries to the device — or, when the caller holds the
    labels on the host (``labels``; ``ohe`` may then be None), ``pcgmix_ctx_salopt_begin_labels``:
    seed, boundaries and a pending step payload from the arguments of one launch, no read-back;
    replay; lambda and knots from numpy's global stream (they do not depend on the labels) while
    the GPU works; ``pcgmix_ctx_salopt_finish``: labels picked up, partners drawn, one H2D copy,
    displacement search and fused splice(+warp) into ``out`` (a captured training step's static
    input) or a new tensor.  Returns (out, mix_indices)."""
    mode, alpha, sigma, n_knots = srec
    B, C, T = data.shape
    idx = data.device.index
    lib = _lib.load()
    ctx = _CTX.get(idx) or step_context(idx)
    stream = _get_raw_stream(idx) if _get_raw_stream is not None \
        else torch.cuda.current_stream(data.device).cuda_stream
    fr_ptr, fr_keep = _frames_ptr(frames, B)
    lab_ptr = None
    if labels is not None:
        labels = np.ascontiguousarray(np.asarray(labels).reshape(-1), dtype=np.int64)
        if labels.shape[0] != B:
            raise ValueError("labels/frames do not match the batch size")
        lab_ptr = labels.ctypes.data
    if out is not None and (out.shape != data.shape or out.dtype != data.dtype
                            or not out.is_contiguous() or out.data_ptr() == data.data_ptr()):
        raise ValueError("out must be a distinct contiguous tensor shaped like data")
    if lab_ptr is not None:
        name = "pcgmix_ctx_salopt_begin_labels"
        err = lib.pcgmix_ctx_salopt_begin_labels(ctx, lab_ptr, g.seed.shape[1], g.seed.data_ptr(),
                                                 fr_ptr, g.fr.data_ptr(), B, T, stream)
    else:
        name = "pcgmix_ctx_salopt_begin"
        err = lib.pcgmix_ctx_salopt_begin(ctx, ohe.data_ptr(), ohe.shape[1], g.seed.data_ptr(), fr_ptr,
                                          g.fr.data_ptr(), B, T, stream)
    if err:
        if err < 0:
            raise ValueError(_SPLICE_ERRORS.get(err, f"{name} error {err}"))
        _lib.check(err, name)
    if torch.cuda.current_device() == idx:
        sal = g.replay(data)
    else:
        with torch.cuda.device(data.device):
            sal = g.replay(data)
    # numpy's global stream, as the reference: seed -> beta -> normal (c_float rounds lam like
    # np.float32, :903); the block was usually drawn ahead by the library (hostprep)
    lam, knots = hostprep.draw_lambda_knots(step, alpha, sigma, B * n_knots * C)
    knots_ptr = knots.ctypes.data if isinstance(knots, np.ndarray) else knots
    if out is None:
        out = torch.empty_like(data)
    mix = np.empty(B, dtype=np.int64)
    _lib.check(lib.pcgmix_ctx_salopt_finish(
        ctx, data.data_ptr(), out.data_ptr(), sal.data_ptr(), g.fr.data_ptr(), lab_ptr, step,
        _c_float(lam), mode, knots_ptr, n_knots, mix.ctypes.data, B, C, T, stream),
        "pcgmix_ctx_salopt_finish")
    return out, mix


def gate_passes(recipe, method: str, step: int, index: int) -> bool:
    """Probability gate of a plain method: ``Random(step).uniform(0,1) < p`` (augmentations.py:
    869-872), drawn in the step context so that the fired step reuses the seeded generator."""
    if recipe[1] >= 1.0:
        return True
    return _lib.load().pcgmix_ctx_gate(_CTX.get(index) or step_context(index), step) < recipe[1]


def blend_targets(target_ohe: torch.Tensor, plan: MixPlan) -> torch.Tensor:
    """'(mixAll)': float blend of the one-hot targets (augmentations.py:915-917, 978-980)."""
    B = target_ohe.shape[0]
    lams = torch.from_numpy((np.ones(B) * plan.lam64).astype("float32")).to(target_ohe.device)
    lt = lams[:, None]
    mix = torch.from_numpy(plan.mix).to(target_ohe.device)
    return target_ohe * lt + target_ohe[mix] * (1 - lt)


_LABEL_PINNED: dict = {}


_SIDE_STREAMS: dict = {}


def labels_from_ohe(target_ohe: torch.Tensor, after: Optional["torch.cuda.Event"] = None) -> np.ndarray:
    """Reverse the one-hot encoding on the host (augmentations.py:501): one D2H copy of the
    (B, classes) matrix, argmax (first maximum, like torch.max) in numpy — no reduce kernel.
    The copy lands in a cached pinned buffer and the launch stream is synchronised: with the
    previous step's kernel still in flight ``tensor.cpu()`` measured 51 us, this 23 us
    (profiles/probes/label_readback.py).

    ``after``: an event recorded on the launch stream.  The copy then runs on a side stream that
    waits for that event only, so work enqueued on the launch stream AFTER the event (the
    saliency graph of a saliency-guided step) does not delay the read-back."""
    t = target_ohe.detach()
    if not t.is_cuda:
        return t.numpy().argmax(axis=1)
    key = (t.device.index, t.dtype)
    buf = _LABEL_PINNED.get(key)
    if buf is None or buf.numel() < t.numel():
        buf = torch.empty(max(4096, t.numel()), dtype=t.dtype, pin_memory=True)
        _LABEL_PINNED[key] = buf
    host = buf[:t.numel()].view(t.shape)
    if after is None:
        host.copy_(t, non_blocking=True)
        torch.cuda.current_stream(t.device).synchronize()
    else:
        side = _SIDE_STREAMS.get(t.device.index)
        if side is None:
            side = _SIDE_STREAMS[t.device.index] = torch.cuda.Stream(t.device)
        side.wait_event(after)
        with torch.cuda.stream(side):
            host.copy_(t, non_blocking=True)
        side.synchronize()
    return host.numpy().argmax(axis=1)


def augment(args, data, target_ohe, frames, wav, step_counter, model, device, RESULTS_ARGS,
            host_labels=None):
    """See module docstring.  Returns ``(data, target_ohe, mix_indices, cut)``; when the method
    does not apply or the gate rejects the step, ``data`` is the very object passed in and
    ``mix_indices`` is ``[]`` (augmentations.py:731-732, 871-872, 938-939).

    ``host_labels`` (extension, keyword only in spirit): the integer class labels as a host
    array.  A training loop that still has the loader's CPU ``target`` can pass it to spare the
    device->host read-back of ``target_ohe`` (the reference's only unavoidable sync,
    augmentations.py:501); results are identical."""
    method = args.method
    step = int(step_counter.count)
    recipe = _RECIPES.get(method, _RECIPES)
    if recipe is _RECIPES:                        # first sight of this method string
        recipe = hostprep.plain_recipe(method, False) \
            if hostprep.select_method(method, is2d=False) is not None else False
        _RECIPES[method] = recipe
    if recipe is False:                           # not one of ours: passthrough (:731-732)
        return data, target_ohe, [], None
    _check_data(data, 3)
    B, C, T = data.shape
    if recipe is not None and B > 0:              # the common case: one library call
        if not gate_passes(recipe, method, step, data.device.index):
            return data, target_ohe, [], None
        out, mix = splice_plain(recipe, data, host_labels, frames, step, target_ohe=target_ohe)
        return out, target_ohe, mix, None
    frames_np = _as_numpy_frames(frames)
    labels = (lambda: labels_from_ohe(target_ohe)) if host_labels is None else np.asarray(host_labels)
    sal = None
    if "(salopt" in method and B > 0:
        # Saliency-guided step: enqueue the frozen model's forward + input gradient +
        # post-processing FIRST, then do the host part of the plan (label read-back, permutation,
        # lambda, B*(k+2)*C normal draws: ~0.1 ms) while the GPU works.  The saliency maps use no
        # host RNG, so the reference's draw order is kept.
        if not hostprep.gate_fires(method, step):
            return data, target_ohe, [], None
        from . import saliency as _sal
        # The label arg-max kernel of the step context goes first: it writes the labels into
        # host-mapped memory (the read-back must wait for whatever produced target_ohe, but not
        # for the saliency pass) and, on the way, the saliency pass's gradient seed; then the
        # captured pass; the host picks the labels up when it needs them — by then the kernel has
        # long finished and the GPU is inside the graph.
        ohe = target_ohe.detach()
        g = _sal.step_graph(args, data, ohe.shape[1]) if ohe.dim() == 2 else None
        ohe_ok = ohe.dim() == 2 and ohe.is_cuda and ohe.dtype == torch.int64 and ohe.is_contiguous() \
            and ohe.shape[0] == B
        srec = hostprep.salopt_recipe(method)
        if g is not None and srec is not None and (ohe_ok or host_labels is not None):
            out, mix = _salopt_step(srec, g, data, ohe if host_labels is None else None, host_labels,
                                    frames, step)
            return out, target_ohe, mix, None
        if g is not None and ohe_ok:
            lib = _lib.load()
            idx = data.device.index
            ctx = _CTX.get(idx) or step_context(idx)
            stream = _raw_stream(data.device)
            _lib.check(lib.pcgmix_ctx_labels_begin(ctx, ohe.data_ptr(), ohe.shape[1], B,
                                                   g.seed.data_ptr(), stream), "pcgmix_ctx_labels_begin")
            with torch.cuda.device(data.device):
                sal = g.run(data, frames_np, keep=False)         # consumed below, in this call
            if host_labels is None:
                def labels():           # asked for by make_plan after its numpy draws
                    out = np.empty(B, dtype=np.int64)
                    _lib.check(lib.pcgmix_ctx_labels_wait(ctx, out.ctypes.data, B, stream),
                               "pcgmix_ctx_labels_wait")
                    return out
        elif host_labels is None:
            mark = torch.cuda.Event()
            mark.record(torch.cuda.current_stream(data.device))
            sal = _sal.get_saliency_maps(args, data.device, data, target_ohe, frames_np, dim=1,
                                         gauss_k_n=101)
            labels = labels_from_ohe(target_ohe, after=mark)
        else:
            sal = _sal.get_saliency_maps(args, data.device, data, target_ohe, frames_np, dim=1,
                                         gauss_k_n=101)
    plan = hostprep.make_plan(method, labels, frames_np, wav, step, B, C, is2d=False)
    if not plan.fired:
        return data, target_ohe, [], None
    out = apply_plan(plan, data, frames_np, sal)
    if plan.mix_all:
        target_ohe = blend_targets(target_ohe, plan)
    return out, target_ohe, plan.mix, None
