"""1D models of the PCGmix hot path, re-declared on PyTorch-ROCm with the reference's
``state_dict`` layout so checkpoints are interchangeable.

  CNN_potes     the "1D-CNN" (Potes et al.), reference models.py:367-465, factory :345-350
  ResNet9       Myrtle ResNet9 with 1D convolutions, reference models.py:520-589

Both keep the reference's ``forward(x, depth=None, pass_part=None)`` signature; the training
loop calls ``model(data, depth=0, pass_part='second')`` (train_model.py:537), which is the plain
full forward pass.  Convolutions run through MIOpen (MFMA paths for the ResNet9 GEMMs).
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


_WARNED: set = set()


def _warn_once(msg: str) -> None:
    if msg not in _WARNED:
        _WARNED.add(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)


class PotesStackFunction(torch.autograd.Function):
    """conv(1->8,k5,p1)+ReLU+pool2 -> conv(8->4,k5,p1)+ReLU+pool2 on (N,T) rows as ONE HIP kernel
    forward and one (+ a 212-block reduction) backward (csrc/pcgmix_potes.hip).

    ``use_masks`` (default): when a gradient will be needed the forward also stores where its
    ReLUs were alive and which element won each max-pool (2 bits per second-layer output; a byte
    per first-layer position if the input gradient is needed) — what autograd keeps for
    nn.ReLU/nn.MaxPool1d, minus the activations.  The weight-gradient backward then recomputes
    only the first layer from the saved input (``pcgmix_potes_stack_bwd_mask_f32``), and the input
    gradient (saliency maps) recomputes nothing (``pcgmix_potes_stack_input_grad_mask_f32``).
    ``use_masks = False``: the kernels that recompute the whole forward per tile and keep nothing
    but the input row (``pcgmix_potes_stack_{bwd,input_grad}_f32``)."""

    use_masks = True
    # A dict while a captured training step records its backward and the optimiser update is the
    # next node (``GraphedTrainStep`` with ``ClipAdam``, one rank): the mask-based weight-gradient
    # backward then leaves its per-block partials un-reduced and reports them here ("partial",
    # "grads", "G") — ``ClipAdam.capture_update`` reduces them inside its own launch
    # (``pcgmix_adam_clip_multi_reduce_dev_f32``: one launch less per replay), or, if it cannot,
    # with ``pcgmix_potes_reduce_f32``.  None everywhere else.
    defer_reduce = None

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, rnd=None, key=None):
        # rnd: a uint8 buffer the forward kernel fills, on the side, with the random bytes the
        # head's dropouts read (pcgmix_hip.h); key: the 64-bit key of the fill — an int, or a
        # device tensor of two int32 words (captured step: the key changes per replay)
        N, T = x.shape
        lib = _lib.load()
        P2 = lib.pcgmix_potes_out_len(T)
        w1c, b1c, w2c, b2c = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        h2 = torch.empty((N, 4, P2), dtype=torch.float32, device=x.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        need_x = ctx.needs_input_grad[0]
        m2 = s1 = None
        # rnd without any gradient (frozen conv stack, head-only fine-tuning in train mode): the
        # mask-saving forward still runs, for the dropout bytes it fills on the side
        if PotesStackFunction.use_masks and N > 0 and (any(ctx.needs_input_grad) or rnd is not None):
            m2 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=x.device)
            if need_x:
                s1 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 1), dtype=torch.uint8,
                                 device=x.device)
            _lib.check(lib.pcgmix_potes_stack_fwd_save_f32(
                x.data_ptr(), w1c.data_ptr(), b1c.data_ptr(), w2c.data_ptr(), b2c.data_ptr(),
                h2.data_ptr(), m2.data_ptr(), s1.data_ptr() if s1 is not None else None, N, T,
                rnd.data_ptr() if rnd is not None else None, rnd.numel() if rnd is not None else 0,
                key.data_ptr() if torch.is_tensor(key) else None,
                0 if (key is None or torch.is_tensor(key)) else int(key), stream),
                "pcgmix_potes_stack_fwd_save_f32")
        elif rnd is not None:
            raise RuntimeError("PotesStackFunction: dropout bytes are filled by the mask-saving "
                               "forward (needs use_masks)")
        else:
            _lib.check(lib.pcgmix_potes_stack_fwd_f32(x.data_ptr(), w1c.data_ptr(), b1c.data_ptr(),
                                                      w2c.data_ptr(), b2c.data_ptr(), h2.data_ptr(),
                                                      N, T, stream), "pcgmix_potes_stack_fwd_f32")
        ctx.save_for_backward(x, w1c, b1c, w2c, b2c, m2, s1)
        return h2

    @staticmethod
    def backward(ctx, grad_h2):
        x, w1, b1, w2, b2, m2, s1 = ctx.saved_tensors
        N, T = x.shape
        lib = _lib.load()
        g = grad_h2.contiguous()
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        gx = gw1 = gb1 = gw2 = gb2 = None
        if ctx.needs_input_grad[0]:                    # saliency: d score / d input
            gx = torch.empty_like(x)
            if s1 is not None:
                _lib.check(lib.pcgmix_potes_stack_input_grad_mask_f32(
                    g.data_ptr(), m2.data_ptr(), s1.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                    gx.data_ptr(), N, T, stream), "pcgmix_potes_stack_input_grad_mask_f32")
            else:
                _lib.check(lib.pcgmix_potes_stack_input_grad_f32(
                    x.data_ptr(), g.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                    b2.data_ptr(), gx.data_ptr(), N, T, stream), "pcgmix_potes_stack_input_grad_f32")
        if any(ctx.needs_input_grad[1:]):
            G = lib.pcgmix_potes_bwd_blocks(N, T)
            partial = torch.empty((G, 212), dtype=torch.float32, device=x.device)
            grads = torch.empty(212, dtype=torch.float32, device=x.device)
            if m2 is not None:
                defer = PotesStackFunction.defer_reduce
                if defer is not None and "partial" in defer:
                    defer = None                       # a second stack in the same step: reduce here
                _lib.check(lib.pcgmix_potes_stack_bwd_mask_f32(
                    x.data_ptr(), g.data_ptr(), m2.data_ptr(), w1.data_ptr(), b1.data_ptr(),
                    w2.data_ptr(), b2.data_ptr(), partial.data_ptr(),
                    grads.data_ptr() if defer is None else None, N, T, stream),
                    "pcgmix_potes_stack_bwd_mask_f32")
                if defer is not None:
                    defer.update(partial=partial, grads=grads, G=G)
            else:
                _lib.check(lib.pcgmix_potes_stack_bwd_f32(
                    x.data_ptr(), g.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                    b2.data_ptr(), partial.data_ptr(), grads.data_ptr(), N, T, stream),
                    "pcgmix_potes_stack_bwd_f32")
            gw1, gb1 = grads[0:40].view(8, 1, 5), grads[40:48]
            gw2, gb2 = grads[48:208].view(4, 8, 5), grads[208:212]
        return gx, gw1, gb1, gw2, gb2, None, None


class SkinnyLinearFunction(torch.autograd.Function):
    """y = x W^T + b for a tall-K, skinny-N layer (the Potes ``dimreduc`` 19968 -> 20) through
    ``pcgmix_skinny_linear_fwd_f32`` (split-K, deterministic); the backward GEMMs are well served
    by hipBLASLt and stay in torch."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, K = x.shape
        O = weight.shape[0]
        lib = _lib.load()
        xc, wc = x.contiguous(), weight.detach().contiguous()
        ks = lib.pcgmix_skinny_linear_splits(B, K)
        partial = torch.empty((ks, B, O), dtype=torch.float32, device=x.device)
        z = torch.empty((B, O), dtype=torch.float32, device=x.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.pcgmix_skinny_linear_fwd_f32(
            xc.data_ptr(), wc.data_ptr(), bias.detach().data_ptr() if bias is not None else None,
            partial.data_ptr(), z.data_ptr(), B, K, O, stream), "pcgmix_skinny_linear_fwd_f32")
        ctx.save_for_backward(xc, wc)
        ctx.has_bias = bias is not None
        return z

    @staticmethod
    def backward(ctx, gz):
        x, w = ctx.saved_tensors
        gx = gz.mm(w) if ctx.needs_input_grad[0] else None
        gw = gz.t().mm(x) if ctx.needs_input_grad[1] else None
        gb = gz.sum(0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        return gx, gw, gb


def dropout_threshold(p: float, max_bits: int = 8):
    """(bits, thr, scale) of the rule the head kernels apply: an element owns ``bits`` uniformly
    random bits and is kept iff their value >= thr.  The fewest bits (1, 2, 4, 8) that represent p
    exactly are used (0.5: 1 bit, 0.25: 2 bits); otherwise 8 bits with p rounded to 1/256ths.
    Kept values are scaled by 2^bits/(2^bits - thr)."""
    for bits in (1, 2, 4, 8):
        if bits > max_bits:
            break
        v = p * (1 << bits)
        if abs(v - round(v)) < 1e-9 or bits == 8:
            thr = min((1 << bits) - 1, max(0, int(round(v))))
            return bits, thr, float(1 << bits) / float((1 << bits) - thr)
    bits = max_bits
    thr = min((1 << bits) - 1, max(0, int(round(p * (1 << bits)))))
    return bits, thr, float(1 << bits) / float((1 << bits) - thr)


def next_dropout_key(device: torch.device) -> int:
    """A 64-bit key for one training forward's dropout bytes, from torch's device generator —
    splitmix64 of (seed, Philox offset), the offset advanced by 4 — on the host, without a launch:
    ``torch.manual_seed`` / ``torch.cuda.manual_seed`` reproduce the masks as they do for
    nn.Dropout (models.py:364, 380), and the reference's reseeding before every optimiser step
    (train_model.py:565) gives every step the same masks here too."""
    gen = torch.cuda.default_generators[device.index if device.index is not None
                                        else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + 4)
    z = (gen.initial_seed() + 0x9E3779B97F4A7C15 * (off // 4 + 1)) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def head_dropout_bytes(B: int, K: int, p1: float = 0.25) -> int:
    """Random bytes one training forward of the head consumes: B*K*bits(p1)/8 for Dropout(p1) on
    the features (padded to 16), B*20 bytes for Dropout(p2) on the hidden layer."""
    bits = dropout_threshold(p1)[0] if p1 > 0.0 else 0
    return ((B * K * bits // 8 + 15) // 16) * 16 + B * 20


class PotesHeadFunction(torch.autograd.Function):
    """Dropout(p1) -> dimreduc Linear(K->20) -> ReLU -> Dropout(p2) -> Linear(20->C) of CNN_potes
    (models.py:364, 376-381, 456-465) as one autograd node: ``pcgmix_potes_head_fwd_f32`` /
    ``pcgmix_potes_head_bwd_f32``.  Both dropouts read uniformly random BYTES — one
    ``random_()`` call on torch's generator (so ``torch.manual_seed`` behaves as with nn.Dropout),
    or ``rnd``, a caller-filled uint8 buffer of ``head_dropout_bytes(B, K)`` bytes (the captured
    training step fills a static one before every replay: RNG calls inside a hipGraph cost two
    extra fill launches per replay) — and are applied where the kernels read their operands: no
    separate masking pass over the feature matrix."""

    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2, p1, p2, training, rnd=None):
        B, K = feat.shape
        C = w2.shape[0]
        dev = feat.device
        lib = _lib.load()
        x = feat.contiguous()
        mask1 = mask2 = None
        thr1 = thr2 = 0
        bits1 = 8
        s1 = s2 = 1.0
        if training and (p1 > 0.0 or p2 > 0.0):
            n = head_dropout_bytes(B, K, p1)
            if rnd is None:
                rnd = torch.empty(n, dtype=torch.uint8, device=dev).random_()
            elif rnd.numel() < n or rnd.dtype != torch.uint8 or not rnd.is_contiguous():
                raise ValueError("rnd must be a contiguous uint8 tensor of head_dropout_bytes(B, K, p1)")
            if p1 > 0.0:
                bits1, thr1, s1 = dropout_threshold(p1)
                mask1 = rnd[:B * K * bits1 // 8]
            if p2 > 0.0:
                _b, thr2, s2 = dropout_threshold(p2, max_bits=8)
                thr2, s2 = (256 * thr2) >> _b, s2           # the hidden mask is read as whole bytes
                mask2 = rnd[n - B * 20:n]
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        ks = lib.pcgmix_skinny_linear_splits(B, K)
        partial = torch.empty((ks, B, 20), dtype=torch.float32, device=dev)
        z = torch.empty((B, 20), dtype=torch.float32, device=dev)
        logits = torch.empty((B, C), dtype=torch.float32, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        opt = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        _lib.check(lib.pcgmix_potes_head_fwd_f32(
            x.data_ptr(), opt(mask1), ctypes.c_float(s1), thr1, bits1, w1c.data_ptr(),
            b1.detach().data_ptr() if b1 is not None else None, opt(mask2), ctypes.c_float(s2), thr2,
            w2c.data_ptr(), b2.detach().data_ptr() if b2 is not None else None, partial.data_ptr(),
            z.data_ptr(), logits.data_ptr(), B, K, C, stream), "pcgmix_potes_head_fwd_f32")
        ctx.save_for_backward(x, w1c, w2c, z, mask1, mask2)
        ctx.drop = (thr1, bits1, s1, thr2, s2)
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        x, w1, w2, z, mask1, mask2 = ctx.saved_tensors
        thr1, bits1, s1, thr2, s2 = ctx.drop
        B, K = x.shape
        C = w2.shape[0]
        dev = x.device
        lib = _lib.load()
        dlogits = dlogits.contiguous()
        dz = torch.empty((B, 20), dtype=torch.float32, device=dev)
        dw2 = torch.empty_like(w2)
        # frozen weights (the saliency model): only dx is wanted, the kernel then never reads x
        need_dw1 = ctx.needs_input_grad[1] or not ctx.needs_input_grad[0]
        dw1 = torch.empty_like(w1) if need_dw1 else None
        db1 = torch.empty(20, dtype=torch.float32, device=dev) if ctx.has_b1 else None
        db2 = torch.empty(C, dtype=torch.float32, device=dev) if ctx.has_b2 else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        opt = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        _lib.check(lib.pcgmix_potes_head_bwd_f32(
            dlogits.data_ptr(), z.data_ptr(), opt(mask2), ctypes.c_float(s2), thr2, w2.data_ptr(),
            x.data_ptr(), opt(mask1), ctypes.c_float(s1), thr1, bits1, w1.data_ptr(),
            dz.data_ptr(), dw2.data_ptr(), opt(db2), opt(db1), opt(dw1), opt(dx), B, K, C,
            stream), "pcgmix_potes_head_bwd_f32")
        return dx, dw1, db1, dw2, db2, None, None, None, None


class PotesHeadLossFunction(torch.autograd.Function):
    """``PotesHeadFunction`` followed by the soft-target cross entropy (CELoss, train_model.py:45-54)
    as ONE autograd node: ``pcgmix_potes_head_loss_{fwd,bwd}_f32``.  Returns (loss, logits); the
    logits are an auxiliary, non-differentiable output (accuracy counters).  Between the split-K
    product and the pass over the features the separate nodes launch four ~5 us kernels for a few
    KB of data; here two.  Whatever gradient arrives at the loss multiplies the stored ones inside
    the backward kernel (no extra launch)."""

    @staticmethod
    def forward(ctx, feat, w1, b1, w2, b2, target, p1, p2, training, rnd=None, defer=False):
        # defer: leave the forward's finalize launch (loss and small gradients from the per-row-block
        # contributions) to the backward's feature pass — the loss is then only valid after
        # backward.  For callers whose forward and backward always run together and who cannot
        # look at the loss in between: a training step being captured (see loss_and_logits).
        B, K = feat.shape
        C = w2.shape[0]
        dev = feat.device
        lib = _lib.load()
        x = feat.contiguous()
        # hard targets may come as uint8 class labels (B,): one byte per row instead of C floats
        hard = target.dtype == torch.uint8 and target.dim() == 1
        if hard and target.shape[0] != B:
            raise ValueError("labels do not match the batch size")
        tgt = target.contiguous() if hard else target.to(torch.float32).contiguous()
        mask1 = mask2 = None
        thr1 = thr2 = 0
        bits1 = 8
        s1 = s2 = 1.0
        if training and (p1 > 0.0 or p2 > 0.0):
            n = head_dropout_bytes(B, K, p1)
            if rnd is None:
                rnd = torch.empty(n, dtype=torch.uint8, device=dev).random_()
            elif rnd.numel() < n or rnd.dtype != torch.uint8 or not rnd.is_contiguous():
                raise ValueError("rnd must be a contiguous uint8 tensor of head_dropout_bytes(B, K, p1)")
            if p1 > 0.0:
                bits1, thr1, s1 = dropout_threshold(p1)
                mask1 = rnd[:B * K * bits1 // 8]
            if p2 > 0.0:
                _b, thr2, s2 = dropout_threshold(p2, max_bits=8)
                thr2 = (256 * thr2) >> _b
                mask2 = rnd[n - B * 20:n]
        w1c, w2c = w1.detach().contiguous(), w2.detach().contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        partial = torch.empty((lib.pcgmix_skinny_linear_splits(B, K), B, 20), **f32)
        z, logits = torch.empty((B, 20), **f32), torch.empty((B, C), **f32)
        dz, loss = torch.empty((B, 20), **f32), torch.empty((), **f32)
        small = torch.empty(C * 20 + C + 20, **f32)
        ws = torch.empty(lib.pcgmix_potes_head_loss_workspace_floats(B), **f32)
        need_dw1 = ctx.needs_input_grad[1] or not ctx.needs_input_grad[0]
        dw1 = torch.empty_like(w1c) if need_dw1 else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        opt = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        defer = bool(defer and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]))
        _lib.check(lib.pcgmix_potes_head_loss_fwd_f32(
            x.data_ptr(), opt(mask1), ctypes.c_float(s1), thr1, bits1, w1c.data_ptr(),
            b1.detach().data_ptr() if b1 is not None else None, opt(mask2), ctypes.c_float(s2), thr2,
            w2c.data_ptr(), b2.detach().data_ptr() if b2 is not None else None, tgt.data_ptr(),
            partial.data_ptr(), z.data_ptr(), logits.data_ptr(), dz.data_ptr(), loss.data_ptr(),
            small.data_ptr(), ws.data_ptr(), opt(dw1), int(defer), int(hard), B, K, C, stream),
            "pcgmix_potes_head_loss_fwd_f32")
        ctx.save_for_backward(x, w1c, dz, small, mask1, dw1)
        # the backward writes the loss through this alias (no version check: under capture nobody
        # can have touched it in between)
        ctx.deferred = (ws, loss.detach()) if defer else (None, None)
        ctx.drop = (thr1, bits1, s1)
        ctx.C = C
        ctx.has_b1, ctx.has_b2 = b1 is not None, b2 is not None
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)        # no zeros for the logits' gradient slot (a fill launch)
        return loss, logits

    @staticmethod
    def backward(ctx, gloss, _glogits):
        x, w1, dz, small, mask1, dw1 = ctx.saved_tensors
        ws, loss = ctx.deferred
        if gloss is None:                       # nothing upstream of the loss
            return (None,) * 11
        # dW1 is accumulated (two row halves, atomicAdd) into the buffer the FORWARD zeroed and is
        # returned as the gradient: a second backward over the same graph would add onto the first
        # result — which may already be p.grad — without an error.  Refuse it.
        if getattr(ctx, "consumed", False):
            raise RuntimeError("PotesHeadLossFunction: backward ran twice over one forward "
                               "(retain_graph / a second autograd.grad): its dW1 buffer is zeroed by "
                               "the forward and consumed by the first backward; run the forward again")
        ctx.consumed = True
        thr1, bits1, s1 = ctx.drop
        B, K = x.shape
        C = ctx.C
        lib = _lib.load()
        g = gloss.to(torch.float32).contiguous()
        small_out = torch.empty_like(small)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        opt = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
        _lib.check(lib.pcgmix_potes_head_loss_bwd_f32(
            dz.data_ptr(), g.data_ptr(), x.data_ptr(), opt(mask1), ctypes.c_float(s1), thr1, bits1,
            w1.data_ptr(), small.data_ptr(), small_out.data_ptr(), opt(dw1), opt(dx), opt(ws), opt(loss),
            B, K, C, stream), "pcgmix_potes_head_loss_bwd_f32")
        dw2 = small_out[:C * 20].view(C, 20)
        db2 = small_out[C * 20:C * 20 + C] if ctx.has_b2 else None
        db1 = small_out[C * 20 + C:] if ctx.has_b1 else None
        return dx, dw1, db1, dw2, db2, None, None, None, None, None, None


def _potes_block(c_in: int, c_out: int, dropout: float = 0.0) -> nn.Sequential:
    # Conv1d(k=5, padding=1) + ReLU + MaxPool(2) [+ Dropout]   (reference models.py:359-365)
    layers = [nn.Conv1d(c_in, c_out, kernel_size=5, padding=1), nn.ReLU(inplace=True),
              nn.MaxPool1d(2)]
    if dropout:
        layers.append(nn.Dropout(dropout))
    return nn.Sequential(*layers)


def potes_flat_features(sig_len: int, width: int = 4, bands: int = 4) -> int:
    """Length of the concatenated feature vector: each band goes conv(k5,p1) -> pool2 twice."""
    n = (sig_len - 2) // 2
    n = (n - 2) // 2
    return bands * width * n


class CNN_potes(nn.Module):
    """Four band-pass channels, each through the SAME branch ``cnn1`` (the reference allocates
    ``cnn2..cnn4`` but never calls them, models.py:444-455; they are kept here so that
    ``state_dict`` keys and parameter counts match — their ``.grad`` stays ``None``)."""

    def __init__(self, c_in: int, c_out: int, layers, linear: int, dropout: float = 0.25):
        super().__init__()
        def branch():
            return nn.Sequential(_potes_block(1, layers[0]),
                                 _potes_block(layers[0], layers[1], dropout=dropout))
        self.cnn1 = branch()
        self.cnn2 = branch()
        self.cnn3 = branch()
        self.cnn4 = branch()
        self.flat1 = nn.Flatten()
        self.flat2 = nn.Flatten()
        self.flat3 = nn.Flatten()
        self.flat4 = nn.Flatten()
        self.dimreduc = nn.Linear(linear, 20)
        self.dropout = nn.Dropout(0.5)
        self.linear = nn.Linear(20, c_out)
        self.fused = True       # use the fused HIP conv stack on a HIP device (see _fused)
        self.dropout_bytes = None   # static random bytes of a captured training step and
        self.dropout_key = None     # the device key they come from (see _static_dropout)

    def _fused(self, x: torch.Tensor) -> bool:
        """The hand-written HIP stack applies to the reference configuration (layers [8,4], float32)
        on a HIP device.  Host tensors (CPU-side tests, gloo rehearsals) and an explicit
        ``self.fused = False`` take torch's ops; a DEVICE tensor that cannot take the HIP kernels
        does so too, but says so — a GPU run must not lose its kernels without a word."""
        if not (self.fused and x.is_cuda):
            return False
        c1, c2 = self.cnn1[0][0], self.cnn1[1][0]
        ok = (x.dtype == torch.float32 and c1.out_channels == 8 and c2.out_channels == 4
              and x.shape[-1] >= 14)
        if not ok:
            _warn_once(f"CNN_potes: input {tuple(x.shape)} {x.dtype} / layers "
                       f"[{c1.out_channels},{c2.out_channels}] cannot use the fused HIP conv stack "
                       "(needs float32, layers [8,4], T >= 14): running torch/MIOpen ops instead")
        return ok

    def _fused_head(self, x: torch.Tensor) -> bool:
        return (self._fused(x) and self.dimreduc.out_features == 20
                and self.linear.in_features == 20 and self.linear.out_features <= 8
                and self.dimreduc.in_features % 4 == 0)

    def _dropout_source(self, x: torch.Tensor):
        """(bytes, key) for the training-mode dropouts of the fused path: the conv stack's forward
        kernel fills ``bytes`` from ``key`` on the side and the head kernels read them.  While a
        training step is being captured (GraphedTrainStep) both are static device tensors — the
        step writes a fresh key before every replay; in eager training ``bytes`` is a scratch
        buffer and ``key`` an int drawn from torch's device generator (``next_dropout_key``).
        (None, None) in eval mode or without autograd: the head then needs none, or draws bytes
        itself (``random_()``)."""
        if not self.training or not torch.is_grad_enabled() or not PotesStackFunction.use_masks:
            return None, None
        if torch.cuda.is_current_stream_capturing():
            if self.dropout_bytes is not None and self.dropout_key is not None:
                return self.dropout_bytes, self.dropout_key
            return None, None
        drop = self.cnn1[1][3] if len(self.cnn1[1]) > 3 else None
        n = head_dropout_bytes(x.shape[0], self.dimreduc.in_features, float(drop.p) if drop else 0.0)
        return (torch.empty((n + 15) // 16 * 16, dtype=torch.uint8, device=x.device),
                next_dropout_key(x.device))

    def _logits_fused(self, x: torch.Tensor) -> torch.Tensor:
        """Whole network on the HIP path: conv stack kernel + head kernels."""
        B, C, T = x.shape
        c1, c2 = self.cnn1[0][0], self.cnn1[1][0]
        rows = x[:, :4, :].reshape(B * 4, T)
        rnd, key = self._dropout_source(x)
        z = PotesStackFunction.apply(rows.contiguous(), c1.weight, c1.bias, c2.weight, c2.bias, rnd, key)
        drop = self.cnn1[1][3] if len(self.cnn1[1]) > 3 else None
        return PotesHeadFunction.apply(z.reshape(B, -1), self.dimreduc.weight, self.dimreduc.bias,
                                       self.linear.weight, self.linear.bias,
                                       float(drop.p) if drop is not None else 0.0,
                                       float(self.dropout.p), self.training, rnd)

    def loss_and_logits(self, x: torch.Tensor, target: torch.Tensor):
        """(soft-target cross entropy, logits) of the whole network in one chain of HIP kernels
        — the conv stack, then head and loss as ONE autograd node (``PotesHeadLossFunction``).
        Needs ``_fused_head(x)``; ``target`` is the (B, classes) one-hot / soft target matrix, or
        — hard targets — a uint8 (B,) tensor of class labels."""
        B, C, T = x.shape
        c1, c2 = self.cnn1[0][0], self.cnn1[1][0]
        rows = x[:, :4, :].reshape(B * 4, T)
        rnd, key = self._dropout_source(x)
        z = PotesStackFunction.apply(rows.contiguous(), c1.weight, c1.bias, c2.weight, c2.bias, rnd, key)
        drop = self.cnn1[1][3] if len(self.cnn1[1]) > 3 else None
        # While a training step is being captured, forward and backward always replay together and
        # nobody can look at the loss in between: one launch less (PotesHeadLossFunction, `defer`).
        defer = torch.is_grad_enabled() and torch.cuda.is_current_stream_capturing()
        return PotesHeadLossFunction.apply(z.reshape(B, -1), self.dimreduc.weight, self.dimreduc.bias,
                                           self.linear.weight, self.linear.bias, target,
                                           float(drop.p) if drop is not None else 0.0,
                                           float(self.dropout.p), self.training, rnd, defer)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        B, C, T = x.shape
        # the four bands share cnn1's weights: run them as one (4B,1,T) batch, then restore the
        # reference's concatenation order [band0 | band1 | band2 | band3] per sample
        rows = x[:, :4, :].reshape(B * 4, T)
        if self._fused(x):
            c1, c2 = self.cnn1[0][0], self.cnn1[1][0]
            z = PotesStackFunction.apply(rows.contiguous(), c1.weight, c1.bias, c2.weight, c2.bias)
            drop = self.cnn1[1][3] if len(self.cnn1[1]) > 3 else None
            if drop is not None:
                z = F.dropout(z, drop.p, self.training)
        else:
            z = self.cnn1(rows.unsqueeze(1))
        z = z.reshape(B, -1)
        if self._fused(x) and self.dimreduc.out_features == 20 and z.shape[1] % 4 == 0:
            z = SkinnyLinearFunction.apply(z, self.dimreduc.weight, self.dimreduc.bias)
        else:
            z = self.dimreduc(z)
        return self.dropout(F.relu(z))

    def forward(self, x, depth=None, pass_part=None):
        if pass_part == "first":
            if depth == 0:
                return x
            return self.features(x)
        if pass_part == "latent_space":
            return self.features(x)
        if pass_part == "second":
            if depth <= 0:
                if self._fused_head(x):
                    return self._logits_fused(x)
                x = self.features(x)
            if depth <= 1:
                x = self.linear(x)
            return x
        if self._fused_head(x):
            return self._logits_fused(x)
        return self.linear(self.features(x))


def CNN_potes_TS(num_channels: int = 4, num_classes: int = 2, dataset: str = "PhysioNet",
                 dropout: float = 0.25, sig_len: int | None = None) -> CNN_potes:
    """Reference factory (models.py:345-350): linear = 9968 for PhysioNet (T=2500), 7968 for UMC.
    ``sig_len`` (an extension) sizes the head for other lengths, e.g. 5000 -> 19968."""
    if sig_len is not None:
        linear = potes_flat_features(sig_len)
    elif dataset == "PhysioNet":
        linear = 9968
    elif dataset == "UMC":
        linear = 7968
    else:
        raise ValueError(dataset)
    return CNN_potes(c_in=num_channels, c_out=num_classes, layers=[8, 4], linear=linear,
                     dropout=dropout)


def _res_block(c_in: int, c_out: int, pool: bool = False) -> nn.Sequential:
    # Conv1d(k=3,p=1) + BatchNorm1d + ReLU [+ MaxPool(2)]     (reference models.py:468-473)
    layers = [nn.Conv1d(c_in, c_out, kernel_size=3, padding=1), nn.BatchNorm1d(c_out),
              nn.ReLU(inplace=True)]
    if pool:
        layers.append(nn.MaxPool1d(2))
    return nn.Sequential(*layers)


FUSED_BN = True      # BatchNorm + ReLU + MaxPool through the HIP kernels (False: torch ops)


class BNReLUPoolFunction(torch.autograd.Function):
    """Training-mode BatchNorm + ReLU + MaxPool on a channels_last activation through
    ``pcgmix_bnrp_{fwd,bwd}_f32``: five passes over the activation per block (forward + backward)
    instead of ten.  ``y`` is (B, C, H, W) channels_last; returns (B, C, H/ph, W/pw) channels_last.
    The running statistics are updated in place, as ``F.batch_norm(training=True)`` does."""

    @staticmethod
    def supported(y: torch.Tensor) -> bool:
        C = y.shape[1] if y.dim() == 4 else 0
        return (y.is_cuda and y.dtype == torch.float32 and y.dim() == 4 and C % 4 == 0
                and C // 4 <= 256 and 256 % (C // 4) == 0 and y.numel() > 0
                and y.is_contiguous(memory_format=torch.channels_last))

    @staticmethod
    def forward(ctx, y, gamma, beta, running_mean, running_var, momentum, eps, ph, pw, skip=None,
                conv_bias=None, batches_tracked=None):
        # conv_bias: the bias of the convolution that produced y, NOT added to y (the normalisation
        # cancels it): it only shifts the running mean, and gets an exact zero gradient — both inside
        # the BatchNorm kernels.  batches_tracked: the module's counter, incremented there too.
        B, C, H, W = y.shape
        if skip is not None and (skip.shape != (B, C, H // ph, W // pw) or skip.dtype != torch.float32
                                 or not skip.is_contiguous(memory_format=torch.channels_last)):
            raise ValueError("skip must be a float32 channels_last tensor shaped like the output")
        lib = _lib.load()
        dev = y.device
        z = torch.empty((B, C, H // ph, W // pw), dtype=torch.float32, device=dev,
                        memory_format=torch.channels_last)
        mean = torch.empty(C, dtype=torch.float32, device=dev)
        invstd = torch.empty(C, dtype=torch.float32, device=dev)
        ws = torch.empty(lib.pcgmix_bnrp_workspace_floats(B, H, W, C), dtype=torch.float32, device=dev)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pcgmix_bnrp_fwd_f32(
            y.data_ptr(), g.data_ptr(), b.data_ptr(),
            running_mean.data_ptr() if running_mean is not None else None,
            running_var.data_ptr() if running_var is not None else None,
            ctypes.c_float(momentum), ctypes.c_float(eps),
            conv_bias.data_ptr() if conv_bias is not None else None,
            batches_tracked.data_ptr() if batches_tracked is not None else None,
            skip.data_ptr() if skip is not None else None, z.data_ptr(), mean.data_ptr(),
            invstd.data_ptr(), ws.data_ptr(), B, H, W, C, ph, pw, stream), "pcgmix_bnrp_fwd_f32")
        ctx.save_for_backward(y, g, b, mean, invstd)
        ctx.pool = (ph, pw)
        ctx.has_skip = skip is not None
        ctx.has_bias = conv_bias is not None
        return z

    @staticmethod
    def backward(ctx, dz):
        y, g, b, mean, invstd = ctx.saved_tensors
        B, C, H, W = y.shape
        ph, pw = ctx.pool
        lib = _lib.load()
        dev = y.device
        dz = dz.contiguous(memory_format=torch.channels_last)
        dx = torch.empty_like(y)                             # preserves channels_last
        dgamma = torch.empty(C, dtype=torch.float32, device=dev)
        dbeta = torch.empty(C, dtype=torch.float32, device=dev)
        dbias = torch.empty(C, dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = torch.empty(lib.pcgmix_bnrp_workspace_floats(B, H, W, C), dtype=torch.float32, device=dev)
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.pcgmix_bnrp_bwd_f32(
            y.data_ptr(), dz.data_ptr(), g.data_ptr(), b.data_ptr(), mean.data_ptr(),
            invstd.data_ptr(), dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
            dbias.data_ptr() if dbias is not None else None, ws.data_ptr(),
            B, H, W, C, ph, pw, stream), "pcgmix_bnrp_bwd_f32")
        # the residual input enters by a plain addition: its gradient is dz itself
        return (dx, dgamma, dbeta, None, None, None, None, None, None,
                (dz if ctx.has_skip else None), dbias, None)


def conv_bn_relu_pool(h, weight4, conv_bias, padding, bn, training: bool, pool, skip=None):
    """Conv -> BatchNorm -> ReLU [-> MaxPool] on a 4-D (channels_last) activation with the
    convolution's bias folded into the BatchNorm instead of added by a separate pass.

    BN(conv(x) + b) == BN(conv(x)): a per-channel constant cancels in (y - mean(y)), so in training
    mode the bias only shows up in the running mean (+ momentum * b per step, added below) and its
    true gradient is zero — through torch it comes out as ~1e-14 rounding noise, which Adam's eps
    (1e-8) swamps; ``beta + 0 * b`` gives it an exact zero gradient so that the optimiser still
    applies weight decay to it, as it does in the reference.  In eval mode the running mean is
    shifted by b instead.  Saves a bias-add pass over the activation and a (B, L) reduction for the
    bias gradient per layer (2.7 ms of a 42 ms ResNet9-1D step)."""
    h = F.conv2d(h, weight4, None, padding=padding)
    batch_stats = training or not bn.track_running_stats
    if bn.momentum is None or bn.momentum >= 1.0 or conv_bias is None:    # no fold
        if conv_bias is not None:
            h = h + conv_bias.view(1, -1, 1, 1)
        if training and bn.track_running_stats:
            bn.num_batches_tracked.add_(1)
        factor = 0.0 if bn.momentum is None else bn.momentum
        if bn.momentum is None and training and bn.track_running_stats:
            factor = 1.0 / float(bn.num_batches_tracked)
        h = F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, bn.bias, batch_stats,
                         factor, bn.eps)
    elif batch_stats:
        if FUSED_BN and training and bn.track_running_stats and BNReLUPoolFunction.supported(h) \
                and conv_bias.is_contiguous() and conv_bias.dtype == torch.float32:
            # counter, running-mean shift (new = (1-m) * old + m * (mean(conv) + b)) and the bias'
            # exact zero gradient all happen inside the BatchNorm kernels: no launch of their own
            ph, pw = (1, 1) if pool is None else ((pool, pool) if isinstance(pool, int) else pool)
            if skip is not None and not skip.is_contiguous(memory_format=torch.channels_last):
                skip = skip.contiguous(memory_format=torch.channels_last)
            return BNReLUPoolFunction.apply(h, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                            float(bn.momentum), float(bn.eps), int(ph), int(pw), skip,
                                            conv_bias, bn.num_batches_tracked)
        if training and bn.track_running_stats:
            bn.num_batches_tracked.add_(1)                  # as nn.BatchNorm.forward does
        beta = bn.bias + 0.0 * conv_bias
        if bn.track_running_stats and bn.momentum < 1.0:
            # new = (1-m) * (old + m/(1-m) * b) + m * mean(conv) = (1-m) * old + m * (mean(conv) + b);
            # shifted BEFORE the call: autograd saves the buffer and rejects a later in-place edit
            with torch.no_grad():
                bn.running_mean.add_(conv_bias, alpha=bn.momentum / (1.0 - bn.momentum))
        if FUSED_BN and training and h.is_cuda:
            _warn_once(f"conv_bn_relu_pool: activation {tuple(h.shape)} {h.dtype} cannot use the HIP "
                       "BatchNorm+ReLU+pool kernels (needs float32 channels_last, C % 4 == 0, "
                       "256 % (C/4) == 0): running torch ops instead")
        h = F.batch_norm(h, bn.running_mean, bn.running_var, bn.weight, beta, True, bn.momentum,
                         bn.eps)
    else:
        h = F.batch_norm(h, bn.running_mean - conv_bias.detach(), bn.running_var, bn.weight,
                         bn.bias, False, 0.0, bn.eps)
    h = F.relu(h, inplace=True)
    if pool is not None:
        h = F.max_pool2d(h, pool)
    return h if skip is None else h + skip


class ResNet9_myrtle(nn.Module):
    """reference models.py:520-589 (the second definition, which shadows the first)."""

    def __init__(self, in_channels: int, num_classes: int, filters, linear: int):
        super().__init__()
        self.conv1 = _res_block(in_channels, filters[0])
        self.conv2 = _res_block(filters[0], filters[1], pool=True)
        self.res1 = nn.Sequential(_res_block(filters[1], filters[1]), _res_block(filters[1], filters[1]))
        self.conv3 = _res_block(filters[1], filters[2], pool=True)
        self.conv4 = _res_block(filters[2], filters[3], pool=True)
        self.res2 = nn.Sequential(_res_block(filters[3], filters[3]), _res_block(filters[3], filters[3]))
        self.pool1d = nn.MaxPool1d(4)
        self.flat = nn.Flatten()
        self.linear = nn.Linear(linear, num_classes)
        # Conv1d weights are kept (O, I, 3) in shape and state_dict but laid out with the input
        # channel innermost, so that their (O, I, 1, 3) view is channels_last as it stands: MIOpen's
        # NHWC kernels otherwise get a re-laid-out copy of every weight in forward, backward-data
        # and backward-weights (26 copies, 0.3 ms of a 32 ms bs=256 step,
        # profiles/r2_resnet1d_step_kernels.csv).  A memory format, not a reshape: values, keys and
        # shapes are unchanged; load_state_dict / .to() / deepcopy keep the strides.
        for m in self.modules():
            if isinstance(m, nn.Conv1d):
                m.weight.data = m.weight.data.permute(0, 2, 1).contiguous().permute(0, 2, 1)

    # ---- execution layout ------------------------------------------------------------------
    # On a HIP device the activations flow as (B, C, 1, L) channels_last tensors through
    # conv2d / batch_norm / max_pool2d with the SAME parameters and buffers (Conv1d weights viewed
    # as (O, I, 1, 3)).  MIOpen's fp32 implicit-GEMM convolutions are NHWC kernels: fed (B, C, L)
    # tensors they are wrapped in batched_transpose launches (8.4 ms of a 55.4 ms bs=256 step at
    # T=5000, profiles/probes/resnet1d_probe.py), and BatchNorm / pooling kernels also run faster
    # with channels innermost: 55.9 -> 42.1 ms per forward+backward
    # (profiles/probes/resnet1d_layout_probe.py), logits equal to 1e-5.
    nhwc = True

    def _block(self, seq, h, skip=None):
        """One conv block; ``skip`` = the residual input added to the block's output (the add is
        fused into the BatchNorm/ReLU kernel on the HIP path)."""
        if h.dim() == 3:
            return seq(h) if skip is None else seq(h) + skip
        pool = (1, seq[3].kernel_size) if len(seq) > 3 else None
        return conv_bn_relu_pool(h, seq[0].weight.unsqueeze(2), seq[0].bias, (0, seq[0].padding[0]),
                                 seq[1], self.training, pool, skip)

    def _stage1(self, out):
        out = self._block(self.conv2, self._block(self.conv1, out))
        return self._block(self.res1[1], self._block(self.res1[0], out), skip=out)   # res1(out) + out

    def _stage2(self, out):
        out = self._block(self.conv4, self._block(self.conv3, out))
        return self._block(self.res2[1], self._block(self.res2[0], out), skip=out)   # res2(out) + out

    def _pool_flat(self, out):
        if out.dim() == 3:
            return self.flat(self.pool1d(out))
        # logical (B, C, L') order, as nn.Flatten of the (B, C, L') tensor gives
        return F.max_pool2d(out, (1, self.pool1d.kernel_size)).squeeze(2).flatten(1)

    def _pool_linear(self, out):
        """MaxPool(4) -> Flatten -> Linear.  On the channels_last path the pooled (B, C, 1, L')
        tensor is NOT re-laid-out to (B, C, L') for the flatten (a 20 MB copy forward and one
        backward at bs 256): its memory order [l][c] is flattened as it is and the Linear's weight
        columns are permuted to match (a 0.6 MB gather)."""
        if out.dim() == 3:
            return self.linear(self.flat(self.pool1d(out)))
        p = F.max_pool2d(out, (1, self.pool1d.kernel_size))
        B, C, _, L = p.shape
        if not p.is_contiguous(memory_format=torch.channels_last) or C * L != self.linear.in_features:
            return self.linear(p.squeeze(2).flatten(1))
        feat = p.permute(0, 2, 3, 1).reshape(B, L * C)                    # a view
        w = self.linear.weight.view(-1, C, L).permute(0, 2, 1).reshape(-1, L * C)
        return F.linear(feat, w, self.linear.bias)

    def forward(self, out, depth=None, pass_part=None):
        if pass_part == "first" and depth == 0:
            return out
        wide = self.nhwc and out.is_cuda and out.dtype == torch.float32 and out.dim() == 3
        if wide:
            out = out.unsqueeze(2).contiguous(memory_format=torch.channels_last)
        act = (lambda t: t.squeeze(2)) if wide else (lambda t: t)    # activations leave as (B,C,L)
        if pass_part == "first":
            out = self._stage1(out)
            if depth == 1:
                return act(out)
            out = self._stage2(out)
            if depth == 2:
                return act(out)
            out = self._pool_flat(out)
            if depth == 3:
                return out
            return self.linear(out)
        if pass_part == "second":
            if depth <= 0:
                out = self._stage1(out)
            if depth <= 1:
                out = self._stage2(out)
            if depth <= 2:
                return self._pool_linear(out)
            if depth <= 3:
                out = self.linear(out)
            return out
        return self._pool_linear(self._stage2(self._stage1(out)))


def resnet9_flat_features(sig_len: int, width: int = 512) -> int:
    """512 channels x (T / 2 / 2 / 2 / 4) positions: 39936 at T=2500, 79872 at T=5000."""
    return width * (sig_len // 2 // 2 // 2 // 4)


def ResNet9(in_channels: int, num_classes: int, filters=(64, 128, 256, 512),
            linear: int = 39936) -> ResNet9_myrtle:
    """Reference factory, models.py:588."""
    return ResNet9_myrtle(in_channels=in_channels, num_classes=num_classes, filters=list(filters),
                          linear=linear)
