"""Per-step training path of the reference's ``train_model.py`` (train_epoch, :490-589) on
MI355X: augment -> forward -> soft-target CE -> backward -> clip_grad_value_ -> Adam ->
OneCycleLR, one process per GPU with DistributedDataParallel over RCCL (the reference's
single-process ``nn.DataParallel``, train_model.py:385, is replaced).

What is kept: the loss (CELoss :45-54, SELCLoss :56-80), optimiser and scheduler setup
(:404-410), gradient value clipping (:557-558), the step counter that seeds the augmentation
(:105-109, :581), the per-epoch reseeding of the loader shuffle (:497), and the
``train_epoch`` signature.  What is deliberately not reproduced: the per-sample ``.item()``
loop (:542-552) — loss and hit counts stay on the device and are read once per epoch — and the
model zoo / plotting / pickling around the loop (out of scope).
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import augmentations, augmentations2d, models, models2d

SPECTROGRAM_DATASETS = ("PhysioNet(spec128)", "UMC(spec128)", "UMC(spec64)")


class SoftCEFunction(torch.autograd.Function):
    """CELoss on a HIP device as one kernel each way (``pcgmix_soft_ce_{fwd,bwd}_f32``) instead of
    log_softmax, mul, sum, neg, mean and their five backward launches."""

    @staticmethod
    def forward(ctx, logits, target):
        import ctypes
        from . import _lib
        lib = _lib.load()
        logits = logits.contiguous()
        target = target.to(torch.float32).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        _lib.check(lib.pcgmix_soft_ce_fwd_f32(logits.data_ptr(), target.data_ptr(), loss.data_ptr(),
                                              logits.shape[0], logits.shape[1], stream),
                   "pcgmix_soft_ce_fwd_f32")
        ctx.save_for_backward(logits, target)
        return loss

    @staticmethod
    def backward(ctx, gout):
        import ctypes
        from . import _lib
        lib = _lib.load()
        logits, target = ctx.saved_tensors
        gout = gout.to(torch.float32).contiguous()
        d = torch.empty_like(logits)
        stream = ctypes.c_void_p(torch.cuda.current_stream(logits.device).cuda_stream)
        _lib.check(lib.pcgmix_soft_ce_bwd_f32(logits.data_ptr(), target.data_ptr(), gout.data_ptr(),
                                              d.data_ptr(), logits.shape[0], logits.shape[1],
                                              stream), "pcgmix_soft_ce_bwd_f32")
        return d, None


class CELoss(nn.Module):
    """Cross-entropy with soft targets: mean_b( -sum_c log_softmax(logits)[b,c] * t[b,c] )
    (train_model.py:45-54)."""

    def __init__(self, num_classes: int):
        super().__init__()
        self.num_classes = num_classes

    def forward(self, logits, target_ohe):
        if logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2 \
                and logits.shape[0] > 0 and not target_ohe.requires_grad:
            return SoftCEFunction.apply(logits, target_ohe)
        return -(F.log_softmax(logits, dim=1) * target_ohe).sum(dim=1).mean()


class SELCLoss(nn.Module):
    """train_model.py:56-80.  Plain CE until epoch ``es``; afterwards the self-ensembled soft
    labels.  With the reference's default (``es = num_epochs + 1`` unless the method contains
    'SELC', :394-402) it is CELoss.  Soft labels live on ``device`` (the reference hard-codes
    ``.cuda()``, :60)."""

    def __init__(self, labels, num_classes: int, es: int = 10, momentum: float = 0.9,
                 device: Optional[torch.device] = None):
        super().__init__()
        labels = torch.as_tensor(np.asarray(labels), dtype=torch.long)
        soft = torch.zeros(len(labels), num_classes, dtype=torch.float)
        soft[torch.arange(len(labels)), labels] = 1
        self.register_buffer("soft_labels", soft.to(device) if device is not None else soft)
        self.num_classes, self.es, self.momentum = num_classes, es, momentum
        self.CEloss = CELoss(num_classes)

    def forward(self, logits, labels, index, epoch, mode):
        if mode == "test" or epoch <= self.es:
            return self.CEloss(logits, labels)
        pred = F.softmax(logits, dim=1)
        index = torch.as_tensor(index, device=self.soft_labels.device)
        with torch.no_grad():
            self.soft_labels[index] = (self.momentum * self.soft_labels[index]
                                       + (1 - self.momentum) * pred.detach())
        return -(torch.log(pred) * self.soft_labels[index]).sum(dim=1).mean()


class step_counter_class:
    """train_model.py:105-109: the only source of augmentation randomness."""

    def __init__(self):
        self.count = 0

    def add(self):
        self.count += 1


class ClipAdam(torch.optim.Adam):
    """``clip_grad_value_(clip)`` + ``torch.optim.Adam`` (L2 weight decay) as one HIP pass per
    parameter (``pcgmix_adam_clip_f32``).  Same state layout (``step``, ``exp_avg``,
    ``exp_avg_sq``) and param-group keys as torch's Adam, so OneCycleLR cycles ``lr`` and
    ``betas`` on it unchanged and checkpoints interchange.  Gradients are NOT modified (torch's
    clip is in place; nothing downstream of the optimiser step reads them)."""

    def __init__(self, params, lr, weight_decay=0.0, clip_value=0.0, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.clip_value = float(clip_value or 0.0)
        self._tables = {}
        self._cap_params, self._cap_grads, self._cap_step, self._cap_dirty = None, None, 0, False

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}                      # the moment tensors were replaced
        if getattr(self, "_cap_params", None):
            raise RuntimeError("ClipAdam: load the state before the step is captured (the graph "
                               "holds the addresses of the old moment tensors)")

    def _init_state(self, params):
        for p in params:
            st = self.state[p]
            if not st:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _table(self, gi, live, grads):
        """Host pointer tables of the group's static tensors (rebuilt when a tensor moved)."""
        import ctypes
        key = tuple(p.data_ptr() for p in live)
        tab = self._tables.get(gi)
        if tab is None or tab[0] != key:
            n = len(live)
            arr = ctypes.c_void_p * n
            tab = (key, arr(*key), arr(*[self.state[p]["exp_avg"].data_ptr() for p in live]),
                   arr(*[self.state[p]["exp_avg_sq"].data_ptr() for p in live]),
                   (ctypes.c_longlong * n)(*[p.numel() for p in live]), arr())
            self._tables[gi] = tab
        gptr = tab[5]
        for i, g in enumerate(grads):
            gptr[i] = g.data_ptr()
        return tab

    @staticmethod
    def _dense_grads(live):
        # flat iteration over raw memory: p, grad, m, v must share one dense layout
        return [p.grad if p.grad.stride() == p.stride() else torch.empty_like(p).copy_(p.grad)
                for p in live]

    @torch.no_grad()
    def step(self, closure=None):
        import ctypes
        from . import _lib
        lib = _lib.load()
        self._flush_captured_steps()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            live = [p for p in group["params"] if p.grad is not None]
            if not live:
                continue
            self._init_state(live)
            for p in live:
                self.state[p]["step"] += 1
            steps = {int(self.state[p]["step"]) for p in live}
            if len(steps) != 1:
                raise RuntimeError("ClipAdam: parameters of one group must share the step count")
            if self._cap_params:
                # an eager step between captured ones (train_epoch: a batch of another shape) counts
                # for the captured launches too: their bias corrections continue from here
                self._cap_step = next(iter(steps))
            tab = self._table(gi, live, self._dense_grads(live))
            dev = live[0].device
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.pcgmix_adam_clip_multi_f32(
                len(live), tab[1], tab[5], tab[2], tab[3], tab[4], ctypes.c_float(self.clip_value),
                ctypes.c_float(float(group["lr"])), ctypes.c_float(b1), ctypes.c_float(b2),
                ctypes.c_float(group["eps"]), ctypes.c_float(group["weight_decay"]),
                steps.pop(), stream), "pcgmix_adam_clip_multi_f32")
        return None

    # ---- the update as a node of a captured training step (GraphedTrainStep) -----------------
    # Kernel arguments are frozen at capture, OneCycleLR moves lr and beta1 every step and the
    # bias corrections move with the step count: the captured launch reads its eight scalars from
    # device memory (``pcgmix_adam_clip_multi_dev_f32``); ``next_hyper`` computes them on the host
    # for the coming replay and advances the step count.
    def can_capture(self) -> bool:
        return len(self.param_groups) == 1

    def prepare_capture(self, params):
        """Before the capture: allocate the moments of ``params`` (the tensors that will carry a
        gradient) — an allocation inside the capture would be re-zeroed by every replay."""
        self._flush_captured_steps()                 # re-capture: the state carries the live count
        self._init_state(params)
        steps = {int(self.state[p]["step"]) for p in params}
        if len(steps) != 1:
            raise RuntimeError("ClipAdam: parameters of one group must share the step count")
        self._cap_params = list(params)
        self._cap_step = steps.pop()

    @torch.no_grad()
    def capture_update(self, hyper_dev: torch.Tensor, deferred: Optional[dict] = None):
        """Inside the capture, after backward: the update of all ``prepare_capture`` tensors.

        ``deferred`` (``models.PotesStackFunction.defer_reduce`` after the backward): the Potes conv
        stack left its 212 gradient columns as G un-reduced partial rows.  When the stack's four
        parameter gradients are views of ``deferred["grads"]`` (autograd keeps the tensors the
        backward returned) the reduction runs as 212 extra blocks of this launch, each of which also
        updates its element; otherwise it is launched on its own first."""
        import ctypes
        from . import _lib
        lib = _lib.load()
        live = self._cap_params
        if any(p.grad is None for p in live):
            raise RuntimeError("ClipAdam.capture_update: a prepared parameter has no gradient")
        self._cap_grads = self._dense_grads(live)            # keep graph memory referenced
        tab = self._table(0, live, self._cap_grads)
        stream = ctypes.c_void_p(torch.cuda.current_stream(live[0].device).cuda_stream)
        if deferred and "partial" in deferred:
            grads, partial, G = deferred["grads"], deferred["partial"], int(deferred["G"])
            lo, hi = grads.data_ptr(), grads.data_ptr() + grads.numel() * 4
            covered = sum(g.numel() for g in self._cap_grads if lo <= g.data_ptr() < hi)
            self._cap_deferred = (partial, grads)            # referenced by the graph
            if covered == grads.numel() and len(live) <= 32:
                _lib.check(lib.pcgmix_adam_clip_multi_reduce_dev_f32(
                    len(live), tab[1], tab[5], tab[2], tab[3], tab[4], hyper_dev.data_ptr(),
                    partial.data_ptr(), grads.data_ptr(), G, stream),
                    "pcgmix_adam_clip_multi_reduce_dev_f32")
                return
            _lib.check(lib.pcgmix_potes_reduce_f32(partial.data_ptr(), grads.data_ptr(), G, stream),
                       "pcgmix_potes_reduce_f32")
        _lib.check(lib.pcgmix_adam_clip_multi_dev_f32(
            len(live), tab[1], tab[5], tab[2], tab[3], tab[4], hyper_dev.data_ptr(), stream),
            "pcgmix_adam_clip_multi_dev_f32")

    def next_hyper(self, out8) -> None:
        """Host: the eight scalars of the NEXT update into ``out8`` (float32[8], numpy), from the
        param group as the scheduler left it; counts the step.  ``state[p]['step']`` is brought
        up to date lazily (``state_dict`` / an eager ``step``)."""
        import ctypes
        from . import _lib
        g = self.param_groups[0]
        self._cap_step += 1
        self._cap_dirty = True
        self._opt_called = True                  # what LRScheduler's wrapper of step() records
        _lib.check(_lib.load().pcgmix_adam_hyper(
            ctypes.c_float(self.clip_value), ctypes.c_float(float(g["lr"])),
            ctypes.c_float(g["betas"][0]), ctypes.c_float(g["betas"][1]), ctypes.c_float(g["eps"]),
            ctypes.c_float(g["weight_decay"]), self._cap_step, out8.ctypes.data), "pcgmix_adam_hyper")

    def _flush_captured_steps(self):
        if getattr(self, "_cap_dirty", False):
            for p in self._cap_params:
                self.state[p]["step"].fill_(float(self._cap_step))
            self._cap_dirty = False

    def state_dict(self):
        self._flush_captured_steps()
        return super().state_dict()


def selc_turning_point(args) -> int:
    """train_model.py:394-402."""
    if "SELC" in args.method and ("mixup" in args.method or "base" in args.method):
        return int(args.num_epochs * 0.4)
    return args.num_epochs + 1


def build_model(args) -> nn.Module:
    """The three models of the hot path (train_model.py:296, 338, 360), head sized for
    ``args.sig_len`` (the reference hard-codes T = 2500)."""
    sig_len = getattr(args, "sig_len", 2500)
    if args.dataset in SPECTROGRAM_DATASETS:
        if args.model != "resnet9":
            raise NotImplementedError(args.model)
        return models2d.ResNet9(num_classes=args.num_classes)
    if args.model == "Potes":
        m = models.CNN_potes_TS(num_channels=args.num_channels, num_classes=args.num_classes,
                                dataset=args.dataset, sig_len=None if sig_len == 2500 else sig_len)
        # cnn2..cnn4 are never called (reference models.py:444-455): their grads stay None in
        # the reference, so Adam and the clipper skip them; freezing them is equivalent and
        # keeps them out of the DDP reducer
        for name in ("cnn2", "cnn3", "cnn4"):
            for p in getattr(m, name).parameters():
                p.requires_grad_(False)
        return m
    if args.model == "resnet9":
        return models.ResNet9(in_channels=args.num_channels, num_classes=args.num_classes,
                              linear=models.resnet9_flat_features(sig_len))
    raise NotImplementedError(f"model {args.model!r} is outside the PCGmix hot path")


def make_optimizer(args, model: nn.Module):
    """train_model.py:404-410."""
    params = [p for p in model.parameters() if p.requires_grad]
    if args.op == "SGD":
        opt = torch.optim.SGD(params, lr=args.lr_max, weight_decay=args.weight_decay)
    elif args.op == "adam":
        def dense(p):       # any non-overlapping dense layout: the kernel walks raw memory
            order = sorted(range(p.dim()), key=lambda d: (-p.stride(d), -p.size(d)))
            return p.numel() == 0 or p.permute(order).is_contiguous()
        if all(p.is_cuda and p.dtype == torch.float32 and dense(p) for p in params):
            # clip + Adam in one HIP pass per tensor (train_step then skips clip_grad_value_)
            opt = ClipAdam(params, lr=args.lr_max, weight_decay=args.weight_decay,
                           clip_value=args.grad_clip)
        else:
            opt = torch.optim.Adam(params, lr=args.lr_max, weight_decay=args.weight_decay)
    else:
        raise ValueError(args.op)
    sched = None
    if args.use_sched:
        sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=args.lr_max,
                                                    total_steps=args.num_steps)
    return opt, sched


def wrap_distributed(model: nn.Module, device: torch.device) -> nn.Module:
    """One process per GPU; gradients are averaged by an all-reduce (RCCL over xGMI when the
    process group is 'nccl').  The whole model fits one bucket (<= 26 MB), so the single
    all-reduce overlaps the tail of backward.  BatchNorm statistics stay per rank, as they are
    per replica under the reference's DataParallel."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    ids = [device.index] if device.type == "cuda" else None
    return nn.parallel.DistributedDataParallel(model, device_ids=ids, bucket_cap_mb=64,
                                               gradient_as_bucket_view=True)


class FlatGradSync:
    """Gradient averaging over ranks without DDP's autograd hooks, so that forward + backward can
    live in a hipGraph: the gradients are packed into one flat buffer by ONE concat kernel (inside
    the captured region), summed by ONE all-reduce (RCCL over xGMI under 'nccl'; 0.8 MB for Potes,
    9.1 MB for ResNet9-1D — a single message per step, the size the per-link-bound ring wants),
    and the optimiser then reads views of that buffer.  The caller scales backward by 1/world
    (``backward_scale``) so the sum IS the mean on every backend.  BatchNorm buffers stay per
    rank, as under the reference's DataParallel (train_model.py:385)."""

    def __init__(self, model: nn.Module, device: torch.device):
        import torch.distributed as dist
        self.dist = dist
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.model, self.device = model, device
        self.params, self.flat, self.views, self._graph_grads = None, None, None, None
        if self.world > 1:                              # same start on every rank, as DDP does
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, 0)

    @property
    def backward_scale(self) -> float:
        return 1.0 / self.world

    def attach(self):
        """Call once after a first backward: fixes the set of parameters that receive gradients
        (CNN_potes allocates cnn2..cnn4 but never uses them) and allocates the flat buffer."""
        self.params = [p for p in self.model.parameters() if p.grad is not None]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, device=self.device, dtype=self.params[0].dtype)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def pack(self):
        """Concatenate the fresh gradients into the flat buffer (capturable: one kernel)."""
        self._graph_grads = [p.grad for p in self.params]            # keep graph memory referenced
        torch.cat([g.reshape(-1) for g in self._graph_grads], out=self.flat)

    def reduce_and_bind(self):
        """Sum over ranks and hand the optimiser the averaged gradients."""
        if self.world > 1:
            self.dist.all_reduce(self.flat)
        for p, v in zip(self.params, self.views):
            p.grad = v


def shard_batch(batch, rank: int, world: int):
    """Give rank r the r-th contiguous slice of every per-sample field (drop_last semantics of
    dataloader_physionet.py:227: the remainder is dropped)."""
    n = len(batch[0]) // world
    sl = slice(rank * n, (rank + 1) * n)
    return tuple(b[sl] for b in batch)


class _SeedView:
    __slots__ = ("count",)

    def __init__(self, count: int):
        self.count = count


def augmentation_counter(args, step_counter):
    """What ``augment()`` is given as its step counter.  The reference derives every random choice
    of a step from ``step_counter.count`` (train_model.py:507, augmentations.py:869-903), so under
    data parallelism every rank draws the same lambda, the same warp knots and the same
    permutation pattern.  ``args.rank_seed = True`` — NOT the reference's behaviour,
    off by default — gives rank r of w the seed ``count * w + r`` instead: distinct
    streams per rank that never collide across steps.  One process: unchanged."""
    if not getattr(args, "rank_seed", False):
        return step_counter
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return step_counter
    return _SeedView(int(step_counter.count) * dist.get_world_size() + dist.get_rank())


def reseed_device_rng(args, device) -> None:
    """train_model.py:565: the reference calls ``torch.cuda.manual_seed_all(args.seed_fix)`` before
    every ``optimizer.step()``.  Its visible effect on a GPU is on the NEXT forward pass: the
    device generator restarts from the same position every step, so every step (and, under
    DataParallel, every replica) draws the same dropout masks.  Kept, for this process's device
    (one process per GPU); host-only, no launch.  ``args.seed_fix`` is set by ``train_model``
    (:217); loops that do not set it keep torch's running stream."""
    seed = getattr(args, "seed_fix", None)
    if seed is not None and device.type == "cuda":
        if getattr(args, "rank_seed", False):           # non-reference: per-rank dropout masks
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                seed = int(seed) + dist.get_rank()
        torch.cuda.manual_seed(int(seed))


def fused_loss_model(model, criterion, data, target_ohe, epoch):
    """The CNN_potes whose head and loss can run as one autograd node for this step, or None:
    an unwrapped CNN_potes on the HIP path, a criterion in its plain-CE phase (CELoss, or SELCLoss
    up to its turning point, train_model.py:66-72), targets that need no gradient."""
    if not isinstance(model, models.CNN_potes) or target_ohe.requires_grad:
        return None
    if isinstance(criterion, SELCLoss):
        if epoch is None or epoch > criterion.es:
            return None
    elif not isinstance(criterion, CELoss):
        return None
    if not (data.dim() == 3 and model._fused_head(data) and data.shape[0] > 0):
        return None
    return model


def train_step(args, model, batch, device, optimizer, scheduler, criterion, epoch, step_counter,
               stats: Optional[dict] = None, sync: Optional["FlatGradSync"] = None):
    """One iteration of the reference's batch loop (train_model.py:498-582) without host syncs.
    ``batch`` = (data, target, frames, wav, sig_qual, indices) as the reference's loaders yield
    it (dataloader_physionet.py:151-172).  Returns the loss tensor (on device, detached)."""
    data, target, frames, wav, _sig_qual, indices = batch
    data = data.to(device, non_blocking=True)
    target_ohe = F.one_hot(target, args.num_classes).to(device, non_blocking=True)
    aug = augmentations2d if args.dataset in SPECTROGRAM_DATASETS else augmentations
    data, target_ohe, _, _ = aug.augment(args, data, target_ohe, frames, wav,
                                         augmentation_counter(args, step_counter), model, device, None,
                                         host_labels=target.numpy() if not target.is_cuda else None)
    fused = fused_loss_model(model, criterion, data, target_ohe, epoch) \
        if getattr(args, "depth", 0) == 0 else None
    if fused is not None:                   # head + loss as one autograd node (two launches, not four)
        loss, out = fused.loss_and_logits(data, target_ohe)
    else:
        out = model(data, depth=getattr(args, "depth", 0), pass_part="second")
        loss = criterion(out, target_ohe, indices, epoch, "train")
    args.depth = 0
    if sync is None:
        loss.backward()
    else:                                   # flat-buffer averaging instead of DDP hooks
        loss.backward(torch.full_like(loss, sync.backward_scale))
        if sync.params is None:
            sync.attach()
        sync.pack()
        sync.reduce_and_bind()
    if args.grad_clip and not isinstance(optimizer, ClipAdam):      # ClipAdam clips in its kernel
        nn.utils.clip_grad_value_([p for p in model.parameters() if p.grad is not None],
                                  clip_value=args.grad_clip)
    reseed_device_rng(args, device)
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    if scheduler is not None:
        scheduler.step()
    step_counter.add()
    if stats is not None:
        with torch.no_grad():
            stats["loss_sum"] += loss.detach()
            stats["hits"] += (out.argmax(1) == target_ohe.argmax(1)).sum()
            stats["seen"] += out.shape[0]
    return loss.detach()


class GraphedTrainStep:
    """The same step as ``train_step`` with forward + loss + backward + gradient clipping
    captured once in a hipGraph (torch.cuda.CUDAGraph) and replayed: at bs=256 the eager Potes
    step issues ~40 launches and is host-bound (~1 ms wall for ~0.45 ms of GPU work).

    Stays eager, around the replay: the augmentation (its index data and lambda are kernel
    arguments that change every step; it writes straight into the graph's static input — plain
    splices, and the saliency-guided ones around the frozen model's own captured pass) and the
    scheduler.  The ClipAdam update is the graph's last node: OneCycleLR moves lr AND beta1 every
    step, so the captured launch reads its eight scalars from device memory (they travel with the
    step's payload).  Under torch.distributed pass the UNWRAPPED model: gradients are packed into
    one flat buffer inside the graph, averaged by one eager all-reduce after the replay
    (``FlatGradSync``) — DDP's hooks cannot be captured — and the update is a SECOND small graph
    replayed behind the collective.  Needs static shapes (the loaders use drop_last=True)."""

    use_tape = True      # direct launches instead of the graph replay where the step allows it
    _TAPE_LAUNCHES = ("pcgmix_potes_stack_fwd_save_f32", "pcgmix_potes_head_loss_fwd_f32",
                      "pcgmix_potes_head_loss_bwd_f32", "pcgmix_potes_stack_bwd_mask_f32",
                      "pcgmix_adam_clip_multi_reduce_dev_f32")

    def __init__(self, args, model, optimizer, scheduler, criterion, device, batch_size, channels,
                 sig_len, sync: Optional[FlatGradSync] = None):
        if args.dataset in SPECTROGRAM_DATASETS:
            raise NotImplementedError("graphed step is wired for the 1D path")
        self.args, self.model, self.opt, self.sched = args, model, optimizer, scheduler
        self.ce = criterion.CEloss if hasattr(criterion, "CEloss") else criterion
        self.es = getattr(criterion, "es", None)
        self.device = device
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.x = torch.zeros(batch_size, channels, sig_len, device=device)
        # Everything small the replay reads and the host decides per step lives in ONE static
        # block, ``aux`` (float32 words): [0:2] dropout key (bits) | [4:12] Adam scalars |
        # [12:12+B*classes] the float targets.  ``_payload`` is its host image; a plain splice
        # carries it with its index block (``pcgmix_ctx_set_payload``: no copy of its own),
        # other steps upload it with one H2D.
        # [12:12+B/4] the class labels as bytes | then the float targets.  Hard targets (everything
        # but '(mixAll)') reach the fused Potes head+loss as those bytes: the payload is then 48 + B
        # bytes and travels in the splice kernel's ARGUMENTS with the index block — no copy at all
        # in front of the replay; other models / soft targets read the float block (one H2D).
        n_t = batch_size * args.num_classes
        n_lab = (batch_size + 15) // 16 * 4                   # words, 16-byte granules
        self.aux = torch.zeros(12 + n_lab + (n_t + 3) // 4 * 4, device=device)
        self.t_u8 = self.aux[12:12 + n_lab].view(torch.uint8)[:batch_size]
        self.t = self.aux[12 + n_lab:12 + n_lab + n_t].view(batch_size, args.num_classes)
        self.t[:, 0] = 1
        self._payload = np.zeros(self.aux.numel(), dtype=np.float32)
        self._pay_key = self._payload[0:2].view(np.uint32)
        self._pay_hyper = self._payload[4:12]
        self._pay_lab = self._payload[12:12 + n_lab].view(np.uint8)[:batch_size]
        self._pay_t = self._payload[12 + n_lab:12 + n_lab + n_t].reshape(batch_size, args.num_classes)
        self._rows = np.arange(batch_size)
        self.labels_mode = bool(                               # same test as _fwd_bwd's
            isinstance(self.ce, CELoss) and "(mixAll)" not in args.method and args.num_classes <= 255
            and fused_loss_model(model, self.ce, self.x, self.t, None) is not None)
        self._pay_bytes = (12 + n_lab) * 4 if self.labels_mode else self._payload.nbytes
        self.sync = sync
        self.bwd_seed = torch.full((), sync.backward_scale if sync else 1.0, device=device)
        # Dropout inside a captured region costs two extra fill launches per replay (torch's
        # graph-safe Philox state) on top of the mask kernels.  The Potes head reads random BYTES
        # instead, from a static buffer that the conv stack's forward kernel fills as a side job
        # (``pcgmix_potes_stack_fwd_save_f32``: a keyed counter hash).  The key is drawn on the
        # host from torch's device generator — (seed, offset), offset advanced by 4 per step — so
        # ``torch.cuda.manual_seed`` behaves as with nn.Dropout, the reference's reseeding before
        # every optimiser step (``reseed_device_rng``) included.
        self.rnd = None
        inner = model.module if hasattr(model, "module") else model
        if isinstance(inner, models.CNN_potes) and device.type == "cuda":
            K = inner.dimreduc.in_features
            drop = inner.cnn1[1][3] if len(inner.cnn1[1]) > 3 else None
            n_rnd = models.head_dropout_bytes(batch_size, K, float(drop.p) if drop else 0.0)
            self.rnd = torch.empty((n_rnd + 15) // 16 * 16, dtype=torch.uint8, device=device)
            inner.dropout_bytes = self.rnd
            inner.dropout_key = self.aux[0:2].view(torch.int32)
        # The optimiser update is the graph's last node when nothing has to happen between
        # backward and update (no gradient all-reduce) and the optimiser can read its scalars
        # from ``aux`` (ClipAdam).
        self.adam_in_graph = isinstance(optimizer, ClipAdam) and optimizer.can_capture() \
            and device.type == "cuda"
        self.graph_update = None        # with a sync: the update as a second graph behind the all-reduce
        # The warm-up passes run the network on the all-zero placeholder batch: they must leave no
        # trace.  Weights are not updated (no optimiser step); BatchNorm running statistics and
        # num_batches_tracked, and the device RNG stream the dropout masks come from, are put
        # back afterwards, so a graphed run starts from exactly the state an eager run starts from.
        buffers = [(b, b.detach().clone()) for b in model.buffers()]
        rng = torch.cuda.get_rng_state(device)
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        import warnings
        warn_always = torch.is_warn_always_enabled()
        torch.set_warn_always(True)                     # the stream-mismatch warning is a warn-once
        try:
            with warnings.catch_warnings(record=True) as caught, torch.cuda.stream(side):
                warnings.simplefilter("always")         # warm-up off the capture (no weight update)
                for _ in range(3):
                    self.opt.zero_grad(set_to_none=True)
                    self._fwd_bwd()
                if sync is not None and sync.params is None:    # once per sync: a second captured
                    sync.attach()                               # step (PipelinedTrainStep) shares it
                if self.adam_in_graph:
                    self.opt.prepare_capture([p for p in self.params if p.grad is not None])
        finally:
            torch.set_warn_always(warn_always)
        torch.cuda.current_stream(device).wait_stream(side)
        stale = [w for w in caught if "AccumulateGrad node's stream does not match" in str(w.message)]
        for w in caught:
            if w not in stale:
                warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        if stale:
            # An autograd graph built in an earlier pass over these parameters is still alive (a loss
            # or output tensor kept by the caller): its AccumulateGrad nodes belong to the stream of
            # that pass.  Backward under capture would then make THAT stream wait on an event of the
            # capturing stream; when it is the legacy default stream, hipStreamEndCapture walks a
            # null stream object and the process dies with SIGSEGV inside libamdhip64
            # (hip::Stream::EndCapture; reproduced with torch ops alone:
            # profiles/probes/capture_defer_probe.py 'puretorch').  Refuse here,
            # before any capture has begun.
            self.opt.zero_grad(set_to_none=True)
            raise RuntimeError(
                "GraphedTrainStep: an autograd graph from an earlier forward pass over this model's "
                "parameters is still alive (e.g. a non-detached loss/output tensor is still "
                "referenced).  Its AccumulateGrad nodes are bound to that pass's stream and would pull "
                "it into the capture (on ROCm: a crash in hipStreamEndCapture).  Drop those tensors "
                "(or .detach() them) before building the captured step.")
        with torch.no_grad():
            for b, saved in buffers:
                b.copy_(saved)
        torch.cuda.set_rng_state(rng, device)
        self.opt.zero_grad(set_to_none=True)
        if sync is not None and sync.world > 1:
            # no collective may be in flight while the graph is captured, and every rank must
            # capture (or fail) together
            torch.cuda.synchronize(device)
            sync.dist.barrier()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls of other threads (RCCL's watchdog) must not abort the capture
        # One rank, ClipAdam as the graph's last node: the conv stack's gradient reduction moves
        # into the optimiser launch (models.PotesStackFunction.defer_reduce).
        fold = {} if (self.adam_in_graph and sync is None
                      and os.environ.get("PCGMIX_NO_REDUCE_FOLD") is None) else None
        # The fused Potes step is five library launches and nothing else (forward, head + loss,
        # feature pass, weight gradients, reduction + update: profiles/r3_train_step_timeline.txt).
        # They are recorded while the capture makes them; ``launch`` then issues them DIRECTLY on the
        # current stream instead of replaying the hipGraph — the graph's launch leaves ~5 us of idle
        # GPU per step that back-to-back kernel launches do not (130.8 -> 125.7 us,
        # profiles/probes/tape_vs_graph.py).  The graph object stays: it owns the buffers.
        from . import _lib
        want_tape = (self.use_tape and fold is not None and self.labels_mode
                     and isinstance(model, models.CNN_potes) and os.environ.get("PCGMIX_NO_TAPE") is None)
        tape = [] if want_tape else None
        with _lib.capture_without_gc(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            models.PotesStackFunction.defer_reduce = fold
            _lib.TAPE = tape
            try:
                self.loss, self.out = self._fwd_bwd()
                if self.adam_in_graph and sync is None:
                    self.opt.capture_update(self.aux[4:12], deferred=fold)
            finally:
                models.PotesStackFunction.defer_reduce = None
                _lib.TAPE = None
        self.tape = None
        if tape is not None and [t[0] for t in tape] == list(self._TAPE_LAUNCHES) \
                and self._tape_covers_capture(tape, fold):
            import ctypes
            # pointer tables are snapshotted: ClipAdam re-uses its ctypes arrays for the next capture
            # (a second slot of PipelinedTrainStep), a graph node would have copied them too
            snap = lambda v: type(v)(*v) if isinstance(v, ctypes.Array) else v      # noqa: E731
            self.tape = [(name, fn, tuple(snap(v) for v in a[:-1])) for name, fn, a in tape]
        if self.adam_in_graph and sync is not None:
            # N > 1: [forward + backward + pack] | one eager all-reduce | [clip + Adam].  The update
            # reads the averaged gradients through views of the flat buffer and its eight scalars
            # from ``aux`` — an N-rank step issues exactly the launches of the one-rank step plus
            # the collective, and no Python optimiser code runs between them.
            for p, v in zip(sync.params, sync.views):
                p.grad = v
            self.graph_update = torch.cuda.CUDAGraph()
            with _lib.capture_without_gc(), torch.cuda.graph(self.graph_update, capture_error_mode="thread_local"):
                self.opt.capture_update(self.aux[4:12])

    def _tape_covers_capture(self, tape, fold) -> bool:
        """The tape replaces the graph only if the five recorded launches ARE the capture: no
        torch-side kernel may sit between them (an AccumulateGrad clone of a head gradient, a
        dense copy of a stride-mismatched gradient — the tape would skip it and the update would
        read stale memory).  Checked at build time: every gradient the captured update reads must
        be (a view of) the very buffer one of the recorded launches was handed — the backward
        launches' outputs, or the deferred-reduction block the update launch fills."""
        import ctypes
        import warnings
        written = set()
        for _name, _fn, a in tape:
            for v in a[:-1]:
                if isinstance(v, int) and v > 4096:
                    written.add(v)
                elif isinstance(v, ctypes.c_void_p) and v.value:
                    written.add(v.value)
        for p, g in zip(self.opt._cap_params, self.opt._cap_grads):
            # (gradients are views: the head's small ones of one block, the stack's of the deferred
            # block — compare the storage a launch was handed, not the view's own address)
            if g is not p.grad or g.untyped_storage().data_ptr() not in written:
                warnings.warn("GraphedTrainStep: the capture holds work besides the five library "
                              "launches (a gradient of %s is not written by them directly); "
                              "replaying the hipGraph instead of launching directly"
                              % (tuple(p.shape),))
                return False
        return True

    def _fwd_bwd(self):
        fused = fused_loss_model(self.model, self.ce, self.x, self.t, None) \
            if isinstance(self.ce, CELoss) else None
        if fused is not None:
            loss, out = fused.loss_and_logits(self.x, self.t_u8 if self.labels_mode else self.t)
        else:
            out = self.model(self.x, depth=0, pass_part="second")
            loss = self.ce(out, self.t)
        loss.backward(self.bwd_seed)
        if self.sync is not None and self.sync.params is not None:
            self.sync.pack()
        elif self.args.grad_clip and not isinstance(self.opt, ClipAdam):
            nn.utils.clip_grad_value_(self.params, clip_value=self.args.grad_clip)
        return loss.detach(), out.detach()

    def _next_key(self) -> None:
        """This step's dropout key from torch's device generator (host only, no launch)."""
        z = models.next_dropout_key(self.device)
        self._pay_key[0] = z & 0xFFFFFFFF
        self._pay_key[1] = z >> 32

    def step(self, batch, epoch, step_counter, stats: Optional[dict] = None):
        """One training step: ``prepare`` (augmentation into the static input, payload) then
        ``launch`` (replay, optimiser, scheduler, counters) on the current stream."""
        self.prepare(batch, epoch, step_counter)
        return self.launch(step_counter, stats)

    def prepare(self, batch, epoch, step_counter):
        """First half of a step, everything in front of the replay: the host image of the static
        block (targets, dropout key, Adam scalars — read from generator / optimiser state as the
        PREVIOUS step's ``launch`` left it, so prepare(k+1) must follow launch(k) on the host) and
        the augmentation launches, which write this object's static input and ``aux`` on the
        CURRENT stream.  ``PipelinedTrainStep`` runs it on a side stream for batch k+1 while the
        graph of batch k replays."""
        from . import hostprep, _lib, saliency as _saliency
        data, target, frames, wav, _sq, _idx = batch
        if self.es is not None and epoch > self.es:
            raise NotImplementedError("SELC phase is not captured; use train_step")
        args = self.args
        self._batch_size = int(data.shape[0])
        data = data.to(self.device, non_blocking=True)
        frames_np = augmentations._as_numpy_frames(frames)
        B, C, T = data.shape
        step = int(augmentation_counter(args, step_counter).count)
        # host image of the static block: one-hot float targets, dropout key, Adam scalars
        labels_np = target.numpy() if not target.is_cuda else target.cpu().numpy()
        if self.labels_mode:
            self._pay_lab[:] = labels_np
        else:
            self._pay_t.fill(0.0)
            self._pay_t[self._rows, labels_np] = 1.0
        if self.rnd is not None and self.model.training:
            self._next_key()
        if self.adam_in_graph:
            self.opt.next_hyper(self._pay_hyper)
        recipe = hostprep.plain_recipe(args.method, False)
        srec = hostprep.salopt_recipe(args.method) if recipe is None else None
        lib, ctx = _lib.load(), augmentations.step_context(data.device.index)
        _lib.check(lib.pcgmix_ctx_set_payload(ctx, self._payload.ctypes.data, self._pay_bytes,
                                              self.aux.data_ptr()), "pcgmix_ctx_set_payload")
        plan = hostprep.MixPlan(fired=False)
        if recipe is not None and B > 0:                # plain splice: one library call
            fired = augmentations.gate_passes(recipe, args.method, step, data.device.index)
            if fired:                                   # ... which carries the payload along
                augmentations.splice_plain(recipe, data, labels_np, frames_np, step, out=self.x)
        elif srec is not None and B > 0:
            # Saliency-guided splice with same-label partners (BASELINE config 3; augmentations.py:
            # 874-928): seed + boundaries + this step's payload in the arguments of one launch,
            # the frozen saliency model's captured pass, then search + splice straight into the
            # training graph's static input — the whole step stays on the stream, no host wait.
            fired = hostprep.gate_fires(args.method, step)
            if fired:
                g = _saliency.step_graph(args, data, args.num_classes)
                if g is None:
                    raise RuntimeError("the captured saliency pass is unavailable (saliency.USE_GRAPHS "
                                       "is off or the stream is capturing)")
                with torch.cuda.device(self.device):
                    augmentations._salopt_step(srec, g, data, None, labels_np, frames_np, step,
                                               out=self.x)
        else:
            if hostprep.select_method(args.method, False):
                plan = hostprep.make_plan(args.method, labels_np, frames_np, wav, step, B, C)
            fired = plan.fired
            if fired:
                sal = None
                if plan.salopt_mode is not None:        # salopt with '(samePCG)' & co.: general path
                    t_dev = F.one_hot(target, args.num_classes).to(self.device, non_blocking=True)
                    sal = _saliency.get_saliency_maps(args, self.device, data, t_dev, frames_np)
                augmentations.apply_plan(plan, data, frames_np, sal, out=self.x)
        if not fired:
            self.x.copy_(data, non_blocking=True)
        # any other step: the payload goes on its own (no-op when the splice took it)
        _lib.check(lib.pcgmix_ctx_flush_payload(
            ctx, torch.cuda.current_stream(self.device).cuda_stream), "pcgmix_ctx_flush_payload")
        if plan.fired and plan.mix_all:                 # float blend of the one-hot rows
            t_ohe = F.one_hot(target, args.num_classes).to(self.device, non_blocking=True)
            self.t.copy_(augmentations.blend_targets(t_ohe, plan))

    def launch(self, step_counter, stats: Optional[dict] = None):
        """Second half: replay the captured forward + backward (+ update), the gradient all-reduce
        and update graph under torch.distributed, scheduler, step counter, statistics."""
        B = self._batch_size
        if self.tape is not None:
            import ctypes
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            for name, fn, a in self.tape:
                err = fn(*a, st)
                if err:
                    from . import _lib
                    _lib.check(err, name)
        else:
            self.graph.replay()
        reseed_device_rng(self.args, self.device)
        if self.graph_update is not None:
            self.sync.reduce_and_bind()
            self.graph_update.replay()
        elif not self.adam_in_graph:
            if self.sync is not None:
                self.sync.reduce_and_bind()
                if self.args.grad_clip and not isinstance(self.opt, ClipAdam):
                    nn.utils.clip_grad_value_(self.sync.params, clip_value=self.args.grad_clip)
            self.opt.step()
        if self.sched is not None:
            self.sched.step()
        step_counter.add()
        if stats is not None:
            with torch.no_grad():
                stats["loss_sum"] += self.loss
                truth = self.t_u8 if self.labels_mode else self.t.argmax(1)
                stats["hits"] += (self.out.argmax(1) == truth).sum()
                stats["seen"] += B
        return self.loss


class PipelinedTrainStep:
    """Two ``GraphedTrainStep`` slots over the same model / optimiser, used alternately so that the
    augmentation of batch k+1 runs on a side stream WHILE the captured training graph of batch k
    replays on the main stream.

    The augmentation of a batch depends on that batch, its labels and the step count only
    (train_model.py:507: ``augment(args, data, target_ohe, frames, wav, step_counter, ...)``), never
    on the weights being trained — the saliency-guided methods use a FROZEN model
    (saliency.py:26-51) — so it can run ahead by one batch without changing a single value; it is
    the data-loader prefetch the reference does not have.  For BASELINE config 3 the augmentation
    (frozen forward + input gradient + search + splice, ~160 us) is as long as the training graph
    (~147 us) and both are issue/latency-bound kernels that leave the other room: measured
    309 -> 225 us per step with the two on separate streams (profiles/probes/overlap_probe.py);
    for the plain splice (10 us) the cross-stream hand-overs cost more than the overlap gains
    (148 -> 166 us), so ``train_model`` uses this class for the saliency-guided methods only.

    Host order is unchanged — prepare(k+1) follows launch(k), so dropout keys, Adam scalars,
    scheduler and step counter advance exactly as in the sequential loop — only the STREAM of the
    augmentation launches differs.  Buffer hazards: slot s's static input / ``aux`` are rewritten
    by prepare(k+2) only after the replay of batch k (same slot) has finished (event), and a replay
    waits for its slot's prepare (event).

        pipe = PipelinedTrainStep(args, model, opt, sched, crit, device, B, C, T, sync=...)
        for batch, nxt in pipe.pairs(loader):           # or: pipe.step(batch, epoch, sc, stats, nxt)
            pipe.step(batch, epoch, step_counter, stats, next_batch=nxt)
    """

    def __init__(self, *a, **kw):
        self.slots = [GraphedTrainStep(*a, **kw), GraphedTrainStep(*a, **kw)]
        self.device = self.slots[0].device
        self.side = torch.cuda.Stream(self.device)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]     # slot prepared (side stream)
        self.done = [torch.cuda.Event(), torch.cuda.Event()]      # slot's replay finished (main stream)
        self._fetched = torch.cuda.Event()
        self.cur = 0
        self.prepared = None            # id() of the batch already prepared into slot `cur`

    @property
    def loss(self):
        return self.slots[self.cur ^ 1].loss          # the step that ran last

    @property
    def x(self):
        return self.slots[0].x                        # (shape of the static input)

    @staticmethod
    def pairs(loader):
        """(batch, next batch or None) over ``loader`` — the one-batch lookahead ``step`` wants."""
        it = iter(loader)
        try:
            cur = next(it)
        except StopIteration:
            return
        for nxt in it:
            yield cur, nxt
            cur = nxt
        yield cur, None

    def _prepare_on_side(self, slot, batch, epoch, step_counter, fetched):
        self.side.wait_event(fetched)                   # the batch itself (loader gather, main stream)
        self.side.wait_event(self.done[slot])           # the slot's previous replay has read its buffers
        with torch.cuda.stream(self.side):
            self.slots[slot].prepare(batch, epoch, step_counter)
            self.ready[slot].record(self.side)
        if batch[0].is_cuda:    # allocated on the main stream (loader gather), read on the side stream
            batch[0].record_stream(self.side)

    def step(self, batch, epoch, step_counter, stats: Optional[dict] = None, next_batch=None):
        main = torch.cuda.current_stream(self.device)
        s = self.cur
        if self.prepared != id(batch):                  # first step, or the caller gave no lookahead
            # inline, on the main stream.  Both slots' prepares share the cached saliency graph
            # (seed, boundaries, activations, map) and the per-device step context (search
            # workspace): the side stream's prepare(k+1) must not start before THIS prepare's last
            # kernel has read them, so the hand-over point below is recorded behind it.
            self.slots[s].prepare(batch, epoch, step_counter)
        else:
            main.wait_event(self.ready[s])
        if next_batch is not None:
            # next_batch exists already (the caller's lookahead fetched it before this call): mark the
            # point on the main stream BEFORE this step's graph — and behind an inline prepare — so
            # that the side stream waits for the batch (and the shared saliency buffers) and not for
            # the graph it is meant to overlap
            self._fetched.record(main)
        self.prepared = None
        loss = self.slots[s].launch(step_counter, stats)
        self.done[s].record(main)
        if next_batch is not None:
            self._prepare_on_side(s ^ 1, next_batch, epoch, step_counter, self._fetched)
            self.prepared = id(next_batch)
        self.cur = s ^ 1
        return loss


def _no_epoch_step():
    return None


class _EpochStep:
    """(key, captured step) kept on the model by ``train_epoch``.  A copy or a pickle of the model
    must not drag the graph along: both see ``None`` in its place."""
    __slots__ = ("key", "step")

    def __init__(self, key, step):
        self.key, self.step = key, step

    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (_no_epoch_step, ())


def _epoch_graphed_step(args, model, optimizer, scheduler, criterion, device, epoch, batch):
    """The captured step ``train_epoch`` replays instead of calling ``train_step``, or None when
    the step has to stay eager.  A caller of the reference's ``train_epoch`` hands over exactly
    what ``GraphedTrainStep`` needs; the graph is built at the first batch and kept for as long as
    model, optimiser, scheduler, criterion, method and batch shape stay the same objects/values
    (the reference creates them once per run, train_model.py:293-410).  ``args.hipgraph = False``
    switches it off.  Eager: CPU, spectrogram datasets, wrapped (DataParallel/DDP) models, a
    criterion other than ``SELCLoss``, epochs past its turning point, an active process group
    (use ``train_model()`` / ``FlatGradSync`` there)."""
    if not getattr(args, "hipgraph", True) or device.type != "cuda":
        return None
    if args.dataset in SPECTROGRAM_DATASETS or not isinstance(criterion, SELCLoss):
        return None
    if not isinstance(model, (models.CNN_potes, models.ResNet9_myrtle)):
        return None
    if epoch > criterion.es or getattr(args, "num_epochs", epoch) > criterion.es:
        return None
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        return None
    data = batch[0]
    if data.dim() != 3:
        return None
    key = (id(optimizer), id(scheduler), id(criterion), tuple(data.shape), args.method,
           args.num_classes, device)
    hit = model.__dict__.get("_pcgmix_epoch_step")      # kept ON the model: dies with it (a module-
    if hit is None or hit.key != key:                   # level table would keep model and graph alive)
        B, C, T = data.shape
        # saliency-guided methods: two captured slots, the augmentation of the next batch on a side
        # stream under the graph of this one (+15 % at BASELINE config 3); one slot otherwise
        cls = PipelinedTrainStep if "(salopt" in args.method else GraphedTrainStep
        hit = _EpochStep(key, cls(args, model, optimizer, scheduler, criterion, device, B, C, T))
        model.__dict__["_pcgmix_epoch_step"] = hit
    return hit.step


def train_epoch(args, model, train_loader, device, optimizer, scheduler, criterion, epoch,
                step_counter, variability_counter=None, EXPERIMENT_ARGS=None):
    """Reference signature (train_model.py:490).  Returns (mean loss, accuracy, lr per step).
    On a GPU the step is the captured one (``_epoch_graphed_step``): the same values as
    ``train_step`` at three times its rate; batches of another shape (a loader without
    drop_last) fall back to the eager step."""
    model.train()
    torch.manual_seed(args.seed * 635410 + step_counter.count)      # :497 fixes this epoch's shuffle
    stats = {"loss_sum": torch.zeros((), device=device), "hits": torch.zeros((), device=device,
                                                                         dtype=torch.long), "seen": 0}
    lrs, n_batches = [], 0
    graphed = None
    for batch, nxt in PipelinedTrainStep.pairs(train_loader):          # one batch of lookahead
        lrs.append(optimizer.param_groups[0]["lr"])
        if n_batches == 0:
            graphed = _epoch_graphed_step(args, model, optimizer, scheduler, criterion, device, epoch, batch)
        if graphed is not None and tuple(batch[0].shape) == tuple(graphed.x.shape):
            if isinstance(graphed, PipelinedTrainStep):
                last = not step_counter.count + 1 < getattr(args, "num_steps", float("inf"))   # :584-586
                ahead = nxt if (not last and nxt is not None
                                and tuple(nxt[0].shape) == tuple(graphed.x.shape)) else None
                graphed.step(batch, epoch, step_counter, stats, next_batch=ahead)
            else:
                graphed.step(batch, epoch, step_counter, stats)
        else:
            train_step(args, model, batch, device, optimizer, scheduler, criterion, epoch,
                       step_counter, stats)
        n_batches += 1
        if not step_counter.count < args.num_steps:                  # :584-586
            break
    loss = float(stats["loss_sum"]) / max(1, n_batches)              # the only host syncs
    acc = float(stats["hits"]) / max(1, stats["seen"])
    return loss, acc, lrs


class performance_metrics_class:
    """The reference's result collector (train_model.py:178-195): a dict of lists with the same
    thirteen keys and the same ``add(key, value)``."""

    def __init__(self):
        self.dict = {k: [] for k in ("steps", "epochs", "times", "train_loss", "train_accuracy",
                                     "test_loss", "test_accuracy", "test_specificity",
                                     "test_sensitivity", "test_precision", "test_recall", "test_f1",
                                     "test_rocauc")}

    def add(self, string, value):
        self.dict[string].append(value)


@torch.no_grad()
def test_data_accuracy(args, model, test_loader, device, criterion=None, epoch=None,
                       performance=None):
    """Evaluation as train_model.py:591-670, reference signature ``(args, model, test_loader,
    device, criterion, epoch, performance)`` (called that way at :455): per recording, average
    the softmax of its heart cycles and take the argmax (:623-633); with '(class_majority)' in
    ``args.method`` the recording's class is the majority of its cycles' arg-max votes, a 0/1 tie
    going to class 1 (:634-647).  Accuracy / sensitivity / specificity / precision / recall / F1
    over recordings, ROC-AUC of the mean class-1 probability.  The per-recording sums are
    segmented sums on the device (index_add), not a Python dict of ``.item()`` values.

    With a ``performance`` object (anything with the reference's ``add(key, value)``,
    :194-195) the eight ``test_*`` values are appended to it in the reference's order
    (:651-669) and the function returns None, as the reference does.  Without one it returns
    the same numbers as a dict.  ``epoch`` is accepted and unused, as in the reference.

    Under '(class_majority)' the reference never collects the mean probabilities, so its own
    ``roc_auc_score`` line (:667) raises IndexError after the other seven values have been
    appended: with ``performance`` this function does exactly that; the dict form carries
    ``rocauc = None`` on that branch."""
    model.eval()
    rec_ids: dict = {}
    rec_label: dict = {}
    parts = []
    loss_sum = torch.zeros((), device=device, dtype=torch.float64)
    n = 0
    for data, target, _frames, wav, _sq, _idx in test_loader:
        data = data.to(device)
        out = model(data)
        prob = F.softmax(out, dim=1)
        for w, t in zip(wav, target.tolist()):
            rec_label.setdefault(rec_ids.setdefault(w, len(rec_ids)), t)   # first cycle's label
        ids = torch.tensor([rec_ids[w] for w in wav], device=device)
        parts.append((ids, prob))
        if criterion is not None:               # :608-609, accumulated on the device
            loss_sum += criterion(out, F.one_hot(target, args.num_classes).to(device),
                                  None, None, "test").double() * len(target)
        n += len(target)
    R = len(rec_ids)
    majority = "(class_majority)" in getattr(args, "method", "")
    acc_p = torch.zeros(R, args.num_classes, device=device)
    acc_n = torch.zeros(R, device=device)
    votes = torch.zeros(R, args.num_classes, device=device, dtype=torch.long)
    for ids, prob in parts:
        acc_p.index_add_(0, ids, prob)
        acc_n.index_add_(0, ids, torch.ones_like(ids, dtype=torch.float))
        if majority:
            votes.index_add_(0, ids, F.one_hot(prob.argmax(1), args.num_classes))
    mean_p = (acc_p / acc_n[:, None]).cpu().numpy()
    if majority:
        pred = np.zeros(R, dtype=np.int64)
        for r, c in enumerate(votes.cpu().numpy()):
            c = np.trim_zeros(c, "b") if c.any() else c[:1]          # what np.bincount returns
            pred[r] = 1 if (len(c) == 2 and c[0] == c[1]) else int(np.argmax(c))
    else:
        pred = mean_p.argmax(1)
    lab = np.asarray([rec_label[i] for i in range(R)])
    tp = int(((pred == 1) & (lab == 1)).sum()); tn = int(((pred == 0) & (lab == 0)).sum())
    fp = int(((pred == 1) & (lab == 0)).sum()); fn = int(((pred == 0) & (lab == 1)).sum())
    prec = tp / max(1, tp + fp)
    rec = tp / max(1, tp + fn)
    # ROC-AUC of the class-1 mean probability (Mann-Whitney U with average ranks for ties)
    auc = None
    if 0 < lab.sum() < R and not majority:
        score = mean_p[:, 1]
        order = np.argsort(score, kind="mergesort")
        ranks = np.empty(R)
        ranks[order] = np.arange(1, R + 1)
        for v in np.unique(score):
            tie = score == v
            ranks[tie] = ranks[tie].mean()
        n1 = int(lab.sum())
        auc = float((ranks[lab == 1].sum() - n1 * (n1 + 1) / 2) / (n1 * (R - n1)))
    n_total = len(test_loader.dataset) if hasattr(test_loader, "dataset") else n      # :653
    ev = {"accuracy": 100.0 * float((pred == lab).sum()) / max(1, R), "sensitivity": 100.0 * rec,
          "specificity": 100.0 * tn / max(1, tn + fp), "precision": prec, "recall": rec,
          "f1": 2 * prec * rec / max(1e-12, prec + rec), "rocauc": auc, "recordings": R,
          "loss": float(loss_sum) / max(1, n_total) if criterion is not None else None}
    if performance is None:
        return ev
    for key in ("accuracy", "loss", "specificity", "sensitivity", "f1", "precision", "recall"):
        performance.add("test_" + key, ev[key])                                       # :651-666
    if majority:
        raise IndexError("'(class_majority)' collects no mean probabilities: the reference's "
                         "roc_auc_score line (train_model.py:667) fails the same way")
    performance.add("test_rocauc", auc)                                               # :667-668
    return None


def train_model(args, dataset, device, use_graph: bool = True, log=print, pipeline: bool = True):
    """The reference's run driver (train_model.py:197-488) reduced to the hot path: seeds (:216-223),
    loaders (:244-248), model (:293-386), criterion/optimiser/scheduler (:390-410), the epoch loop
    (:428-478) with evaluation at the reference's 11 "plot epochs", and ``model.pth`` written with the
    reference's DataParallel key prefix (:481-482) so that saliency-guided runs of either code
    base can load it (saliency.py:50).  Plots, the pickle of the performance dict and the model
    zoo are out of scope.  Returns the performance dict.

    ``dataset`` is the dictionary ``dataloader_physionet.file2dict`` returns (time series:
    ``args.dataset = 'PhysioNet'``; log-mel images, one per cycle: ``'PhysioNet(spec128)'``,
    train_model.py:228-233 -> ``dataloader_physionet2d``); it is selected by ``physionet_dataloader``
    and kept resident on ``device``.  The step runs as a captured
    hipGraph (``use_graph``); under torch.distributed each rank trains on its shard of every batch
    and gradients are averaged by one all-reduce per step (``FlatGradSync`` around the graph, DDP
    for the eager step)."""
    import random
    import time as _time

    import torch.distributed as dist
    from . import dataloader_physionet as dlp
    from . import saliency as _sal

    spectro = args.dataset == "PhysioNet(spec128)"
    if args.dataset != "PhysioNet" and not spectro:
        raise NotImplementedError("train_model drives the PhysioNet time-series and spectrogram paths")
    seed_fix = 4                                                   # :217
    args.seed_fix = seed_fix
    torch.manual_seed(seed_fix)
    random.seed(seed_fix)
    np.random.seed(seed_fix)
    args.device = device
    if not hasattr(args, "classical_space"):
        args.classical_space = False
    if spectro:                                                    # :228-233
        from . import dataloader_physionet2d as dlp2
        loaders = dlp2.physionet_dataloader(args, dataset)
    else:
        loaders = dlp.physionet_dataloader(args, dataset)
    train_loader, train_labels = loaders.run("train", seed_fix)
    test_loader = loaders.run("valid" if args.valid else "test", None)
    args.sig_len = int(train_loader.data.shape[-1])
    torch.manual_seed(seed_fix)                                    # :293 initial weights
    model = build_model(args).to(device)
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    rank, world = (dist.get_rank(), dist.get_world_size()) if distributed else (0, 1)
    args.num_steps = args.num_epochs * (len(train_loader.dataset) // args.batch_size)   # :390
    criterion = SELCLoss(train_labels, args.num_classes, es=selc_turning_point(args), device=device)
    # (the spectrogram step is 80 ms of MIOpen convolutions: nothing for a graph to win, it stays eager)
    graphable = use_graph and device.type == "cuda" and args.num_epochs <= criterion.es and not spectro
    if not graphable:
        model = wrap_distributed(model, device)
    optimizer, scheduler = make_optimizer(args, model)
    step_counter = step_counter_class()
    graphed = None
    if graphable:
        # Saliency-guided methods: two captured slots, the augmentation of batch k+1 (as long as the
        # training graph itself) overlaps the graph of batch k: 297 -> 258 us per step.  For the plain
        # splice (10 us) the cross-stream hand-overs cost more than the overlap gains (148 -> 166 us,
        # profiles/r3_pipeline_ab.txt): one slot.
        cls = PipelinedTrainStep if pipeline and "(salopt" in args.method else GraphedTrainStep
        graphed = cls(args, model, optimizer, scheduler, criterion, device,
                      args.batch_size // world, args.num_channels, args.sig_len,
                      sync=FlatGradSync(model, device) if distributed else None)
    performance = performance_metrics_class()                                           # :421
    perf = performance.dict
    plot_epochs = set(np.linspace(1, args.num_epochs, 11).astype("int").tolist())       # :424
    args.depth = 0
    t_sum = 0.0
    for epoch in range(1, args.num_epochs + 1):
        t0 = _time.time()
        model.train()
        torch.manual_seed(args.seed * 635410 + step_counter.count)                      # :497
        stats = {"loss_sum": torch.zeros((), device=device),
                 "hits": torch.zeros((), device=device, dtype=torch.long), "seen": 0}
        n_batches = 0
        shard = (lambda b: shard_batch(b, rank, world)) if distributed else (lambda b: b)
        pairs = PipelinedTrainStep.pairs(shard(b) for b in train_loader)
        for batch, nxt in pairs:
            if isinstance(graphed, PipelinedTrainStep):
                last = not step_counter.count + 1 < args.num_steps        # :584-586 stops behind this one
                graphed.step(batch, epoch, step_counter, stats, next_batch=None if last else nxt)
            elif graphed is not None:
                graphed.step(batch, epoch, step_counter, stats)
            else:
                train_step(args, model, batch, device, optimizer, scheduler, criterion, epoch,
                           step_counter, stats)
            n_batches += 1
            if not step_counter.count < args.num_steps:
                break
        t_sum += _time.time() - t0
        if epoch in plot_epochs:
            performance.add("epochs", epoch)                                            # :447-455
            performance.add("steps", step_counter.count)
            performance.add("train_loss", float(stats["loss_sum"]) / max(1, n_batches))
            performance.add("train_accuracy", 100.0 * float(stats["hits"]) / max(1, stats["seen"]))
            try:
                test_data_accuracy(args, model, test_loader, device, criterion, epoch, performance)
            except IndexError:                    # '(class_majority)': the reference stops here (:667)
                performance.add("test_rocauc", None)
            performance.add("times", t_sum)
            if rank == 0 and log is not None:
                log(f"epoch {epoch:3d} step {step_counter.count:6d} train loss "
                    f"{perf['train_loss'][-1]:.4f} acc {perf['train_accuracy'][-1]:.1f}%  "
                    f"test acc {perf['test_accuracy'][-1]:.1f}%")
    out_dir = getattr(args, "EXPERIMENTS", None)
    if out_dir and rank == 0:
        exp = _sal.experiment_dir(args)
        os.makedirs(exp, exist_ok=True)
        inner = model.module if hasattr(model, "module") else model
        torch.save({"module." + k: v for k, v in inner.state_dict().items()},
                   os.path.join(exp, "model.pth"))
    perf["model"] = model
    return perf


class SyntheticCycleLoader:
    """Stands in for physionet_dataloader(...).run('train') (dataloader_physionet.py:204-229):
    yields the same 6-tuple (data, target, frames, wav, sig_qual, index) of CPU tensors, shuffled
    with torch's global generator and ``drop_last=True``, from a synthetic pool."""

    def __init__(self, pool, batch_size: int):
        self.x, self.frames, self.labels, self.wav = pool
        self.batch_size = batch_size
        self.dataset = range(len(self.labels))

    def __len__(self):
        return len(self.labels) // self.batch_size

    def __iter__(self):
        perm = torch.randperm(len(self.labels))
        for i in range(len(self)):
            idx = perm[i * self.batch_size:(i + 1) * self.batch_size]
            ii = idx.numpy()
            yield (torch.from_numpy(self.x[ii]), torch.from_numpy(self.labels[ii]),
                   torch.from_numpy(self.frames[ii]), tuple(self.wav[j] for j in ii),
                   torch.ones(len(ii), dtype=torch.long), idx)
