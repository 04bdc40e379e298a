"""Host prologue of one PCGmix augmentation call: everything that is integer or random.

The reference derives all randomness of a step from ``step_counter.count`` through CPython's
``random.Random`` and numpy's legacy global ``RandomState`` (augmentations.py:869, 936, 500-514,
659-666, 677).  Those streams *define* parity, are O(B) work, and are therefore kept on the
host and called in the reference's order:

    Random(step).uniform          probability gate                 augmentations.py:869-872
    Random(step).sample per group partner permutation              augmentations.py:500-514
    np.random.seed(step); beta    lambda (global numpy stream!)    augmentations.py:659-666
    Random(step).randint          '(rand)' placement offsets       augmentations.py:305-337
    np.random.normal              magnitude-warp knots             augmentations.py:677

The result is a small ``MixPlan`` of index/scalar data that the device kernels consume; no
waveform data is touched here.
"""
from __future__ import annotations

import functools
import random
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

from . import _lib

# Methods this package implements, in the reference's dispatch order (augmentations.py:864, 931).
PCGMIX_METHODS_1D = ("durmixmagwarp", "durratiomixup")
# 2D dispatch order: augmentations2d.py:286 (cutout), :325 (timemask), :361 (freqmask), :397
PCGMIX_METHODS_2D = ("durmixcutout", "durmixtimemask", "durmixfreqmask", "durratiomixup")

# Every name the reference dispatcher knows (augmentations.py:700-729, augmentations2d.py:269-281).
# A method string that names one of these but none of ours is refused loudly instead of being
# passed through un-augmented.
_REFERENCE_METHODS_1D = (
    "durratiocutmix", "lengthcutmix", "datasetcutmix", "wav-durratiocutmix", "wavcutmix",
    "lc-nointrusion", "labelcutmix", "swapsysdia", "s1s2mask", "cont-cutmix", "saliency-cutmix",
    "latentmixup", "manifold-cutmix(ch)", "manifold-cutmix", "manifold-cutout(ch)",
    "manifold-cutout", "cutmix(ch)", "cutmix", "cutout(ch)", "cutout", "gaussiannoise",
    "magnitudewarp", "timewarp", "mixup", "timemask", "durratiomixup", "durmixmagwarp",
    "respiratoryscale", "durmixrespscale")
_REFERENCE_METHODS_2D = (
    "durratiocutmix", "cutmix", "mixup", "latentmixup", "freqmask", "timemask", "cutout",
    "durratiomixup", "durmixfreqmask", "durmixtimemask", "durmixcutout")
# Branches the reference tests BEFORE ours (augmentations.py:731-862; augmentations2d.py:286-395)
_EARLIER_1D = ("durmixrespscale", "respiratoryscale", "timemask")
_EARLIER_2D = ()
_UNSUPPORTED_SELECTORS = ("(sameCVD)", "(closestbins=", "(closestknn=")


@dataclass
class MixPlan:
    """Index/scalar description of one augmentation step (host memory only)."""
    fired: bool
    name: str = ""
    step: int = 0
    mix: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))   # (B,) partner of b
    lam64: float = float("nan")                # np.random.beta result
    lam32: np.float32 = np.float32("nan")      # what multiplies the waveforms
    rand_off: Optional[np.ndarray] = None      # int32 (B,4) for '(rand)'
    salopt_mode: Optional[int] = None          # 0 = '(saloptenv', 1 = '(saloptsum'
    knots: Optional[np.ndarray] = None         # float64 (B, n_knots, C) as numpy drew them
    n_knots: int = 0
    mix_all: bool = False                      # '(mixAll)': targets are blended too
    zero_rect: Optional[np.ndarray] = None     # int32 (B,4) [row0,row1,col0,col1): 2D mask variants


@functools.lru_cache(maxsize=256)
def select_method(method: str, is2d: bool) -> Optional[str]:
    """Which of our branches the reference's if-chain would reach, or None for passthrough.

    Raises NotImplementedError for reference augmentations outside this package's scope."""
    ours = PCGMIX_METHODS_2D if is2d else PCGMIX_METHODS_1D
    known = _REFERENCE_METHODS_2D if is2d else _REFERENCE_METHODS_1D
    if not any(m in method for m in known):
        return None                                            # augmentations.py:731-732
    earlier = _EARLIER_2D if is2d else _EARLIER_1D
    hit = next((m for m in ours if m in method), None)
    if hit is None or any(e in method for e in earlier):
        raise NotImplementedError(
            f"method {method!r} selects a reference augmentation outside the PCGmix hot path; "
            f"this package implements {ours} only")
    for sel in _UNSUPPORTED_SELECTORS:
        if sel in method and not is2d:
            raise NotImplementedError(f"partner selector {sel!r} is out of scope (SURVEY.md §2)")
    return hit


@functools.lru_cache(maxsize=256)
def parse_probability(method: str) -> float:
    """Text after the last '+' (augmentations.py:865-868)."""
    parts = method.split("+")
    return float(parts[-1]) if len(parts) > 1 else 1.0


@functools.lru_cache(maxsize=256)
def parse_alpha(method: str, name: str) -> float:
    """'(alpha=a)' immediately in front of the method name (augmentations.py:897-899)."""
    parts = method.split("(alpha=")
    return float(parts[1].split(")" + name)[0]) if len(parts) > 1 else 1.0


@functools.lru_cache(maxsize=256)
def parse_magwarp(method: str):
    """'durmixmagwarp(sigma,knot)' (augmentations.py:919-923); defaults 0.2, 4."""
    sigma, knot = 0.2, 4
    parts = method.split("durmixmagwarp(")
    if len(parts) > 1:
        sigma = float(parts[1].split(",")[0])
        knot = int(method.split(",")[1].split(")")[0])
    return sigma, knot


@functools.lru_cache(maxsize=256)
def plain_recipe(method: str, is2d: bool):
    """(name, p, alpha, sigma, knots) when ``method`` is a plain splice — same-label partners, no
    '(rand)' offsets, no saliency, no 2D mask — i.e. what ``pcgmix_splice_same_label_f32`` does in
    one call; None otherwise (the general ``make_plan`` path handles those)."""
    name = select_method(method, is2d)
    if name is None:
        return None
    if is2d:
        if name != "durratiomixup" or "(salopt" in method:
            return None
        return name, parse_probability(method), 1.0, 0.0, 0          # augmentations2d.py:411
    if any(t in method for t in ("(rand)", "(salopt", "(samePCG)", "(sameDataset)", "(mixAll)")):
        return None
    sigma, knot = parse_magwarp(method) if name == "durmixmagwarp" else (0.0, -2)
    return name, parse_probability(method), parse_alpha(method, name), sigma, knot + 2


@functools.lru_cache(maxsize=256)
def salopt_recipe(method: str):
    """(mode, alpha, sigma, knots) when ``method`` is a saliency-guided splice with same-label
    partners — what ``pcgmix_ctx_salopt_begin/_finish`` do around the saliency pass; None otherwise
    ('(samePCG)', '(sameDataset)', '(mixAll)': the general ``make_plan`` path)."""
    name = select_method(method, False)
    if name is None or "(salopt" not in method:
        return None
    if any(t in method for t in ("(samePCG)", "(sameDataset)", "(mixAll)")):
        return None
    if "(saloptenv" in method:
        mode = 0
    elif "(saloptsum" in method:
        mode = 1
    else:
        raise NotImplementedError("only (saloptenv…) and (saloptsum…) exist in the reference")
    sigma, knot = parse_magwarp(method) if name == "durmixmagwarp" else (0.0, -2)
    return mode, parse_alpha(method, name), sigma, knot + 2


def gate_fires(method: str, step: int) -> bool:
    """Fresh ``Random(step)``; the method runs iff u < p (augmentations.py:869-872).  The draw is
    the library's bit-exact restatement of ``random.Random(step).uniform(0, 1)``."""
    return _lib.load().pcgmix_py_uniform01(int(step)) < parse_probability(method)


def shuffle_within_groups(keys, step: int) -> np.ndarray:
    """Partner permutation: group positions by key and permute every group with a fresh
    ``Random(step).sample`` (augmentations.py:500-514).  The draw itself runs in the library's
    exact restatement of CPython's sampler (pcgmix_partner_permutation_i64): ~10x cheaper than
    ``random.sample`` and bit-identical to it (tests/test_host_logic.py)."""
    if isinstance(keys, np.ndarray) and keys.dtype.kind in "iu":
        lo, hi = (int(keys.min()), int(keys.max())) if keys.size else (0, 0)
        if hi - lo < 4096:                      # class labels: ids are the labels themselves
            gid, n_groups = keys - lo, hi - lo + 1          # (empty groups are allowed)
        else:
            uniq, gid = np.unique(keys, return_inverse=True)
            n_groups = int(uniq.shape[0])
    else:
        table: dict = {}
        gid = np.fromiter((table.setdefault(k, len(table)) for k in keys), dtype=np.int32,
                          count=len(keys))
        n_groups = len(table)
    gid = np.ascontiguousarray(gid, dtype=np.int32)
    mix = np.empty(gid.shape[0], dtype=np.int64)
    lib = _lib.load()
    _lib.check(lib.pcgmix_partner_permutation_i64(gid.ctypes.data, gid.shape[0], n_groups,
                                                  int(step), mix.ctypes.data),
               "pcgmix_partner_permutation_i64")
    return mix


def partner_indices(method: str, labels: np.ndarray, wav: Sequence[str], step: int,
                    is2d: bool = False) -> np.ndarray:
    """Partner selection with the reference's override order (augmentations.py:877-896)."""
    labels = np.asarray(labels).reshape(-1)
    mix = shuffle_within_groups(labels.astype(np.int64, copy=False), step)  # same label
    if is2d:
        return mix                                                          # augmentations2d.py:410
    if "(samePCG)" in method:                                               # augmentations.py:528
        mix = shuffle_within_groups(list(wav), step)
    if "(sameDataset)" in method:                                           # augmentations.py:542
        mix = shuffle_within_groups([f"{w[0]}_{int(t)}" for w, t in zip(wav, labels)], step)
    if "(mixAll)" in method:                                                # augmentations.py:883
        mix = shuffle_within_groups(np.zeros(len(labels), dtype=np.int64), step)  # one group
    return mix


def rand_offsets(frames: np.ndarray, mix: np.ndarray, step: int) -> np.ndarray:
    """'(rand)': offset of the shorter state inside the longer one,
    ``Random(step).randint(0, |gap|)`` with a fresh generator per (sample, state), so the value
    depends on (step, |gap|) only (augmentations.py:305-337)."""
    lens = np.diff(frames, axis=1)
    gap = np.abs(lens[mix] - lens)
    lib = _lib.load()
    uniq, inv = np.unique(gap, return_inverse=True)
    vals = np.array([lib.pcgmix_py_randint0(int(step), int(g)) for g in uniq], dtype=np.int32)
    return vals[inv].reshape(gap.shape)


def mask_rectangles(method: str, name: str, frames: np.ndarray, step: int, n_rows: int,
                    n_cols: int) -> np.ndarray:
    """Zeroed rectangle per sample for durmixcutout / durmixtimemask / durmixfreqmask
    (augmentations2d.py:309-323, 348-358, 384-394): region sizes from
    ``Random(step+131071).uniform``, positions from ``Random(step+13119).uniform``; the time span
    is a fraction of each sample's own cycle length (``int(frac * f[-1])``), the frequency span is
    one row range for the whole batch.  Rows count along the flattened (channel, frequency) axis
    of the kernel call; the reference's images have one channel."""
    def clamp01(v):
        return min(max(v, 0), 1)
    t_max = f_max = 0.2
    key = name[len("durmix"):] + "("
    parts = method.split(key)
    if len(parts) > 1:
        if name == "durmixcutout":
            t_max = clamp01(float(parts[1].split(",")[0]))
            f_max = clamp01(float(method.split(",")[1].split(")")[0]))
        else:
            t_max = f_max = clamp01(float(parts[1].split(")")[0]))
    B = frames.shape[0]
    rect = np.zeros((B, 4), dtype=np.int32)
    rect[:, 1] = n_rows
    rect[:, 3] = n_cols
    if name in ("durmixcutout", "durmixtimemask"):
        gap = random.Random(step + 131071).uniform(0, t_max)
        frac1 = random.Random(step + 13119).uniform(0, 1 - gap)
        frac2 = frac1 + gap
        beat = frames[:, 4].astype(np.float64)
        rect[:, 2] = (frac1 * beat).astype(np.int64)          # int() truncation
        rect[:, 3] = (frac2 * beat).astype(np.int64)
    if name in ("durmixcutout", "durmixfreqmask"):
        fgap = random.Random(step + 131071).uniform(0, f_max)
        h1 = int(n_rows * random.Random(step + 13119).uniform(0, 1 - fgap))
        rect[:, 0] = h1
        rect[:, 1] = min(n_rows, h1 + int(fgap * n_rows))
    return rect


# ---- numpy's global stream: lambda and the warp knots ---------------------------------------------
_NPDRAW = None        # (handle, lam, knots pointer, hit) of the process-wide draw object
_NP_GLOBAL = None     # (bit generator of numpy's global RandomState, address of its MT19937 state)


def _npdraw():
    global _NPDRAW
    if _NPDRAW is None:
        import ctypes
        import os
        h = ctypes.c_void_p()
        look = max(0, min(3, int(os.environ.get("PCGMIX_NPDRAW_LOOKAHEAD", "2"))))
        if _lib.load().pcgmix_npdraw_create(ctypes.byref(h), look):
            raise RuntimeError("pcgmix_npdraw_create failed")
        _NPDRAW = (h, ctypes.c_double(), ctypes.c_void_p(), ctypes.c_int())
    return _NPDRAW


def _numpy_global_state():
    """(lock, address) of the MT19937 state behind ``np.random.seed/beta/normal``, or None when
    the global RandomState runs on another bit generator (``np.random.set_bit_generator``)."""
    global _NP_GLOBAL
    bg = np.random.get_bit_generator()
    if _NP_GLOBAL is None or _NP_GLOBAL[0] is not bg:
        addr = bg.ctypes.state_address if type(bg).__name__ == "MT19937" else None
        _NP_GLOBAL = (bg, bg.lock, addr)
    return _NP_GLOBAL[1], _NP_GLOBAL[2]


def draw_lambda_knots(step: int, alpha: float, sigma: float, count: int):
    """``np.random.seed(step); lam = np.random.beta(alpha, alpha)`` (augmentations.py:661-663) and,
    for ``count`` > 0, ``np.random.normal(1.0, sigma, count)`` right behind it (:677) — the same
    doubles, and numpy's GLOBAL stream left exactly where the reference leaves it (asserted by
    tests/test_host_logic.py against numpy itself).  Returns (lam, knots) with knots = the host
    address of ``count`` float64 (library memory, valid until the next call), an ndarray on the
    fallback path, or None.

    The 6144 normals of a (256, 6, 4) block cost numpy ~110 us per step — ten times the kernel they
    feed.  The library restates the legacy stream (csrc/pcgmix_nprand.hip) and draws the blocks of
    the NEXT steps on worker threads (they depend on (step, alpha, sigma, count) only); a matching
    call picks its block up and writes the final generator state into numpy's own state memory.
    Outside that restatement's contract (alpha <= 0: the reference does not seed; alpha > 1:
    gamma-based beta; an odd count: numpy's Gaussian cache would be left full; a foreign bit
    generator) the draws are numpy's own calls."""
    if count == 0 and alpha > 0.0:          # plain splice: numpy's own two calls, nothing else (0.5 us less)
        np.random.seed(step)
        return float(np.random.beta(alpha, alpha)), None
    lock, addr = _numpy_global_state()
    if 0.0 < alpha <= 1.0 and count > 0 and not (count & 1) and addr is not None \
            and 0 <= step <= 0xFFFFFFFF:
        import ctypes
        h, lam, kp, hit = _npdraw()
        np.random.seed(step)                # key (overwritten below) AND the empty Gaussian cache
        with lock:
            err = _lib.load().pcgmix_npdraw_step(h, step, alpha, sigma, count, addr,
                                                 ctypes.byref(lam), ctypes.byref(kp), ctypes.byref(hit))
        if err:
            raise RuntimeError("pcgmix_npdraw_step refused a draw inside its contract")
        return lam.value, kp.value
    if alpha > 0.0:
        np.random.seed(step)                # global stream, as the reference (side effect kept)
        lam = float(np.random.beta(alpha, alpha))
    else:
        lam = 1.0
    knots = np.random.normal(loc=1.0, scale=sigma, size=count) if count else None
    return lam, knots


def knots_array(knots, shape) -> np.ndarray:
    """``draw_lambda_knots``'s knots as an owned float64 array of ``shape`` (as numpy fills it)."""
    if isinstance(knots, np.ndarray):
        return knots.reshape(shape)
    import ctypes
    n = int(np.prod(shape))
    return np.frombuffer((ctypes.c_double * n).from_address(knots), dtype=np.float64).reshape(shape).copy()


def validate_frames(frames: np.ndarray, sig_len: int) -> None:
    """The reference silently mis-slices (and usually raises a shape error) when a cycle runs
    past the padded length; refuse such input up front."""
    if frames.ndim != 2 or frames.shape[1] != 5:
        raise ValueError(f"frames must be (B, 5), got {frames.shape}")
    if frames.size == 0:
        return
    if (frames[:, 1:] < frames[:, :-1]).any() or int(frames.min()) < 0:
        raise ValueError("frames must be non-decreasing and non-negative")
    if int(frames.max()) > sig_len:
        raise ValueError(f"heart cycle ends at {int(frames.max())} > signal length {sig_len}")


def make_plan(method: str, labels, frames: np.ndarray, wav: Sequence[str], step: int,
              batch: int, channels: int, is2d: bool = False, n_cols: int = 0) -> MixPlan:
    """Everything random/integer for one step, in the reference's RNG order.

    ``labels`` may be an array or a zero-argument callable returning one: they are needed only
    when the gate fires (on a GPU they cost a device->host sync, augmentations.py:501), so a
    callable lets rejected steps skip the sync."""
    name = select_method(method, is2d)
    if name is None or not gate_fires(method, step):
        return MixPlan(fired=False, step=step)
    if frames.shape[0] != batch:
        raise ValueError("labels/frames do not match the batch size")
    plan = MixPlan(fired=True, name=name, step=step)
    # numpy's global stream first (lambda, then the warp knots right behind it): neither depends
    # on the labels, and python's random.Random (partners, offsets, masks) is a separate stream,
    # so the order BETWEEN the two streams is free — a callable `labels` that has to wait for the
    # GPU is asked as late as possible, after the ~0.1 ms of normal draws.
    alpha = 1.0 if is2d else parse_alpha(method, name)                      # augmentations2d.py:411
    sigma, knot = parse_magwarp(method) if (not is2d and name == "durmixmagwarp") else (0.0, -2)
    # seed -> beta (augmentations.py:661-663) -> normal right behind it (:677)
    plan.lam64, knots = draw_lambda_knots(step, alpha, sigma, batch * (knot + 2) * channels)
    plan.lam32 = np.float32(plan.lam64)                                     # augmentations.py:903
    if knot + 2:
        plan.n_knots = knot + 2
        plan.knots = knots_array(knots, (batch, knot + 2, channels))
    if callable(labels):
        labels = labels()
    labels = np.asarray(labels).reshape(-1)
    if labels.shape[0] != batch:
        raise ValueError("labels/frames do not match the batch size")
    plan.mix = partner_indices(method, labels, wav, step, is2d)
    if is2d and name != "durratiomixup":
        plan.zero_rect = mask_rectangles(method, name, frames, step, channels, n_cols)
    if not is2d and "(rand)" in method and "(salopt" not in method:
        plan.rand_off = rand_offsets(frames, plan.mix, step)
    # saliency-guided placement: 1D augmentations.py:905-913; 2D only under durratiomixup
    # (augmentations2d.py:416-423 — the mask variants never look at '(salopt')
    if (not is2d or name == "durratiomixup") and "(salopt" in method:
        if "(saloptenv" in method:
            plan.salopt_mode = 0
        elif "(saloptsum" in method:
            plan.salopt_mode = 1
        else:
            raise NotImplementedError("only (saloptenv…) and (saloptsum…) exist in the reference")
    if not is2d:
        plan.mix_all = "(mixAll)" in method
    return plan


def _gpu_numa_node(torch, index):
    pr = torch.cuda.get_device_properties(index)
    bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
    with open("/sys/bus/pci/devices/%s/numa_node" % bus) as f:
        return bus, int(f.read())


def _slice_for(cpus, ordinal, n_cpus):
    """``n_cpus`` of a node's CPU list for the ``ordinal``-th GPU of that node.  The first half of
    the list are the physical cores where SMT siblings are listed behind them; the node's first
    eight CPUs (interrupts, housekeeping) are left alone when there is room."""
    phys = cpus[:max(2, len(cpus) // 2)]
    n = max(2, min(n_cpus, len(phys)))
    usable = phys[8:] if len(phys) - 8 >= n else phys
    slot = ordinal % max(1, len(usable) // n)
    return usable[slot * n:(slot + 1) * n]


def bind_host_threads(device_index: int = 0, local_rank: int = 0, n_cpus: int = 8) -> str:
    """Pin the calling process to ``n_cpus`` CPUs of the NUMA node its GPU hangs off (a slice of its
    own for each GPU of that node), and say what was done.  Threads created afterwards — the
    draw-ahead workers of §3.6 — inherit the mask.  The strict-signature step is a host loop with one
    device->host hand-over per call: left to the scheduler on a two-socket box the process migrates
    between 256 CPUs and the step is 22.7-24.4 us; on four CPUs next to the GPU 20.4-20.5 us, on the
    remote socket 22.6-22.7 (MI355X box, ``profiles/r4_host_affinity.txt``).  The reference does not
    pin anything; a launcher would normally do this (``numactl``), ``torch.distributed.run`` does
    not.  ``local_rank`` only breaks ties where the other GPUs' nodes cannot be read.  No-op (with
    the reason in the returned string) wherever the topology cannot be read or leaves fewer than two
    CPUs."""
    import os
    if not hasattr(os, "sched_setaffinity"):
        return "no affinity API"
    try:
        import torch
        bus, node = _gpu_numa_node(torch, device_index)
        if node < 0:
            return "GPU %s reports no NUMA node" % bus
        try:        # which of this node's GPUs am I (device properties only: no context is created)
            ordinal = sum(1 for j in range(device_index) if _gpu_numa_node(torch, j)[1] == node)
        except Exception:
            ordinal = local_rank
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            cpus = []
            for part in f.read().strip().split(","):
                if "-" in part:
                    a, b = part.split("-")
                    cpus += list(range(int(a), int(b) + 1))
                elif part:
                    cpus.append(int(part))
    except Exception as exc:                                   # no sysfs, no such device, ...
        return "topology not readable (%s)" % type(exc).__name__
    allowed = os.sched_getaffinity(0)
    cpus = [c for c in cpus if c in allowed]
    if len(cpus) < 2:
        return "fewer than two CPUs of node %d allowed" % node
    chosen = _slice_for(cpus, ordinal, n_cpus)
    os.sched_setaffinity(0, chosen)
    return "GPU %s on NUMA node %d: pinned to CPUs %d-%d" % (bus, node, chosen[0], chosen[-1])
