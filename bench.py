#!/usr/bin/env python3
"""Headline benchmark: augmented PCG samples/s of the drop-in ``augment()`` on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): synthetic 2.5 s @ 2 kHz heart cycles, batch 256 per GPU,
``durratiomixup`` feeding the 1D-CNN, i.e. (256, 4, 5000) float32 — the Potes 1D-CNN needs its
four band-pass channels.  One step = one ``augment()`` call through the
reference's own call signature (device-resident input -> device-resident output, labels read
back, partner permutation drawn, one fused HIP launch — made before the labels are known, its
blocks wait for the index records the host writes: DESIGN.md §3.6 "armed").  Every rank augments
its own batch (no collective on the data path: weak scaling) and pins itself to CPUs of its GPU's
NUMA node (``config.host_affinity``; PCGMIX_BENCH_NO_AFFINITY=1 leaves it to the scheduler).

One JSON line is printed by rank 0.  Besides the contract keys it carries
  roofline      the splice kernel against the HBM roofline (12*C*T algorithmic bytes/sample and the
                bytes this batch needs), kernel time measured with HIP events on the launch stream
                around back-to-back launches of the same splice body (the timed steps' own kernel
                contains a wait for the host: ``roofline.step_kernel``)
  cpu_baseline  the CPU oracle (reference structure: per-sample Python loop) timed on this
                host on a bounded number of batches of the same workload
  extra         secondary measurements (other shapes/methods, saturating batch, train step/s)
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pcgmix_amd  # noqa: E402,F401
from pcgmix_amd import augmentations, hostprep, synthetic, train_model as tm  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


class Args:
    def __init__(self, method):
        self.method = method
        self.num_classes = 2
        self.batch_size = 256
        self.sample_rate = 2000
        self.model = "Potes"
        self.dataset = "PhysioNet"
        self.num_channels = 4


class StepCounter:
    def __init__(self):
        self.count = 0

    def add(self):
        self.count += 1


def strict_json(obj):
    """json.dumps that never emits NaN/Infinity (not JSON): non-finite floats become null."""
    def clean(o):
        if isinstance(o, float):
            return o if np.isfinite(o) else None
        if isinstance(o, dict):
            return {k: clean(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [clean(v) for v in o]
        if isinstance(o, (np.floating, np.integer)):
            return clean(o.item())
        return o
    return json.dumps(clean(obj), allow_nan=False)


def make_device_batch(B, C, T, rate, seed, device):
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=rate, seed=seed)
    data = torch.from_numpy(x).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
    return x, data, tgt, torch.from_numpy(frames), labels, wav


def settle_heap():
    """Before a timed region: collect, then move everything that is alive (this process has run
    other legs, compiled graphs, loaded torch) to the permanent generation.  The collector stays
    ON inside the region, but a full collection no longer walks the whole heap — in a 20 ms region
    one such pass (several ms here) read as +10-30 us per step of a 200 us step."""
    gc.collect()
    gc.freeze()


HOST_AFFINITY = None
ORIGINAL_CPUS = None   # the job's CPU mask before main() pinned the process
STAMPS = [0.0] * 4096     # preallocated: per-call host clock reads of a traced region

_SPIN = {}


def settle_clocks(device, ms=40.0):
    """Bring the GPU to its loaded power state before a leg's warm-up steps: ~40 ms of neutral
    work (a float32 GEMM and a streaming pass over 256 MB, nothing of the package), then a
    synchronize.  Why: a leg starts after tens of ms of host-only work (graph capture, heap
    settling) with the GPU at idle clocks, and the clocks climb for the first milliseconds of
    load — in a rocprofv3 trace of `--steps 20 --warmup 5` the same kernel of the train step
    shortened monotonically from 78.3 to 74.9 us over the 25 steps of the leg, the step from 172
    to 164.5 us (round 3, gpurun_out/prof_r3_bench).  A 4 ms region measured the ramp, not the
    step.  This is not part of the W warm-up steps and not of the K timed ones; the JSON line
    records it (`config.pre_settle`)."""
    st = _SPIN.get(device)
    if st is None:
        st = _SPIN[device] = (torch.randn(4096, 4096, device=device), torch.empty(4096, 4096, device=device),
                              torch.zeros(64 << 20, device=device))
    a, c, big = st
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            torch.mm(a, a, out=c)
            big.add_(1.0)
        torch.cuda.synchronize(device)


def call_trace(t0, stamps, n, dt, steps):
    """Host-side return time of each call of a timed region (the calls are asynchronous: this is
    when the host came back, the queue drains behind it): first / last call and the region's tail
    (synchronize + barrier), in us — shows whether a short region is at steady state."""
    d = [1e6 * (b - a) for a, b in zip([t0] + stamps[:n - 1], stamps[:n])]
    return {"first_call_us": d[0], "last_call_us": d[-1], "median_call_us": float(np.median(d)),
            "max_call_us": max(d), "drain_us": 1e6 * (t0 + dt - stamps[n - 1]) if n == steps else None,
            "first_5_calls_us": d[:5]}


def run_augment_steps(method, data, tgt, frames, wav, device, steps, warmup, barrier,
                      host_labels=None, trace=None):
    """``steps`` calls of the drop-in ``augment()`` through the reference's positional signature
    (``host_labels``: the keyword extension that spares the label read-back, for the split of the
    step time reported under extra.host_split)."""
    args, sc = Args(method), StepCounter()
    kw = {} if host_labels is None else {"host_labels": host_labels}
    # Heap settling (tens of ms of host-only work) comes BEFORE the warm-up: nothing but the
    # barrier sits between the last warm-up call and t0, so a 20-step region is steady state.
    settle_heap()
    settle_clocks(device)     # A/B on one box, four fresh processes each: 24.9 against 25.4 us per step
    out = None
    for _ in range(warmup):
        # bound to `out` exactly as in the timed loop: while the previous output is still alive the
        # next call needs a SECOND 20 MB block from torch's caching allocator — with the result
        # discarded here, that hipMalloc (~40 us) landed on the second timed call (round 3: calls
        # of 40 / 63 / 25 / 24 us at the head of every region; profiles/r3_region_start_probe.txt)
        out = augmentations.augment(args, data, tgt, frames, wav, sc, None, device, "", **kw)
        sc.add()
    barrier()
    torch.cuda.synchronize()
    stamps = STAMPS if trace is not None else None
    t0 = time.perf_counter()
    if stamps is None:
        for _ in range(steps):
            out = augmentations.augment(args, data, tgt, frames, wav, sc, None, device, "", **kw)
            sc.add()
    else:                                           # same loop + one clock read per call (~40 ns)
        n = 0
        for _ in range(steps):
            out = augmentations.augment(args, data, tgt, frames, wav, sc, None, device, "", **kw)
            sc.add()
            if n < len(stamps):
                stamps[n] = time.perf_counter()
                n += 1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    if trace is not None:
        trace.update(call_trace(t0, stamps, min(steps, len(stamps)), dt, steps))
    return dt, out


def exact_mix_bytes(frames, mix, C, T):
    """Bytes one launch of the splice kernel must move for THIS batch: every own element read
    once and every output element written once (8*C*T per sample) plus the partner row inside
    the blended ranges only, sum_k min(len_k(b), len_k(mix[b])) elements per channel row
    (augmentations.py:294-304).  The contract's 12*C*T counts the whole partner
    row; the kernel never reads the part of it that is not blended."""
    frames = np.asarray(frames, dtype=np.int64)
    lens = np.diff(frames, axis=1)
    blended = int(np.minimum(lens, lens[np.asarray(mix)]).sum())
    return 4.0 * C * (2.0 * frames.shape[0] * T + blended)


def karg_unroll(B, C, T, warp):
    """Unroll of the kernarg instantiation the drop-in step launches for this problem (plain
    splice, B <= 256: the index block travels in the kernel arguments), or 0."""
    import ctypes
    from pcgmix_amd import _lib
    if warp or os.environ.get("PCGMIX_NO_KARG"):
        return 0
    u = ctypes.c_int()
    return u.value if _lib.load().pcgmix_mix_karg_variant(B, C, T, ctypes.byref(u)) else 0


def mix_kernel_name(B, C, T, n_knots):
    """Name of the instantiation the drop-in step launches for this problem (asked from the
    library: the choice of lane width and unroll lives there).  n_knots = 0: no warp."""
    u = karg_unroll(B, C, T, n_knots)
    if u:
        return f"pcgmix::mix_warp_karg_kernel<false, {u}>"
    return mix_warp_kernel_name(B, C, T, n_knots)


def mix_warp_kernel_name(B, C, T, n_knots, zero_rect=False):
    """Name of the instantiation pcgmix_mix_warp_f32 launches (index block in device memory)."""
    import ctypes
    from pcgmix_amd import _lib
    buf = ctypes.create_string_buffer(96)
    _lib.check(_lib.load().pcgmix_mix_kernel_name(B, C, T, int(n_knots), int(bool(zero_rect)), 1,
                                                  buf, 96), "pcgmix_mix_kernel_name")
    return buf.value.decode()


def _n_knots(plan):
    return 0 if plan.knots is None else int(plan.knots.shape[1])


def kernel_back_to_back_ms(method, B, C, T, rate, device, iters=200, per_launch=False, info=None):
    """The fused kernel alone, launched back to back from one prepared plan (no host prologue
    between launches): the figure to compare with rocprofv3's per-kernel average."""
    if B * C * T > 64_000_000:          # saturating batches: random payload made on device
        frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=rate, seed=0)
        data = torch.randn(B, C, T, device=device)
    else:
        x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=rate, seed=0)
        data = torch.from_numpy(x).to(device)
    plan = hostprep.make_plan(method, labels, frames, wav, 1, B, C)
    if info is not None:
        info["exact_bytes"] = exact_mix_bytes(frames, plan.mix, C, T)
        info["kernel"] = mix_kernel_name(B, C, T, _n_knots(plan))
    with torch.cuda.device(device):
        dev, offs = augmentations.upload_plan(plan, frames, device)
    base = dev.data_ptr()
    out = torch.empty_like(data)
    knots_ptr = op_ptr = None
    if plan.knots is not None:
        op = augmentations.spline_operator(device, T, plan.n_knots)
        knots_ptr, op_ptr = base + offs["knots"], op.data_ptr()

    def launch():
        augmentations.launch_mix(data, out, base + offs["frames"], base + offs["mix"], None,
                                 float(plan.lam32), knots_ptr, op_ptr, plan.n_knots, B, C, T)
    if karg_unroll(B, C, T, plan.knots is not None):
        # the instantiation the drop-in step runs at this size: index block in the kernel arguments
        import ctypes
        from pcgmix_amd import _lib
        fr16 = np.ascontiguousarray(frames, dtype=np.int16)
        mx16 = np.ascontiguousarray(plan.mix, dtype=np.int16)
        lib, lam_c = _lib.load(), ctypes.c_float(float(plan.lam32))
        stream = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

        def launch():                                        # noqa: F811
            _lib.check(lib.pcgmix_mix_karg_f32(data.data_ptr(), out.data_ptr(), fr16.ctypes.data,
                                               mx16.ctypes.data, lam_c, B, C, T, stream),
                       "pcgmix_mix_karg_f32")
    for _ in range(10):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    interval = e0.elapsed_time(e1) / iters
    if not per_launch:
        return interval
    # per-launch duration: an event pair around every launch while the queue stays full, so the
    # pair brackets the kernel itself (start -> end, what rocprofv3 --kernel-trace reports);
    # consecutive kernels of one stream overlap their ramp-up/drain, so this is slightly longer
    # than the launch-to-launch interval above
    pairs = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        pairs.append((a, b))
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in pairs])), interval


class _EveryAllowedCpu:
    """The CPU baseline gets the whole host back: every thread of this process (torch's intra-op
    pool may exist already, pinned with the rest in main()) on the mask the job started with for
    the duration, each thread's own mask restored afterwards."""

    def __enter__(self):
        self.saved = {}
        if ORIGINAL_CPUS is None or not hasattr(os, "sched_setaffinity"):
            return self
        for tid in os.listdir("/proc/self/task"):
            try:
                self.saved[int(tid)] = os.sched_getaffinity(int(tid))
                os.sched_setaffinity(int(tid), ORIGINAL_CPUS)
            except OSError:
                pass
        return self

    def __exit__(self, *exc):
        for tid, mask in self.saved.items():
            try:
                os.sched_setaffinity(tid, mask)
            except OSError:
                pass
        return False


def cpu_baseline(method, B, C, T, rate, budget_s=12.0):
    with _EveryAllowedCpu():
        return _cpu_baseline(method, B, C, T, rate, budget_s)


def _cpu_baseline(method, B, C, T, rate, budget_s):
    """CPU oracle on the host cores of this box: whole batches of the benchmark workload until
    about ``budget_s`` seconds of CPU work have been timed.  torch's intra-op thread count is
    first probed (1, 8, all cores; 3 batches each) and the fastest setting is used, so the
    baseline is not handicapped by oversubscription on tiny slice ops."""
    from oracle import pcgmix_oracle as O
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=rate, seed=0)
    ncpu = os.cpu_count() or 1
    probe = {}
    for nt in sorted({1, min(8, ncpu), ncpu}):
        torch.set_num_threads(nt)
        O.augment(method, x, labels, frames, wav, 0)
        t0 = time.perf_counter()
        for i in range(3):
            O.augment(method, x, labels, frames, wav, i)
        probe[nt] = (time.perf_counter() - t0) / 3
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    n, t0 = 0, time.perf_counter()
    while True:
        O.augment(method, x, labels, frames, wav, n + 1)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 2000:
            break
    return {"value": B * n / dt, "unit": "samples/s", "cores": best, "kind": "port",
            "ms_per_batch": 1e3 * dt / n, "host_cpus": ncpu,
            "thread_probe_ms": {str(k): 1e3 * v for k, v in probe.items()},
            "sample": f"{n} batches of {method} ({B},{C},{T}) through oracle/pcgmix_oracle.py "
                      f"(reference structure: per-sample loop of torch CPU slice ops) in {dt:.1f} s"}


class TrainArgs:
    def __init__(self, method, model, B, C, T, steps):
        self.dataset, self.model, self.method = "PhysioNet", model, method
        self.num_epochs, self.batch_size, self.op, self.use_sched = 50, B, "adam", True
        self.lr_max, self.weight_decay, self.grad_clip, self.seed = 0.01, 1e-4, 0.1, 4
        self.num_classes, self.num_channels, self.sig_len, self.depth = 2, C, T, 0
        self.num_steps, self.sample_rate = steps, 2000


PROGRESS = {"leg": "", "step": -1}     # what the watchdog reports when a collective hangs


def build_train_step(method, model_name, B, C, T, rate, device, total_steps, rank, use_graph=True):
    """Model, optimiser, (captured) step function for the train legs.  Separate from the timed
    loop so that every rank can report whether its capture worked BEFORE anyone enters a
    gradient all-reduce (see ``agree``)."""
    args = TrainArgs(method, model_name, B, C, T, total_steps)
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=rate, seed=100 + rank)
    batch = (torch.from_numpy(x).to(device), torch.from_numpy(labels), torch.from_numpy(frames), wav,
             torch.ones(B, dtype=torch.long), torch.arange(B))
    torch.manual_seed(4)
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    graphed = use_graph and device.type == "cuda"
    model = tm.build_model(args).to(device)
    if not graphed:
        model = tm.wrap_distributed(model, device)
    model.train()
    opt, sched = tm.make_optimizer(args, model)
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    if graphed:
        # what train_model() runs: one captured slot; for the saliency-guided methods two, the
        # augmentation of the next batch on a side stream while this batch's graph replays
        # (PCGMIX_BENCH_NO_PIPELINE=1: one slot, for A/B)
        if os.environ.get("PCGMIX_BENCH_NO_PIPELINE") or "(salopt" not in method:
            g = tm.GraphedTrainStep(args, model, opt, sched, crit, device, B, C, T,
                                    sync=tm.FlatGradSync(model, device) if distributed else None)
            step = lambda: g.step(batch, 0, sc)                              # noqa: E731
        else:
            g = tm.PipelinedTrainStep(args, model, opt, sched, crit, device, B, C, T,
                                      sync=tm.FlatGradSync(model, device) if distributed else None)
            step = lambda: g.step(batch, 0, sc, None, next_batch=batch)      # noqa: E731
    else:
        step = lambda: tm.train_step(args, model, batch, device, opt, sched, crit, 0, sc)  # noqa: E731
    slots = getattr(g, "slots", [g]) if graphed else []
    return step, {"model": model_name, "method": method, "batch_per_gpu": B, "shape": [B, C, T],
                  "hipgraph": bool(graphed),
                  # the captured step issued as its recorded library launches instead of a graph
                  # replay (fused Potes step, one rank; train_model.GraphedTrainStep)
                  "direct_launches": bool(slots) and all(s_.tape is not None for s_ in slots),
                  "pipelined": bool(graphed) and "(salopt" in method
                  and not os.environ.get("PCGMIX_BENCH_NO_PIPELINE")}


def run_train_steps(step, info, steps, warmup, barrier, tag):
    PROGRESS["leg"] = tag
    settle_heap()                                   # before the warm-up (see run_augment_steps)
    settle_clocks(torch.device("cuda", torch.cuda.current_device()))
    for i in range(warmup):
        PROGRESS["step"] = i - warmup
        step()
    barrier()
    torch.cuda.synchronize()
    stamps, n = STAMPS, 0
    t0 = time.perf_counter()
    for i in range(steps):
        PROGRESS["step"] = i
        loss = step()
        if n < len(stamps):
            stamps[n] = time.perf_counter()
            n += 1
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    gc.unfreeze()
    PROGRESS["leg"] = ""
    return dict(info, steps_per_s=steps / dt, ms_per_step=1e3 * dt / steps, loss=float(loss),
                call_trace=call_trace(t0, stamps, n, dt, steps))


def train_steps_per_s(method, model_name, B, C, T, rate, device, steps, warmup, barrier, rank,
                      use_graph=True, agree=None, tag="train"):
    """Full training step (train_model.py:498-582): augment + forward + soft CE + backward +
    clip + Adam + OneCycleLR, batch resident in HBM.  world > 1: one gradient all-reduce per step
    (FlatGradSync around the hipGraph; DDP for the eager step).

    ``agree(ok) -> bool`` (N>1): logical AND over ranks.  A rank whose graph capture fails must
    not fall back to the eager step alone while the others wait in the captured step's
    all-reduce, so the decision is taken together, after the capture and before the first step."""
    err = None
    try:
        step, info = build_train_step(method, model_name, B, C, T, rate, device, steps + warmup + 1,
                                      rank, use_graph)
    except Exception as e:          # noqa: BLE001
        if not use_graph:
            raise
        err, step = e, None
        print(f"[bench] rank {rank}: graphed {tag} step could not be built: {e!r}", file=sys.stderr)
    ok = agree(err is None) if agree is not None else err is None
    if not ok:
        step, info = build_train_step(method, model_name, B, C, T, rate, device, steps + warmup + 1,
                                      rank, use_graph=False)
        info["graph_error"] = repr(err)[:300] if err is not None else "another rank failed to capture"
    return run_train_steps(step, info, steps, warmup, barrier, tag)


def potes_kernel_times(device, B=256, T=5000, iters=100):
    """Fused Potes conv stack alone: forward and weight-gradient backward, us per launch, and
    the HBM-roofline fraction for their algorithmic bytes (4*T + 16*P2 per band row, each)."""
    from pcgmix_amd import models
    N = 4 * B
    m = models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=T).to(device)
    c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
    params = [c1.weight, c1.bias, c2.weight, c2.bias]
    x = torch.randn(N, T, device=device)
    out = {}
    h = models.PotesStackFunction.apply(x, *params)
    g = torch.randn_like(h)

    def timeit(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    with torch.no_grad():
        out["fwd_us"] = timeit(lambda: models.PotesStackFunction.apply(x, *params))

    def bwd():
        hh = models.PotesStackFunction.apply(x, *params)
        torch.autograd.grad((hh * g).sum(), params)
    out["fwd_bwd_us"] = timeit(bwd)
    nbytes = N * (4 * T + 16 * h.shape[-1])
    out["alg_bytes_each"] = nbytes
    out["fwd_GBs"] = nbytes / out["fwd_us"] / 1e3
    return out


def secondary_kernel_times(device, B=256, iters=50):
    """The other kernels of the path at bs=256, us per launch and fraction of the HBM roofline for
    their algorithmic bytes: log-mel 4T+4*128*128, saliency post 4CT+4T,
    displacement scan 8T, 2D splice 12*F*W per sample."""
    from pcgmix_amd import augmentations2d, frontend, saliency
    out = {}

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3

    def entry(us, nbytes):
        return {"us": us, "GBs": nbytes / us / 1e3, "frac_of_8TBs": nbytes / us / 1e3 / HBM_PEAK_GBS}

    T, C = 5000, 4
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=0)
    data = torch.from_numpy(x).to(device)
    fr = torch.from_numpy(frames.astype(np.int32)).to(device)
    x1 = data[:, 0, :].contiguous()
    out["logmel_256x5000"] = entry(timeit(lambda: frontend.logmel(x1, frames)), B * (4 * T + 4 * 128 * 128))
    # The log-mel kernel is bound by its float64 STFT GEMM, not by memory: folded real transform,
    # two products (re, im) of M = 72 bins (padded), K = n_fft/2 = 68, N = 160 frames per item
    #  -> 2 * 2*72*68*160 = 3.13 MFLOP per cycle on v_mfma_f64_16x16x4_f64.
    lm_flop = B * 2 * (2.0 * 72 * 68 * 160)
    lm = out["logmel_256x5000"]
    lm["roofline"] = {"bound": "mfma_f64", "achieved": lm_flop / lm["us"] / 1e6, "unit": "TFLOP/s",
                      "peak": 78.6, "frac": lm_flop / lm["us"] / 1e6 / 78.6,
                      "measured_issue_rate_peak": 64.0,
                      "frac_of_measured_rate": lm_flop / lm["us"] / 1e6 / 64.0,
                      "flop_per_launch": lm_flop,
                      "executed_flop_per_launch": B * 2 * (2.0 * 64 * 36 * 160),
                      "note": "peak = data-sheet FP64 matrix rate; the instruction's measured issue rate on "
                              "this part is 64 TFLOP/s with four waves per SIMD feeding it (profiles/probes/f64_valu_rate.hip; "
                              "32 with one dependent chain per wave, profiles/probes/mfma_f64_rate.hip).  flop_per_launch "
                              "keeps rounds 1-3's definition (real-input fold, K = 68); since round 4 the "
                              "kernel EXECUTES executed_flop_per_launch: even and odd bins as separate "
                              "products with K = 36 (second fold, DESIGN.md §3.4) + 5 bins on the f64 VALU"}
    grad = torch.randn_like(data)
    out["saliency_post_256x4x5000"] = entry(timeit(lambda: saliency.saliency_post(grad, fr.data_ptr())),
                                            B * (4 * C * T + 4 * T))
    sal = saliency.saliency_post(grad, fr.data_ptr())
    mix = torch.from_numpy(np.random.RandomState(0).permutation(B).astype(np.int32)).to(device)
    for mode, name in ((0, "env"), (1, "sum")):
        out[f"salopt_disp_{name}_256x5000"] = entry(
            timeit(lambda: saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, mode, B, T,
                                                          max_len=int(np.diff(frames, axis=1).max()))),
            B * 8 * T)
        # the same search with the host copies of the boundaries and partners handed over (the
        # reference's situation: CPU arrays): blocks with candidates only, longest chain first
        mix_np = np.random.RandomState(0).permutation(B).astype(np.int32)
        fr_np = frames.astype(np.int32)
        ml = int(np.diff(frames, axis=1).max())
        out[f"salopt_disp_{name}_hosted_256x5000"] = entry(
            timeit(lambda: saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(), 0.37, mode, B, T,
                                                          max_len=ml, frames_host=fr_np, mix_host=mix_np)),
            B * 8 * T)
    spec, fs = frontend.logmel(x1, frames)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
    sc = StepCounter()
    a2 = Args("durratiomixup")

    def mix2d():
        augmentations2d.augment(a2, spec, tgt, fs, wav, sc, None, device, "", host_labels=labels)
        sc.add()
    out["augment2d_256x1x128x128"] = entry(timeit(mix2d), B * 12 * 128 * 128)
    return out


def cfg3_salopt(device, steps=100, warmup=10, B=256, C=4, T=5000, reps=3):
    """BASELINE.json configs[2]: (saloptenv)durmixmagwarp(0.2,4) — saliency from a frozen copy of
    the 1D-CNN (one fwd+bwd through torch), then the three HIP kernels (saliency post-processing,
    displacement search, splice+warp) with no host round trip in between."""
    from pcgmix_amd import models, saliency
    method = "(saloptenv)durmixmagwarp(0.2,4)"
    _, data, tgt, frames, labels, wav = make_device_batch(B, C, T, 2000, 7, device)
    torch.manual_seed(4)
    saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=T).to(device))
    try:    # a few repeats: the first one carries one-time costs (graph capture, pinned buffers)
        dts = [run_augment_steps(method, data, tgt, frames, wav, device, steps, warmup, lambda: None)[0]
               for _ in range(reps)]
    finally:
        saliency.set_saliency_model(None)
    dt = sorted(dts)[len(dts) // 2]                       # median repeat
    return {"method": method, "shape": [B, C, T], "samples_per_s": B * steps / dt,
            "ms_per_step": 1e3 * dt / steps, "ms_per_step_repeats": [1e3 * d / steps for d in dts],
            "steps": steps, "saliency_model": "CNN_potes (frozen copy)"}


def cfg3_train(device, steps, warmup, barrier, rank, B=256, C=4, T=5000):
    """BASELINE.json configs[2] as a TRAINING step: (saloptenv)durmixmagwarp(0.2,4) + the 1D-CNN,
    captured (GraphedTrainStep: seed/boundaries/payload kernel -> the frozen saliency model's
    captured pass -> search + splice into the graph's static input -> replay) and eager.  The
    saliency model is a frozen random-init CNN_potes (reference: the 'base' run's model.pth)."""
    from pcgmix_amd import models, saliency
    method = "(saloptenv)durmixmagwarp(0.2,4)"
    torch.manual_seed(5)
    saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=T).to(device))
    try:
        tr = train_steps_per_s(method, "Potes", B, C, T, 2000, device, steps, warmup, barrier, rank,
                               tag="train_cfg3")
        eager = train_steps_per_s(method, "Potes", B, C, T, 2000, device, min(steps, 200), warmup,
                                  barrier, rank, use_graph=False, tag="train_cfg3_eager")
        tr["eager_steps_per_s"] = eager["steps_per_s"]
    finally:
        saliency.set_saliency_model(None)
    return tr


def cfg4_spectrogram(device, steps=12, warmup=4, B=256, T=5000):
    """BASELINE.json configs[3]: waveform (256,1,5000) -> HIP log-mel (256,1,128,128) -> 2D
    durratiomixup on the spectrogram columns -> ResNet9-2D train step (MIOpen convolutions)."""
    from pcgmix_amd import augmentations2d, frontend
    x, frames, labels, wav = synthetic.make_batch(B, 1, T, sample_rate=2000, seed=3)
    wave = torch.from_numpy(x).to(device)
    args = TrainArgs("durratiomixup", "resnet9", B, 1, T, steps + warmup + 1)
    args.dataset = "PhysioNet(spec128)"
    torch.manual_seed(4)
    model = tm.build_model(args).to(device).train()
    opt, sched = tm.make_optimizer(args, model)
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    target = torch.from_numpy(labels)

    def step():
        spec, fspec = frontend.logmel(wave, frames)
        batch = (spec, target, torch.from_numpy(fspec), wav, None, torch.arange(B))
        return tm.train_step(args, model, batch, device, opt, sched, crit, 0, sc)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # the front end + 2D splice alone
    a2, sc2 = Args("durratiomixup"), StepCounter()
    tgt = torch.nn.functional.one_hot(target, 2).to(device)
    for _ in range(3):
        spec, fspec = frontend.logmel(wave, frames)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(50):
        spec, fspec = frontend.logmel(wave, frames)
        augmentations2d.augment(a2, spec, tgt, fspec, wav, sc2, None, device, "", host_labels=labels)
        sc2.add()
    torch.cuda.synchronize()
    dfe = (time.perf_counter() - t1) / 50
    return {"steps_per_s": steps / dt, "ms_per_step": 1e3 * dt / steps, "loss": float(loss),
            "model": "ResNet9-2D", "shape": [B, 1, 128, 128],
            "frontend_plus_mix_ms": 1e3 * dfe, "frontend_plus_mix_samples_per_s": B / dfe}


def measured_traffic(method, B, C, T):
    """HBM-side bytes per launch of the splice kernel as rocprofv3's PMC passes counted them
    (profiles/r*_mix_roofline.json: FETCH_SIZE doubled per the guide's gfx950 correction +
    WRITE_SIZE, separate --pmc passes, collected by profiles/mix_pmc_probe.py) — read from the
    committed file, NOT measured in this run, and only for exactly this workload; else None."""
    import glob
    key = f"{method} ({B},{C},{T})"
    if karg_unroll(B, C, T, "magwarp" in method):
        key = f"{method} [kernarg] ({B},{C},{T})"           # the instantiation that runs at this size
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_mix_roofline.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        w = d.get("workloads", {}).get(key)
        if w and w.get("hbm_bytes_per_launch"):
            return w["hbm_bytes_per_launch"], f"file profiles/{os.path.basename(path)} " \
                                              f"(collected at {d.get('collected_at', '?')})"
    return None, None


def armed_step_stats(device):
    """[armed steps, checked after a stream synchronisation, relaunched] of this device's step context."""
    import ctypes
    from pcgmix_amd import _lib
    out = (ctypes.c_longlong * 3)()
    if _lib.load().pcgmix_ctx_armed_stats(augmentations.step_context(device.index), out):
        return None
    return list(out)


def roofline_entry(method, B, C, T, kern_ms, info):
    """Roofline object of one splice launch.  ``achieved``/``frac`` use the bytes THIS batch needs
    (own read + write + partner read inside blended ranges, ``exact_mix_bytes``); the contract's
    12*C*T model, which counts the whole partner row, is reported beside it."""
    exact = info["exact_bytes"]
    model = 12.0 * C * T * B
    achieved = exact / (kern_ms * 1e-3) / 1e9
    traffic, src = measured_traffic(method, B, C, T)
    working_set = 8.0 * B * C * T
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src,
           "kernel": info["kernel"], "kernel_ms": kern_ms,
           "algorithmic_bytes_per_launch": exact,
           "bytes_definition": "4*C*(2*B*T + sum_b sum_k min(len_k(b), len_k(partner b))): own read "
                               "+ write + partner read inside blended ranges, this batch",
           "contract_model_12CT": {"bytes_per_launch": model,
                                   "achieved": model / (kern_ms * 1e-3) / 1e9,
                                   "frac": model / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
           "working_set_bytes": working_set,
           "residency": ("input + output fit the 256 MiB Infinity Cache: back-to-back launches are "
                         "served on-die, this is NOT an HBM-rate measurement (see the saturating "
                         "legs in extra)") if working_set < 256 * 2**20 else
                        "input + output exceed the 256 MiB Infinity Cache: HBM-rate measurement",
           "achievable_copy_GBs": 6290.0}
    if traffic:
        out["frac_on_counter_bytes"] = traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out


def launch_ranks(n, argv):
    """Start ``python -m torch.distributed.run --nproc-per-node n bench.py <argv>`` as a child,
    stream its stdout/stderr through, return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 1000 steps = 24 ms of the headline loop / 0.16 s of the train leg: a 200-step region (5 ms)
    # still carried its start-up in the average (23.3-26 us where 100k steps run at 22.0)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--sig-len", type=int, default=5000)
    ap.add_argument("--method", default="durratiomixup")
    ap.add_argument("--no-extra", action="store_true", help="skip secondary measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-train", action="store_true", help="skip the train-step leg")
    ap.add_argument("--kernels-only", action="store_true",
                    help="only the back-to-back kernel table (diagnostic)")
    ap.add_argument("--profile-host", action="store_true",
                    help="cProfile the host side of the timed steps to stderr (diagnostic)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher.  Nothing in this process has
        # touched the GPU yet (imports only), and it never will: the N ranks are CHILDREN
        # (torch.distributed.run, one process per GPU), rank 0's JSON line is relayed and the
        # child's exit code is ours.  No exec of a GPU-initialised process anywhere.
        raise SystemExit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one process per GPU; the modulo only matters for rehearsals with more ranks than GPUs
    # (PCGMIX_DIST_BACKEND=gloo on a 1-GPU box) — the driver's runs have one GPU per rank
    local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # Host placement: this process (and the library's draw-ahead threads, created later) on a few CPUs
    # of the NUMA node the GPU hangs off — what a launcher's numactl would do; disclosed in
    # config.host_affinity.  PCGMIX_BENCH_NO_AFFINITY=1 leaves placement to the scheduler.
    global HOST_AFFINITY, ORIGINAL_CPUS
    ORIGINAL_CPUS = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    HOST_AFFINITY = ("left to the scheduler (PCGMIX_BENCH_NO_AFFINITY)" if os.environ.get("PCGMIX_BENCH_NO_AFFINITY")
                     else hostprep.bind_host_threads(local, int(os.environ.get("LOCAL_RANK", "0"))))
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("PCGMIX_DIST_BACKEND", "nccl")     # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    B, C, T, rate = a.batch, a.channels, a.sig_len, 2000
    if a.kernels_only:
        print("potes:", potes_kernel_times(device), flush=True)
        for k, v in secondary_kernel_times(device).items():
            print(f"{k:32s} {v['us']:9.1f} us  {v['GBs']:8.1f} GB/s", flush=True)
        for m, b, c, t in (("durratiomixup", 256, 1, 5000), ("durratiomixup", 256, 4, 5000),
                           ("durmixmagwarp(0.2,4)", 256, 1, 5000), ("durmixmagwarp(0.2,4)", 256, 4, 5000),
                           ("durratiomixup", 4096, 4, 5000), ("durmixmagwarp(0.2,4)", 4096, 4, 5000),
                           ("durratiomixup", 16384, 4, 5000), ("durmixmagwarp(0.2,4)", 16384, 4, 5000)):
            ms = kernel_back_to_back_ms(m, b, c, t, rate, device, iters=50 if b > 1000 else 200)
            print(f"{m:24s} ({b},{c},{t})  {ms * 1e3:9.2f} us  {12.0 * b * c * t / ms / 1e6:8.1f} GB/s",
                  flush=True)
        return
    _, data, tgt, frames, labels, wav = make_device_batch(B, C, T, rate, seed=rank, device=device)

    if a.profile_host:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        run_augment_steps(a.method, data, tgt, frames, wav, device, 5, 5, barrier)
        pr.enable()
        run_augment_steps(a.method, data, tgt, frames, wav, device, a.steps, 0, barrier)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumtime").print_stats(25)
    # the timed region carries nothing but the drop-in calls (no event recording inside it)
    headline_trace = {}
    dt, _ = run_augment_steps(a.method, data, tgt, frames, wav, device, a.steps, a.warmup, barrier,
                              trace=headline_trace)
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    value = world * B * a.steps / dt

    # Kernel time for the roofline: one HIP event pair on the launch stream around 200 launches
    # of the same kernel on this workload, issued back to back so the GPU never waits for the
    # host, divided by 200.  Launches of one stream run in order, so this average is the kernel's
    # duration plus the dispatch gap; rocprofv3's per-dispatch mean agrees with it within 5 %
    # (profiles/).  Also reported, not used: the median of event pairs recorded around every
    # single launch (each pair adds 2-6 us of marker overhead, different from box to box).
    kinfo = {}
    per_launch_pair_ms, kern_ms = kernel_back_to_back_ms(a.method, B, C, T, rate, device,
                                                         per_launch=True, info=kinfo)
    roof = roofline_entry(a.method, B, C, T, kern_ms, kinfo)
    roof["per_launch_event_pair_ms"] = per_launch_pair_ms
    roof["timing"] = ("HIP events on the launch stream around 200 back-to-back launches "
                      "(in-order stream, queue kept full), divided by 200")
    armed = armed_step_stats(device)
    if armed and armed[0] > 0:
        roof["step_kernel"] = {
            "name": "pcgmix::mix_armed_kernel<%d>" % max(1, karg_unroll(B, C, T, False) or 1),
            "armed_steps": armed[0], "checked_after_sync": armed[1], "relaunched": armed[2],
            "note": "the timed augment() steps launch this instantiation of the same splice body (mix_body) "
                    "BEFORE its index block exists: block (0,0) does the label arg-max, the other blocks wait for "
                    "the records the host writes once it has drawn the partners.  Its duration under rocprofv3 "
                    "therefore contains the host's reaction (label arg-max 1.0 + host 4.4 + relay 1.0 us of ~12, "
                    "profiles/r4_armed_step.txt); `kernel` above is the two-launch instantiation of the same "
                    "body issued back to back, which is what a roofline can be read from"}
    result = {
        "metric": "augmented PCG samples/s", "value": value, "unit": "samples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{a.method} augment() on synthetic 2.5 s @ 2 kHz PCG cycles, "
                               f"({B},{C},{T}) float32 per GPU (BASELINE.json configs[1])",
                   "batch_per_gpu": B, "channels": C, "sig_len": T, "method": a.method,
                   "parallelism": f"dp{world}",
                   "host_affinity": HOST_AFFINITY,
                   "pre_settle": "before the W warm-up steps of every leg: heap collected + frozen, then 40 ms of "
                                 "neutral GPU work (GEMM + streaming add) to reach loaded clocks, see "
                                 "bench.settle_clocks"},
        "roofline": roof,
        "call_trace": headline_trace,
    }

    # train step/s (second half of BASELINE.json's metric): 1D-CNN, bs 256 per GPU; N>1: one
    # gradient all-reduce per step.  N>1 also runs BASELINE.json configs[4]'s per-rank workload
    # (durmixmagwarp(0.2,4) + ResNet9-1D, bs 256 per GPU -> global 2048 at N=8).
    guard = None
    if not a.no_train and world > 1:
        # The N>1 train legs are the one part of this file that cannot be rehearsed with RCCL on
        # a one-GPU box.  If a collective hangs (a rank died, mismatched calls), nobody may wait
        # for ever and nobody may report success: after 240 s every rank says where it was on
        # stderr, rank 0 still prints the line it has (marked with the error) and the process
        # exits NON-ZERO without waiting for the stuck call.
        import threading

        def give_up():
            print(f"[bench] rank {rank}: train leg {PROGRESS['leg']!r} stuck at step "
                  f"{PROGRESS['step']} (collective pending?) after 240 s; giving up",
                  file=sys.stderr, flush=True)
            if rank == 0:
                result["error"] = (f"train leg {PROGRESS['leg']!r} did not finish within 240 s "
                                   f"(step {PROGRESS['step']}); process exits with code 3")
                print(strict_json(result), flush=True)
            os._exit(3)
        guard = threading.Timer(240.0, give_up)
        guard.daemon = True
        guard.start()

    def agree(ok):
        if dist is None:
            return ok
        t = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def max_over_ranks(tr):
        if dist is not None:
            t = torch.tensor([tr["ms_per_step"]], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            tr["ms_per_step"] = float(t.item())
            tr["steps_per_s"] = 1e3 / tr["ms_per_step"]
        tr["global_batch"] = tr["batch_per_gpu"] * world
        tr["samples_per_s"] = tr["steps_per_s"] * tr["global_batch"]
        return tr

    if not a.no_train:
        # as many steps as the headline leg: a 50-step region (9 ms) carried ~0.4 ms of
        # start-up and drain, i.e. 3-4 % on the step time
        n_tr, w_tr = max(20, a.steps), max(5, a.warmup)
        # An exception here is the same code failing on every rank (a rank-local capture failure
        # is settled inside train_steps_per_s through ``agree``): record it, keep the headline.
        try:
            train = train_steps_per_s(a.method, "Potes", B, C, T, rate, device, n_tr, w_tr,
                                      barrier, rank, agree=agree)
            if world == 1 and "graph_error" not in train:
                eager = train_steps_per_s(a.method, "Potes", B, C, T, rate, device, n_tr, w_tr,
                                          barrier, rank, use_graph=False)
                train["eager_steps_per_s"] = eager["steps_per_s"]
            result["train"] = max_over_ranks(train)
        except Exception as e:          # noqa: BLE001
            print(f"[bench] train leg failed on rank {rank}: {e!r}", file=sys.stderr)
            result["train"] = {"error": repr(e)[:300]}
        if world > 1:
            try:
                cfg5 = train_steps_per_s("durmixmagwarp(0.2,4)", "resnet9", 256, 4, 5000, rate,
                                         device, 10, 3, barrier, rank, agree=agree, tag="train_cfg5")
                cfg5["config"] = "BASELINE.json configs[4]: durmixmagwarp(0.2,4) + ResNet9-1D, " \
                                 f"bs 256 per GPU (global {256 * world}), one gradient all-reduce per step"
                result["train_cfg5"] = max_over_ranks(cfg5)
            except Exception as e:      # noqa: BLE001
                print(f"[bench] cfg5 leg failed on rank {rank}: {e!r}", file=sys.stderr)
                result["train_cfg5"] = {"error": repr(e)[:300]}
        if guard is not None:
            guard.cancel()

    if rank == 0 and world == 1:
        extra = {}

        def leg(tag, fn):
            """Secondary measurements must never cost the headline line: a failing leg is
            recorded as an error string instead of propagating."""
            try:
                extra[tag] = fn()
            except Exception as e:          # noqa: BLE001
                extra[tag] = {"error": repr(e)[:300]}
                print(f"[bench] extra leg {tag} failed: {e!r}", file=sys.stderr)

        if not a.no_extra:
            def kernel_leg(m, b, c, t):
                ki = {}
                ms = kernel_back_to_back_ms(m, b, c, t, rate, device, iters=50 if b > 1000 else 200,
                                            info=ki)
                r = roofline_entry(m, b, c, t, ms, ki)
                return {"kernel_ms": ms, "kernel": r["kernel"], "GBs": r["achieved"],
                        "frac_of_8TBs": r["frac"], "bytes_per_launch": r["algorithmic_bytes_per_launch"],
                        "GBs_12CT_model": r["contract_model_12CT"]["achieved"],
                        "frac_12CT_model": r["contract_model_12CT"]["frac"],
                        "traffic": r["traffic"], "traffic_source": r["traffic_source"],
                        "frac_on_counter_bytes": r.get("frac_on_counter_bytes"),
                        "infinity_cache_resident": r["working_set_bytes"] < 256 * 2**20}
            for tag, (m, b, c, t) in {
                "mix_256x1x5000": ("durratiomixup", 256, 1, 5000),
                "magwarp_256x1x5000": ("durmixmagwarp(0.2,4)", 256, 1, 5000),
                "magwarp_256x4x5000": ("durmixmagwarp(0.2,4)", 256, 4, 5000),
                "mix_sat_16384x4x5000": ("durratiomixup", 16384, 4, 5000),
                "magwarp_sat_16384x4x5000": ("durmixmagwarp(0.2,4)", 16384, 4, 5000),
            }.items():
                leg(tag, lambda m=m, b=b, c=c, t=t: kernel_leg(m, b, c, t))
            _, d1, t1, f1, _, w1 = make_device_batch(256, 1, 5000, rate, 0, device)

            def augment_leg(m, dd, tt, ff, ww):
                tr = {}
                dte, _ = run_augment_steps(m, dd, tt, ff, ww, device, a.steps, a.warmup, barrier, trace=tr)
                return {"samples_per_s": 256 * a.steps / dte, "ms_per_step": 1e3 * dte / a.steps,
                        "call_trace": tr}
            for tag, m, (dd, tt, ff, ww) in (
                ("augment_mix_256x1x5000", "durratiomixup", (d1, t1, f1, w1)),
                ("augment_magwarp_256x1x5000", "durmixmagwarp(0.2,4)", (d1, t1, f1, w1)),
                ("augment_magwarp_256x4x5000", "durmixmagwarp(0.2,4)", (data, tgt, frames, wav)),
            ):
                leg(tag, lambda m=m, dd=dd, tt=tt, ff=ff, ww=ww: augment_leg(m, dd, tt, ff, ww))
            def host_split():
                """Where a strict-signature step goes: the same loop with the labels handed over
                on the host (no read-back, no host wait), the read-back on its own (D2H of the
                one-hot matrix + stream sync through torch, for scale), and the kernel."""
                n = max(a.steps, 200)
                d_strict, _ = run_augment_steps(a.method, data, tgt, frames, wav, device, n, 20, barrier)
                d_host, _ = run_augment_steps(a.method, data, tgt, frames, wav, device, n, 20, barrier,
                                              host_labels=labels)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    augmentations.labels_from_ohe(tgt)
                d_rb = time.perf_counter() - t0
                return {"strict_signature_us": 1e6 * d_strict / n,
                        "host_labels_no_readback_us": 1e6 * d_host / n,
                        "readback_inside_step_us": 1e6 * (d_strict - d_host) / n,
                        "torch_readback_alone_us": 1e6 * d_rb / n,
                        "kernel_us": 1e3 * kern_ms,
                        "note": "host_labels_no_readback is pure host + launch cost (the GPU keeps "
                                "up: kernel < host); strict adds the label read-back the "
                                "reference's signature forces (augmentations.py:501)"}
            leg("host_split", host_split)
            leg("potes_stack", lambda: potes_kernel_times(device))
            leg("secondary_kernels", lambda: secondary_kernel_times(device))
            leg("cfg3_salopt", lambda: cfg3_salopt(device))
            if not a.no_train:
                # PCGmix+ (the paper's durmixmagwarp) on the 1D-CNN: the same captured step with the
                # warp fused into the splice; its knots are drawn ahead by the library
                leg("train_potes_magwarp", lambda: max_over_ranks(train_steps_per_s(
                    "durmixmagwarp(0.2,4)", "Potes", B, C, T, rate, device, max(20, a.steps),
                    max(5, a.warmup), barrier, rank, tag="train_potes_magwarp")))
                leg("cfg3_train", lambda: cfg3_train(device, max(20, a.steps), max(5, a.warmup), barrier,
                                                     rank))
                leg("cfg4_spectrogram", lambda: cfg4_spectrogram(device))
                leg("train_resnet9_1d_magwarp", lambda: train_steps_per_s(
                    "durmixmagwarp(0.2,4)", "resnet9", 256, 4, 5000, rate, device, 20, 5, barrier,
                    rank))
        result["extra"] = extra
        if not a.no_cpu:
            try:
                result["cpu_baseline"] = cpu_baseline(a.method, B, C, T, rate)
                result["extra"]["gpu_over_cpu"] = value / result["cpu_baseline"]["value"]
                if not a.no_extra:
                    # the reference's CPU path for PCGmix+: B*C scipy CubicSpline constructions per
                    # batch (augmentations.py:674-683), next to extra.augment_magwarp_*
                    cb = cpu_baseline("durmixmagwarp(0.2,4)", B, C, T, rate, budget_s=8.0)
                    result["extra"]["cpu_baseline_magwarp"] = cb
                    aug = result["extra"].get(f"augment_magwarp_{B}x{C}x{T}", {})
                    if aug.get("samples_per_s"):
                        result["extra"]["gpu_over_cpu_magwarp"] = aug["samples_per_s"] / cb["value"]
            except Exception as e:      # noqa: BLE001
                print(f"[bench] cpu_baseline failed: {e!r}", file=sys.stderr)
                result["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": 0,
                                          "kind": "port", "sample": "failed: " + repr(e)[:200]}
    if rank == 0:
        print(strict_json(result))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
