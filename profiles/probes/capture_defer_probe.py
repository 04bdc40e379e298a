#!/usr/bin/env python3
"""Root-cause probe for round 2's segfault in ``CUDAGraph.capture_end`` (gpurun_out/r2_t30.log,
r2_t31.log: ``test_head_loss_finalize_deferred_under_capture``, deterministic, also as the first
test of a fresh process).  ``capture_probe.py`` could not reproduce it because it never passed
``defer=True``.  This probe restates the crashed body — eager forward(False) + autograd.grad, then
the deferred form captured DIRECTLY with ``torch.cuda.graph`` (default capture_error_mode, no
side-stream warm-up, ``torch.autograd.grad`` inside the capture) — and isolates one difference
per variant.  Every variant runs ONCE in its own child process with a native-stack SIGSEGV
handler (segv_bt.c), so a crash is evidence (which library, which call) and not a lost run.

    python profiles/probes/capture_defer_probe.py [variant ...] > gpurun_out/capture_defer_probe.txt
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))

CODE = r'''
import ctypes, os, sys
so = os.path.join(sys.argv[2], "segv_bt.so")
ctypes.CDLL(so).segv_bt_install()
import torch, torch.nn.functional as F
sys.path.insert(0, sys.argv[3])
import pcgmix_amd
from pcgmix_amd import models
variant = sys.argv[1]
flags = set(variant.split("+"))
device = torch.device("cuda", 0)
torch.manual_seed(8)
if "puretorch" in flags:
    # no pcgmix kernel at all: the same shape of program with torch ops only
    w = torch.randn(4096, device=device, requires_grad=True)
    x = torch.randn(4096, device=device)
    y0 = (w * x).sum()
    g0 = torch.autograd.grad(y0, w)              # y0 stays alive: so does its AccumulateGrad(w)
    if "fresh" in flags:
        y0, g0 = y0.detach().clone(), [g.clone() for g in g0]
    torch.cuda.synchronize()
    print("eager ok", flush=True)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        y1 = (w * x).sum()
        g1 = torch.autograd.grad(y1, w)
        print("body done", flush=True)
    print("captured", flush=True)
    graph.replay()
    torch.cuda.synchronize()
    print("replayed; equal:", torch.equal(y1.detach(), y0.detach()) and torch.equal(g1[0], g0[0]), flush=True)
    sys.exit(0)
B, K, C = 100, 19968, 2
w1 = (torch.randn(20, K, device=device) * 0.01).requires_grad_(True)
b1 = torch.randn(20, device=device).requires_grad_(True)
w2 = torch.randn(C, 20, device=device).requires_grad_(True)
b2 = torch.randn(C, device=device).requires_grad_(True)
feat = torch.randn(B, K, device=device).requires_grad_(True)
t = F.one_hot(torch.randint(0, C, (B,), device=device), C).float()
gs = torch.tensor(0.5, device=device)
params = (feat, w1, b1, w2, b2)

def forward(defer):
    return models.PotesHeadLossFunction.apply(feat, w1, b1, w2, b2, t, 0.0, 0.0, True, None, defer)

def grads_of(loss):
    if "backward" in flags:
        for p in params:
            p.grad = None
        loss.backward(gs)
        return [p.grad for p in params]
    return list(torch.autograd.grad(loss, params, gs))

loss0, logits0 = forward(False)
grads0 = grads_of(loss0)
if "backward" in flags:
    grads0 = [g.clone() for g in grads0]
if "fresh" in flags:                      # drop the eager pass's autograd graph before capturing
    loss0, logits0 = loss0.detach().clone(), logits0.detach().clone()
    grads0 = [g.detach().clone() for g in grads0]
    for p in params:
        p.grad = None
    import gc
    gc.collect()
if "warm" in flags:                       # side-stream warm-up of the deferred form first
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        l, _ = forward(True)
        grads_of(l)
    torch.cuda.current_stream(device).wait_stream(side)
torch.cuda.synchronize()
print("eager ok", flush=True)
graph = torch.cuda.CUDAGraph()
mode = "thread_local" if "tl" in flags else ("relaxed" if "relaxed" in flags else "global")
defer = "nodefer" not in flags
try:
    with torch.cuda.graph(graph, capture_error_mode=mode):
        if "raise" in flags:              # hypothesis: an exception inside the block, then capture_end
            l, _ = forward(False)
            raise KeyError("deliberate")
        loss1, logits1 = forward(defer)
        if "fill" in flags:
            loss1.detach().fill_(-1.0)
        grads1 = grads_of(loss1)
        print("body done", flush=True)
except KeyError as e:
    print("exception left the capture block cleanly:", repr(e), flush=True)
    sys.exit(0)
print("captured", flush=True)
graph.replay()
torch.cuda.synchronize()
ok = torch.equal(loss1.detach(), loss0.detach()) and torch.equal(logits1, logits0) and \
    all(torch.equal(a, b) for a, b in zip(grads1, grads0))
print("replayed; bit-identical to the eager two-launch form:", ok, flush=True)
'''

# Round-3 run 1 (gpurun_out/r3_capture_defer_probe.txt): "fill", "fill+tl", "fill+warm",
# "fill+backward" and "nodefer+fill" all die in hip::Stream::EndCapture (null `this`: a parallel
# capture stream that is the legacy default stream), "raise" leaves cleanly -> neither `defer`, nor
# the capture mode, nor a missing warm-up, nor an exception in the block is the cause.  What they
# share is the eager pass's autograd graph still alive (loss0) while the same leaves are used
# under capture.  Run 2 isolates that: "fresh" = the same body with the stale graph dropped,
# "puretorch" = the stale-graph pattern with torch ops only, "puretorch+fresh" its control.
VARIANTS = ("fill+fresh", "fresh", "puretorch", "puretorch+fresh")


def main():
    so = os.path.join(HERE, "segv_bt.so")
    if not os.path.exists(so):
        subprocess.run(["gcc", "-O1", "-g", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "segv_bt.c")],
                       check=True)
    for v in (sys.argv[1:] or VARIANTS):
        r = subprocess.run([sys.executable, "-c", CODE, v, HERE, ROOT], capture_output=True, text=True,
                           timeout=300)
        print(f"--- variant {v!r}: rc {r.returncode}", flush=True)
        print(r.stdout.strip(), flush=True)
        err = [ln for ln in r.stderr.splitlines() if "amdgpu.ids" not in ln]
        print("\n".join(err[-70:]), flush=True)


if __name__ == "__main__":
    main()
