"""Fused Potes classifier head and soft-target cross entropy (HIP) against the same maths in torch
float32 (floating-point kernels: torch fp32 is the reference; tolerances stated per assert)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib, models, train_model as tm

pytestmark = pytest.mark.gpu


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


@pytest.mark.parametrize("B,K,C,drop", [(256, 19968, 2, True), (32, 9968, 2, True), (7, 4096, 3, False),
                                        (1, 1028, 2, True), (300, 2052, 8, True), (17, 64, 2, False)])
def test_head_kernels_match_torch(B, K, C, drop, device):
    """C ABI with explicit dropout masks == torch autograd of the same composition."""
    torch.manual_seed(B + K)
    lib = _lib.load()
    feat = torch.randn(B, K, device=device).relu_().requires_grad_(True)
    w1 = (torch.randn(20, K, device=device) / K ** 0.5).requires_grad_(True)
    b1 = torch.randn(20, device=device).requires_grad_(True)
    w2 = (torch.randn(C, 20, device=device) / 4).requires_grad_(True)
    b2 = torch.randn(C, device=device).requires_grad_(True)
    p1, p2 = 0.25, 0.5
    if drop:
        m1 = (torch.rand(B, K, device=device) > p1)
        m2 = (torch.rand(B, 20, device=device) > p2).to(torch.uint8)
        s1, s2 = 1 / (1 - p1), 1 / (1 - p2)
    else:
        m1 = m2 = None
        s1 = s2 = 1.0
    x = feat * m1 * s1 if drop else feat
    z_t = F.linear(x, w1, b1)
    h = z_t.relu() * (m2 * s2 if drop else 1.0)
    logits_t = F.linear(h, w2, b2)
    dl = torch.randn(B, C, device=device)
    logits_t.backward(dl)

    ks = lib.pcgmix_skinny_linear_splits(B, K)
    partial = torch.empty(ks, B, 20, device=device)
    z = torch.empty(B, 20, device=device)
    logits = torch.empty(B, C, device=device)
    xd = feat.detach().contiguous()                  # the kernels take the features BEFORE dropout
    m1b = m1.to(torch.uint8).contiguous() if drop else None      # 0/1 bytes, threshold 1
    thr = 1 if drop else 0
    opt = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    _lib.check(lib.pcgmix_potes_head_fwd_f32(xd.data_ptr(), opt(m1b), ctypes.c_float(s1), thr, 8,
                                             w1.data_ptr(), b1.data_ptr(), opt(m2), ctypes.c_float(s2),
                                             thr, w2.data_ptr(), b2.data_ptr(), partial.data_ptr(),
                                             z.data_ptr(), logits.data_ptr(), B, K, C, _stream(device)),
               "fwd")
    assert torch.allclose(z, z_t, rtol=1e-4, atol=2e-5), float((z - z_t).abs().max())
    assert torch.allclose(logits, logits_t, rtol=1e-4, atol=2e-5)
    dz = torch.empty(B, 20, device=device)
    dw2, db2, db1 = torch.empty(C, 20, device=device), torch.empty(C, device=device), torch.empty(20, device=device)
    dw1, dx = torch.empty(20, K, device=device), torch.empty(B, K, device=device)
    _lib.check(lib.pcgmix_potes_head_bwd_f32(dl.data_ptr(), z.data_ptr(), opt(m2), ctypes.c_float(s2), thr,
                                             w2.data_ptr(), xd.data_ptr(), opt(m1b), ctypes.c_float(s1), thr, 8,
                                             w1.data_ptr(), dz.data_ptr(), dw2.data_ptr(), db2.data_ptr(),
                                             db1.data_ptr(), dw1.data_ptr(), dx.data_ptr(), B, K, C,
                                             _stream(device)), "bwd")
    for got, want, name in ((dw2, w2.grad, "dw2"), (db2, b2.grad, "db2"), (db1, b1.grad, "db1"),
                            (dw1, w1.grad, "dw1"), (dx, feat.grad, "dx")):
        scale = float(want.abs().max()) + 1e-6
        assert float((got - want).abs().max()) <= 2e-5 * max(1.0, scale) + 1e-4 * scale, name


def test_model_fused_head_matches_unfused(device):
    """Whole CNN_potes, dropout off: HIP stack + HIP head == torch modules, logits and all grads."""
    torch.manual_seed(3)
    m = models.CNN_potes_TS(4, 2, "PhysioNet").to(device).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    x = torch.randn(16, 4, 2500, device=device)
    t = F.one_hot(torch.randint(0, 2, (16,), device=device), 2)
    res = []
    for fused in (True, False):
        m.fused = fused
        m.zero_grad(set_to_none=True)
        out = m(x, depth=0, pass_part="second")
        loss = tm.CELoss(2)(out, t)
        loss.backward()
        res.append((out.detach(), loss.detach(),
                    {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-4, atol=1e-5)
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6)
    assert res[0][2].keys() == res[1][2].keys()
    for k in res[0][2]:
        a, b = res[0][2][k], res[1][2][k]
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-6 + 1e-4 * float(b.abs().max())), k


def test_fused_head_dropout_statistics_and_seeding(device):
    """Training mode: masks come from torch's generator — same seed, same logits; the kept
    fraction of the 20-wide hidden layer is ~1-p."""
    torch.manual_seed(0)
    m = models.CNN_potes_TS(4, 2, "PhysioNet").to(device).train()
    x = torch.randn(64, 4, 2500, device=device)
    torch.manual_seed(11)
    a = m(x, depth=0, pass_part="second")
    torch.manual_seed(11)
    b = m(x, depth=0, pass_part="second")
    torch.manual_seed(12)
    c = m(x, depth=0, pass_part="second")
    assert torch.equal(a, b) and not torch.equal(a, c)
    m.eval()
    e1, e2 = m(x, depth=0, pass_part="second"), m(x, depth=0, pass_part="second")
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("B,C", [(256, 2), (1, 2), (1000, 2), (37, 5)])
def test_soft_ce_matches_torch(B, C, device):
    torch.manual_seed(B)
    for soft in (False, True):
        logits = (torch.randn(B, C, device=device) * 3).requires_grad_(True)
        if soft:
            tgt = torch.rand(B, C, device=device)
            tgt = tgt / tgt.sum(1, keepdim=True) * 1.3          # need not sum to 1 (mixAll blends do)
        else:
            tgt = F.one_hot(torch.randint(0, C, (B,), device=device), C)
        want = -(F.log_softmax(logits, dim=1) * tgt).sum(dim=1).mean()
        (want * 0.7).backward()
        gw = logits.grad.clone()
        logits.grad = None
        got = tm.CELoss(C)(logits, tgt)
        assert got.grad_fn is not None and "SoftCE" in type(got.grad_fn).__name__
        (got * 0.7).backward()
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-6)
        assert torch.allclose(logits.grad, gw, rtol=1e-4, atol=1e-7)


def test_head_backward_is_deterministic(device):
    """dW1 is accumulated by two row halves with float atomicAdd onto a zeroed buffer: exactly two
    addends per element, so the result must not depend on arrival order — bit-identical reruns."""
    torch.manual_seed(4)
    m = models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=5000).to(device).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    x = torch.randn(256, 4, 5000, device=device)
    t = F.one_hot(torch.randint(0, 2, (256,), device=device), 2)
    grads = []
    for _ in range(3):
        m.zero_grad(set_to_none=True)
        tm.CELoss(2)(m(x, depth=0, pass_part="second"), t).backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]) and torch.equal(grads[0][k], grads[2][k]), k


@pytest.mark.parametrize("training", [False, True])
def test_head_input_gradient_with_frozen_weights(training, device):
    """Saliency maps differentiate through a FROZEN model (saliency.py:47-61): the head's backward
    then produces dx only (pcgmix_potes_head_bwd_f32 with dw1 = NULL: x is not read, no dW1
    accumulation).  Same dx as with trainable weights, and as torch's."""
    from pcgmix_amd import models
    torch.manual_seed(3)
    B, K = 48, 9968
    lin1, lin2 = torch.nn.Linear(K, 20).to(device), torch.nn.Linear(20, 2).to(device)
    feat = torch.randn(B, K, device=device)
    r = torch.randn(B, 2, device=device)
    outs = []
    for frozen in (True, False):
        for p in list(lin1.parameters()) + list(lin2.parameters()):
            p.requires_grad_(not frozen)
        x = feat.clone().requires_grad_(True)
        torch.manual_seed(11)                                     # same dropout masks
        lo = models.PotesHeadFunction.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias,
                                            0.25, 0.5, training)
        (gx,) = torch.autograd.grad((lo * r).sum(), x)
        outs.append((lo.detach(), gx))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    if not training:
        x = feat.clone().requires_grad_(True)
        ref = lin2(torch.relu(lin1(x)))
        (gt,) = torch.autograd.grad((ref * r).sum(), x)
        assert torch.allclose(outs[0][1], gt, rtol=1e-4, atol=1e-6 * float(gt.abs().max()) + 1e-9)


@pytest.mark.parametrize("p1,p2", [(0.25, 0.5), (0.5, 0.25), (0.1, 0.5), (0.0625, 0.3)])
def test_random_bit_dropout_is_torch_dropout_with_the_same_mask(p1, p2, device):
    """Training mode reads uniformly random bits (an element owns 1, 2, 4 or 8 of them and is kept
    iff their value >= thr): against torch ops given the masks those bits define; the keep
    probabilities are exact for the reference's p = 0.25 (2 bits) and 0.5 (1 bit)."""
    from pcgmix_amd import models
    torch.manual_seed(4)
    B, K = 64, 9968
    lin1, lin2 = torch.nn.Linear(K, 20).to(device), torch.nn.Linear(20, 2).to(device)
    feat = torch.randn(B, K, device=device).relu_()
    rnd = torch.empty(models.head_dropout_bytes(B, K, p1), dtype=torch.uint8, device=device).random_()
    x = feat.clone().requires_grad_(True)
    lo = models.PotesHeadFunction.apply(x, lin1.weight, lin1.bias, lin2.weight, lin2.bias, p1, p2,
                                        True, rnd)
    r = torch.randn(B, 2, device=device)
    g = torch.autograd.grad((lo * r).sum(), [x, lin1.weight, lin2.weight])
    assert models.dropout_threshold(0.25) == (2, 1, 4 / 3) and models.dropout_threshold(0.5) == (1, 1, 2.0)
    bits, thr, sc = models.dropout_threshold(p1)
    e = torch.arange(B * K, device=device)
    val = (rnd[(e * bits) >> 3].to(torch.int32) >> ((e * bits) & 7)) & ((1 << bits) - 1)
    m1 = (val >= thr).float().view(B, K) * sc
    b2, t2, s2 = models.dropout_threshold(p2)
    m2 = (rnd[rnd.numel() - B * 20:].view(B, 20).to(torch.int32) >= ((256 * t2) >> b2)).float() * s2
    assert abs(float((m1 > 0).float().mean()) - (1 - thr / (1 << bits))) < 5e-3
    assert abs(float((m2 > 0).float().mean()) - (1 - t2 / (1 << b2))) < 0.06
    xt = feat.clone().requires_grad_(True)
    ref = lin2(torch.relu(lin1(xt * m1)) * m2)
    gt = torch.autograd.grad((ref * r).sum(), [xt, lin1.weight, lin2.weight])
    assert torch.allclose(lo, ref, rtol=1e-4, atol=1e-5)
    for a, b in zip(g, gt):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6 + 1e-4 * float(b.abs().max()))


@pytest.mark.parametrize("training,soft", [(True, False), (True, True), (False, False)])
def test_fused_head_loss_equals_head_then_celoss(training, soft, device):
    """PotesHeadLossFunction (head + soft-target CE as one autograd node, two launches between the
    split-K product and the feature pass) == PotesHeadFunction followed by CELoss: loss, logits and
    every gradient, with dropout on (same generator state -> same random bytes) and with a loss
    gradient other than 1."""
    from pcgmix_amd import models
    torch.manual_seed(5)
    m = models.CNN_potes_TS(4, 2, "PhysioNet").to(device).train(training)
    x = torch.randn(24, 4, 2500, device=device)
    t = F.one_hot(torch.randint(0, 2, (24,), device=device), 2).float()
    if soft:
        t = 0.7 * t + 0.3 * t.flip(0)
    gscale = torch.tensor(0.37, device=device)
    res = []
    for fused in (True, False):
        m.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        torch.manual_seed(77)
        if fused:
            loss, logits = m.loss_and_logits(xi, t)
        else:
            logits = m(xi, depth=0, pass_part="second")
            loss = tm.CELoss(2)(logits, t)
        loss.backward(gscale)
        res.append((loss.detach(), logits.detach(), xi.grad.clone(),
                    {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}))
    assert torch.allclose(res[0][0], res[1][0], rtol=1e-5, atol=1e-6)
    assert torch.allclose(res[0][1], res[1][1], rtol=1e-5, atol=1e-6)
    assert torch.allclose(res[0][2], res[1][2], rtol=1e-4, atol=1e-7 + 1e-4 * float(res[1][2].abs().max()))
    assert res[0][3].keys() == res[1][3].keys()
    for k in res[0][3]:
        a, b = res[0][3][k], res[1][3][k]
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-7 + 1e-4 * float(b.abs().max())), k


def test_head_loss_finalize_deferred(device):
    """`defer`: the fused head+loss forward leaves its finalize launch (loss and small gradients
    from the per-row-block contributions) to the backward's feature pass — what the captured
    training step asks for (CNN_potes.loss_and_logits while capturing).  The loss is written by
    the backward; loss, logits and every gradient are bit-identical to the two-launch form.
    (The captured form itself: test_graphed_step_matches_eager* in test_train_gpu.py.)"""
    from pcgmix_amd import models
    from torch.profiler import profile, ProfilerActivity
    torch.manual_seed(8)
    B, K, C = 100, 19968, 2                                  # 25 row blocks: uneven segments
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

    loss0, logits0 = forward(False)
    grads0 = torch.autograd.grad(loss0, params, gs)
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        loss1, logits1 = forward(True)
        loss1.detach().fill_(-1.0)                           # the forward has not written it ...
        grads1 = torch.autograd.grad(loss1, params, gs)
        torch.cuda.synchronize()
    names = [e.name for e in prof.events() if "potes_tail_loss" in e.name]
    assert names and not any("finalize" in n for n in names), names      # one launch less
    assert torch.equal(loss1.detach(), loss0.detach())       # ... the backward does
    assert torch.equal(logits1, logits0)
    for a, b in zip(grads1, grads0):
        assert torch.equal(a, b)
    # (the Function trusts `defer`; CNN_potes.loss_and_logits sets it only with autograd on and
    # a capture in progress)


def test_head_loss_takes_hard_targets_as_uint8_labels(device):
    """target = uint8 (B,) class labels == the float one-hot matrix of the same labels: loss,
    logits and gradients bit-identical (what the captured step sends: B bytes instead of B*C floats)."""
    from pcgmix_amd import models
    torch.manual_seed(3)
    B, K, C = 50, 9968, 2
    w1 = (torch.randn(20, K, device=device) * 0.01).requires_grad_(True)
    b1 = torch.randn(20, device=device).requires_grad_(True)
    w2 = torch.randn(C, 20, device=device).requires_grad_(True)
    b2 = torch.randn(C, device=device).requires_grad_(True)
    feat = torch.randn(B, K, device=device).requires_grad_(True)
    labels = torch.randint(0, C, (B,), device=device)
    params = (feat, w1, b1, w2, b2)
    res = []
    for tgt in (F.one_hot(labels, C).float(), labels.to(torch.uint8)):
        loss, logits = models.PotesHeadLossFunction.apply(feat, w1, b1, w2, b2, tgt, 0.0, 0.0, True, None)
        res.append((loss.detach(), logits, torch.autograd.grad(loss, params)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.equal(a, b)
    with pytest.raises(ValueError):
        models.PotesHeadLossFunction.apply(feat, w1, b1, w2, b2, labels[:-1].to(torch.uint8), 0.0, 0.0,
                                           True, None)


def test_head_loss_finalize_deferred_under_capture(device):
    """The deferred head+loss captured DIRECTLY with torch.cuda.graph (what CNN_potes.loss_and_logits
    does while a training step is being captured) and replayed: loss, logits and every gradient
    bit-identical to the eager two-launch form.

    Round 2's version of this test died with SIGSEGV in CUDAGraph.capture_end.  Cause (round 3,
    profiles/probes/capture_defer_probe.py + profiles/r3_capture_defer_probe_run*.txt): it kept the
    eager pass's ``loss0`` — and with it that pass's autograd graph, whose AccumulateGrad nodes
    belong to the legacy default stream — alive while the same leaves went through backward under
    capture.  The autograd engine then makes the default stream wait on an event recorded in the
    capturing stream, HIP registers the NULL stream as a parallel capture stream, and
    hip::Stream::EndCapture dereferences it (fault address 0x308).  Not `defer`, not the capture
    mode, not a missing warm-up: a torch-ops-only program crashes the same way.  So: keep only
    detached copies of the eager results and let the eager graph die before capturing."""
    from pcgmix_amd import models
    torch.manual_seed(8)
    B, K, C = 100, 19968, 2                                  # 25 row blocks: uneven segments
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

    def eager_reference():
        loss, logits = forward(False)
        grads = torch.autograd.grad(loss, params, gs)
        return loss.detach().clone(), logits.detach().clone(), [g.detach().clone() for g in grads]

    loss0, logits0, grads0 = eager_reference()               # the eager graph is gone on return
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):                            # warm-up of the deferred form
        l, _ = forward(True)
        torch.autograd.grad(l, params, gs)
        del l
    torch.cuda.current_stream(device).wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss1, logits1 = forward(True)
        loss1.detach().fill_(-1.0)                           # the forward has not written it ...
        grads1 = torch.autograd.grad(loss1, params, gs)
    loss1.detach().fill_(-2.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss1.detach(), loss0)                # ... the backward's feature pass does
    assert torch.equal(logits1, logits0)
    for a, b in zip(grads1, grads0):
        assert torch.equal(a, b)


def test_head_loss_backward_twice_is_refused(device):
    """dW1 accumulates into a buffer the forward zeroed: a second backward over the same forward
    would silently double it.  It raises instead."""
    from pcgmix_amd import models
    torch.manual_seed(2)
    B, K, C = 8, 400, 2
    w1 = torch.randn(20, K, device=device, requires_grad=True)
    w2 = torch.randn(C, 20, device=device, requires_grad=True)
    feat = torch.randn(B, K, device=device, requires_grad=True)
    t = F.one_hot(torch.randint(0, C, (B,), device=device), C).float()
    loss, _ = models.PotesHeadLossFunction.apply(feat, w1, None, w2, None, t, 0.0, 0.0, True, None)
    g1 = torch.autograd.grad(loss, (w1,), retain_graph=True)[0].clone()
    with pytest.raises(RuntimeError, match="backward ran twice"):
        torch.autograd.grad(loss, (w1,))
    loss2, _ = models.PotesHeadLossFunction.apply(feat, w1, None, w2, None, t, 0.0, 0.0, True, None)
    assert torch.equal(torch.autograd.grad(loss2, (w1,))[0], g1)
