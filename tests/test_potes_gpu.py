"""Fused Potes conv stack (HIP) against the same stack through torch/MIOpen in float32
(a floating-point kernel: the torch fp32 path is the reference here; tolerances stated)."""
import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import models

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[True, False], ids=["masks", "recompute"])
def backward_kind(request):
    """Both backward families: from the forward's saved ReLU/pool routing (default), and the
    kernels that recompute the forward per tile."""
    old = models.PotesStackFunction.use_masks
    models.PotesStackFunction.use_masks = request.param
    yield request.param
    models.PotesStackFunction.use_masks = old


def make(T, device, seed=0):
    torch.manual_seed(seed)
    m = models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=None if T == 2500 else T).to(device)
    return m


@pytest.mark.parametrize("B,T", [(8, 2500), (4, 5000), (3, 1037), (2, 1030), (5, 526), (2, 14), (1, 23)])
def test_forward_matches_torch(B, T, device):
    m = make(T, device).eval()
    x = torch.randn(B, 4, T, device=device)
    m.fused = True
    y_f = m(x)
    m.fused = False
    y_t = m(x)
    assert torch.allclose(y_f, y_t, rtol=1e-4, atol=1e-5), float((y_f - y_t).abs().max())
    # the stack itself, element-wise
    c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
    h_f = models.PotesStackFunction.apply(x.reshape(B * 4, T), c1.weight, c1.bias, c2.weight, c2.bias)
    h_t = m.cnn1(x.reshape(B * 4, 1, T))
    assert h_f.shape == h_t.shape
    assert torch.allclose(h_f, h_t, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("B,T", [(8, 2500), (4, 5000), (3, 1037), (2, 1030), (6, 526), (2, 270), (1, 23)])
def test_weight_gradients_match_torch(B, T, backward_kind, device):
    """dL/d{w1,b1,w2,b2} of a random linear functional of the stack output (dropout off)."""
    m = make(T, device, seed=1).eval()
    x = torch.randn(B, 4, T, device=device)
    c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
    params = [c1.weight, c1.bias, c2.weight, c2.bias]
    h_t = m.cnn1(x.reshape(B * 4, 1, T))
    r = torch.randn_like(h_t)
    g_t = torch.autograd.grad((h_t * r).sum(), params)
    h_f = models.PotesStackFunction.apply(x.reshape(B * 4, T), *params)
    g_f = torch.autograd.grad((h_f * r).sum(), params)
    for a, b, name in zip(g_f, g_t, ("w1", "b1", "w2", "b2")):
        scale = float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= 2e-4 * scale, (name, float((a - b).abs().max()), scale)


def test_training_step_equivalence(backward_kind, device):
    """One Adam step with the fused stack == one with the torch stack (dropout disabled)."""
    outs = []
    for fused in (True, False):
        m = make(2500, device, seed=2).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        m.fused = fused
        opt = torch.optim.Adam([p for p in m.parameters()], lr=1e-2)
        torch.manual_seed(5)
        x = torch.randn(16, 4, 2500, device=device)
        t = torch.nn.functional.one_hot(torch.randint(0, 2, (16,), device=device), 2).float()
        loss = -(torch.log_softmax(m(x), 1) * t).sum(1).mean()
        loss.backward()
        opt.step()
        outs.append((float(loss), [p.detach().clone() for p in m.cnn1.parameters()] +
                     [m.dimreduc.weight.detach().clone()]))
    assert abs(outs[0][0] - outs[1][0]) < 1e-5
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.allclose(a, b, rtol=1e-3, atol=2e-4)   # Adam normalises: sign/ratio sensitive


@pytest.mark.parametrize("B,T", [(8, 2500), (4, 5000), (3, 1037), (2, 1030), (6, 526), (2, 270), (1, 23), (2, 14)])
def test_input_gradient_matches_torch(B, T, backward_kind, device):
    """dL/dx through the fused stack == through the torch stack (saliency maps differentiate the
    class score with respect to the input, saliency.py:52-61)."""
    m = make(T, device, seed=3).eval()
    c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
    params = [c1.weight, c1.bias, c2.weight, c2.bias]
    x1 = torch.randn(B * 4, T, device=device, requires_grad=True)
    x2 = x1.detach().clone().requires_grad_(True)
    h_t = m.cnn1(x1.unsqueeze(1))
    r = torch.randn_like(h_t)
    (g_t,) = torch.autograd.grad((h_t * r).sum(), x1)
    h_f = models.PotesStackFunction.apply(x2, *params)
    (g_f,) = torch.autograd.grad((h_f * r).sum(), x2)
    scale = float(g_t.abs().max()) + 1e-6
    assert float((g_f - g_t).abs().max()) <= 1e-4 * scale
    # whole model, saliency style
    xa = torch.randn(B, 4, T, device=device, requires_grad=True)
    xb = xa.detach().clone().requires_grad_(True)
    m.fused = True
    m(xa)[:, 0].sum().backward()
    m.fused = False
    m(xb)[:, 0].sum().backward()
    assert torch.allclose(xa.grad, xb.grad, rtol=1e-3, atol=1e-5 * float(xb.grad.abs().max()) + 1e-9)


@pytest.mark.parametrize("B,K", [(256, 19968), (32, 9968), (7, 4096), (1, 1028), (3, 19968)])
def test_skinny_linear_matches_torch(B, K, device):
    torch.manual_seed(B + K)
    lin = torch.nn.Linear(K, 20).to(device)
    x = torch.randn(B, K, device=device, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    z_t = lin(x)
    z_f = models.SkinnyLinearFunction.apply(x2, lin.weight, lin.bias)
    assert torch.allclose(z_f, z_t, rtol=1e-4, atol=1e-4 * float(z_t.abs().max()))
    r = torch.randn_like(z_t)
    g_t = torch.autograd.grad((z_t * r).sum(), [x, lin.weight, lin.bias])
    g_f = torch.autograd.grad((z_f * r).sum(), [x2, lin.weight, lin.bias])
    for a, b in zip(g_f, g_t):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()) + 1e-7)


def test_saved_masks_equal_recomputed_routing(device):
    """The forward with masks returns the same activations as the plain forward, and the two
    backward families agree with each other far below the torch tolerance (they can only differ
    where a ReLU/pool decision is an exact near-tie under another summation order)."""
    m = make(5000, device, seed=4).eval()
    c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
    params = [c1.weight, c1.bias, c2.weight, c2.bias]
    x = torch.randn(64, 5000, device=device)
    res = {}
    old = models.PotesStackFunction.use_masks
    try:
        for kind in (True, False):
            models.PotesStackFunction.use_masks = kind
            xi = x.clone().requires_grad_(True)
            h = models.PotesStackFunction.apply(xi, *params)
            r = torch.randn(h.shape, device=device, generator=torch.Generator(device).manual_seed(1))
            res[kind] = (h.detach(), torch.autograd.grad((h * r).sum(), [xi] + params))
    finally:
        models.PotesStackFunction.use_masks = old
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert float((a - b).abs().max()) <= 2e-5 * (float(b.abs().max()) + 1e-6)


def _counter_hash(n_words, key):
    """numpy restatement of counter_hash (csrc/pcgmix_potes.hip): murmur3's 32-bit finaliser, keyed."""
    k0, k1 = np.uint32(key & 0xFFFFFFFF), np.uint32(key >> 32)
    with np.errstate(over="ignore"):
        h = np.arange(n_words, dtype=np.uint32) * np.uint32(0x9E3779B1) + k0
        h ^= h >> np.uint32(16)
        h *= np.uint32(0x85EBCA6B)
        h ^= h >> np.uint32(13)
        h ^= k1
        h *= np.uint32(0xC2B2AE35)
        h ^= h >> np.uint32(16)
    return h


@pytest.mark.parametrize("key_on_device", [False, True])
def test_forward_fills_dropout_bytes(key_on_device, device):
    """The mask-saving forward also fills the head's dropout bytes: word w = counter_hash(key, w),
    key by value or from device memory; the forward's own outputs do not change; the bytes are
    uniform enough for dropout (mean, per-bit balance) and differ between keys."""
    import ctypes
    from pcgmix_amd import _lib
    lib = _lib.load()
    N, T = 64, 2500
    m = make(T, device)
    c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
    x = torch.randn(N, T, device=device)
    P2 = lib.pcgmix_potes_out_len(T)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)

    def forward(rnd, key):
        h2 = torch.empty(N, 4, P2, device=device)
        m2 = torch.empty(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=device)
        kd = torch.tensor([key & 0xFFFFFFFF, key >> 32], dtype=torch.int64).to(torch.int32).to(device) \
            if (key_on_device and rnd is not None) else None
        _lib.check(lib.pcgmix_potes_stack_fwd_save_f32(
            x.data_ptr(), c1.weight.data_ptr(), c1.bias.data_ptr(), c2.weight.data_ptr(),
            c2.bias.data_ptr(), h2.data_ptr(), m2.data_ptr(), None, N, T,
            rnd.data_ptr() if rnd is not None else None, rnd.numel() if rnd is not None else 0,
            kd.data_ptr() if kd is not None else None, 0 if kd is not None else key, st), "fwd")
        return h2, m2

    h_ref, m_ref = forward(None, 0)
    words = []
    for key, nbytes in ((0x0123456789ABCDEF, 16), (0xFEDCBA9876543210, 3 * 1024 * 1024 + 16)):
        rnd = torch.zeros(nbytes + 16, dtype=torch.uint8, device=device)
        h2, m2 = forward(rnd[:nbytes], key)
        got = rnd.cpu().numpy()
        assert (got[nbytes:] == 0).all()
        w = got[:nbytes].view(np.uint32)
        assert np.array_equal(w, _counter_hash(nbytes // 4, key))
        assert torch.equal(h2, h_ref) and torch.equal(m2, m_ref)
        words.append(w)
    big = words[1]
    assert abs(big.view(np.uint8).mean() - 127.5) < 0.5
    bits = np.unpackbits(big.view(np.uint8)[:1 << 20]).reshape(-1, 8).mean(0)
    assert np.abs(bits - 0.5).max() < 0.01
    assert (_counter_hash(4096, 1) != _counter_hash(4096, 2)).mean() > 0.99
    with pytest.raises(RuntimeError):                       # not a multiple of 16 bytes
        forward(torch.zeros(24, dtype=torch.uint8, device=device), 1)


@pytest.mark.parametrize("T", [5000, 1036])
def test_forward_staging_paths_agree(T, device):
    """The matrix-core forward stages a tile with aligned 16-byte loads when the rows are 16-byte
    aligned (T % 4 == 0 and an aligned base) and with 4-byte loads otherwise: the same rows at a
    base shifted by one float must give bit-identical activations and routing bytes (m2, s1)."""
    import ctypes
    from pcgmix_amd import _lib
    lib = _lib.load()
    N = 24
    m = make(T, device, seed=6).eval()
    c1, c2 = m.cnn1[0][0], m.cnn1[1][0]
    buf = torch.randn(N * T + 1, device=device)
    x_al = buf[:N * T].clone()                       # 16-byte aligned base (fresh allocation)
    x_un = buf[1:]                                   # the same layout 4 bytes off an aligned base
    x_un.copy_(x_al)
    assert x_al.data_ptr() % 16 == 0 and x_un.data_ptr() % 16 == 4
    P2 = lib.pcgmix_potes_out_len(T)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    out = []
    for x in (x_al, x_un):
        h2 = torch.empty(N, 4, P2, device=device)
        m2 = torch.zeros(lib.pcgmix_potes_mask_bytes(N, T, 2), dtype=torch.uint8, device=device)
        s1 = torch.zeros(lib.pcgmix_potes_mask_bytes(N, T, 1), dtype=torch.uint8, device=device)
        _lib.check(lib.pcgmix_potes_stack_fwd_save_f32(
            x.data_ptr(), c1.weight.data_ptr(), c1.bias.data_ptr(), c2.weight.data_ptr(),
            c2.bias.data_ptr(), h2.data_ptr(), m2.data_ptr(), s1.data_ptr(), N, T, None, 0, None, 0, st), "fwd")
        out.append((h2, m2, s1))
    for a, b in zip(*out):
        assert torch.equal(a, b)
    ref = m.cnn1(x_al.reshape(N, 1, T))
    assert torch.allclose(out[0][0], ref, rtol=1e-4, atol=1e-5)
