"""Fused BatchNorm(train) + ReLU + MaxPool kernels (HIP, NHWC) against the torch composition in
float64 (floating-point kernels; tolerances stated per assert)."""
import pytest
import torch
import torch.nn.functional as F

import pcgmix_amd  # noqa: F401
from pcgmix_amd import models

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,C,H,W,ph,pw", [
    (8, 64, 1, 2500, 1, 2),      # ResNet9-1D conv2: pooled
    (8, 128, 1, 1250, 1, 1),     # res block: no pooling
    (4, 512, 1, 625, 1, 2),      # odd length: the last column is in no window
    (3, 256, 1, 37, 1, 1),
    (4, 128, 16, 16, 2, 2),      # ResNet9-2D
    (2, 64, 9, 7, 2, 2),         # ragged 2-D: leftover row and column
    (1, 4, 1, 8, 1, 4),
    (2, 1024, 1, 16, 1, 2),
])
def test_bn_relu_pool_matches_torch_float64(B, C, H, W, ph, pw, device):
    torch.manual_seed(B * C + W)
    y = (torch.randn(B, C, H, W, device=device) * 1.7 + 0.3).contiguous(memory_format=torch.channels_last)
    y.requires_grad_(True)
    gamma = (torch.rand(C, device=device) + 0.5).requires_grad_(True)
    beta = (torch.randn(C, device=device) * 0.3).requires_grad_(True)
    rm, rv = torch.randn(C, device=device) * 0.1, torch.rand(C, device=device) + 0.5
    rm2, rv2 = rm.clone().double(), rv.clone().double()
    assert models.BNReLUPoolFunction.supported(y)
    z = models.BNReLUPoolFunction.apply(y, gamma, beta, rm, rv, 0.1, 1e-5, ph, pw)
    assert z.shape == (B, C, H // ph, W // pw) and z.is_contiguous(memory_format=torch.channels_last)
    dz = torch.randn_like(z)
    z.backward(dz)

    yd = y.detach().double().requires_grad_(True)
    gd, bd = gamma.detach().double().requires_grad_(True), beta.detach().double().requires_grad_(True)
    want = F.relu(F.batch_norm(yd, rm2, rv2, gd, bd, True, 0.1, 1e-5))
    if (ph, pw) != (1, 1):
        want = F.max_pool2d(want, (ph, pw))
    want.backward(dz.double())
    assert torch.allclose(z.double(), want, rtol=1e-4, atol=2e-5), float((z.double() - want).abs().max())
    assert torch.allclose(rm.double(), rm2, rtol=1e-5, atol=1e-6)
    assert torch.allclose(rv.double(), rv2, rtol=1e-5, atol=1e-6)
    for got, ref, name in ((y.grad, yd.grad, "dx"), (gamma.grad, gd.grad, "dgamma"), (beta.grad, bd.grad, "dbeta")):
        scale = float(ref.abs().max()) + 1e-12
        err = float((got.double() - ref).abs().max())
        # a ReLU / arg-max decision that sits within rounding of a tie may flip for single elements
        bad = ((got.double() - ref).abs() > 1e-4 * scale + 1e-6).float().mean().item()
        assert bad <= 1e-4, (name, err, scale, bad)


def test_bn_relu_pool_is_deterministic(device):
    torch.manual_seed(0)
    y = torch.randn(8, 128, 1, 2500, device=device).contiguous(memory_format=torch.channels_last)
    g, b = torch.rand(128, device=device) + 0.5, torch.randn(128, device=device)
    outs = []
    for _ in range(2):
        yy = y.clone().requires_grad_(True)
        z = models.BNReLUPoolFunction.apply(yy, g, b, None, None, 0.1, 1e-5, 1, 2)
        z.square().sum().backward()
        outs.append((z.detach().clone(), yy.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
