"""Saliency post-processing and displacement search against goldens recorded from the reference
(tests/golden/salopt_*.npz: raw input gradient, the reference's saliency maps, the displacement
every optimal_displacement_* call returned, and augment()'s final output)."""
import ctypes

import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import augmentations, models, saliency
from conftest import GOLDEN, Args, StepCounter, golden_files, load_golden
from oracle import pcgmix_oracle as O

pytestmark = pytest.mark.gpu
CASES = golden_files("salopt_")


def dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device)


@pytest.mark.parametrize("path", CASES[:1], ids=lambda p: p.split("/")[-1][:-4])
def test_saliency_post_matches_reference(path, device):
    g = load_golden(path)
    grad = torch.from_numpy(g["grad"]).to(device)
    fr = dev_i32(g["frames"], device)
    sal = saliency.saliency_post(grad, fr.data_ptr()).cpu().numpy()
    # float32 convolution in a different summation order than oneDNN's: not bit-identical
    assert np.abs(sal - g["sal"]).max() <= 2e-6
    assert sal.min() == 0.0 and sal.max() == 1.0


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("/")[-1][:-4])
def test_displacements_bit_exact_given_reference_saliency(path, device):
    """Same saliency input -> the integer displacement must be the reference's, which requires
    numpy's float32 pairwise summation order inside the kernel."""
    g = load_golden(path)
    B, T = g["sal"].shape
    sal = torch.from_numpy(g["sal"]).to(device)
    fr, mix = dev_i32(g["frames"], device), dev_i32(g["mix"], device)
    mode = 0 if "(saloptenv" in g["method"] else 1
    disp = saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(),
                                          float(np.float32(g["lam"])), mode, B, T)
    assert np.array_equal(disp.cpu().numpy().astype(np.int64), g["disp"])


@pytest.mark.parametrize("mode", [0, 1])
def test_displacements_random_saliency_vs_oracle(mode, device):
    """Dense random saliency (many near-ties) at 2 kHz state lengths, against the numpy oracle."""
    from pcgmix_amd import synthetic
    B, T = 24, 5000
    frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=2000, seed=5)
    frames[3] = frames[2]                                   # equal lengths -> 0
    rs = np.random.RandomState(mode)
    sal = rs.rand(B, T).astype(np.float32)
    sal[np.arange(T)[None, :] >= frames[:, 4:5]] = 0
    mix = rs.permutation(B)
    lam = np.float32(0.3713)
    lam_np = np.full((1, 1), lam, dtype=np.float32)
    method = "(saloptenv)" if mode == 0 else "(saloptsum)"
    ref = np.stack([O.salopt_displacements(sal[i], sal[mix[i]], frames[i], frames[mix[i]], lam_np, method)
                    for i in range(B)])
    fr, mx = dev_i32(frames, device), dev_i32(mix, device)      # keep the buffers alive
    got = saliency.optimal_displacements(torch.from_numpy(sal).to(device), fr.data_ptr(),
                                         mx.data_ptr(), float(lam), mode, B, T)
    assert np.array_equal(got.cpu().numpy().astype(np.int64), ref)


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("/")[-1][:-4])
def test_salopt_augment_end_to_end(path, device):
    """augment() with the golden's frozen Potes checkpoint.  The model's backward runs on MIOpen,
    so gradients differ from the CPU reference in the last bits; saliency stays within 1e-4 and a
    displacement may flip only where two candidates tie to within that noise."""
    g = load_golden(path)
    sd = np.load(GOLDEN + "/potes_state_seed1234.npz")
    model = models.CNN_potes_TS(4, 2, "PhysioNet")
    model.load_state_dict({k: torch.from_numpy(sd[k]) for k in sd.files})
    saliency.set_saliency_model(model.to(device))
    try:
        data = torch.from_numpy(g["x"]).to(device)
        tgt = torch.nn.functional.one_hot(torch.from_numpy(g["labels"]), 2).to(device)
        sal = saliency.get_saliency_maps(Args(g["method"]), device, data, tgt, g["frames"])
        assert np.abs(sal.cpu().numpy() - g["sal"]).max() <= 1e-4
        y, _, mix, _ = augmentations.augment(Args(g["method"]), data, tgt, torch.from_numpy(g["frames"]),
                                             g["wav"], StepCounter(g["step"]), None, device, "")
    finally:
        saliency.set_saliency_model(None)
    assert np.array_equal(mix, g["mix"])
    got = y.cpu().numpy()
    rows_ok = (np.abs(got - g["y"]).max(axis=(1, 2)) <= 1e-4)
    assert rows_ok.mean() >= 0.75, f"{(~rows_ok).sum()} of {len(rows_ok)} rows differ"


def test_bad_arguments(device):
    lib = pcgmix_amd._lib.load()
    z = torch.zeros(8, device=device)
    assert lib.pcgmix_saliency_post_f32(z.data_ptr(), z.data_ptr(), z.data_ptr(), 100,
                                        ctypes.c_double(12.0), 1, 1, 8, None) != 0     # even ksize
    assert lib.pcgmix_salopt_disp_f32(z.data_ptr(), z.data_ptr(), z.data_ptr(), ctypes.c_float(0.5),
                                      2, z.data_ptr(), 1, 8, None) != 0               # bad mode


def test_displacements_full_batch_vs_oracle(device):
    """BASELINE.json configs[2] size (B=256, T=5000): every displacement equals the oracle's."""
    from pcgmix_amd import synthetic
    B, T = 256, 5000
    frames, labels, wav = synthetic.make_index_data(B, T, sample_rate=2000, seed=9)
    rs = np.random.RandomState(3)
    sal = rs.rand(B, T).astype(np.float32) ** 2
    sal[np.arange(T)[None, :] >= frames[:, 4:5]] = 0
    mix = rs.permutation(B)
    lam = np.float32(0.6180339)
    lam_np = np.full((1, 1), lam, dtype=np.float32)
    fr, mx = dev_i32(frames, device), dev_i32(mix, device)
    sal_d = torch.from_numpy(sal).to(device)
    for mode, tag in ((0, "(saloptenv)"), (1, "(saloptsum)")):
        ref = np.stack([O.salopt_displacements(sal[i], sal[mix[i]], frames[i], frames[mix[i]], lam_np, tag)
                        for i in range(0, B, 4)])                       # every 4th sample: ~4 s of numpy
        got = saliency.optimal_displacements(sal_d, fr.data_ptr(), mx.data_ptr(), float(lam), mode, B, T)
        assert np.array_equal(got.cpu().numpy().astype(np.int64)[::4], ref)


def test_graphed_saliency_equals_eager(device):
    """hipGraph replay of the frozen model's fwd + input gradient + post-processing == eager."""
    from pcgmix_amd import synthetic
    torch.manual_seed(0)
    model = models.CNN_potes_TS(4, 2, "PhysioNet").to(device)
    saliency.set_saliency_model(model)
    try:
        outs = {}
        for use in (False, True):
            saliency.USE_GRAPHS = use
            res = []
            for seed in (1, 2, 3):
                x, frames, labels, wav = synthetic.make_batch(16, 4, 2500, seed=seed)
                data = torch.from_numpy(x).to(device)
                tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
                res.append(saliency.get_saliency_maps(Args("x"), device, data, tgt, frames).cpu())
            outs[use] = res
        for a, b in zip(outs[False], outs[True]):
            assert torch.allclose(a, b, atol=1e-6)
    finally:
        saliency.USE_GRAPHS = True
        saliency.set_saliency_model(None)
