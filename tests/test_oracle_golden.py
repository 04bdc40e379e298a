"""The CPU oracle must reproduce, bit for bit, what the real reference produced
(tests/golden/*.npz were written by tests/golden/make_golden.py running the reference)."""
import numpy as np
import pytest

from conftest import golden_files, load_golden
from oracle import pcgmix_oracle as O

CASES = (golden_files("mix1d_") + golden_files("mix2d_") + golden_files("mask2d_")
         + golden_files("salopt_") + golden_files("salopt2d_"))


def test_golden_inventory():
    assert len(golden_files("mix1d_")) >= 25
    assert len(golden_files("mix2d_")) == 3
    assert len(golden_files("salopt_")) == 3
    assert len(golden_files("salopt2d_")) == 3


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_matches_reference(path):
    g = load_golden(path)
    r = O.augment(g["method"], g["x"], g["labels"], g["frames"], g["wav"], g["step"],
                  saliency_maps=g.get("sal"))
    assert r["fired"] == bool(g["fired"])
    if not r["fired"]:
        assert r["y"] is g["x"] and int(g["same_object"]) == 1 and g["mix"].size == 0
        return
    assert np.array_equal(r["mix"], g["mix"])                      # partner indices: bit-exact
    assert r["lam"] == float(g["lam"])
    if "knots" in g:
        assert np.array_equal(r["knots"].ravel(), g["knots"].ravel())
    assert np.array_equal(r["y"], g["y"])                          # waveforms: bit-exact
    assert np.array_equal(np.asarray(r["target"], np.float64), np.asarray(g["target_out"], np.float64))
    if "disp" in g:
        assert np.array_equal(r["disp"], g["disp"])                # displacements: bit-exact


@pytest.mark.parametrize("path", golden_files("salopt_")[:1], ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_saliency_post(path):
    g = load_golden(path)
    assert np.array_equal(O.saliency_post(g["grad"], g["frames"]), g["sal"])


def test_oracle_saliency_post2d():
    """The spectrogram branch of the saliency post-processing (saliency.py:93-113) against the maps
    the reference returned for the recorded input gradient of its ResNet9-2D."""
    g = load_golden(golden_files("salopt2d_")[0])
    assert np.array_equal(O.saliency_post2d(g["grad"], g["frames"]), g["sal"])


def test_oracle_ce_soft():
    import torch
    rs = np.random.RandomState(0)
    logits = rs.randn(16, 2).astype(np.float32)
    t = np.eye(2)[rs.randint(0, 2, 16)] * 0.7 + 0.15
    ref = -(torch.log_softmax(torch.from_numpy(logits), 1) * torch.from_numpy(t)).sum(1).mean()
    assert abs(O.ce_soft(logits, t) - float(ref)) < 1e-6


@pytest.mark.parametrize("path", golden_files("salopt_") + golden_files("salopt2d_"),
                         ids=lambda p: p.split("/")[-1][:-4])
def test_oracle_displacement_objective_is_what_the_search_maximises(path):
    """``displacement_objective`` (used by the GPU tests to prove near-ties) evaluated at every
    candidate reproduces the reference's recorded arg-max: first strict maximum."""
    g = load_golden(path)
    lam_np = np.full((1, 1), np.float32(g["lam"]), dtype=np.float32)
    fr, checked = g["frames"], 0
    for i in range(len(fr)):
        j = g["mix"][i]
        for k in range(4):
            n1, n2 = fr[i, k + 1] - fr[i, k], fr[j, k + 1] - fr[j, k]
            if n1 == n2:
                assert g["disp"][i, k] == 0
                continue
            s1, s2 = g["sal"][i][fr[i, k]:fr[i, k + 1]], g["sal"][j][fr[j, k]:fr[j, k + 1]]
            vals = [O.displacement_objective(s1, s2, lam_np, d, g["method"]) for d in range(abs(n1 - n2) + 1)]
            best, arg = float("-inf"), 0
            for d, v in enumerate(vals):
                if v > best:
                    best, arg = v, d
            assert arg == g["disp"][i, k]
            checked += 1
    assert checked > (10 if g["x"].ndim == 3 else 5)


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_oracle_stft_stage_matches_torch_stft(pad_mode):
    """The log-mel restatement is unpinned against librosa (absent offline); its STFT stage at
    least is cross-checked against an independent implementation, torch.stft in float64, for both
    paddings of the centred frames."""
    import torch
    y = np.random.RandomState(0).randn(5000).astype(np.float32)
    S = O.stft_complex(y, 136, 34, pad_mode)
    win = torch.hann_window(136, periodic=True, dtype=torch.float64)
    T = torch.stft(torch.from_numpy(y).double(), 136, 34, 136, win, center=True, pad_mode=pad_mode,
                   return_complex=True).numpy()
    assert S.shape == T.shape == (69, 148)
    assert np.abs(S - T).max() <= 1e-11


def test_oracle_recording_level_logmel_is_consistent():
    """logmel_recording of a recording that IS one cycle starting at sample 0 equals the
    per-cycle restatement (same padding, same maximum); column boundaries use Python's round."""
    rs = np.random.RandomState(2)
    y = rs.randn(5000).astype(np.float32)
    fr = np.array([0, 300, 900, 1200, 4100])
    a, rel = O.logmel_recording(y, fr, [0])
    b, fs = O.logmel(y[None, :], fr[None, :])
    assert np.array_equal(rel[0], fs[0]) and np.array_equal(a[0], b[0])
    assert rel[0, 4] == round(4100 * 148 / 5000)
