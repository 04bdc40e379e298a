"""HIP log-mel front end against the numpy restatement of librosa 0.9.2 semantics
(oracle.logmel — parity with librosa itself is UNPINNED: it is not installed and the reference
stores no spectrogram)."""
import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import frontend, synthetic
from oracle import pcgmix_oracle as O

pytestmark = pytest.mark.gpu


def test_filterbank_structure():
    """125 of the 128 filters have one non-zero FFT bin, 3 have two."""
    w = O.mel_filterbank(2000.0, 136, 128, 25.0, 1000.0)
    nnz = (w > 0).sum(1)
    assert (nnz == 1).sum() == 125 and (nnz == 2).sum() == 3 and (nnz == 0).sum() == 0
    assert (w[:, [0, 1, 68]] == 0).all()


@pytest.mark.parametrize("B,seed", [(8, 0), (32, 3)])
def test_logmel_matches_restatement(B, seed, device):
    x, frames, _, _ = synthetic.make_batch(B, 1, 5000, sample_rate=2000, seed=seed)
    # heart-sound-like content: band-limited bursts instead of white noise in S1/S2
    t = np.arange(5000) / 2000.0
    x[:, 0] *= (0.2 + np.abs(np.sin(2 * np.pi * 3.0 * t)))[None, :].astype(np.float32)
    ref, fs_ref = O.logmel(x[:, 0], frames)
    spec, fs = frontend.logmel(torch.from_numpy(x).to(device), frames)
    got = spec.cpu().numpy()[:, 0]
    assert np.array_equal(fs, fs_ref)                      # column boundaries: bit-exact
    assert got.shape == (B, 128, 128)
    err = np.abs(got - ref)
    assert err.max() <= 1e-4, err.max()                    # north_star tolerance on spectrograms
    cols = np.arange(128)[None, None, :] >= fs[:, 4][:, None, None]
    assert (got[np.broadcast_to(cols, got.shape)] == 0).all()


def test_logmel_silence_and_bad_arguments(device):
    x = torch.zeros(2, 5000, device=device)
    frames = np.array([[0, 200, 600, 800, 1800]] * 2)
    spec, fs = frontend.logmel(x, frames)
    # all-zero input: every band sits at amin, ref is amin too -> 0 dB -> (0 - mean)/std inside
    inside = spec[0, 0, :, : int(fs[0, 4])].cpu().numpy()
    assert np.allclose(inside, (0.0 - frontend.TRAIN_MEAN) / frontend.TRAIN_STD, atol=1e-6)
    with pytest.raises(ValueError):
        frontend.logmel(torch.zeros(2, 2, 5000, device=device), frames)
    with pytest.raises(ValueError):
        frontend.logmel(torch.zeros(2, 5000), frames)
