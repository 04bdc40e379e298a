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


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
@pytest.mark.parametrize("B,seed", [(8, 0), (32, 3)])
def test_logmel_matches_restatement(B, seed, pad_mode, device):
    """Per-cycle front end vs the numpy restatement, for both paddings of the centred frames
    (which one librosa 0.9.2 defaults to is the open point of the restatement: a parameter, with
    'constant' the default; tolerance 1e-4 = north_star's bound on spectrograms)."""
    x, frames, _, _ = synthetic.make_batch(B, 1, 5000, sample_rate=2000, seed=seed)
    # heart-sound-like content: band-limited bursts instead of white noise in S1/S2
    t = np.arange(5000) / 2000.0
    x[:, 0] *= (0.2 + np.abs(np.sin(2 * np.pi * 3.0 * t)))[None, :].astype(np.float32)
    ref, fs_ref = O.logmel(x[:, 0], frames, pad_mode=pad_mode)
    spec, fs = frontend.logmel(torch.from_numpy(x).to(device), frames, pad_mode=pad_mode)
    got = spec.cpu().numpy()[:, 0]
    assert np.array_equal(fs, fs_ref)                      # column boundaries: bit-exact
    assert got.shape == (B, 128, 128)
    err = np.abs(got - ref)
    assert err.max() <= 1e-4, err.max()                    # north_star tolerance on spectrograms
    cols = np.arange(128)[None, None, :] >= fs[:, 4][:, None, None]
    assert (got[np.broadcast_to(cols, got.shape)] == 0).all()


@pytest.mark.parametrize("sample_rate,T", [(2000, 5000), (1000, 2500), (1000, 1800)])
def test_logmel_boundaries_in_arguments_equal_boundaries_in_memory(sample_rate, T, device):
    """pcgmix_logmel_hostframes_f32 (cycle ends in the kernel arguments: what frontend.logmel calls)
    == pcgmix_logmel_f32 (boundaries read from device memory, with frames_out), bit for bit, at
    n_fft = 136 (nine k-steps, the register ring) and at 68 (the general loop); and both
    against the restatement."""
    import ctypes
    from pcgmix_amd import _lib
    from pcgmix_amd.augmentations import upload_array
    B = 6
    x, frames, _, _ = synthetic.make_batch(B, 1, T, sample_rate=sample_rate, seed=2)
    xd = torch.from_numpy(x[:, 0].copy()).to(device)
    spec, fs = frontend.logmel(xd, frames, sample_rate=sample_rate)
    n_fft, hop = frontend.stft_params(sample_rate)
    lib = _lib.load()
    tables = frontend.logmel_tables(xd.device, n_fft, 128, sample_rate)
    fr = upload_array(frames.astype(np.int32), xd.device)
    spec2 = torch.empty_like(spec)
    fo = torch.empty((B, 5), dtype=torch.int32, device=device)
    _lib.check(lib.pcgmix_logmel_f32(xd.data_ptr(), fr.data_ptr(), tables.data_ptr(), spec2.data_ptr(),
                                     fo.data_ptr(), B, T, n_fft, hop, 128, ctypes.c_float(frontend.TRAIN_MEAN),
                                     ctypes.c_float(frontend.TRAIN_STD), 128, 0,
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)),
               "pcgmix_logmel_f32")
    assert torch.equal(spec, spec2)
    assert np.array_equal(fo.cpu().numpy().astype(np.int64), fs)
    ref, fs_ref = O.logmel(x[:, 0], frames, n_fft=n_fft, hop=hop, sr=float(sample_rate))
    assert np.array_equal(fs, fs_ref)
    assert np.abs(spec.cpu().numpy()[:, 0] - ref).max() <= 1e-4


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


def test_pad_modes_differ_only_at_the_edges(device):
    x, frames, _, _ = synthetic.make_batch(4, 1, 5000, sample_rate=2000, seed=1)
    x[:, 0, 2000:2400] *= 6.0                              # the item's maximum is in the interior
    frames[:, 4] = 4990                                    # keep every column
    xd = torch.from_numpy(x).to(device)
    a, _ = frontend.logmel(xd, frames, pad_mode="constant")
    b, _ = frontend.logmel(xd, frames, pad_mode="reflect")
    a, b = a.cpu().numpy()[:, 0], b.cpu().numpy()[:, 0]
    assert frontend.DEFAULT_PAD_MODE == "constant"
    assert np.abs(a[:, :, :2] - b[:, :, :2]).max() > 1e-3  # frames 0, 1 see the padding
    assert np.abs(a[:, :, 2:128] - b[:, :, 2:128]).max() <= 1e-5
    with pytest.raises(ValueError):
        frontend.logmel(xd, frames, pad_mode="edge")


def _recordings(seed, n_rec, sample_rate=2000):
    """Synthetic recordings in the shape databuilder.ipynb cell 6 sees them: a waveform of several
    consecutive heart cycles, the boundaries of ALL heart states in samples, and the indices of
    the boundaries at which a full cycle starts."""
    rs = np.random.RandomState(seed)
    ys, bounds, starts = [], [], []
    for r in range(n_rec):
        n_cyc = int(rs.randint(3, 12))
        fr = synthetic.make_frames(n_cyc, sample_rate / 1000.0, rs)          # (n_cyc, 5) relative
        lead = int(rs.randint(0, 900))
        b = [lead]
        for c in range(n_cyc):
            b += list(b[-1] + np.diff(fr[c]))
        tail = int(rs.randint(70, 1500))
        n = b[-1] + tail
        t = np.arange(n) / sample_rate
        y = (rs.standard_normal(n) * (0.05 + np.abs(np.sin(2 * np.pi * 1.3 * t + r)))).astype(np.float32)
        y *= float(rs.uniform(0.05, 3.0))                                  # recordings differ in level
        ys.append(y)
        bounds.append(np.asarray(b, dtype=np.int64))
        starts.append(list(range(0, 4 * n_cyc, 4)))
    return ys, bounds, starts


@pytest.mark.parametrize("pad_mode", ["constant", "reflect"])
def test_logmel_recordings_match_restatement(pad_mode, device):
    """The reference's order of operations (databuilder.ipynb cell 6:81-101, 127-142): ONE
    transform per recording, dB relative to the recording's maximum, per-cycle column slices,
    zero-padding after normalisation — against oracle.logmel_recording; column boundaries
    bit-exact, values within 1e-4."""
    ys, bounds, starts = _recordings(11, 9)
    y = torch.from_numpy(np.concatenate(ys)).to(device)
    spec, fs, rec_of = frontend.logmel_recordings(y, [len(v) for v in ys], bounds, starts,
                                                  pad_mode=pad_mode)
    got = spec.cpu().numpy()[:, 0]
    k = 0
    for r, (yy, b, st) in enumerate(zip(ys, bounds, starts)):
        ref, rel = O.logmel_recording(yy, b, st, pad_mode=pad_mode)
        n = len(st)
        assert np.array_equal(fs[k:k + n], rel) and (rec_of[k:k + n] == r).all()
        err = np.abs(got[k:k + n] - ref).max()
        assert err <= 1e-4, (r, err)
        k += n
    assert k == got.shape[0] and got.shape[1:] == (128, 128)


def test_recording_level_differs_from_per_cycle_as_documented(device):
    """Why both granularities exist: the same cycle cut out of a recording-level spectrogram
    differs from the per-cycle transform of its samples at the cycle's edge columns (neighbouring
    samples instead of padding) and by the dB reference (recording maximum vs item maximum); in
    the interior, when the cycle starts on the recording's frame grid, the two differ by that
    constant only."""
    rs = np.random.RandomState(5)
    hop = 34
    fr = synthetic.make_frames(3, 2.0, rs)
    b = [5 * hop]                                           # cycle 0 starts on the frame grid
    for c in range(3):
        b += list(b[-1] + np.diff(fr[c]))
    n = b[-1] + 400
    y = rs.standard_normal(n).astype(np.float32)
    y[b[4]:b[8]] *= 4.0                                     # the loudest part is in cycle 1
    bounds = np.asarray(b, dtype=np.int64)
    spec, fs, _ = frontend.logmel_recordings(torch.from_numpy(y).to(device), [n], [bounds], [[0, 4, 8]])
    item = np.zeros((1, 5000), np.float32)
    seg = y[b[0]:b[4]]
    item[0, :len(seg)] = seg
    per_cycle, fs1 = frontend.logmel(torch.from_numpy(item).to(device), (bounds[:5] - b[0])[None, :])
    a, c = spec[0, 0].cpu().numpy(), per_cycle[0, 0].cpu().numpy()
    m = int(min(fs[0, 4], fs1[0, 4])) - 3
    diff = a[:, 3:m] - c[:, 3:m]
    assert np.abs(diff).max() > 1e-2                        # not the same image ...
    assert np.ptp(diff) <= 2e-3                             # ... but a constant apart inside
    assert np.abs(a[:, :2] - c[:, :2] - diff.mean()).max() > 1e-2   # and not at the edge columns
