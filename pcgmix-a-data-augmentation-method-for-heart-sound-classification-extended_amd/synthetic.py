"""Synthetic PCG heart-cycle frames (BASELINE.md §2 "Inputs").

The PhysioNet-2016 pickle the reference trains on is not redistributable and its
pre-processing is not in the reference repo (databuilder.ipynb cell 25:86-91), so
every parity test and every benchmark runs on frames shaped like the reference's
loader output (dataloader_physionet.py:151-172): one heart cycle per row,
zero-padded to ``sig_len``, with cumulative state boundaries
``frames = [0, S1end, sysEnd, S2end, cycleEnd]``.

Draw order is part of the contract (goldens depend on it):
``RandomState(seed)`` -> S1, systole, S2, diastole lengths (each a length-B
``randint``) -> ``standard_normal((B, C, T))`` -> labels ``randint(0, 2, B)``.
"""
from __future__ import annotations

import numpy as np

# state-length ranges in 1 kHz samples, [lo, hi) — 0.55–1.35 s heart cycles
STATE_RANGES_1KHZ = ((80, 160), (150, 350), (70, 140), (250, 700))


def make_frames(batch: int, rate_scale: float, rs: np.random.RandomState) -> np.ndarray:
    """int64 (B, 5) cumulative boundaries; lengths scale with sample rate / 1 kHz."""
    lens = np.stack([rs.randint(lo, hi, size=batch) for lo, hi in STATE_RANGES_1KHZ], axis=1)
    lens = np.maximum(1, np.floor(lens * rate_scale + 0.5).astype(np.int64))
    frames = np.zeros((batch, 5), dtype=np.int64)
    frames[:, 1:] = np.cumsum(lens, axis=1)
    return frames


def make_batch(batch: int, channels: int, sig_len: int, sample_rate: int = 1000,
               seed: int = 0, rate_scale: float | None = None):
    """Returns (x float32 (B,C,T), frames int64 (B,5), labels int64 (B,), wav tuple).

    ``rate_scale`` overrides sample_rate/1000 (used by small test shapes so the
    cycle still fits in ``sig_len``).
    """
    rs = np.random.RandomState(seed)
    scale = sample_rate / 1000.0 if rate_scale is None else rate_scale
    frames = make_frames(batch, scale, rs)
    if int(frames[:, 4].max()) > sig_len:
        raise ValueError(f"cycle end {int(frames[:, 4].max())} exceeds sig_len {sig_len}")
    x = rs.standard_normal((batch, channels, sig_len)).astype(np.float32)
    t = np.arange(sig_len)[None, None, :]
    x[np.broadcast_to(t >= frames[:, 4][:, None, None], x.shape)] = 0.0
    labels = rs.randint(0, 2, size=batch).astype(np.int64)
    # recording ids: a/b/c… dataset letter + number, ~4 cycles per recording
    wav = tuple(f"{'abcdef'[i % 6]}{(i // 4):04d}" for i in range(batch))
    return x, frames, labels, wav


def make_index_data(batch: int, sig_len: int, sample_rate: int = 1000, seed: int = 0):
    """Boundaries, labels and recording ids only (for batches whose waveforms are generated
    directly on the device).  NOT the same label stream as ``make_batch`` (no waveform draw)."""
    rs = np.random.RandomState(seed)
    frames = make_frames(batch, sample_rate / 1000.0, rs)
    if int(frames[:, 4].max()) > sig_len:
        raise ValueError("cycle does not fit")
    labels = rs.randint(0, 2, size=batch).astype(np.int64)
    wav = tuple(f"{'abcdef'[i % 6]}{(i // 4):04d}" for i in range(batch))
    return frames, labels, wav


def spec_frames(frames: np.ndarray, n_cols: int, sig_len: int) -> np.ndarray:
    """Waveform boundaries -> spectrogram-column boundaries, Python banker's
    ``round`` as databuilder.ipynb cell 6:101 does."""
    out = np.zeros_like(frames)
    for i in range(frames.shape[0]):
        for k in range(frames.shape[1]):
            out[i, k] = int(round(int(frames[i, k]) * n_cols / sig_len))
    return out
