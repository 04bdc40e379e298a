"""On-device STFT -> log-mel front end for the 2D (spectrogram) path.

The reference builds its 128x128 log-mel images offline with librosa
(databuilder.ipynb cell 6:19-23, 81-101, 127-142) and loads them from a pickle
(dataloader_physionet2d.py:22-32).  Here the transform runs per batch on the GPU
(``pcgmix_logmel_f32``) from the 2 kHz heart-cycle waveform, with the reference's settings:
hop = int(2000*2.2/128) = 34, n_fft = 4*hop = 136, 128 mel bands from 25 Hz to 1 kHz, dB
relative to the item's maximum, 80 dB floor, fixed mean/std normalisation, the cycle's columns
kept and zero-filled up to 128.  Parity with librosa itself is unpinned.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from .augmentations import _raw_stream, upload_array

SPEC_LEN_S = 2.2                         # databuilder.ipynb cell 6:19
SPEC_FRAMES = 128                        # cell 6:22
FMIN, FMAX = 25.0, 1000.0                # cell 6:20-21
TRAIN_MEAN = -59.606563568115234         # fixed normalisation constants of the reference
TRAIN_STD = 15.96771240234375


def stft_params(sample_rate: int):
    hop = int(sample_rate * SPEC_LEN_S / SPEC_FRAMES)          # cell 6:83
    return 4 * hop, hop                                        # n_fft = hop*4, cell 6:86


def spec_frames(frames: np.ndarray, sig_len: int, hop: int) -> np.ndarray:
    """Waveform boundaries -> spectrogram columns: ``round(f * n_frames / len(y))`` with Python's
    round-half-even (cell 6:101)."""
    n_frames = 1 + sig_len // hop
    v = frames.astype(np.int64) * n_frames / float(sig_len)
    return np.rint(v).astype(np.int64)


_TABLES: dict = {}     # (device index, n_fft, n_mels, fmin, fmax, sr) -> device blob


def logmel_tables(device: torch.device, n_fft: int, n_mels: int, sample_rate: float) -> torch.Tensor:
    """Constant tables of the transform (windowed DFT matrix in MFMA operand order, Slaney filter
    bank, filter spans), built on the host by ``pcgmix_logmel_tables`` and kept on ``device``."""
    key = (device.index, n_fft, n_mels, FMIN, FMAX, float(sample_rate))
    blob = _TABLES.get(key)
    if blob is None:
        lib = _lib.load()
        host = np.zeros(lib.pcgmix_logmel_tables_size(n_fft, n_mels), dtype=np.uint8)
        _lib.check(lib.pcgmix_logmel_tables(n_fft, n_mels, ctypes.c_float(FMIN), ctypes.c_float(FMAX),
                                            ctypes.c_float(sample_rate), host.ctypes.data),
                   "pcgmix_logmel_tables")
        blob = torch.from_numpy(host).to(device)
        _TABLES[key] = blob
    return blob


def logmel(x: torch.Tensor, frames, sample_rate: int = 2000, n_mels: int = SPEC_FRAMES,
           width: int = SPEC_FRAMES, mean: float = TRAIN_MEAN, std: float = TRAIN_STD):
    """x: float32 device tensor (B, T) or (B, 1, T); frames: (B,5) host boundaries.
    Returns (spec (B,1,n_mels,width) on device, frames_spec int64 (B,5) host array)."""
    if x.dim() == 3:
        if x.shape[1] != 1:
            raise ValueError("log-mel takes one channel per item")
        x = x[:, 0, :]
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_contiguous() or not x.is_cuda:
        raise ValueError("x must be a contiguous float32 (B, T) device tensor")
    B, T = x.shape
    frames_np = frames.detach().cpu().numpy() if isinstance(frames, torch.Tensor) else np.asarray(frames)
    n_fft, hop = stft_params(sample_rate)
    lib = _lib.load()
    with torch.cuda.device(x.device):
        tables = logmel_tables(x.device, n_fft, n_mels, sample_rate)
        fr = upload_array(frames_np.astype(np.int32), x.device)
        spec = torch.empty((B, 1, n_mels, width), dtype=torch.float32, device=x.device)
        stream = _raw_stream(x.device)
        _lib.check(lib.pcgmix_logmel_f32(x.data_ptr(), fr.data_ptr(), tables.data_ptr(),
                                         spec.data_ptr(), None, B, T, n_fft, hop, n_mels,
                                         ctypes.c_float(mean), ctypes.c_float(std), width,
                                         ctypes.c_void_p(stream)), "pcgmix_logmel_f32")
    return spec, spec_frames(frames_np, T, hop)
