"""On-device STFT -> log-mel front end for the 2D (spectrogram) path.

The reference builds its 128x128 log-mel images offline with librosa
(databuilder.ipynb cell 6:19-23, 81-101, 127-142) and loads them from a pickle
(dataloader_physionet2d.py:22-32).  Here the transform runs per batch on the GPU
(``pcgmix_logmel_f32``) from the 2 kHz heart-cycle waveform, with the reference's settings:
hop = int(2000*2.2/128) = 34, n_fft = 4*hop = 136, 128 mel bands from 25 Hz to 1 kHz, dB
relative to the maximum, 80 dB floor, fixed mean/std normalisation, the cycle's columns kept and
zero-filled up to 128.  Parity with librosa itself is UNPINNED: librosa is not
available offline and the reference stores no spectrogram.

Two entry points:

``logmel``             per heart-cycle item of a batch (already cut cycles, e.g. the time-series
                       loader's output): one transform per item, dB relative to the item's own
                       maximum.
``logmel_recordings``  the reference's own order of operations (cell 6:81-101, 127-142): one
                       transform over each WHOLE recording, dB relative to the recording's
                       maximum, then every cycle's columns ``[round(f0*n/len), round(f4*n/len))``
                       are sliced out and zero-padded.  Cycles cut this way differ from ``logmel``
                       of the same samples at the cycle edges (neighbouring samples instead of
                       padding) and in the dB reference.

``pad_mode`` ('constant' = zeros, or 'reflect') is the padding of the centred frames at the
signal's ends.  Which one librosa 0.9.2's ``melspectrogram`` defaults to could not be verified
offline; 'constant' (librosa's default since its 0.9 series, to the best of our knowledge) is
the default here and 'reflect' (the 0.8 default) is selectable.  It only affects the first and
last two frames of a signal.
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


PAD_MODES = {"constant": 0, "reflect": 1}
DEFAULT_PAD_MODE = "constant"


def _pad_code(pad_mode: str) -> int:
    try:
        return PAD_MODES[pad_mode]
    except KeyError:
        raise ValueError(f"pad_mode must be one of {sorted(PAD_MODES)}, got {pad_mode!r}") from None


def logmel(x: torch.Tensor, frames, sample_rate: int = 2000, n_mels: int = SPEC_FRAMES,
           width: int = SPEC_FRAMES, mean: float = TRAIN_MEAN, std: float = TRAIN_STD,
           pad_mode: str = DEFAULT_PAD_MODE):
    """x: float32 device tensor (B, T) or (B, 1, T); frames: (B,5) host boundaries.
    Returns (spec (B,1,n_mels,width) on device, frames_spec int64 (B,5) host array)."""
    pad_code = _pad_code(pad_mode)
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
        spec = torch.empty((B, 1, n_mels, width), dtype=torch.float32, device=x.device)
        stream = _raw_stream(x.device)
        fr32 = np.ascontiguousarray(frames_np, dtype=np.int32)
        if fr32.shape != (B, 5):
            raise ValueError("frames must be (B, 5)")
        if B <= 1024 and T <= 32767:
            # the cycle ends ride in the kernel arguments: no upload, no copy kernel before the launch
            _lib.check(lib.pcgmix_logmel_hostframes_f32(
                x.data_ptr(), fr32.ctypes.data, tables.data_ptr(), spec.data_ptr(), B, T, n_fft, hop,
                n_mels, ctypes.c_float(mean), ctypes.c_float(std), width, pad_code,
                ctypes.c_void_p(stream)), "pcgmix_logmel_hostframes_f32")
        else:
            fr = upload_array(fr32, x.device)
            _lib.check(lib.pcgmix_logmel_f32(x.data_ptr(), fr.data_ptr(), tables.data_ptr(),
                                             spec.data_ptr(), None, B, T, n_fft, hop, n_mels,
                                             ctypes.c_float(mean), ctypes.c_float(std), width, pad_code,
                                             ctypes.c_void_p(stream)), "pcgmix_logmel_f32")
    return spec, spec_frames(frames_np, T, hop)


def recording_plan(lengths, boundaries, seg_starts, hop: int, width: int, tile_frames: int):
    """Integer bookkeeping of the per-recording front end (host, O(cycles)): for recording r of
    ``lengths[r]`` samples, ``boundaries[r]`` are its heart-state boundaries in samples
    (databuilder.ipynb cell 6:51-52) and ``seg_starts[r]`` the indices i of the boundaries at which
    a heart cycle starts (cell 6:57-72): cycle = boundaries[i : i+5].

    Returns dict(rec_off, rec_len, tiles (n_tiles,4), cycles (n_cycles,4), scratch_cols,
    frames_spec (n_cycles,5) cycle-relative column boundaries as cell 6:130, rec_of_cycle).
    ``frames_spec = round(f * n_frames / len(y))`` with Python's round-half-even, cell 6:101."""
    rec_off, tiles, cycles, fspec, rec_of = [], [], [], [], []
    off = col = 0
    five = np.arange(5, dtype=np.int64)
    for r, n in enumerate(lengths):                 # numpy per recording: O(recordings) Python steps
        n = int(n)
        n_frames = 1 + n // hop
        rec_off.append(off)
        f0 = np.arange(0, n_frames, tile_frames, dtype=np.int64)
        tiles.append(np.stack([np.full_like(f0, r), f0, np.minimum(tile_frames, n_frames - f0), col + f0], 1))
        b = np.asarray(boundaries[r], dtype=np.int64)
        cols = np.rint(b * n_frames / float(n)).astype(np.int64)          # cell 6:101
        st = np.asarray(seg_starts[r], dtype=np.int64).reshape(-1)
        if st.size:
            c = cols[st[:, None] + five]                                   # (cycles, 5)
            keep = np.minimum(np.minimum(np.maximum(c[:, 4] - c[:, 0], 0), width), n_frames - c[:, 0])
            cycles.append(np.stack([np.full_like(keep, r), col + c[:, 0], np.maximum(keep, 0),
                                    np.zeros_like(keep)], 1))
            fspec.append(c - c[:, :1])                                     # cell 6:130
            rec_of.append(np.full(st.size, r, dtype=np.int64))
        off += n
        col += n_frames
    cat = lambda parts, width_: (np.concatenate(parts) if parts else np.zeros((0, width_), np.int64))
    return {"rec_off": np.asarray(rec_off, dtype=np.int64),
            "rec_len": np.asarray(lengths, dtype=np.int32),
            "tiles": cat(tiles, 4).astype(np.int32).reshape(-1, 4),
            "cycles": cat(cycles, 4).astype(np.int32).reshape(-1, 4),
            "scratch_cols": int(col),
            "frames_spec": cat(fspec, 5).astype(np.int64).reshape(-1, 5),
            "rec_of_cycle": (np.concatenate(rec_of) if rec_of else np.zeros(0, np.int64))}


def logmel_recordings(y: torch.Tensor, lengths, boundaries, seg_starts, sample_rate: int = 2000,
                      n_mels: int = SPEC_FRAMES, width: int = SPEC_FRAMES, mean: float = TRAIN_MEAN,
                      std: float = TRAIN_STD, pad_mode: str = DEFAULT_PAD_MODE):
    """The reference's per-recording log-mel (databuilder.ipynb cell 6:81-101, 127-142) on device.

    y: float32 device tensor, the recordings back to back (sum(lengths) samples);
    lengths / boundaries / seg_starts: see ``recording_plan``.
    Returns (spec (n_cycles,1,n_mels,width) on device, frames_spec int64 (n_cycles,5) host,
    rec_of_cycle int64 (n_cycles,) host)."""
    pad_code = _pad_code(pad_mode)
    if y.dim() != 1 or y.dtype != torch.float32 or not y.is_contiguous() or not y.is_cuda:
        raise ValueError("y must be a contiguous float32 1-D device tensor")
    if int(np.sum(lengths)) != y.numel():
        raise ValueError("lengths do not add up to the number of samples in y")
    n_fft, hop = stft_params(sample_rate)
    if min(int(n) for n in lengths) <= n_fft // 2:
        raise ValueError("a recording is shorter than half a transform window")
    lib = _lib.load()
    plan = recording_plan(lengths, boundaries, seg_starts, hop, width, lib.pcgmix_logmel_tile_frames())
    n_cycles = plan["cycles"].shape[0]
    dev = y.device
    with torch.cuda.device(dev):
        tables = logmel_tables(dev, n_fft, n_mels, sample_rate)
        rec_off = upload_array(plan["rec_off"], dev)
        rec_len = upload_array(plan["rec_len"], dev)
        tiles = upload_array(plan["tiles"], dev)
        cycles = upload_array(plan["cycles"] if n_cycles else np.zeros((1, 4), np.int32), dev)
        scratch = torch.empty((n_mels, plan["scratch_cols"]), dtype=torch.float32, device=dev)
        ref_pow = torch.empty(len(lengths), dtype=torch.int32, device=dev)
        spec = torch.empty((n_cycles, 1, n_mels, width), dtype=torch.float32, device=dev)
        _lib.check(lib.pcgmix_logmel_recordings_f32(
            y.data_ptr(), rec_off.data_ptr(), rec_len.data_ptr(), len(lengths), tiles.data_ptr(),
            plan["tiles"].shape[0], cycles.data_ptr(), n_cycles, tables.data_ptr(), scratch.data_ptr(),
            plan["scratch_cols"], ref_pow.data_ptr(), spec.data_ptr() if n_cycles else scratch.data_ptr(),
            n_fft, hop, n_mels, ctypes.c_float(mean), ctypes.c_float(std), width, pad_code,
            ctypes.c_void_p(_raw_stream(dev))), "pcgmix_logmel_recordings_f32")
    return spec, plan["frames_spec"], plan["rec_of_cycle"]
