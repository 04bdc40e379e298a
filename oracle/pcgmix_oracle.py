"""CPU oracle for the PCGmix hot path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement, in this project's own words, of what the upstream reference computes
on the per-batch augmentation path.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package never
does (its kernels fail loudly when the HIP library is missing — there is no CPU
fallback).

Parity pinning: every function below is checked bit-for-bit against golden vectors
that were produced by running the real reference in the build container
(tests/golden/make_golden.py; tests/test_oracle_golden.py).  The log-mel restatement
(``logmel``) is the exception — librosa 0.9.2 is not available offline, the reference
holds no stored spectrogram, so that function is "parity unpinned" (SURVEY.md §8c).

The structure deliberately follows the reference (per-sample Python loop for the
splice, one scipy CubicSpline per (sample, channel) for the warp, one candidate loop per
(sample, state) for the saliency displacement) because it doubles as the timed CPU
baseline: its cost profile must be the reference's, not a vectorised
rewrite's.

Reference citations are ``file:line`` into the upstream repository.
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch
from scipy.interpolate import CubicSpline

PCGMIX_METHODS = ("durmixmagwarp", "durratiomixup")
PCGMIX_METHODS_2D = ("durmixcutout", "durmixtimemask", "durmixfreqmask", "durratiomixup")


# --------------------------------------------------------------------------- parsing
def parse_probability(method: str) -> float:
    """augmentations.py:865-868 / 932-935: text after the last '+' is the gate p."""
    parts = method.split("+")
    return float(parts[-1]) if len(parts) > 1 else 1.0


def gate_fires(method: str, step: int) -> bool:
    """augmentations.py:869-872 / 936-939: fresh Random(step); fire iff u < p."""
    return random.Random(step).uniform(0, 1) < parse_probability(method)


def parse_alpha(method: str, name: str) -> float:
    """augmentations.py:897-899 / 958-960."""
    parts = method.split("(alpha=")
    return float(parts[1].split(")" + name)[0]) if len(parts) > 1 else 1.0


def parse_magwarp(method: str):
    """augmentations.py:919-923: defaults sigma 0.2, 4 knots."""
    sigma, knot = 0.2, 4
    if len(method.split("durmixmagwarp(")) > 1:
        sigma = float(method.split("durmixmagwarp(")[1].split(",")[0])
        knot = int(method.split(",")[1].split(")")[0])
    return sigma, knot


# --------------------------------------------------------------------------- partners
def _shuffle_groups(keys, step: int) -> np.ndarray:
    """Group indices by key in order of first appearance; permute each group with a
    FRESH Random(step).sample (augmentations.py:500-514, 528-540, 542-556)."""
    groups = {}
    for i, k in enumerate(keys):
        groups.setdefault(k, []).append(i)
    mix = np.arange(len(keys))
    for idx in groups.values():
        mix[idx] = random.Random(step).sample(list(mix[idx]), len(idx))
    return mix


def mix_indices(method: str, labels: np.ndarray, wav, step: int) -> np.ndarray:
    """Partner selection with the reference's override order
    (augmentations.py:877-896 / 943-957); later substrings win."""
    mix = _shuffle_groups([int(v) for v in labels], step)            # same label :500
    if "(samePCG)" in method:
        mix = _shuffle_groups(list(wav), step)                       # same recording :528
    if "(sameDataset)" in method:
        mix = _shuffle_groups([f"{w[0]}_{int(t)}" for w, t in zip(wav, labels)], step)  # :542
    if "(mixAll)" in method:
        mix = np.asarray(random.Random(step).sample(list(np.arange(len(labels))), len(labels)))
    return mix


def get_lambda(alpha: float, step: int) -> float:
    """augmentations.py:659-666.  Reseeds the GLOBAL numpy stream (side effect kept:
    magnitude_warp's normal draw continues from here)."""
    if alpha > 0.0:
        np.random.seed(step)
        return np.random.beta(alpha, alpha)
    return 1.0


# --------------------------------------------------------------------------- splice
def splice_plain(d1, d2, f1, f2, lam, method: str, step: int):
    """augmentations.py:289-337 (1D) / augmentations2d.py:206-221 (2D, last axis).
    d1/d2 are torch CPU tensors (..., T); lam is a float32 tensor of shape (1,1[,1])."""
    out = d1.clone()
    for k in range(4):
        len1, len2 = int(f1[k + 1] - f1[k]), int(f2[k + 1] - f2[k])
        n = min(len1, len2)
        a, b = int(f1[k]), int(f2[k])
        if "(rand)" in method:
            gap = len2 - len1
            off = random.Random(step).randint(0, abs(gap))
            if gap >= 0:
                b += off
            else:
                a += off
        out[..., a:a + n] = out[..., a:a + n] * lam + d2[..., b:b + n] * (1 - lam)
    return out


def displacement_env(s1: np.ndarray, s2: np.ndarray, lam) -> int:
    """augmentations.py:60-93: first strict maximum of the float32 envelope sum."""
    n1, n2 = len(s1), len(s2)
    best, arg = float("-inf"), 0
    if n1 > n2:
        pad = np.pad(s2, (0, n1 - n2), "constant")
        for d in range(n1 - n2 + 1):
            sh = np.roll(pad, d)[d:d + n2]          # == s2; kept for the reference's cost profile
            cur = np.sum(s1[:d]) + np.sum(np.maximum(s1[d:d + n2], sh)) + np.sum(s1[d + n2:])
            if cur > best:
                best, arg = cur, d
    else:
        pad = np.pad(s1, (0, n2 - n1), "constant")
        for d in range(n2 - n1 + 1):
            sh = np.roll(pad, d)[d:d + n1]          # == s1
            cur = np.sum(np.maximum(s2[d:d + n1], sh))
            if cur > best:
                best, arg = cur, d
    return arg


def displacement_sum(s1: np.ndarray, s2: np.ndarray, lam) -> int:
    """augmentations.py:95-128.  ``lam`` is the (1,1) float32 array the reference passes
    (lams_out[0].numpy()), so the products broadcast to shape (1, n) before np.sum."""
    n1, n2 = len(s1), len(s2)
    best, arg = float("-inf"), 0
    if n1 > n2:
        pad = np.pad(s2, (0, n1 - n2), "constant")
        for d in range(n1 - n2 + 1):
            sh = np.roll(pad, d)[d:d + n2]          # == s2
            cur = np.sum(s1[:d]) + np.sum(s1[d:d + n2] * lam + sh * (1 - lam)) + np.sum(s1[d + n2:])
            if cur > best:
                best, arg = cur, d
    else:
        pad = np.pad(s1, (0, n2 - n1), "constant")
        for d in range(n2 - n1 + 1):
            sh = np.roll(pad, d)[d:d + n1]          # == s1
            cur = np.sum(sh * lam + s2[d:d + n1] * (1 - lam))
            if cur > best:
                best, arg = cur, d
    return arg


def displacement_objective(s1: np.ndarray, s2: np.ndarray, lam, d: int, method: str):
    """J(d) of ONE candidate exactly as the loops above form it (augmentations.py:70-76, 84-90,
    105-111, 119-125): numpy float32 sums.  Used by the tests to show that a displacement which
    differs from the reference's is a near-tie of the reference's own objective."""
    n1, n2 = len(s1), len(s2)
    env = "(saloptenv" in method
    if n1 > n2:
        mid = np.maximum(s1[d:d + n2], s2) if env else s1[d:d + n2] * lam + s2 * (1 - lam)
        return np.sum(s1[:d]) + np.sum(mid) + np.sum(s1[d + n2:])
    return np.sum(np.maximum(s2[d:d + n1], s1) if env else s1 * lam + s2[d:d + n1] * (1 - lam))


def salopt_displacements(sal1, sal2, f1, f2, lam_np, method: str) -> np.ndarray:
    """Displacement per heart state (0 where lengths agree); augmentations.py:210-287."""
    fn = displacement_env if "(saloptenv" in method else displacement_sum
    disp = np.zeros(4, dtype=np.int64)
    for k in range(4):
        len1, len2 = int(f1[k + 1] - f1[k]), int(f2[k + 1] - f2[k])
        if len1 != len2:
            disp[k] = fn(sal1[f1[k]:f1[k + 1]], sal2[f2[k]:f2[k + 1]], lam_np)
    return disp


def splice_salopt(d1, d2, f1, f2, sal1, sal2, lam, method: str):
    """augmentations.py:210-287: place the shorter state inside the longer one at the
    saliency-optimal displacement, then blend."""
    out = d1.clone()
    lam_np = lam.detach().cpu().numpy()
    disp = salopt_displacements(sal1, sal2, f1, f2, lam_np, method)
    for k in range(4):
        len1, len2 = int(f1[k + 1] - f1[k]), int(f2[k + 1] - f2[k])
        a, b, n = int(f1[k]), int(f2[k]), min(len1, len2)
        if len1 > len2:
            a += int(disp[k])
        elif len1 < len2:
            b += int(disp[k])
        out[..., a:a + n] = out[..., a:a + n] * lam + d2[..., b:b + n] * (1 - lam)
    return out, disp


# --------------------------------------------------------------------------- warp
def magnitude_warp(x: np.ndarray, sigma: float, knot: int, return_knots: bool = False):
    """augmentations.py:674-683.  x is (B, T, C) float32; draws from the GLOBAL numpy
    stream; float64 spline * float32 sample, rounded once into a float32 array."""
    B, T, C = x.shape
    steps = np.arange(T)
    knots = np.random.normal(loc=1.0, scale=sigma, size=(B, knot + 2, C))
    brk = np.linspace(0, T - 1.0, num=knot + 2)
    ret = np.zeros_like(x)
    for i in range(B):
        w = np.array([CubicSpline(brk, knots[i, :, c])(steps) for c in range(C)]).T
        ret[i] = x[i] * w
    return (ret, knots) if return_knots else ret


# --------------------------------------------------------------------------- saliency
def gaussian_taps(n: int, sigma: float):
    """saliency.py:15-18 (not renormalised)."""
    return [1 / (sigma * math.sqrt(2 * math.pi)) * math.exp(-float(r) ** 2 / (2 * sigma ** 2))
            for r in range(-int(n / 2), int(n / 2) + 1)]


def saliency_post(grad: np.ndarray, frames: np.ndarray, gauss_k_n: int = 101) -> np.ndarray:
    """saliency.py:63-91 (dim=1): |grad| -> zero tail -> sum channels -> Gaussian
    smoothing ('same', zero padded) -> zero tail -> per-row (s-min)/max -> NaN->0.
    Uses torch CPU ops, as the reference does, so the float32 results are its results."""
    sal = torch.from_numpy(np.abs(grad))
    for s, f in zip(sal, frames):
        s[:, int(f[-1]):] = 0
    sal = torch.sum(sal, dim=1)[:, None, :]
    sigma = (12 / 101) * gauss_k_n
    kern = torch.FloatTensor([[gaussian_taps(gauss_k_n, sigma)]])
    sal = torch.nn.functional.conv1d(sal, kern, padding="same")
    for s, f in zip(sal, frames):
        s[:, int(f[-1]):] = 0
    shape = sal.size()
    sal = sal.view(sal.size(0), -1)
    sal -= sal.min(1, keepdim=True)[0]
    sal /= sal.max(1, keepdim=True)[0]
    sal = torch.nan_to_num(sal.view(shape), nan=0.0)
    return np.squeeze(sal.numpy())


def saliency_post2d(grad: np.ndarray, frames: np.ndarray) -> np.ndarray:
    """saliency.py:93-113 (dim=2): |grad| (B,1,F,W) -> zero the columns t >= f[-1] -> sum over the
    frequency axis -> 11-tap Gaussian, sigma 1 ('same', zero padded) -> per sample: zero the tail,
    then (s - min)/max over the cycle's OWN columns only -> NaN->0.  torch CPU ops, as the
    reference."""
    sal = torch.from_numpy(np.abs(grad))
    for s, f in zip(sal, frames):
        s[:, :, int(f[-1]):] = 0
    sal = torch.sum(sal, dim=2)[:, None, :]
    sal = torch.squeeze(sal, 2)
    kern = torch.FloatTensor([[gaussian_taps(11, 1)]])
    sal = torch.nn.functional.conv1d(sal, kern, padding="same")
    for s, f in zip(sal, frames):
        e = int(f[-1])
        s[:, e:] = 0
        s[:, :e] -= s[:, :e].min()
        s[:, :e] /= s[:, :e].max()
    sal = torch.nan_to_num(sal, nan=0.0)
    return np.squeeze(sal.numpy())


def input_gradient(model: torch.nn.Module, x: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """saliency.py:52-61: d(score of the true class)/d(input), model in eval mode."""
    model.eval()
    data = torch.from_numpy(x.copy()).requires_grad_()
    out = model(data)
    scores = out.gather(1, torch.from_numpy(labels).view(-1, 1)).squeeze()
    scores.backward(torch.ones_like(scores))
    return data.grad.detach().numpy()


# --------------------------------------------------------------------------- augment
def augment(method: str, x: np.ndarray, labels: np.ndarray, frames: np.ndarray, wav,
            step: int, saliency_maps: np.ndarray | None = None, num_classes: int = 2):
    """The durmixmagwarp / durratiomixup branches of augmentations.py:864-981 (1D, x is
    (B,C,T)) and the durratiomixup branch of augmentations2d.py:397-427 (x is (B,1,F,W); with
    '(saloptenv' / '(saloptsum' in the method the saliency-guided splice of :125-204, maps (B,W)).

    Returns dict(y, target, mix, fired, lam, knots, disp).  ``y is x`` when the method
    does not apply or the gate rejects (the reference returns the input object).
    """
    target = np.eye(num_classes, dtype=np.int64)[labels]
    res = dict(y=x, target=target, mix=np.zeros(0, np.int64), fired=False, lam=float("nan"),
               knots=np.zeros(0), disp=None)
    is2d = x.ndim == 4
    # 2D dispatch order: augmentations2d.py:286 (cutout), :325 (timemask), :361 (freqmask), :397
    names = PCGMIX_METHODS_2D if is2d else PCGMIX_METHODS
    name = next((m for m in names if m in method), None)           # dispatch order :864,:931
    if name is None or not gate_fires(method, step):
        return res
    B = x.shape[0]
    # the reference recovers the labels from the one-hot tensor with torch (:501); kept for the
    # cost profile (bench.py times this function as the CPU baseline)
    labels = torch.from_numpy(target).max(1, keepdim=True)[1].numpy()[:, 0]
    mix = mix_indices(method, labels, wav, step) if not is2d else \
        _shuffle_groups([int(v) for v in labels], step)              # augmentations2d.py:410
    alpha = parse_alpha(method, name) if not is2d else 1.0            # augmentations2d.py:411
    lam64 = get_lambda(alpha, step)
    lams = torch.from_numpy((np.ones(B) * lam64).astype("float32"))
    lam = lams[:, None, None, None][0] if is2d else lams[:, None, None][0]
    data = torch.from_numpy(x)
    y = torch.zeros(x.shape)
    disp = np.zeros((B, 4), dtype=np.int64)
    partners, partner_frames = data[mix], frames[mix]     # gathered copies, as :909 / :970 make
    # 2D: only the durratiomixup branch looks at '(salopt' (augmentations2d.py:416-423)
    salopt = "(salopt" in method and (not is2d or name == "durratiomixup")
    for i, (d1, f1, d2, f2) in enumerate(zip(data, frames, partners, partner_frames)):
        if salopt:
            y[i], disp[i] = splice_salopt(d1, d2, f1, f2, saliency_maps[i],
                                          saliency_maps[mix][i], lam, method)
        else:
            y[i] = splice_plain(d1, d2, f1, f2, lam,
                                "" if is2d else method, step)    # 2D has no (rand) variant
    if "(mixAll)" in method and not is2d:                             # :915-917 / :978-980
        lt = lams[:, None]
        t = torch.from_numpy(target)
        target = (t * lt + t[mix] * (1 - lt)).numpy()
    knots = np.zeros(0)
    y = y.numpy()
    if is2d and name != "durratiomixup":
        y = mask_2d(y, frames, method, name, step)
    if name == "durmixmagwarp":                                       # :919-928
        sigma, knot = parse_magwarp(method)
        y, knots = magnitude_warp(np.transpose(y, (0, 2, 1)), sigma, knot, return_knots=True)
        y = np.ascontiguousarray(np.transpose(y, (0, 2, 1)))
    res.update(y=y, target=target, mix=mix, fired=True, lam=lam64, knots=knots,
               disp=disp if salopt else None)
    return res


def mask_2d(y: np.ndarray, frames: np.ndarray, method: str, name: str, step: int) -> np.ndarray:
    """Zeroed rectangle after the 2D splice: durmixcutout / durmixtimemask / durmixfreqmask,
    augmentations2d.py:309-323, 348-358, 384-394.  Region sizes come from
    Random(step+131071).uniform, positions from Random(step+13119).uniform; the time span is a
    fraction of each sample's own cycle length, the frequency span is the same for the batch."""
    F = y.shape[2]

    def clamp01(v):
        return min(max(v, 0), 1)
    t_max = f_max = 0.2
    key = name[len("durmix"):] + "("
    if len(method.split(key)) > 1:
        if name == "durmixcutout":
            t_max = clamp01(float(method.split(key)[1].split(",")[0]))
            f_max = clamp01(float(method.split(",")[1].split(")")[0]))
        else:
            t_max = f_max = clamp01(float(method.split(key)[1].split(")")[0]))
    y = y.copy()
    if name in ("durmixcutout", "durmixtimemask"):
        gap = random.Random(step + 131071).uniform(0, t_max)
        frac1 = random.Random(step + 13119).uniform(0, 1 - gap)
        frac2 = frac1 + gap
    if name in ("durmixcutout", "durmixfreqmask"):
        fgap = random.Random(step + 131071).uniform(0, f_max)
        h1 = int(F * random.Random(step + 13119).uniform(0, 1 - fgap))
        h2 = min(F, h1 + int(fgap * F))
    for i in range(y.shape[0]):
        beat = frames[i][-1]
        if name == "durmixfreqmask":
            y[i, :, h1:h2, :] = 0
        else:
            c0, c1 = int(frac1 * beat), int(frac2 * beat)
            if name == "durmixtimemask":
                y[i, :, :, c0:c1] = 0
            else:
                y[i, :, h1:h2, c0:c1] = 0
    return y


# --------------------------------------------------------------------------- loss
def ce_soft(logits: np.ndarray, target_ohe: np.ndarray) -> float:
    """train_model.py:45-54: mean over the batch of -sum(log_softmax * target)."""
    z = logits - logits.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    return float((-(logp * target_ohe).sum(1)).mean())


# --------------------------------------------------------------------------- log-mel (UNPINNED)
# Restatement of librosa 0.9.2 as called by databuilder.ipynb cell 6:81-101, 127-142.  librosa is
# not available offline and the reference stores no spectrogram, so NOTHING below is checked
# against the reference's own output: "parity unpinned".  The HIP kernel is tested against this
# restatement only.
LOGMEL_MEAN = -59.606563568115234        # databuilder.ipynb cell 6 (train_mean, train_std)
LOGMEL_STD = 15.96771240234375


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=float)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep,
                    f / f_sp)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=float)
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft, n_mels, fmin, fmax):
    """librosa.filters.mel(htk=False, norm='slaney', dtype=float32)."""
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def _mel_db(y: np.ndarray, n_fft, hop, basis, pad_mode):
    """librosa.feature.melspectrogram(center=True, power=2) of one signal followed by the
    un-referenced part of power_to_db: returns (mel power (n_mels, n_frames) float32, 10*log10(
    max(amin, .)) of it).  ``pad_mode``: numpy.pad mode of the centred frames ('constant' or
    'reflect'; which one librosa 0.9.2 defaults to is the open point of this restatement)."""
    n_frames = 1 + len(y) // hop
    n = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    yp = np.pad(y, n_fft // 2, mode=pad_mode)
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    S = np.fft.rfft(window[:, None] * yp[idx], axis=0).astype(np.complex64)
    P = np.abs(S) ** 2.0
    mel = np.einsum("ft,mf->mt", P, basis, optimize=True)
    return mel, 10.0 * np.log10(np.maximum(1e-10, mel))


def stft_complex(y: np.ndarray, n_fft=136, hop=34, pad_mode="constant") -> np.ndarray:
    """The STFT stage alone, float64 (before librosa's complex64 rounding): cross-checked against
    torch.stft in tests/test_oracle_golden.py for both pad modes."""
    n_frames = 1 + len(y) // hop
    n = np.arange(n_fft)
    window = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)
    yp = np.pad(y.astype(np.float64), n_fft // 2, mode=pad_mode)
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    return np.fft.rfft(window[:, None] * yp[idx], axis=0)


def logmel(x: np.ndarray, frames: np.ndarray, n_fft=136, hop=34, n_mels=128, fmin=25.0,
           fmax=1000.0, sr=2000.0, mean=LOGMEL_MEAN, std=LOGMEL_STD, W=128, pad_mode="constant"):
    """x (B,T) float32 heart cycles, frames (B,5) -> (spec (B,n_mels,W) float32, frames_spec).
    Per item: centred STFT (``pad_mode`` padding, periodic Hann, float64 FFT rounded to
    complex64), power, Slaney mel, power_to_db(ref=np.max over THIS item, amin 1e-10, top_db 80),
    (x-mean)/std, keep columns < round(f4*n_frames/T), zero-fill up to W."""
    B, T = x.shape
    n_frames = 1 + T // hop
    basis = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    out = np.zeros((B, n_mels, W), dtype=np.float32)
    fspec = np.zeros_like(frames)
    for b in range(B):
        mel, db = _mel_db(x[b], n_fft, hop, basis, pad_mode)
        db -= 10.0 * np.log10(np.maximum(1e-10, np.max(mel)))
        db = np.maximum(db, db.max() - 80.0)
        db = ((db - mean) / std).astype(np.float32)
        fspec[b] = [int(round(int(f) * n_frames / T)) for f in frames[b]]
        c4 = min(int(fspec[b, 4]), W)
        out[b, :, :c4] = db[:, :c4]
    return out, fspec


def logmel_recording(y: np.ndarray, boundaries, seg_starts, n_fft=136, hop=34, n_mels=128,
                     fmin=25.0, fmax=1000.0, sr=2000.0, mean=LOGMEL_MEAN, std=LOGMEL_STD, W=128,
                     pad_mode="constant"):
    """databuilder.ipynb cell 6:81-101, 127-142 for ONE recording y: mel spectrogram of the whole
    recording, power_to_db(ref=np.max) over the whole recording (:93), (x-mean)/std (:99), column
    boundaries round(f*n_frames/len(y)) (:101); per cycle starting at boundary index i: columns
    [fs[i], fs[i+4]) (:134), zero-padded on the right to W columns (:141-142; a slice wider than
    W is cut at W here — the reference would store the wider image).
    Returns (specs (n_cycles, n_mels, W) float32, frames_spec (n_cycles, 5) cycle-relative)."""
    basis = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    mel, db = _mel_db(y, n_fft, hop, basis, pad_mode)
    db -= 10.0 * np.log10(np.maximum(1e-10, np.max(mel)))
    db = np.maximum(db, db.max() - 80.0)
    db = ((db - mean) / std).astype(np.float32)
    fs = [round(int(f) * db.shape[1] / len(y)) for f in boundaries]
    specs = np.zeros((len(seg_starts), n_mels, W), dtype=np.float32)
    rel = np.zeros((len(seg_starts), 5), dtype=np.int64)
    for j, i in enumerate(seg_starts):
        sl = db[:, fs[i]:fs[i + 4]][:, :W]
        specs[j, :, :sl.shape[1]] = sl
        rel[j] = np.asarray(fs[i:i + 5]) - fs[i]
    return specs, rel
