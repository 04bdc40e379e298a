"""Parity of the HIP splice/warp kernel (through the drop-in augment() and the C ABI) with the
golden vectors produced by the reference and with the CPU oracle.

Bars (BASELINE.json north_star): partner/segment indices bit-exact; mixed waveforms within
1e-4.  The splice alone is in fact bit-exact (same three fp32 roundings); the fp64 spline is
evaluated from a precomputed linear operator, so warped outputs may differ from scipy's by
one float32 ulp in rare elements — the tolerance below states that."""
import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import augmentations, augmentations2d, synthetic
from conftest import Args, StepCounter, golden_files, load_golden
from oracle import pcgmix_oracle as O

pytestmark = pytest.mark.gpu
WAVE_TOL = 1e-4          # north_star tolerance on mixed waveforms / spectrograms


def run(mod, g, device, method=None, x=None):
    x = g["x"] if x is None else x
    data = torch.from_numpy(x).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(g["labels"]), 2).to(device)
    out = mod.augment(Args(method or g["method"]), data, tgt, torch.from_numpy(g["frames"]),
                      g["wav"], StepCounter(g["step"]), None, device, "")
    return data, tgt, out


@pytest.mark.parametrize("path", golden_files("mix1d_") + golden_files("mix2d_") + golden_files("mask2d_"),
                         ids=lambda p: p.split("/")[-1][:-4])
def test_augment_matches_reference_goldens(path, device):
    g = load_golden(path)
    mod = augmentations2d if g["x"].ndim == 4 else augmentations
    data, tgt, (y, t_out, mix, cut) = run(mod, g, device)
    assert cut is None
    if not g["fired"]:
        assert y is data and t_out is tgt and mix == []          # same objects, as the reference
        return
    assert y is not data and y.data_ptr() != data.data_ptr()
    assert torch.equal(data.cpu(), torch.from_numpy(g["x"]))     # input untouched
    assert np.array_equal(mix, g["mix"])                         # indices: bit-exact
    got = y.cpu().numpy()
    if "magwarp" in g["method"]:
        assert np.abs(got - g["y"]).max() <= WAVE_TOL
        ulp = np.abs(got.view(np.int32).astype(np.int64) - g["y"].view(np.int32).astype(np.int64))
        assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3
    else:
        assert np.array_equal(got, g["y"])                       # splice: bit-exact
    assert np.array_equal(t_out.cpu().numpy().astype(np.float64), g["target_out"].astype(np.float64))


@pytest.mark.parametrize("method", ["durratiomixup", "durmixmagwarp(0.2,4)", "(rand)durmixmagwarp(0.1,3)"])
@pytest.mark.parametrize("shape", [(6, 3, 637), (5, 1, 1001), (4, 2, 130)])
def test_odd_lengths_scalar_path(method, shape, device):
    """T % 4 != 0 takes the one-element-per-lane kernel."""
    B, C, T = shape
    x, frames, labels, wav = synthetic.make_batch(B, C, T, seed=3, rate_scale=T / 1400.0)
    ref = O.augment(method, x, labels, frames, wav, 17)
    g = dict(x=x, labels=labels, frames=frames, wav=wav, step=17, method=method)
    _, _, (y, _, mix, _) = run(augmentations, g, device)
    assert np.array_equal(mix, ref["mix"])
    assert np.abs(y.cpu().numpy() - ref["y"]).max() <= (WAVE_TOL if "magwarp" in method else 0.0)


@pytest.mark.parametrize("method,shape", [("durratiomixup", (256, 4, 5000)),
                                          ("durmixmagwarp(0.2,4)", (256, 1, 5000)),
                                          ("durratiomixup", (32, 4, 2500))])
def test_full_size_against_oracle(method, shape, device):
    """BASELINE.json configs at full size; the oracle finishes these in well under a second."""
    B, C, T = shape
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000 if T == 5000 else 1000, seed=0)
    ref = O.augment(method, x, labels, frames, wav, 5)
    g = dict(x=x, labels=labels, frames=frames, wav=wav, step=5, method=method)
    _, _, (y, _, mix, _) = run(augmentations, g, device)
    assert np.array_equal(mix, ref["mix"])
    got = y.cpu().numpy()
    if "magwarp" in method:
        assert np.abs(got - ref["y"]).max() <= WAVE_TOL
    else:
        assert np.array_equal(got, ref["y"])


def test_properties_large_batch(device):
    """Size-independent properties at a batch the oracle is not run on (B=4096):
    (1) partners are a same-label permutation; (2) outside every blended range y == x bit for
    bit; (3) a sample whose partner is itself is x*lam + x*(1-lam) inside its cycle; (4) the
    zero padding stays zero; (5) a second call with the same step is bit-identical."""
    B, C, T = 4096, 1, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=1)
    labels[:] = 0
    labels[7] = 1                                   # singleton class -> partner is itself
    g = dict(x=x, labels=labels, frames=frames, wav=wav, step=3, method="durratiomixup")
    _, _, (y, _, mix, _) = run(augmentations, g, device)
    _, _, (y2, _, _, _) = run(augmentations, g, device)
    assert torch.equal(y, y2)
    y = y.cpu().numpy()
    assert sorted(mix.tolist()) == list(range(B)) and (labels[mix] == labels).all() and mix[7] == 7
    lens = np.diff(frames, axis=1)
    n = np.minimum(lens, lens[mix])
    t = np.arange(T)[None, :]
    blended = np.zeros((B, T), bool)
    for k in range(4):
        blended |= (t >= frames[:, k:k + 1]) & (t < frames[:, k:k + 1] + n[:, k:k + 1])
    assert np.array_equal(y[:, 0][~blended], x[:, 0][~blended])
    assert (y[:, 0][t >= frames[:, 4:5]] == 0).all()
    lam = np.float32(O.get_lambda(1.0, 3))
    own = x[7, 0, :frames[7, 4]]
    assert np.array_equal(y[7, 0, :frames[7, 4]], own * lam + own * (np.float32(1) - lam))


def test_rejects_bad_inputs(device):
    x, frames, labels, wav = synthetic.make_batch(4, 1, 2500, seed=2)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
    a = Args("durratiomixup")
    with pytest.raises(ValueError):
        augmentations.augment(a, torch.from_numpy(x), tgt, frames, wav, StepCounter(0), None, device, "")
    with pytest.raises(ValueError):
        augmentations.augment(a, torch.from_numpy(x).to(device).double(), tgt, frames, wav,
                              StepCounter(0), None, device, "")
    with pytest.raises(ValueError):
        augmentations.augment(a, torch.from_numpy(x).to(device).transpose(0, 1), tgt, frames, wav,
                              StepCounter(0), None, device, "")
    bad = frames.copy(); bad[0, 4] = 2501
    with pytest.raises(ValueError):
        augmentations.augment(a, torch.from_numpy(x).to(device), tgt, bad, wav, StepCounter(0),
                              None, device, "")


# ------------------------------------------------------------------ edge cases
def _aug(method, x, frames, labels, wav, step, device):
    g = dict(x=x, labels=labels, frames=frames, wav=wav, step=step, method=method)
    _, _, (y, _, mix, _) = run(augmentations, g, device)
    return y.cpu().numpy(), mix


@pytest.mark.parametrize("method", ["durratiomixup", "durmixmagwarp(0.2,4)"])
def test_degenerate_frames(method, device):
    """Zero-length states, a cycle that fills the whole row, an empty cycle, batch of one."""
    T = 512
    rs = np.random.RandomState(0)
    frames = np.array([[0, 0, 100, 100, 300],        # S1 and S2 empty
                       [0, 50, 50, 200, 512],        # systole empty, cycle ends exactly at T
                       [0, 0, 0, 0, 0],              # empty cycle
                       [0, 120, 260, 330, 500],
                       [0, 1, 2, 3, 4]], dtype=np.int64)
    B = frames.shape[0]
    x = rs.standard_normal((B, 3, T)).astype(np.float32)
    labels = np.zeros(B, dtype=np.int64)
    wav = tuple("abcde")
    for step in (0, 3):
        ref = O.augment(method, x, labels, frames, wav, step)
        y, mix = _aug(method, x, frames, labels, wav, step, device)
        assert np.array_equal(mix, ref["mix"])
        assert np.abs(y - ref["y"]).max() <= (WAVE_TOL if "magwarp" in method else 0.0)
    # batch of one: the only partner is the sample itself
    ref = O.augment(method, x[:1], labels[:1], frames[3:4], wav[:1], 2)
    y, mix = _aug(method, x[:1].copy(), frames[3:4], labels[:1], wav[:1], 2, device)
    assert mix.tolist() == [0] and np.abs(y - ref["y"]).max() <= (WAVE_TOL if "magwarp" in method else 0.0)


def test_empty_batch(device):
    x = np.zeros((0, 2, 64), dtype=np.float32)
    y, mix = _aug("durratiomixup", x, np.zeros((0, 5), np.int64), np.zeros(0, np.int64), (), 1, device)
    assert y.shape == (0, 2, 64) and len(mix) == 0


def test_batch_beyond_grid_y_limit(device):
    """B > 32768 takes the gridDim.z split of the sample index."""
    B, C, T = 40000, 1, 64
    x, frames, labels, wav = synthetic.make_batch(B, C, T, seed=8, rate_scale=0.04)
    ref = O.augment("durratiomixup", x, labels, frames, wav, 4)
    y, mix = _aug("durratiomixup", x, frames, labels, wav, 4, device)
    assert np.array_equal(mix, ref["mix"]) and np.array_equal(y, ref["y"])


def test_two_streams_and_second_device_free(device):
    """Launches follow torch's CURRENT stream: results on a side stream are identical."""
    x, frames, labels, wav = synthetic.make_batch(16, 2, 2500, seed=12)
    y0, _ = _aug("durmixmagwarp(0.2,4)", x, frames, labels, wav, 9, device)
    s = torch.cuda.Stream(device)
    with torch.cuda.stream(s):
        y1, _ = _aug("durmixmagwarp(0.2,4)", x, frames, labels, wav, 9, device)
    s.synchronize()
    assert np.array_equal(y0, y1)


def test_full_size_2d_against_oracle(device):
    """BASELINE.json configs[3] shape: 2D durratiomixup on (256,1,128,128) spectrogram columns."""
    from pcgmix_amd import frontend
    B = 256
    _, frames, labels, wav = synthetic.make_batch(B, 1, 5000, sample_rate=2000, seed=21)
    fs = frontend.spec_frames(frames, 5000, 34)
    rs = np.random.RandomState(5)
    x = rs.standard_normal((B, 1, 128, 128)).astype(np.float32)
    for method, step in (("durratiomixup", 2), ("durmixcutout(0.4,0.3)", 7)):
        ref = O.augment(method, x, labels, fs, wav, step)
        g = dict(x=x, labels=labels, frames=fs, wav=wav, step=step, method=method)
        _, _, (y, _, mix, _) = run(augmentations2d, g, device)
        assert np.array_equal(mix, ref["mix"]) and np.array_equal(y.cpu().numpy(), ref["y"])


def test_step_payload_rides_with_the_splice(device):
    """pcgmix_ctx_set_payload: bytes handed over before a plain step arrive at their device
    address with that step's single H2D copy (block (0,0,0) of the splice kernel forwards them);
    one shot; the splice itself is unchanged."""
    from pcgmix_amd import _lib, hostprep
    lib = _lib.load()
    x, frames, labels, _ = synthetic.make_batch(16, 4, 2500, seed=12)
    data = torch.from_numpy(x).to(device)
    recipe = hostprep.plain_recipe("durmixmagwarp(0.2,4)", False)
    ref, mix_ref = augmentations.splice_plain(recipe, data, labels, frames, 5)
    ctx = augmentations.step_context(device.index)
    for nbytes in (8, 48, 4096 + 24):                        # not multiples of 16 as well
        pay = np.random.RandomState(nbytes).randint(0, 256, nbytes).astype(np.uint8)
        dst = torch.full(((nbytes + 15) // 16 * 16 + 16,), 7, dtype=torch.uint8, device=device)
        _lib.check(lib.pcgmix_ctx_set_payload(ctx, pay.ctypes.data, nbytes, dst.data_ptr()), "payload")
        out, mix = augmentations.splice_plain(recipe, data, labels, frames, 5)
        got = dst.cpu().numpy()
        assert np.array_equal(got[:nbytes], pay)
        assert (got[nbytes:(nbytes + 15) // 16 * 16] == 0).all() and (got[-16:] == 7).all()
        assert torch.equal(out, ref) and np.array_equal(mix, mix_ref)
        dst.fill_(9)                                         # consumed: the next step carries nothing
        augmentations.splice_plain(recipe, data, labels, frames, 5)
        assert (dst.cpu().numpy() == 9).all()
    # a payload whose step does not come is sent on its own
    pay = np.arange(40, dtype=np.uint8)
    dst = torch.full((64,), 7, dtype=torch.uint8, device=device)
    st = torch.cuda.current_stream(device).cuda_stream
    _lib.check(lib.pcgmix_ctx_set_payload(ctx, pay.ctypes.data, 40, dst.data_ptr()), "payload")
    _lib.check(lib.pcgmix_ctx_flush_payload(ctx, st), "flush")
    _lib.check(lib.pcgmix_ctx_flush_payload(ctx, st), "flush")          # nothing pending: no-op
    got = dst.cpu().numpy()
    assert np.array_equal(got[:40], pay) and (got[40:48] == 0).all() and (got[48:] == 7).all()
    with pytest.raises(RuntimeError):                       # destination not 16-byte aligned
        _lib.check(lib.pcgmix_ctx_set_payload(ctx, pay.ctypes.data, 8, dst.data_ptr() + 4), "payload")


def test_large_plan_upload_goes_through_the_fetch_kernel(device):
    """Index data above 16 KB (the warp knots at bs 256: 49 KB) is copied by one launch of the
    library's fetch kernel from pinned staging instead of an SDMA transfer; the results are those
    of the oracle.  Covers pcgmix_fetch_h2d with torch's pinned memory, the general plan path
    ('(rand)': upload_plan) and the one-call path (the step context's own staging)."""
    from pcgmix_amd import _lib
    lib = _lib.load()
    n = 50_000
    src = torch.empty(65536, dtype=torch.uint8, pin_memory=True)
    src.numpy()[:] = np.random.RandomState(0).randint(0, 256, 65536)
    dst = torch.zeros(65536, dtype=torch.uint8, device=device)
    _lib.check(lib.pcgmix_fetch_h2d(src.data_ptr(), dst.data_ptr(), n,
                                    torch.cuda.current_stream(device).cuda_stream), "fetch")
    got = dst.cpu().numpy()
    n16 = (n + 15) // 16 * 16
    assert np.array_equal(got[:n16], src.numpy()[:n16]) and (got[n16:] == 0).all()
    assert lib.pcgmix_fetch_h2d(src.data_ptr() + 4, dst.data_ptr(), 64, None) != 0   # misaligned
    B, C, T = 256, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=17)
    data = torch.from_numpy(x).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
    for method in ("(rand)durmixmagwarp(0.2,4)", "durmixmagwarp(0.2,4)"):
        assert B * 6 * C * 8 > augmentations._BLIT_LIMIT
        y, _, mix, _ = augmentations.augment(Args(method), data, tgt, torch.from_numpy(frames), wav,
                                             StepCounter(11), None, device, "")
        ref = O.augment(method, x, labels, frames, wav, 11)
        assert np.array_equal(mix, ref["mix"])
        assert np.abs(y.cpu().numpy() - ref["y"]).max() <= WAVE_TOL


@pytest.mark.parametrize("B,C,T", [(256, 4, 5000), (7, 1, 2500), (33, 128, 128)])
def test_kernarg_splice_equals_the_copy_path(B, C, T, device):
    """pcgmix_mix_karg_f32 (index block as int16 in the kernel arguments, what the drop-in step
    launches for plain batches up to 256 samples) == pcgmix_mix_warp_f32 on an uploaded index
    block, bit for bit; shapes it cannot take are refused."""
    import ctypes
    from pcgmix_amd import _lib
    lib = _lib.load()
    if T >= 2500:
        x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000 if T == 5000 else 1000, seed=B)
    else:                                                    # spectrogram-shaped: boundaries in columns
        rs = np.random.RandomState(B)
        x = rs.standard_normal((B, C, T)).astype(np.float32)
        frames = np.concatenate([np.zeros((B, 1), np.int64),
                                 np.sort(rs.randint(1, T + 1, (B, 4)), axis=1)], axis=1)
    mix = np.random.RandomState(B).permutation(B)
    lam = float(np.float32(0.2718))
    data = torch.from_numpy(x).to(device)
    fr = torch.from_numpy(frames.astype(np.int32)).to(device)
    mx = torch.from_numpy(mix.astype(np.int32)).to(device)
    ref, out = torch.empty_like(data), torch.empty_like(data)
    augmentations.launch_mix(data, ref, fr.data_ptr(), mx.data_ptr(), None, lam, None, None, 0, B, C, T)
    fr16, mx16 = frames.astype(np.int16), mix.astype(np.int16)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.pcgmix_mix_karg_f32(data.data_ptr(), out.data_ptr(), fr16.ctypes.data, mx16.ctypes.data,
                                       ctypes.c_float(lam), B, C, T, st), "karg")
    assert torch.equal(out, ref) and not torch.equal(out, data)
    u = ctypes.c_int()
    assert lib.pcgmix_mix_karg_variant(B, C, T, ctypes.byref(u)) == 1 and u.value in (1, 2)
    assert lib.pcgmix_mix_karg_variant(257, C, T, ctypes.byref(u)) == 0
    assert lib.pcgmix_mix_karg_variant(B, C, 32768, ctypes.byref(u)) == 0
    assert lib.pcgmix_mix_karg_f32(data.data_ptr(), out.data_ptr(), fr16.ctypes.data, mx16.ctypes.data,
                                   ctypes.c_float(lam), 257, C, T, st) != 0
    assert lib.pcgmix_mix_karg_f32(data.data_ptr(), out.data_ptr(), fr16.ctypes.data, mx16.ctypes.data,
                                   ctypes.c_float(lam), B, C, T - 1, st) != 0       # T % 4


def _random_frames(rs, B, T):
    """Monotone boundaries 0 = f0 <= f1 <= f2 <= f3 <= f4 <= T per row, ragged: cycle lengths from a
    few samples to the whole row, some states empty."""
    f = np.zeros((B, 5), dtype=np.int64)
    for b in range(B):
        end = int(rs.randint(4, T + 1)) if rs.rand() > 0.1 else T
        cuts = np.sort(rs.randint(0, end + 1, 3))
        if rs.rand() < 0.15:
            cuts[rs.randint(0, 3)] = cuts[rs.randint(0, 3)]          # an empty state now and then
            cuts = np.sort(cuts)
        f[b] = [0, cuts[0], cuts[1], cuts[2], end]
    return f


_FUZZ_METHODS = ["durratiomixup", "durratiomixup+0.7", "(alpha=0.5)durratiomixup", "(rand)durratiomixup",
                 "durmixmagwarp(0.2,4)", "durmixmagwarp(0.05,7)+0.8", "(rand)durmixmagwarp(0.1,3)",
                 "(alpha=2.0)durmixmagwarp(0.3,2)", "(mixAll)durmixmagwarp(0.3,5)", "(mixAll)durratiomixup",
                 "(samePCG)durratiomixup", "(sameDataset)durmixmagwarp(0.2,4)"]


@pytest.mark.parametrize("case", range(36))
def test_random_cases_against_oracle(case, device):
    """Seeded differential test: random batch size, channel count, row length (odd and even, the
    vector and the scalar kernel), ragged cycles, label mix, recording ids, step and method string
    of the grammar the reference parses; partner indices and targets exact, the splice bit-exact,
    the warped waveform within the north_star tolerance; a rejected gate returns the input object."""
    rs = np.random.RandomState(1000 + case)
    B = int(rs.choice([1, 2, 3, 5, 8, 13, 32, 57]))
    C = int(rs.choice([1, 2, 4]))
    T = int(rs.choice([64, 130, 333, 512, 1001, 1400, 2500]))
    method = _FUZZ_METHODS[case % len(_FUZZ_METHODS)]
    step = int(rs.randint(0, 5000))
    x = rs.standard_normal((B, C, T)).astype(np.float32)
    frames = _random_frames(rs, B, T)
    labels = rs.randint(0, 2, B).astype(np.int64)
    wav = tuple(f"{'abcdef'[rs.randint(0, 6)]}{rs.randint(0, 4):04d}" for _ in range(B))
    ref = O.augment(method, x, labels, frames, wav, step)
    g = dict(x=x, labels=labels, frames=frames, wav=wav, step=step, method=method)
    data, tgt, (y, t_out, mix, _) = run(augmentations, g, device)
    if not ref["fired"]:
        assert y is data and t_out is tgt and len(mix) == 0
        return
    assert np.array_equal(np.asarray(mix), ref["mix"])
    assert np.array_equal(t_out.cpu().numpy().astype(np.float64), ref["target"].astype(np.float64)) \
        if "(mixAll)" not in method else np.allclose(t_out.cpu().numpy(), ref["target"], rtol=0, atol=1e-7)
    err = np.abs(y.cpu().numpy() - ref["y"]).max() if B else 0.0
    assert err <= (WAVE_TOL if "magwarp" in method else 0.0), (method, (B, C, T), err)


_FUZZ_METHODS_2D = ["durratiomixup", "durratiomixup+0.6", "durmixcutout(0.4,0.3)", "durmixcutout(0.9,1.0)+0.9",
                    "durmixtimemask(0.5)", "durmixtimemask", "durmixfreqmask(0.25)", "durmixfreqmask"]


@pytest.mark.parametrize("case", range(16))
def test_random_2d_cases_against_oracle(case, device):
    """The same differential test for the spectrogram path (augmentations2d.py:286-427): random
    (B,1,F,W) images — W a multiple of four or not —, ragged column boundaries, mask variants with
    and without parameters and gates; everything bit-exact (splice and zeroed rectangles)."""
    rs = np.random.RandomState(2000 + case)
    B = int(rs.choice([1, 2, 5, 16, 37]))
    F = int(rs.choice([16, 40, 128]))
    W = int(rs.choice([32, 75, 128, 130]))
    method = _FUZZ_METHODS_2D[case % len(_FUZZ_METHODS_2D)]
    step = int(rs.randint(0, 5000))
    x = rs.standard_normal((B, 1, F, W)).astype(np.float32)
    frames = _random_frames(rs, B, W)
    labels = rs.randint(0, 2, B).astype(np.int64)
    wav = tuple(f"a{rs.randint(0, 9):04d}" for _ in range(B))
    ref = O.augment(method, x, labels, frames, wav, step)
    g = dict(x=x, labels=labels, frames=frames, wav=wav, step=step, method=method)
    data, tgt, (y, t_out, mix, _) = run(augmentations2d, g, device)
    if not ref["fired"]:
        assert y is data and t_out is tgt and len(mix) == 0
        return
    assert np.array_equal(np.asarray(mix), ref["mix"])
    assert np.array_equal(y.cpu().numpy(), ref["y"]), (method, (B, F, W))


def test_step_context_entry_points_refuse_a_capturing_stream(device):
    """The step-context entry points stage through pinned slots, wait for a label read-back and
    carry per-step data in kernel arguments: none of that may be recorded into a hipGraph.  On a
    capturing stream they return hipErrorStreamCaptureUnsupported before touching anything (VERDICT
    r2 item 2: no first-call work inside a capture), the capture itself stays valid, and the same
    call works again afterwards."""
    import ctypes
    from pcgmix_amd import _lib
    lib = _lib.load()
    B, C, T = 8, 2, 512
    x, frames, labels, wav = synthetic.make_batch(B, C, T, seed=4, rate_scale=T / 1400.0)
    data = torch.from_numpy(x).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
    fr = torch.from_numpy(frames)
    args, sc = Args("durratiomixup"), StepCounter(5)
    y0, _, mix0, _ = augmentations.augment(args, data, tgt, fr, wav, sc, None, device, "")   # context exists
    torch.cuda.synchronize()
    ctx = augmentations.step_context(data.device.index)
    probe = torch.zeros(4, device=device)
    pay_host = np.arange(8, dtype=np.float32)
    pay_dev = torch.zeros(8, device=device)
    _lib.check(lib.pcgmix_ctx_set_payload(ctx, pay_host.ctypes.data, pay_host.nbytes, pay_dev.data_ptr()),
               "pcgmix_ctx_set_payload")             # pending: flush has something to send
    errors = {}
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        probe.add_(1.0)                                # the capture holds one real node
        st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        try:
            augmentations.augment(args, data, tgt, fr, wav, sc, None, device, "")
            errors["augment"] = None
        except RuntimeError as e:
            errors["augment"] = str(e)
        errors["flush"] = lib.pcgmix_ctx_flush_payload(ctx, st)
    capture_unsupported = 900                          # hipErrorStreamCaptureUnsupported
    assert errors["augment"] is not None and f"hipError_t {capture_unsupported}" in errors["augment"], errors
    assert errors["flush"] == capture_unsupported, errors
    g.replay()
    torch.cuda.synchronize()
    assert probe.tolist() == [1.0] * 4                 # (capture does not execute; one replay)
    assert float(pay_dev.abs().sum()) == 0.0           # nothing of the payload went out under capture
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.pcgmix_ctx_flush_payload(ctx, st), "pcgmix_ctx_flush_payload")
    torch.cuda.synchronize()
    assert np.array_equal(pay_dev.cpu().numpy(), pay_host)
    y1, _, mix1, _ = augmentations.augment(args, data, tgt, fr, wav, sc, None, device, "")
    assert np.array_equal(mix0, mix1) and torch.equal(y0, y1)


@pytest.mark.parametrize("method,C", [("durmixmagwarp(0.2,4)", 4), ("durmixmagwarp(0.2,4)", 1),
                                      ("(alpha=0.5)durmixmagwarp(0.1,3)", 4), ("durratiomixup", 4),
                                      ("(rand)durmixmagwarp(0.2,4)", 2)])
def test_augment_leaves_numpys_global_stream_where_the_reference_does(method, C, device):
    """The reference reseeds numpy's GLOBAL stream in every call (augmentations.py:662) and leaves
    it behind its beta / normal draws (:677).  The knots come from the library's restatement of
    that stream, drawn ahead on worker threads (csrc/pcgmix_nprand.hip) — ``np.random.get_state()``
    after ``augment()`` must still equal the state after numpy's own seed -> beta -> normal, and
    the output the oracle's (which calls numpy) over consecutive steps."""
    from pcgmix_amd import hostprep
    B, T = 16, 2500
    x, frames, labels, wav = synthetic.make_batch(B, C, T, seed=21)
    name = hostprep.select_method(method, False)
    alpha = hostprep.parse_alpha(method, name)
    sigma, knot = hostprep.parse_magwarp(method) if name == "durmixmagwarp" else (0.0, -2)
    for step in (40, 41, 42, 43, 41, 7):
        ref = O.augment(method, x, labels, frames, wav, step)
        np.random.seed(step)
        np.random.beta(alpha, alpha)
        if knot + 2:
            np.random.normal(1.0, sigma, size=(B, knot + 2, C))
        want = np.random.get_state()
        np.random.seed(777)
        g = dict(x=x, labels=labels, frames=frames, wav=wav, step=step, method=method)
        _, _, (y, _, mix, _) = run(augmentations, g, device)
        got = np.random.get_state()
        assert got[0] == want[0] and np.array_equal(got[1], want[1]) and tuple(got[2:]) == tuple(want[2:])
        assert np.array_equal(mix, ref["mix"])
        assert np.abs(y.cpu().numpy() - ref["y"]).max() <= (WAVE_TOL if knot + 2 else 0.0)
