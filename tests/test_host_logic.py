"""Host prologue (method DSL, gate, partner permutation, RNG order) against the goldens that
the running reference produced, plus the C-ABI surface — no GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib, hostprep
from conftest import ROOT, golden_files, load_golden

CASES = golden_files("mix1d_") + golden_files("salopt_")


@pytest.mark.parametrize("path", CASES + golden_files("mix2d_") + golden_files("mask2d_"),
                         ids=lambda p: p.split("/")[-1][:-4])
def test_plan_matches_reference(path):
    g = load_golden(path)
    is2d = g["x"].ndim == 4
    B = g["x"].shape[0]
    C = 1 if is2d else g["x"].shape[1]
    C = g["x"].shape[1] * g["x"].shape[2] if is2d else C
    plan = hostprep.make_plan(g["method"], g["labels"], g["frames"], g["wav"], g["step"], B, C,
                              is2d=is2d, n_cols=g["x"].shape[-1])
    assert plan.fired == bool(g["fired"])
    if not plan.fired:
        return
    assert np.array_equal(plan.mix, g["mix"])                 # segment/partner indices bit-exact
    assert plan.lam64 == float(g["lam"])
    if g["knots"].size:
        assert np.array_equal(plan.knots, g["knots"].reshape(plan.knots.shape))
    else:
        assert plan.knots is None


def test_gate_statistics_and_labels_not_needed_when_rejected():
    fired = 0
    for step in range(200):
        called = []
        plan = hostprep.make_plan("durratiomixup+0.2", lambda: called.append(1) or np.zeros(4, int),
                                  np.zeros((4, 5), np.int64), ("a",) * 4, step, 4, 1)
        fired += plan.fired
        assert bool(called) == plan.fired        # rejected steps never touch the labels
    assert fired == 43


def test_rand_offsets_depend_on_gap_only():
    import random
    frames = np.array([[0, 10, 30, 40, 90], [0, 14, 30, 45, 80]], dtype=np.int64)
    off = hostprep.rand_offsets(frames, np.array([1, 0]), step=9)
    lens = np.diff(frames, axis=1)
    for b, m in ((0, 1), (1, 0)):
        for k in range(4):
            assert off[b, k] == random.Random(9).randint(0, abs(int(lens[m, k] - lens[b, k])))


@pytest.mark.parametrize("seed", [0, 1, 77, 1234, 2**31 - 1, 2**32 + 7])
def test_c_sampler_is_cpython_random(seed):
    """The library's MT19937/sample/uniform/randint restatement against CPython itself."""
    import random
    lib = _lib.load()
    assert lib.pcgmix_py_uniform01(seed) == random.Random(seed).uniform(0, 1)
    for hi in (0, 1, 2, 5, 255, 256, 1345, 2**33):
        assert lib.pcgmix_py_randint0(seed, hi) == random.Random(seed).randint(0, hi)
    rs = np.random.RandomState(seed % 2**31)
    for B in (1, 2, 3, 17, 64, 255, 256, 1000):
        keys = rs.randint(0, 3, B)
        got = hostprep.shuffle_within_groups(keys, seed)
        ref = np.arange(B)
        for k in np.unique(keys):
            idx = [i for i in range(B) if keys[i] == k]
            ref[idx] = random.Random(seed).sample(idx, len(idx))
        assert np.array_equal(got, ref)
    names = [f"w{i % 5}" for i in range(40)]
    got = hostprep.shuffle_within_groups(names, seed)
    ref = np.arange(40)
    for k in set(names):
        idx = [i for i in range(40) if names[i] == k]
        ref[idx] = random.Random(seed).sample(idx, len(idx))
    assert np.array_equal(got, ref)


def test_unknown_method_is_passthrough_and_foreign_method_is_refused():
    assert hostprep.select_method("base", False) is None
    assert hostprep.select_method("durmixmagwarp(0.2,4)+0.4", False) == "durmixmagwarp"
    assert hostprep.select_method("(saloptenv)durratiomixup", False) == "durratiomixup"
    with pytest.raises(NotImplementedError):
        hostprep.select_method("cutmix", False)
    with pytest.raises(NotImplementedError):
        hostprep.select_method("(sameCVD)durratiomixup", False)
    assert hostprep.select_method("durmixmagwarp(0.2,4)", True) is None   # unknown to the 2D dispatcher
    assert hostprep.select_method("durmixcutout(0.2,0.3)", True) == "durmixcutout"
    with pytest.raises(NotImplementedError):
        hostprep.select_method("cutout", True)


def test_validate_frames():
    ok = np.array([[0, 5, 9, 12, 20]])
    hostprep.validate_frames(ok, 20)
    with pytest.raises(ValueError):
        hostprep.validate_frames(ok, 19)
    with pytest.raises(ValueError):
        hostprep.validate_frames(np.array([[0, 5, 4, 12, 20]]), 30)


# ------------------------------------------------------------------ C ABI surface
def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pcgmix_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcgmix_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = declared_symbols()
    assert len(names) >= 8
    lib = ctypes.CDLL(_lib.LIB_PATH)          # loads without a GPU (no device call is made)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pcgmix_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert _lib.load().pcgmix_abi_version() == _lib.ABI_VERSION


@pytest.mark.parametrize("T,n", [(2500, 6), (5000, 6), (640, 8), (640, 4), (128, 3), (64, 2), (5000, 12)])
def test_spline_operator_matches_scipy(T, n):
    """knots -> PPoly coefficients is linear; the library's operator must reproduce scipy's
    not-a-knot CubicSpline (magnitude_warp, augmentations.py:674-683)."""
    from scipy.interpolate import CubicSpline
    lib = _lib.load()
    op = np.empty(lib.pcgmix_spline_operator_size(n))
    assert lib.pcgmix_spline_operator_f64(T, n, op.ctypes.data) == 0
    brk = np.linspace(0, T - 1.0, num=n)
    assert np.array_equal(op[:n], brk)                       # numpy's linspace rounding
    M = op[n:].reshape(n - 1, 4, n)
    rs = np.random.RandomState(T + n)
    t = np.arange(T)
    for _ in range(5):
        k = rs.normal(1.0, 0.2, n)
        cs = CubicSpline(brk, k)
        coef = M @ k                                         # (pieces, 4)
        assert np.allclose(coef.T, cs.c, rtol=1e-11, atol=1e-13 * np.abs(cs.c).max())
        p = np.clip(np.searchsorted(brk, t, "right") - 1, 0, n - 2)
        s = t - brk[p]
        w = coef[p, 3] + coef[p, 2] * s + coef[p, 1] * (s * s) + coef[p, 0] * (s * s * s)
        assert np.abs(w - cs(t)).max() < 5e-13


def test_spline_operator_rejects_bad_arguments():
    lib = _lib.load()
    assert lib.pcgmix_spline_operator_size(1) == 0
    buf = np.empty(64)
    assert lib.pcgmix_spline_operator_f64(100, 1, buf.ctypes.data) != 0
    assert lib.pcgmix_spline_operator_f64(1, 4, buf.ctypes.data) != 0


def test_plan_edge_batches():
    one = hostprep.make_plan("durmixmagwarp(0.2,4)", np.array([1]), np.array([[0, 5, 9, 12, 20]]),
                             ("a",), 3, 1, 2)
    assert one.fired and one.mix.tolist() == [0] and one.knots.shape == (1, 6, 2)
    empty = hostprep.make_plan("durratiomixup", np.zeros(0, np.int64), np.zeros((0, 5), np.int64),
                               (), 3, 0, 1)
    assert empty.fired and empty.mix.shape == (0,)
    with pytest.raises(ValueError):
        hostprep.make_plan("durratiomixup", np.zeros(3, np.int64), np.zeros((2, 5), np.int64),
                           ("a", "b"), 0, 2, 1)


def test_mask_rectangles_match_reference_zero_pattern():
    """The rectangles derived on the host cover exactly the elements the reference zeroed
    (goldens mask2d_*: inputs are dense noise, so zeros in y are the mask)."""
    for path in golden_files("mask2d_"):
        g = load_golden(path)
        if not g["fired"]:
            continue
        B, _, F, W = g["x"].shape
        plan = hostprep.make_plan(g["method"], g["labels"], g["frames"], g["wav"], g["step"], B, F,
                                  is2d=True, n_cols=W)
        r = plan.zero_rect
        rows, cols = np.arange(F)[None, :, None], np.arange(W)[None, None, :]
        mask = ((rows >= r[:, 0, None, None]) & (rows < r[:, 1, None, None])
                & (cols >= r[:, 2, None, None]) & (cols < r[:, 3, None, None]))
        assert np.array_equal(mask, g["y"][:, 0] == 0), path


def test_plain_recipe_selects_the_one_call_path():
    """Which method strings take pcgmix_splice_same_label_f32 (plain same-label splice) and which
    fall back to the general plan; parsed fields follow the reference's string rules."""
    from pcgmix_amd import hostprep as H
    assert H.plain_recipe("durratiomixup", False) == ("durratiomixup", 1.0, 1.0, 0.0, 0)
    assert H.plain_recipe("durratiomixup+0.2", False)[1] == 0.2
    assert H.plain_recipe("(alpha=0.4)durratiomixup", False)[2] == 0.4
    assert H.plain_recipe("durmixmagwarp(0.3,5)+0.5", False) == ("durmixmagwarp", 0.5, 1.0, 0.3, 7)
    assert H.plain_recipe("durmixmagwarp", False)[3:] == (0.2, 6)
    for m in ("(rand)durratiomixup", "(saloptenv)durmixmagwarp(0.2,4)", "(samePCG)durratiomixup",
              "(sameDataset)durratiomixup", "(mixAll)durratiomixup"):
        assert H.plain_recipe(m, False) is None, m
    assert H.plain_recipe("base", False) is None and H.plain_recipe("base", True) is None
    assert H.plain_recipe("(alpha=0.4)durratiomixup", True) == ("durratiomixup", 1.0, 1.0, 0.0, 0)
    assert H.plain_recipe("durmixcutout(0.2,0.2)", True) is None


def _recording_plan_loop(lengths, boundaries, seg_starts, hop, width, tile_frames):
    """databuilder.ipynb cell 6:93-101, 127-134 as a plain loop per tile and per cycle — the checker
    for the vectorised ``frontend.recording_plan``."""
    rec_off, tiles, cycles, fspec, rec_of = [], [], [], [], []
    off = col = 0
    for r, n in enumerate(lengths):
        n = int(n)
        n_frames = 1 + n // hop
        rec_off.append(off)
        for f0 in range(0, n_frames, tile_frames):
            tiles.append((r, f0, min(tile_frames, n_frames - f0), col + f0))
        b = np.asarray(boundaries[r], dtype=np.int64)
        cols = [int(round(float(v) * n_frames / float(n))) for v in b]      # Python round: half-even
        for i in seg_starts[r]:
            c = cols[i:i + 5]
            keep = max(min(max(c[4] - c[0], 0), width, n_frames - c[0]), 0)
            cycles.append((r, col + c[0], keep, 0))
            fspec.append([v - c[0] for v in c])
            rec_of.append(r)
        off += n
        col += n_frames
    return rec_off, tiles, cycles, fspec, rec_of, col


def test_recording_plan_matches_loop_restatement():
    from pcgmix_amd import frontend
    rng = np.random.default_rng(3)
    lengths = rng.integers(3000, 90000, 40)
    bounds, starts = [], []
    for k, n in enumerate(lengths):
        b = np.cumsum(rng.integers(250, 550, 300))
        b = b[b < n - 200]
        bounds.append(b)
        starts.append([] if k % 7 == 3 else list(range(0, max(len(b) - 4, 0), 4)))
    plan = frontend.recording_plan(lengths, bounds, starts, 34, 128, 128)
    rec_off, tiles, cycles, fspec, rec_of, col = _recording_plan_loop(lengths, bounds, starts, 34, 128, 128)
    assert plan["scratch_cols"] == col
    assert np.array_equal(plan["rec_off"], np.asarray(rec_off, dtype=np.int64))
    assert np.array_equal(plan["tiles"], np.asarray(tiles, dtype=np.int32).reshape(-1, 4))
    assert np.array_equal(plan["cycles"], np.asarray(cycles, dtype=np.int32).reshape(-1, 4))
    assert np.array_equal(plan["frames_spec"], np.asarray(fspec, dtype=np.int64).reshape(-1, 5))
    assert np.array_equal(plan["rec_of_cycle"], np.asarray(rec_of, dtype=np.int64))
    assert plan["tiles"].dtype == np.int32 and plan["cycles"].dtype == np.int32
    # no cycles at all: empty, correctly shaped tables
    empty = frontend.recording_plan([5000], [np.asarray([100, 600, 1100])], [[]], 34, 128, 128)
    assert empty["cycles"].shape == (0, 4) and empty["frames_spec"].shape == (0, 5) and empty["rec_of_cycle"].shape == (0,)


# ---- numpy's legacy global stream restated in C (csrc/pcgmix_nprand.hip, round 4) ---------------
def _numpy_reference(step, alpha, sigma, count):
    """What the reference does (augmentations.py:661-663, 677) and where it leaves the stream."""
    if alpha > 0.0:
        np.random.seed(step)
        lam = np.random.beta(alpha, alpha)
    else:
        lam = 1.0
    knots = np.random.normal(loc=1.0, scale=sigma, size=count) if count else None
    return lam, knots, np.random.get_state()


def _same_state(a, b):
    return a[0] == b[0] and np.array_equal(a[1], b[1]) and tuple(a[2:]) == tuple(b[2:])


@pytest.mark.parametrize("alpha", [1.0, 0.5, 0.2, 0.05])
def test_np_stream_primitives_equal_numpy(alpha):
    """seed -> beta(alpha, alpha) -> normal(1, sigma, n) on a private state == numpy, bit for bit,
    and the state behind it == numpy's (key, position)."""
    lib = _lib.load()
    st = (ctypes.c_uint32 * 625)()
    lam = ctypes.c_double()
    rs = np.random.RandomState(11)
    for seed in list(range(40)) + [int(v) for v in rs.randint(0, 2 ** 32 - 1, 40, dtype=np.int64)] \
            + [2 ** 32 - 1]:
        n = int(rs.choice([0, 2, 48, 312, 1536, 6144]))
        want_lam, want, state = _numpy_reference(seed, alpha, 0.2, n)
        assert lib.pcgmix_np_seed(st, seed) == 0
        assert lib.pcgmix_np_beta(st, alpha, alpha, ctypes.byref(lam)) == 0
        out = np.empty(n)
        assert lib.pcgmix_np_normal_fill(st, 1.0, 0.2, n, out.ctypes.data) == 0
        assert lam.value == want_lam
        if n:
            assert np.array_equal(out, want)
        mine = np.frombuffer(st, dtype=np.uint32)
        assert np.array_equal(mine[:624], state[1]) and int(mine[624]) == int(state[2])
    # outside the contract: refused, the caller draws with numpy
    assert lib.pcgmix_np_normal_fill(st, 1.0, 0.2, 7, np.empty(7).ctypes.data) == 1
    assert lib.pcgmix_np_beta(st, 2.0, 2.0, ctypes.byref(lam)) == 1


def test_np_normal_fill_from_an_unaligned_stream_position():
    """The block-wise trial evaluation assumes a position that is a multiple of four words; any
    other position (a caller who drew something else first) goes word by word."""
    lib = _lib.load()
    addr = np.random.get_bit_generator().ctypes.state_address
    for pre in (1, 2, 3, 5, 311, 623):
        np.random.seed(99)
        np.random.randint(0, 2 ** 31, size=pre)            # one 32-bit word each
        want = np.random.normal(1.0, 0.3, 500)
        state = np.random.get_state()
        np.random.seed(99)
        np.random.randint(0, 2 ** 31, size=pre)
        out = np.empty(500)
        assert lib.pcgmix_np_normal_fill(addr, 1.0, 0.3, 500, out.ctypes.data) == 0
        assert np.array_equal(out, want) and _same_state(np.random.get_state(), state)


@pytest.mark.parametrize("lookahead", ["0", "2"])
def test_draw_lambda_knots_equals_numpy_and_leaves_the_global_stream_there(lookahead, monkeypatch):
    """hostprep.draw_lambda_knots == the reference's seed -> beta -> normal: same lambda, same
    knots, and ``np.random.get_state()`` afterwards == after numpy's own calls (the reference's
    side effect, augmentations.py:662) — for consecutive steps (blocks drawn ahead by the worker
    threads), repeated and out-of-order steps, changing shapes, and the fallbacks (alpha <= 0,
    alpha > 1, odd counts)."""
    monkeypatch.setenv("PCGMIX_NPDRAW_LOOKAHEAD", lookahead)
    monkeypatch.setattr(hostprep, "_NPDRAW", None)
    cases = [(s, 1.0, 0.2, 6144) for s in range(12)]
    cases += [(5, 1.0, 0.2, 6144), (5, 1.0, 0.2, 6144), (3, 1.0, 0.2, 6144), (4, 1.0, 0.2, 1536),
              (5, 0.5, 0.2, 1536), (6, 0.5, 0.3, 1536), (7, 0.5, 0.3, 48), (8, 0.05, 0.3, 48)]
    cases += [(9, 2.0, 0.2, 48), (10, 0.0, 0.2, 48), (11, 1.0, 0.2, 45), (12, 1.0, 0.2, 0),
              (13, 1.0, 0.2, 48), (14, 1.0, 0.2, 48)]
    for step, alpha, sigma, n in cases:
        np.random.seed(12345)                   # (alpha <= 0 continues whatever stream there is)
        want_lam, want, state = _numpy_reference(step, alpha, sigma, n)
        np.random.seed(12345)
        lam, knots = hostprep.draw_lambda_knots(step, alpha, sigma, n)
        assert lam == want_lam, (step, alpha)
        if n:
            assert np.array_equal(hostprep.knots_array(knots, (n,)), want), (step, alpha, n)
        else:
            assert knots is None
        assert _same_state(np.random.get_state(), state), (step, alpha, n)
    if lookahead != "0":
        h = hostprep._NPDRAW[0]
        misses = ctypes.c_longlong()
        hits = _lib.load().pcgmix_npdraw_stats(h, ctypes.byref(misses))
        assert hits >= 11                       # steps 1..11 of the run were drawn ahead


def test_draw_lambda_knots_with_a_foreign_bit_generator_falls_back_to_numpy():
    old = np.random.get_bit_generator()
    try:
        np.random.set_bit_generator(np.random.PCG64(1))
        lam, knots = hostprep.draw_lambda_knots(3, 1.0, 0.2, 48)
        assert isinstance(knots, np.ndarray) and knots.shape == (48,)
    finally:
        np.random.set_bit_generator(old)
    np.random.seed(3)
    want = (np.random.beta(1.0, 1.0), np.random.normal(1.0, 0.2, 48))
    lam, knots = hostprep.draw_lambda_knots(3, 1.0, 0.2, 48)
    assert lam == want[0] and np.array_equal(hostprep.knots_array(knots, (48,)), want[1])


def test_npdraw_cancels_wrong_guesses_and_survives_fork():
    """The draw-ahead object guesses the next steps' keys (step + 1, + 2 with the same shape).  A
    caller that jumps — another step, another shape, another alpha — makes it abandon the running
    jobs (one polled flag per generator block) and draw inline; every result still equals numpy's.
    A forked child has the object but not its threads: it must draw inline, not wait for them."""
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.pcgmix_npdraw_create(ctypes.byref(h), 3) == 0
    lam, kp, hit = ctypes.c_double(), ctypes.c_void_p(), ctypes.c_int()

    def draw(seed, alpha, sigma, n):
        assert lib.pcgmix_npdraw_step(h, seed, alpha, sigma, n, None, ctypes.byref(lam), ctypes.byref(kp),
                                      ctypes.byref(hit)) == 0
        got = np.frombuffer((ctypes.c_double * n).from_address(kp.value), dtype=np.float64).copy() if n else None
        np.random.seed(seed)
        want_lam = np.random.beta(alpha, alpha)
        want = np.random.normal(1.0, sigma, n) if n else None
        assert lam.value == want_lam and (n == 0 or np.array_equal(got, want)), (seed, alpha, sigma, n)
        return hit.value
    rs = np.random.RandomState(2)
    hits = 0
    for i in range(60):                              # big blocks in flight, abandoned by the next call
        hits += draw(int(rs.randint(0, 10 ** 6)), float(rs.choice([1.0, 0.5, 0.1])), 0.2,
                     int(rs.choice([48, 6144, 200000, 2])))
    for s in range(100, 130):                        # a regular sequence: everything but its head hits
        hits += draw(s, 1.0, 0.2, 6144)
    assert hits >= 27
    # refused arguments leave the object usable
    assert lib.pcgmix_npdraw_step(h, 1, 2.0, 0.2, 48, None, ctypes.byref(lam), ctypes.byref(kp), None) == 1
    assert lib.pcgmix_npdraw_step(h, 1, 1.0, 0.2, 7, None, ctypes.byref(lam), ctypes.byref(kp), None) == 1
    draw(5, 1.0, 0.2, 48)
    pid = os.fork()
    if pid == 0:                                     # child: no worker threads here
        try:
            for s in (131, 132, 7):
                draw(s, 1.0, 0.2, 6144)
            os._exit(0)
        except BaseException:
            os._exit(1)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0
    draw(131, 1.0, 0.2, 6144)                        # the parent's workers are still there
    lib.pcgmix_npdraw_destroy(h)


def test_salopt_launch_plan_covers_every_slice_longest_first():
    """pcgmix_salopt_plan (host): the planned launch of the displacement search lists exactly the
    slices that have candidates — slice z of a pair holds d = 256 z .. 256 z + 255 (+ 1024 ...) of
    d = 0 .. |n1 - n2| — plus slice 0 of every pair without a search, ordered by the length of the
    block's chain of sums (own state's length x passes) in steps of 16, longest first; no plan
    beyond 256 samples or 1408 blocks."""
    import ctypes
    from pcgmix_amd import _lib, synthetic
    lib = _lib.load()
    for B, seed in ((256, 0), (37, 3), (1, 5)):
        frames = synthetic.make_index_data(B, 5000, sample_rate=2000, seed=seed)[0].astype(np.int32)
        mix = np.random.RandomState(seed).permutation(B).astype(np.int32)
        ids = np.zeros(4096, dtype=np.uint16)
        n = lib.pcgmix_salopt_plan(frames.ctypes.data, mix.ctypes.data, B, 5000, 0, ids.ctypes.data, len(ids))
        L = np.diff(frames, axis=1)
        want, cost = set(), {}
        for b in range(B):
            for k in range(4):
                n1, n2 = int(L[b, k]), int(L[mix[b], k])
                dm = abs(n1 - n2)
                zs = [0] if dm == 0 else [z for z in range(4) if z * 256 <= dm]
                for z in zs:
                    bid = (b << 4) | (k << 2) | z
                    want.add(bid)
                    cost[bid] = 0 if dm == 0 else min((n1 * ((dm - z * 256) // 1024 + 1)) >> 4, 511)
        assert n == len(want) and set(ids[:n].tolist()) == want
        c = [cost[i] for i in ids[:n].tolist()]
        assert c == sorted(c, reverse=True)
    # too many samples / too many blocks: no plan
    frames = synthetic.make_index_data(300, 5000, sample_rate=2000, seed=0)[0].astype(np.int32)
    mix = np.arange(300, dtype=np.int32)[::-1].copy()
    ids = np.zeros(8, dtype=np.uint16)
    assert lib.pcgmix_salopt_plan(frames.ctypes.data, mix.ctypes.data, 300, 5000, 0, ids.ctypes.data, 8) == 0
    wide = np.where((np.arange(256) % 2 == 0)[:, None], np.array([0, 100, 200, 300, 400]),
                    np.array([0, 1000, 2000, 3000, 4000])).astype(np.int32)
    mixw = (np.arange(256) ^ 1).astype(np.int32)
    assert lib.pcgmix_salopt_plan(wide.ctypes.data, mixw.ctypes.data, 256, 5000, 0, ids.ctypes.data, 8) == 0
    assert lib.pcgmix_salopt_plan(None, mixw.ctypes.data, 256, 5000, 0, ids.ctypes.data, 8) < 0


def _mel_table_layout(n_fft, n_mels):
    """The blob layout of pcgmix_logmel_tables (pcgmix_logmel.hip: mel_tables)."""
    n_bins = n_fft // 2 + 1
    tpp = n_bins // 32
    rem = n_bins - 32 * tpp
    if tpp >= 1 and rem <= 8:
        n_left = rem
    else:
        tpp, n_left = (n_bins + 31) // 32, 0
    ksteps = (n_fft // 4 + 1 + 3) // 4
    o = 2 * tpp * ksteps * 2 * 64 * 8
    off_wts = o
    o = (o + n_mels * n_bins * 4 + 7) & ~7
    off_kr = o
    o = (o + n_mels * 8 + 7) & ~7
    off_left = o
    o += n_left * (n_fft // 2 + 8) * 16
    off_win = o
    o += (n_fft // 2 + 1) * 8
    return dict(n_bins=n_bins, tpp=tpp, n_left=n_left, ksteps=ksteps, off_wts=off_wts, off_kr=off_kr,
                off_left=off_left, off_win=off_win, off_meta=o, total=(o + 8 + 15) & ~15)


@pytest.mark.parametrize("sr,n_fft", [(2000.0, 136), (1000.0, 68), (4000.0, 272)])
def test_logmel_tables_are_the_windowed_transform(sr, n_fft):
    """pcgmix_logmel_tables (host): the doubly folded twiddle fragments (window on the data side,
    even / odd bins, columns n = 0 .. n_fft/4 with weights 1/2 at both ends), the single-fold rows of
    the bins beyond the tiles, the window and bin_lo / n_left_used, evaluated in numpy exactly as the
    kernel evaluates them, give rfft(hann * frame) for every bin a mel filter reads."""
    import ctypes
    from pcgmix_amd import _lib
    lib = _lib.load()
    n_mels = 128
    lay = _mel_table_layout(n_fft, n_mels)
    assert lib.pcgmix_logmel_tables_size(n_fft, n_mels) == lay["total"]
    blob = np.zeros(lay["total"], dtype=np.uint8)
    assert lib.pcgmix_logmel_tables(n_fft, n_mels, ctypes.c_float(25.0), ctypes.c_float(sr / 2),
                                    ctypes.c_float(sr), blob.ctypes.data) == 0
    N, H, Q = n_fft, n_fft // 2, n_fft // 4
    tpp, ks, n_bins = lay["tpp"], lay["ksteps"], lay["n_bins"]
    afrag = blob[:lay["off_wts"]].view(np.float64).reshape(tpp, ks, 4, 64)
    kr = blob[lay["off_kr"]:lay["off_kr"] + n_mels * 8].view(np.int32).reshape(n_mels, 2)
    left = blob[lay["off_left"]:lay["off_win"]].view(np.float64).reshape(lay["n_left"], H + 8, 2)
    win = blob[lay["off_win"]:lay["off_meta"]].view(np.float64)
    used, bin_lo = blob[lay["off_meta"]:lay["off_meta"] + 8].view(np.int32)
    assert np.allclose(win, 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(H + 1) / N), atol=1e-15)
    nonempty = kr[:, 1] >= kr[:, 0]
    lo_bin, hi_bin = int(kr[nonempty, 0].min()), int(kr[nonempty, 1].max())
    assert bin_lo % 2 == 0 and 0 <= bin_lo <= lo_bin and hi_bin < bin_lo + 32 * tpp + lay["n_left"]
    rs = np.random.RandomState(n_fft)
    x = rs.randn(N + 1).astype(np.float32).astype(np.float64)          # x[N] is not part of the frame
    ref = np.fft.rfft(0.5 * (1 - np.cos(2 * np.pi * np.arange(N) / N)) * x[:N])
    # the kernel's B operands, column n = 0 .. Q (column 0 pairs x[0] with itself)
    n = np.arange(Q + 1)
    hi_n = np.where(n == 0, 0, N - n)
    u, v = win[n] * (x[n] + x[hi_n]), win[n] * (x[n] - x[hi_n])
    m = H - n
    u2, v2 = win[m] * (x[m] + x[H + n]), win[m] * (x[m] - x[H + n])
    ops = [u + u2, v - v2, u - u2, v + v2]                              # even re, even im, odd re, odd im
    got = {}
    for tp in range(tpp):
        for row in range(16):
            for par in range(2):
                b = bin_lo + 2 * (16 * tp + row) + par
                lanes = row + 16 * np.arange(4)                       # lane = row + 16 * (n & 3), k-step n >> 2
                coef = lambda q: np.array([afrag[tp, nn >> 2, q, lanes[nn & 3]] for nn in range(4 * ks)])
                cre, cim = coef(2 * par), coef(2 * par + 1)
                assert (cre[Q + 1:] == 0).all() and (cim[Q + 1:] == 0).all()
                if b < n_bins:
                    got[b] = complex(np.dot(cre[:Q + 1], ops[2 * par]), np.dot(cim[:Q + 1], ops[2 * par + 1]))
    k = np.arange(1, H + 1)
    for lb in range(lay["n_left"]):
        b = bin_lo + 32 * tpp + lb
        assert (left[lb, H:] == 0).all()                               # the zero columns behind the row
        if b < n_bins:
            got[b] = complex(np.dot(left[lb, :H, 0], x[k] + x[N - k]), np.dot(left[lb, :H, 1], x[k] - x[N - k]))
    for b in range(lo_bin, hi_bin + 1):
        assert b in got, b
        assert abs(got[b] - ref[b]) <= 1e-12 * max(1.0, np.abs(ref).max()), (b, got[b], ref[b])
    assert used == max(0, min(lay["n_left"], hi_bin - (bin_lo + 32 * tpp) + 1))


def test_bind_host_threads_is_a_noop_where_the_topology_is_unreadable():
    """No GPU here: the helper must leave the mask alone and say why (bench.py prints the string)."""
    import os
    from pcgmix_amd import hostprep
    before = os.sched_getaffinity(0)
    msg = hostprep.bind_host_threads(0)
    assert os.sched_getaffinity(0) == before
    assert "pinned" not in msg and msg


def test_cpu_slices_of_one_node_do_not_overlap():
    """Eight ranks of an 8-GPU host (four GPUs per socket, 64 cores + 64 SMT siblings per node):
    every GPU of a node gets its own eight physical cores, none of them the node's first eight."""
    from pcgmix_amd import hostprep
    node1 = list(range(64, 128)) + list(range(192, 256))
    seen = set()
    for ordinal in range(4):
        got = hostprep._slice_for(node1, ordinal, 8)
        assert len(got) == 8 and not (set(got) & seen) and min(got) >= 72 and max(got) < 128
        seen |= set(got)
    assert hostprep._slice_for(node1, 0, 8) == list(range(72, 80))
    # a small cgroup: whatever is there, at least two CPUs, never an empty mask
    assert hostprep._slice_for([3, 4, 5, 6], 5, 8) == [3, 4]
    assert hostprep._slice_for(list(range(16)), 1, 8) == list(range(8))


def test_capture_without_gc_restores_the_collector():
    import gc
    from pcgmix_amd import _lib
    assert gc.isenabled()
    with _lib.capture_without_gc():
        assert not gc.isenabled()
    assert gc.isenabled()
    gc.disable()
    try:
        with _lib.capture_without_gc():
            assert not gc.isenabled()
        assert not gc.isenabled()          # a caller that had it off keeps it off
    finally:
        gc.enable()
