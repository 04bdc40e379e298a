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


def _golden_potes(device):
    sd = np.load(GOLDEN + "/potes_state_seed1234.npz")
    model = models.CNN_potes_TS(4, 2, "PhysioNet")
    model.load_state_dict({k: torch.from_numpy(sd[k]) for k in sd.files})
    return model.to(device)


def _check_against_reference_golden(g, sal_gpu, disp_gpu, y_gpu, mix, eps_max=1e-5, vs_reference=True):
    """Everything the saliency-guided step produced on the GPU against (i) the oracle run on the
    GPU's own saliency maps and (ii) the reference's recorded run.

    The frozen model's backward runs through the HIP kernels, so its input gradient differs from
    the CPU reference's in the last bits and the saliency maps by eps = max|sal_gpu - sal_ref|
    (<= 1e-5 asserted below).  The displacement is an arg-max of a float32 objective
    J(d): it may differ from the recorded one ONLY at a near-tie.  For every state whose
    displacement differs this asserts, on the REFERENCE's saliency maps,
        0 <= J_ref(d_ref) - J_ref(d_gpu) <= 2 * (n1 + n2) * eps + float32 rounding slack
    (each map entry moves by at most eps, so J moves by at most (n1+n2)*eps at either candidate),
    and prints the exception.  Rows whose four displacements equal the recorded ones must match
    the recorded output to 1e-4."""
    method, frames = g["method"], g["frames"]
    assert np.array_equal(mix, g["mix"])
    eps = float(np.abs(sal_gpu - g["sal"]).max())
    if vs_reference:
        assert eps <= eps_max   # Potes: measured <= 2e-6; the near-tie bound below scales with it
    ref = O.augment(method, g["x"], g["labels"], frames, g["wav"], g["step"], saliency_maps=sal_gpu)
    # (i) the GPU chain == the oracle fed the same saliency: indices bit-exact, waveform 1e-4
    assert np.array_equal(ref["mix"], mix)
    assert np.array_equal(disp_gpu, ref["disp"]), "displacement kernel != oracle on identical saliency"
    assert np.abs(y_gpu - ref["y"]).max() <= 1e-4
    if not vs_reference:        # (maps from a non-deterministic model backward: part (i) only)
        return 0
    # (ii) against the reference's recorded run
    lam_np = np.full((1, 1), np.float32(g["lam"]), dtype=np.float32)
    differing = np.argwhere(disp_gpu != g["disp"])
    for i, k in differing:
        j = g["mix"][i]
        s1 = g["sal"][i][frames[i, k]:frames[i, k + 1]]
        s2 = g["sal"][j][frames[j, k]:frames[j, k + 1]]
        j_ref = float(O.displacement_objective(s1, s2, lam_np, int(g["disp"][i, k]), method))
        j_gpu = float(O.displacement_objective(s1, s2, lam_np, int(disp_gpu[i, k]), method))
        bound = 2.0 * (len(s1) + len(s2)) * eps + 1e-5 * max(1.0, abs(j_ref))
        import warnings
        warnings.warn(f"[salopt] proven near-tie: sample {i} state {k}: d_gpu={disp_gpu[i, k]} "
                      f"d_ref={g['disp'][i, k]} J_ref(d_ref)-J_ref(d_gpu)={j_ref - j_gpu:.3e} "
                      f"bound={bound:.3e} eps={eps:.2e}")     # reaches the log under -q too
        assert -1e-5 * max(1.0, abs(j_ref)) <= j_ref - j_gpu <= bound, (i, k, j_ref, j_gpu, bound)
    same = np.ones(len(frames), dtype=bool)
    same[differing[:, 0]] = False
    assert np.abs(y_gpu[same] - g["y"][same]).max() <= 1e-4
    return int(len(differing))


@pytest.mark.parametrize("path", CASES, ids=lambda p: p.split("/")[-1][:-4])
def test_salopt_augment_end_to_end(path, device):
    """BASELINE.json configs[2] end to end: augment() with the golden's frozen Potes checkpoint
    (handed over with set_saliency_model) against the reference's recorded run; see
    ``_check_against_reference_golden`` for what may differ and why."""
    g = load_golden(path)
    saliency.set_saliency_model(_golden_potes(device))
    try:
        data = torch.from_numpy(g["x"]).to(device)
        tgt = torch.nn.functional.one_hot(torch.from_numpy(g["labels"]), 2).to(device)
        B, C, T = data.shape
        sal = saliency.get_saliency_maps(Args(g["method"]), device, data, tgt, g["frames"])
        y, _, mix, _ = augmentations.augment(Args(g["method"]), data, tgt, torch.from_numpy(g["frames"]),
                                             g["wav"], StepCounter(g["step"]), None, device, "")
        fr, mx = dev_i32(g["frames"], device), dev_i32(mix, device)
        mode = 0 if "(saloptenv" in g["method"] else 1
        disp = saliency.optimal_displacements(sal, fr.data_ptr(), mx.data_ptr(),
                                              float(np.float32(g["lam"])), mode, B, T)
        torch.cuda.synchronize()
    finally:
        saliency.set_saliency_model(None)
    _check_against_reference_golden(g, sal.cpu().numpy(), disp.cpu().numpy().astype(np.int64),
                                    y.cpu().numpy(), mix)


def test_salopt_checkpoint_path_reaches_baseline_model(device, tmp_path):
    """The reference's way of getting the saliency model (saliency.py:26-51): no model is handed
    over; ``<EXPERIMENTS>/<experiment_dir of the 'base' run>/model.pth`` is read, its keys carry
    DataParallel's ``module.`` prefix.  Same maps and same augmentation as with the injected
    model; the second call is served from the cache; a rewritten checkpoint is picked up."""
    import argparse
    g = load_golden(CASES[2])
    sd = np.load(GOLDEN + "/potes_state_seed1234.npz")
    args = argparse.Namespace(
        dataset="PhysioNet", model="Potes", method=g["method"], num_epochs=50, batch_size=8,
        n_fraction=1.0, op="adam", use_sched=True, lr_max=0.01, train_balance=True, num_channels=4,
        grad_clip=0.1, seed_data=1100001, valid=False, seed=4, EXPERIMENTS=str(tmp_path),
        num_classes=2, sample_rate=1000)
    base = argparse.Namespace(**vars(args))
    base.method = "base"
    exp = saliency.experiment_dir(base)
    import os
    os.makedirs(exp)
    torch.save({"module." + k: torch.from_numpy(sd[k]) for k in sd.files}, os.path.join(exp, "model.pth"))
    saliency.set_saliency_model(None)
    saliency._LOADED.clear()
    data = torch.from_numpy(g["x"]).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(g["labels"]), 2).to(device)
    B, C, T = data.shape
    sal = saliency.get_saliency_maps(args, device, data, tgt, g["frames"])
    assert len(saliency._LOADED) == 1
    loaded = next(iter(saliency._LOADED.values()))[1]
    assert not loaded.training and not any(p.requires_grad for p in loaded.parameters())
    y, _, mix, _ = augmentations.augment(args, data, tgt, torch.from_numpy(g["frames"]), g["wav"],
                                         StepCounter(g["step"]), None, device, str(tmp_path))
    assert next(iter(saliency._LOADED.values()))[1] is loaded          # cached, not re-read
    fr, mx = dev_i32(g["frames"], device), dev_i32(mix, device)
    disp = saliency.optimal_displacements(sal, fr.data_ptr(), mx.data_ptr(),
                                          float(np.float32(g["lam"])), 0, B, T)
    _check_against_reference_golden(g, sal.cpu().numpy(), disp.cpu().numpy().astype(np.int64),
                                    y.cpu().numpy(), mix)
    # identical to the injected-model path
    saliency.set_saliency_model(_golden_potes(device))
    try:
        sal2 = saliency.get_saliency_maps(args, device, data, tgt, g["frames"])
    finally:
        saliency.set_saliency_model(None)
    assert torch.equal(sal, sal2)
    # a missing baseline run is an error, not a silent passthrough
    base.seed = 5
    args5 = argparse.Namespace(**vars(args))
    args5.seed = 5
    with pytest.raises(FileNotFoundError):
        saliency.get_saliency_maps(args5, device, data, tgt, g["frames"])


def test_salopt_full_size_vs_oracle(device):
    """BASELINE.json configs[2] at full size: (256,4,5000) '(saloptenv)durmixmagwarp(0.2,4)'.  The
    GPU's saliency maps are copied to the host and handed to the oracle; partner indices and
    displacements must be bit-exact, the warped waveform within 1e-4."""
    from pcgmix_amd import synthetic
    method, step = "(saloptenv)durmixmagwarp(0.2,4)", 17
    B, C, T = 256, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=31)
    torch.manual_seed(11)
    saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=T).to(device))
    try:
        data = torch.from_numpy(x).to(device)
        tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
        a = Args(method, sample_rate=2000, batch_size=B)
        sal = saliency.get_saliency_maps(a, device, data, tgt, frames)
        y, _, mix, _ = augmentations.augment(a, data, tgt, torch.from_numpy(frames), wav,
                                             StepCounter(step), None, device, "")
        sal_h = sal.cpu().numpy()
        ref = O.augment(method, x, labels, frames, wav, step, saliency_maps=sal_h)
        fr, mx = dev_i32(frames, device), dev_i32(mix, device)
        disp = saliency.optimal_displacements(sal, fr.data_ptr(), mx.data_ptr(),
                                              float(np.float32(ref["lam"])), 0, B, T)
        torch.cuda.synchronize()
    finally:
        saliency.set_saliency_model(None)
    assert sal_h.min() == 0.0 and sal_h.max() == 1.0 and np.isfinite(sal_h).all()
    assert np.array_equal(mix, ref["mix"])
    assert np.array_equal(disp.cpu().numpy().astype(np.int64), ref["disp"])
    assert (ref["disp"] > 0).mean() > 0.5                       # the search did real work
    assert np.abs(y.cpu().numpy() - ref["y"]).max() <= 1e-4


def test_bad_arguments(device):
    lib = pcgmix_amd._lib.load()
    z = torch.zeros(8, device=device)
    assert lib.pcgmix_saliency_post_f32(z.data_ptr(), z.data_ptr(), z.data_ptr(), 100,
                                        ctypes.c_double(12.0), 1, 1, 8, None) != 0     # even ksize
    assert lib.pcgmix_salopt_disp_f32(z.data_ptr(), z.data_ptr(), z.data_ptr(), ctypes.c_float(0.5),
                                      2, z.data_ptr(), z.data_ptr(), 0, 1, 8, None) != 0  # bad mode
    assert lib.pcgmix_salopt_disp_f32(z.data_ptr(), z.data_ptr(), z.data_ptr(), ctypes.c_float(0.5),
                                      0, z.data_ptr(), None, 0, 1, 8, None) != 0       # no workspace


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


def _index_batch(kind, B, T, rs):
    """(frames, mix) for the planned-launch tests: 'bench' = the benchmark's synthetic cycles,
    'equal' = every pair has equal state lengths (no search anywhere), 'wide' = gaps beyond 768
    samples in every state pair (more blocks than a plan holds at B = 256)."""
    from pcgmix_amd import synthetic
    if kind == "bench":
        frames = synthetic.make_index_data(B, T, sample_rate=2000, seed=3)[0]
    elif kind == "equal":
        frames = np.tile(np.array([0, 300, 1100, 1400, 2800]), (B, 1))
    else:
        a = np.array([0, 100, 200, 300, 400])
        w = np.array([0, 1000, 2000, 3000, 4000])
        frames = np.where((np.arange(B) % 2 == 0)[:, None], a, w)
    mix = rs.permutation(B) if kind != "wide" else (np.arange(B) ^ 1)
    return frames.astype(np.int64), mix.astype(np.int32)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("kind,B", [("bench", 256), ("bench", 37), ("equal", 64), ("wide", 256), ("wide", 64),
                                    ("bench", 1), ("bench", 300)])
def test_planned_search_equals_the_full_grid(kind, B, mode, device):
    """pcgmix_salopt_disp_hosted_f32 (host copies of boundaries and partners: the launch is the
    list of blocks with candidates, longest chain first) == pcgmix_salopt_disp_f32, bit for bit;
    B = 300 and the 'wide' batch at B = 256 (4096 blocks > kDispPlanMax) fall back to the grid."""
    T = 5000
    rs = np.random.RandomState(11 + B)
    frames, mix = _index_batch(kind, B, T, rs)
    sal = torch.from_numpy(rs.rand(B, T).astype(np.float32)).to(device)
    fr, mx = dev_i32(frames, device), dev_i32(mix, device)
    ref = saliency.optimal_displacements(sal, fr.data_ptr(), mx.data_ptr(), 0.37, mode, B, T)
    for ml in (0, int(np.diff(frames, axis=1).max())):
        got = saliency.optimal_displacements(sal, fr.data_ptr(), mx.data_ptr(), 0.37, mode, B, T, max_len=ml,
                                             frames_host=frames, mix_host=mix)
        assert torch.equal(got, ref)
    if kind == "bench" and B == 37:       # and against the oracle
        lam_np = np.full((1, 1), 0.37, dtype=np.float32)
        method = "(saloptenv)" if mode == 0 else "(saloptsum)"
        s = sal.cpu().numpy()
        want = np.stack([O.salopt_displacements(s[i], s[mix[i]], frames[i], frames[mix[i]], lam_np, method)
                         for i in range(B)])
        assert np.array_equal(got.cpu().numpy().astype(np.int64), want)


@pytest.mark.parametrize("mode", [0, 1])
def test_displacements_long_gaps_vs_oracle(mode, device):
    """Length gaps beyond one pass of the candidate split (> 1024 candidates per state) and gaps
    that leave some of the split's blocks without a candidate, against the numpy oracle."""
    B, T = 6, 9000
    frames = np.array([[0, 300, 2800, 3100, 8900], [0, 200, 700, 1000, 3400], [0, 310, 2000, 2300, 6000],
                       [0, 100, 350, 500, 900], [0, 300, 2800, 3100, 8900], [0, 250, 1400, 1650, 1660]],
                      dtype=np.int64)
    rs = np.random.RandomState(10 + mode)
    sal = (rs.rand(B, T) ** 3).astype(np.float32)
    sal[np.arange(T)[None, :] >= frames[:, 4:5]] = 0
    mix = np.array([1, 0, 3, 2, 5, 4])
    lam = np.float32(0.4142)
    lam_np = np.full((1, 1), lam, dtype=np.float32)
    tag = "(saloptenv)" if mode == 0 else "(saloptsum)"
    ref = np.stack([O.salopt_displacements(sal[i], sal[mix[i]], frames[i], frames[mix[i]], lam_np, tag)
                    for i in range(B)])
    fr, mx = dev_i32(frames, device), dev_i32(mix, device)
    got = saliency.optimal_displacements(torch.from_numpy(sal).to(device), fr.data_ptr(), mx.data_ptr(),
                                         float(lam), mode, B, T)
    assert np.array_equal(got.cpu().numpy().astype(np.int64), ref)
    assert ref.max() > 1024 or (np.abs(np.diff(frames, axis=1)[mix] - np.diff(frames, axis=1)).max() > 1024)


@pytest.mark.parametrize("B,T", [(8, 2500), (5, 5000)])
def test_direct_potes_input_gradient_equals_autograd(B, T, device):
    """The frozen CNN_potes' input gradient as a fixed chain of launches (no autograd, fused
    tail: dz = (z > 0) * (seed W2) without forming the logits) == the autograd path through
    PotesStackFunction / PotesHeadFunction, for one-hot and for soft seeds."""
    torch.manual_seed(B)
    model = models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=None if T == 2500 else T).to(device).eval()
    for p in model.parameters():
        p.requires_grad_(False)
    x = torch.randn(B, 4, T, device=device)
    tgt = torch.nn.functional.one_hot(torch.randint(0, 2, (B,)), 2).to(device)
    assert saliency._potes_direct(model, x) is model
    for seed in (saliency.class_seed(tgt), torch.rand(B, 2, device=device)):
        direct = saliency.input_gradient_seeded(model, x, seed)
        old = models.PotesStackFunction.use_masks
        models.PotesStackFunction.use_masks = False          # forces the autograd path
        try:
            assert saliency._potes_direct(model, x) is None
            recompute = saliency.input_gradient_seeded(model, x, seed)
        finally:
            models.PotesStackFunction.use_masks = old
        xa = x.detach().requires_grad_(True)
        with torch.enable_grad():
            (auto,) = torch.autograd.grad(model(xa), xa, seed)   # autograd, saved-routing kernels
        assert torch.equal(direct, auto)
        scale = float(auto.abs().max())
        assert float((direct - recompute).abs().max()) <= 1e-5 * scale
    model.train()                                            # training mode: not the direct chain
    assert saliency._potes_direct(model, x) is None


def test_label_kernel_writes_the_gradient_seed(device):
    """pcgmix_ctx_labels_begin's by-product: float one-hot of the FIRST maximum of every row."""
    from pcgmix_amd import _lib
    lib = _lib.load()
    B, K = 300, 3
    rs = np.random.RandomState(0)
    ohe = np.zeros((B, K), dtype=np.int64)
    ohe[np.arange(B), rs.randint(0, K, B)] = 1
    ohe[5] = 0                                               # all equal: first column wins
    ohe[6] = 1
    t = torch.from_numpy(ohe).to(device)
    seed = torch.full((B, K), 7.0, device=device)
    ctx = augmentations.step_context(device.index)
    st = torch.cuda.current_stream(device).cuda_stream
    _lib.check(lib.pcgmix_ctx_labels_begin(ctx, t.data_ptr(), K, B, seed.data_ptr(), st), "begin")
    labels = np.empty(B, dtype=np.int64)
    _lib.check(lib.pcgmix_ctx_labels_wait(ctx, labels.ctypes.data, B, st), "wait")
    assert np.array_equal(labels, ohe.argmax(1))
    assert torch.equal(seed, saliency.class_seed(t))
    assert seed[5].tolist() == [1.0, 0.0, 0.0] and seed[6].tolist() == [1.0, 0.0, 0.0]


@pytest.mark.parametrize("mode,warp", [(0, True), (1, False)])
def test_search_and_splice_in_one_call(mode, warp, device):
    """pcgmix_salopt_mix_warp_f32 (the splice kernel reduces the search's per-block results
    itself) == pcgmix_salopt_disp_f32 followed by pcgmix_mix_warp_f32, bit for bit."""
    from pcgmix_amd import _lib, synthetic
    lib = _lib.load()
    B, C, T = 48, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=21)
    frames[7] = frames[3]
    rs = np.random.RandomState(3)
    sal = rs.rand(B, T).astype(np.float32)
    sal[np.arange(T)[None, :] >= frames[:, 4:5]] = 0
    mix = rs.permutation(B)
    lam = float(np.float32(0.4321))
    data = torch.from_numpy(x).to(device)
    sal_d = torch.from_numpy(sal).to(device)
    fr, mx = dev_i32(frames, device), dev_i32(mix, device)
    knots = op = None
    n_knots = 0
    if warp:
        n_knots = 6
        knots = torch.from_numpy(rs.normal(1.0, 0.2, (B, n_knots, C))).to(device)
        op = augmentations.spline_operator(device, T, n_knots)
    max_len = int(np.diff(frames, axis=1).max())
    disp = saliency.optimal_displacements(sal_d, fr.data_ptr(), mx.data_ptr(), lam, mode, B, T,
                                          max_len=max_len)
    ref = torch.empty_like(data)
    augmentations.launch_mix(data, ref, fr.data_ptr(), mx.data_ptr(), disp.data_ptr(), lam,
                             knots.data_ptr() if warp else None, op.data_ptr() if warp else None,
                             n_knots, B, C, T)
    out = torch.empty_like(data)
    disp2 = torch.full((B, 4), -1, dtype=torch.int32, device=device)
    ws = torch.empty(lib.pcgmix_salopt_workspace_bytes(B) // 8, dtype=torch.int64, device=device)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    _lib.check(lib.pcgmix_salopt_mix_warp_f32(
        data.data_ptr(), out.data_ptr(), sal_d.data_ptr(), fr.data_ptr(), mx.data_ptr(),
        ctypes.c_float(lam), mode, knots.data_ptr() if warp else None, op.data_ptr() if warp else None,
        n_knots, ws.data_ptr(), max_len, disp2.data_ptr(), B, C, T, st), "salopt_mix")
    assert torch.equal(disp, disp2)
    assert int((disp != 0).sum()) > B                        # the search did move things
    assert torch.equal(out, ref)


def test_salopt_fast_path_equals_general_path(device):
    """The two-call step on the context (pcgmix_ctx_salopt_begin/_finish: seed from the label
    kernel, knots fetched by the search kernel, no finalize launch) == the general plan path
    (make_plan + upload_plan + pcgmix_salopt_mix_warp_f32), in strict mode and with the labels
    handed over; malformed boundaries raise ValueError from either."""
    from pcgmix_amd import hostprep, synthetic
    torch.manual_seed(1)
    saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet", sig_len=5000).to(device))
    try:
        x, frames, labels, wav = synthetic.make_batch(40, 4, 5000, sample_rate=2000, seed=31)
        data = torch.from_numpy(x).to(device)
        tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
        for method in ("(saloptenv)durmixmagwarp(0.2,4)", "(saloptsum)durratiomixup"):
            a = Args(method)
            assert hostprep.salopt_recipe(method) is not None
            fast = augmentations.augment(a, data, tgt, frames, wav, StepCounter(3), None, device, "")
            fast_h = augmentations.augment(a, data, tgt, frames, wav, StepCounter(3), None, device, "",
                                           host_labels=labels)
            real = hostprep.salopt_recipe
            hostprep.salopt_recipe = lambda m: None            # forces the general path
            try:
                slow = augmentations.augment(a, data, tgt, frames, wav, StepCounter(3), None, device, "")
            finally:
                hostprep.salopt_recipe = real
            for got in (fast, fast_h):
                assert np.array_equal(got[2], slow[2]) and torch.equal(got[0], slow[0])
                assert got[1] is tgt and got[3] is None
            assert not torch.equal(fast[0], data)
        assert hostprep.salopt_recipe("(samePCG)(saloptenv)durratiomixup") is None
        bad = frames.copy()
        bad[2, 4] = 5001
        with pytest.raises(ValueError):
            augmentations.augment(Args("(saloptenv)durratiomixup"), data, tgt, bad, wav, StepCounter(3),
                                  None, device, "")
        bad = frames.copy()
        bad[1, 2] = bad[1, 1] - 1
        with pytest.raises(ValueError):
            augmentations.augment(Args("(saloptenv)durratiomixup"), data, tgt, bad, wav, StepCounter(3),
                                  None, device, "")
        ok = augmentations.augment(Args("(saloptenv)durratiomixup"), data, tgt, frames, wav, StepCounter(3),
                                   None, device, "")          # the context is still usable afterwards
        assert ok[0].shape == data.shape
    finally:
        saliency.set_saliency_model(None)


def test_salopt_fast_path_beyond_the_kernarg_batch(device):
    """B > 256: boundaries and partners no longer fit the kernel arguments — the two-call step
    then sends them through the context's staging ring; results == the general plan path."""
    from pcgmix_amd import hostprep, synthetic
    torch.manual_seed(2)
    saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet").to(device))
    try:
        B = 300
        x, frames, labels, wav = synthetic.make_batch(B, 4, 2500, seed=44)
        data = torch.from_numpy(x).to(device)
        tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
        a = Args("(saloptsum)durmixmagwarp(0.2,4)")
        fast = augmentations.augment(a, data, tgt, frames, wav, StepCounter(9), None, device, "")
        real = hostprep.salopt_recipe
        hostprep.salopt_recipe = lambda m: None
        try:
            slow = augmentations.augment(a, data, tgt, frames, wav, StepCounter(9), None, device, "")
        finally:
            hostprep.salopt_recipe = real
        assert np.array_equal(fast[2], slow[2]) and torch.equal(fast[0], slow[0])
        plain = augmentations.augment(Args("durratiomixup"), data, tgt, frames, wav, StepCounter(9), None,
                                      device, "")               # plain step beyond 256: staging slot
        ref = O.augment("durratiomixup", x, labels, frames, wav, 9)
        assert np.array_equal(plain[2], ref["mix"])
        assert np.abs(plain[0].cpu().numpy() - ref["y"]).max() <= 1e-4
    finally:
        saliency.set_saliency_model(None)


def test_salopt_begin_with_host_labels_all_paths(device):
    """pcgmix_ctx_salopt_begin_labels (round 3): labels handed over on the host, no read-back.
    B <= 256: seed, boundaries and a pending payload ride in ONE launch's arguments; B > 256: a
    staged [frames | labels] block and a kernel.  Both == the read-back form; a payload set on
    the context arrives at its destination (small batch) or stays pending for flush (large);
    malformed boundaries and out-of-range labels are refused."""
    import ctypes
    from pcgmix_amd import _lib, synthetic
    lib = _lib.load()
    torch.manual_seed(2)
    saliency.set_saliency_model(models.CNN_potes_TS(4, 2, "PhysioNet").to(device))
    try:
        for B in (48, 300):
            x, frames, labels, wav = synthetic.make_batch(B, 4, 2500, seed=41 + B)
            data = torch.from_numpy(x).to(device)
            tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
            a = Args("(saloptenv)durmixmagwarp(0.2,4)")
            ref = augmentations.augment(a, data, tgt, frames, wav, StepCounter(5), None, device, "")
            ctx = augmentations.step_context(device.index)
            pay = np.arange(40, dtype=np.float32) + B
            dst = torch.zeros(40, device=device)
            _lib.check(lib.pcgmix_ctx_set_payload(ctx, pay.ctypes.data, pay.nbytes, dst.data_ptr()), "payload")
            got = augmentations.augment(a, data, tgt, frames, wav, StepCounter(5), None, device, "",
                                        host_labels=labels)
            assert np.array_equal(got[2], ref[2]) and torch.equal(got[0], ref[0])
            if B <= 256:                       # rode in the begin kernel's arguments
                assert np.array_equal(dst.cpu().numpy(), pay)
            else:                              # still pending: goes on its own
                assert float(dst.abs().sum()) == 0.0
                _lib.check(lib.pcgmix_ctx_flush_payload(
                    ctx, ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "flush")
                assert np.array_equal(dst.cpu().numpy(), pay)
            bad = frames.copy()
            bad[1, 4] = 2501
            with pytest.raises(ValueError):
                augmentations.augment(a, data, tgt, bad, wav, StepCounter(5), None, device, "",
                                      host_labels=labels)
            wrong = labels.copy()
            wrong[0] = 7                       # not a class of the (B, 2) seed
            with pytest.raises(RuntimeError):
                augmentations.augment(a, data, tgt, frames, wav, StepCounter(5), None, device, "",
                                      host_labels=wrong)
    finally:
        saliency.set_saliency_model(None)


# ---- spectrograms: saliency.get_saliency_maps(dim=2) and '(salopt…)durratiomixup' in 2D (round 4) ----
CASES2D = golden_files("salopt2d_")


def _golden_args2d(method, experiments):
    import argparse
    return argparse.Namespace(
        dataset="PhysioNet(spec128)", model="resnet9", method=method, num_epochs=50, batch_size=6,
        n_fraction=1.0, op="adam", use_sched=True, lr_max=0.01, train_balance=True, num_channels=1,
        grad_clip=0.1, seed_data=1100001, valid=False, seed=4, EXPERIMENTS=experiments,
        num_classes=2, sample_rate=1000)


def _write_resnet2d_base_checkpoint(tmp_path):
    """The golden's frozen saliency model: models2d.ResNet9 under torch.manual_seed(4321) — the same
    weights the reference's factory draws (make_golden_salopt2d.py; test_model_goldens) — saved the
    way the reference's 'base' run saves it (DataParallel prefix, train_model.py:481-482)."""
    import os
    import sys
    sys.path.insert(0, GOLDEN)
    from make_golden_salopt2d import SEED2D
    from pcgmix_amd import models2d
    base = _golden_args2d("base", str(tmp_path))
    exp = saliency.experiment_dir(base)
    os.makedirs(exp, exist_ok=True)
    torch.manual_seed(SEED2D)
    net = models2d.ResNet9(num_classes=2)
    torch.save({"module." + k: v for k, v in net.state_dict().items()}, os.path.join(exp, "model.pth"))


def test_saliency_post2d_matches_reference(device):
    """pcgmix_saliency_post2d_f32 on the reference's recorded input gradient == the reference's
    maps (saliency.py:93-113) up to the float32 summation order (128 rows, 11 taps)."""
    g = load_golden(CASES2D[0])
    grad = torch.from_numpy(g["grad"]).to(device)
    fr = dev_i32(g["frames"], device)
    sal = saliency.saliency_post2d(grad, fr.data_ptr()).cpu().numpy()
    assert sal.shape == g["sal"].shape == (6, 128)
    assert np.abs(sal - g["sal"]).max() <= 2e-6
    for b, f in enumerate(g["frames"]):
        assert (sal[b, f[4]:] == 0).all() and sal[b, :f[4]].min() == 0.0 and sal[b, :f[4]].max() == 1.0
    # and against the oracle on a ragged batch: empty cycle, full-width cycle, W not a power of two
    rs = np.random.RandomState(5)
    grad2 = rs.standard_normal((5, 1, 40, 96)).astype(np.float32)
    fr2 = np.array([[0, 5, 9, 20, 96], [0, 1, 2, 3, 4], [0, 10, 30, 50, 61], [0, 2, 4, 6, 8],
                    [0, 20, 40, 60, 95]], dtype=np.int64)
    got = saliency.saliency_post2d(torch.from_numpy(grad2).to(device), dev_i32(fr2, device).data_ptr())
    assert np.abs(got.cpu().numpy() - O.saliency_post2d(grad2.copy(), fr2)).max() <= 2e-6


@pytest.mark.parametrize("path", CASES2D, ids=lambda p: p.split("/")[-1][:-4])
def test_displacements2d_bit_exact_given_reference_saliency(path, device):
    """The search kernel on the reference's recorded (B, W) maps: short states (2 .. 40 columns,
    the sequential n < 8 and the 8-accumulator branches of numpy's pairwise sum)."""
    g = load_golden(path)
    B, W = g["sal"].shape
    sal = torch.from_numpy(g["sal"]).to(device)
    fr, mix = dev_i32(g["frames"], device), dev_i32(g["mix"], device)
    mode = 0 if "(saloptenv" in g["method"] else 1
    disp = saliency.optimal_displacements(sal, fr.data_ptr(), mix.data_ptr(),
                                          float(np.float32(g["lam"])), mode, B, W)
    assert np.array_equal(disp.cpu().numpy().astype(np.int64), g["disp"])


@pytest.mark.parametrize("deterministic", [False, True], ids=["miopen-default", "miopen-deterministic"])
@pytest.mark.parametrize("path", CASES2D, ids=lambda p: p.split("/")[-1][:-4])
def test_salopt2d_augment_end_to_end(path, deterministic, device, tmp_path):
    """augmentations2d.augment with '(saloptenv|saloptsum)durratiomixup' (augmentations2d.py:416-423)
    end to end against the reference's recorded run: the ResNet9-2D 'base' checkpoint is read from
    where ``utils.experiment_dir`` puts it, its input gradient comes from MIOpen instead of oneDNN
    (maps within 1e-5; measured 4e-7 with MIOpen's deterministic algorithms), displacements may differ from the recorded ones only at proven
    near-ties (``_check_against_reference_golden``), partners and lambda are exact."""
    g = load_golden(path)
    _write_resnet2d_base_checkpoint(tmp_path)
    saliency.set_saliency_model(None)
    saliency._LOADED.clear()
    args = _golden_args2d(g["method"], str(tmp_path))
    data = torch.from_numpy(g["x"]).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(g["labels"]), 2).to(device)
    B, _, F, W = data.shape
    from pcgmix_amd import augmentations2d
    # The maps augment() itself used are taken from inside the call (robust against any
    # run-to-run variation of the model's backward; with the deterministic algorithms this test
    # switches on there is none).
    used = []
    real = saliency.get_saliency_maps

    def recording(*a, **k):
        used.append(real(*a, **k))
        return used[-1]
    saliency.get_saliency_maps = recording
    saliency.DETERMINISTIC_FROZEN_PASS = deterministic   # True: parity mode; 40-77x slower at bs 256, hence opt-in
    try:
        y, t_out, mix, cut = augmentations2d.augment(args, data, tgt, torch.from_numpy(g["frames"]), g["wav"],
                                                     StepCounter(g["step"]), None, device, str(tmp_path))
    finally:
        saliency.get_saliency_maps = real
        saliency.DETERMINISTIC_FROZEN_PASS = None
    assert cut is None and t_out is tgt and y.shape == data.shape
    assert len(used) == 1
    sal = used[0]
    assert sal.shape == (B, W)
    fr, mx = dev_i32(g["frames"], device), dev_i32(mix, device)
    mode = 0 if "(saloptenv" in g["method"] else 1
    disp = saliency.optimal_displacements(sal, fr.data_ptr(), mx.data_ptr(),
                                          float(np.float32(g["lam"])), mode, B, W)
    torch.cuda.synchronize()
    # The input gradient of an eight-convolution network through MIOpen's fp32 kernels instead of
    # oneDNN's.  With MIOpen's default algorithm choice the maps were 3.1e-5 from the reference's on
    # one box and 1.8e-3 on another, and moved between two passes over the same batch; this test
    # asks for the deterministic algorithms (saliency.DETERMINISTIC_FROZEN_PASS): 4.2e-7,
    # bit-reproducible (profiles/r4_sal2d_determinism.txt).
    import warnings
    eps = float(np.abs(sal.cpu().numpy() - g["sal"]).max())
    warnings.warn(f"[salopt2d] max |saliency map - reference| = {eps:.2e}")
    # MIOpen's default selection is not reproducible from box to box (3e-5 on one, 1.8e-3 on another):
    # that leg checks the chain against the oracle on the maps augment() itself used and only REPORTS
    # the distance to the reference's maps; the deterministic leg holds the reference to 1e-5.
    _check_against_reference_golden(g, sal.cpu().numpy(), disp.cpu().numpy().astype(np.int64),
                                    y.cpu().numpy(), mix, eps_max=1e-5, vs_reference=deterministic)


def test_salopt2d_on_reference_saliency_is_exact(device):
    """With the reference's recorded maps injected, the 2D saliency-guided splice is the
    reference's output bit for bit (no model in the loop: search + offset splice only)."""
    from pcgmix_amd import hostprep
    for path in CASES2D:
        g = load_golden(path)
        B, _, F, W = g["x"].shape
        plan = hostprep.make_plan(g["method"], g["labels"], g["frames"], g["wav"], g["step"], B, F,
                                  is2d=True, n_cols=W)
        assert plan.fired and plan.salopt_mode == (0 if "(saloptenv" in g["method"] else 1)
        assert np.array_equal(plan.mix, g["mix"]) and plan.lam64 == float(g["lam"])
        data = torch.from_numpy(g["x"]).to(device)
        y = augmentations.apply_plan(plan, data.view(B, F, W), g["frames"],
                                     torch.from_numpy(g["sal"]).to(device)).view(B, 1, F, W)
        assert np.array_equal(y.cpu().numpy(), g["y"])


def test_mask_variants_ignore_salopt_in_2d(device):
    """Only the durratiomixup branch looks at '(salopt' (augmentations2d.py:416-423): the mask
    variants (:286-395) splice at offset 0 whatever the method string says."""
    from pcgmix_amd import augmentations2d
    g = load_golden(golden_files("mask2d_")[0])
    data = torch.from_numpy(g["x"]).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(g["labels"]), 2).to(device)
    out = []
    for m in (g["method"], "(saloptenv)" + g["method"]):
        y, _, mix, _ = augmentations2d.augment(Args(m), data, tgt, torch.from_numpy(g["frames"]), g["wav"],
                                               StepCounter(g["step"]), None, device, "")
        out.append(y.cpu().numpy())
    assert np.array_equal(out[0], out[1]) and np.array_equal(out[0], g["y"])


# ---- a7 at BASELINE config 3's size with a TRAINED saliency model (round 4) ----
def _trained_potes(device, B, T, steps=300):
    """A Potes 1D-CNN trained for ``steps`` steps on separable synthetic cycles (class-1 cycles
    carry a three times louder 80-200 Hz band, as conftest.learnable_dataset): the saliency maps of
    a trained model are peaked, unlike a random-init one's."""
    import argparse
    from pcgmix_amd import synthetic, train_model as tm
    args = argparse.Namespace(dataset="PhysioNet", model="Potes", method="base", num_epochs=50,
                              batch_size=B, op="adam", use_sched=True, lr_max=0.003, weight_decay=1e-4,
                              grad_clip=0.1, seed=4, num_classes=2, num_channels=4, sig_len=T, depth=0,
                              num_steps=steps, sample_rate=2000)
    torch.manual_seed(11)
    net = tm.build_model(args).to(device).train()
    opt, sched = tm.make_optimizer(args, net)
    crit = tm.SELCLoss(np.zeros(B * 4, int), 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    pool = []
    for i in range(4):
        x, frames, labels, wav = synthetic.make_batch(B, 4, T, sample_rate=2000, seed=600 + i)
        x[labels == 1, 2] *= 3.0
        pool.append((torch.from_numpy(x).to(device), torch.from_numpy(labels), torch.from_numpy(frames), wav,
                     torch.ones(B, dtype=torch.long), torch.arange(B)))
    losses = []
    for s in range(steps):
        losses.append(tm.train_step(args, net, pool[s % 4], device, opt, sched, crit, 1, sc))
    first, last = float(torch.stack(losses[:10]).mean()), float(torch.stack(losses[-10:]).mean())
    assert last < 0.5 * first, (first, last)            # it did learn
    cpu = tm.build_model(args)                          # the same architecture for the CPU side
    cpu.load_state_dict({k: v.detach().cpu() for k, v in net.state_dict().items()})
    return net.eval(), cpu.eval()


@pytest.mark.parametrize("mode", ["(saloptenv)", "(saloptsum)"])
def test_displacement_flip_rate_with_a_trained_saliency_model(mode, device, record_property):
    """north_star: "bit-exact for segment index selection".  The search itself is (given identical
    maps, the tests above); end to end the frozen model's backward runs on HIP instead of oneDNN,
    the maps differ in the last bits and an arg-max over a float32 objective can flip at a
    near-tie.  Round 3 knew the rate (2.5 %) only for a random-init model on 8-sample batches.
    Here, at BASELINE config 3's size — (256, 4, 5000), 1024 (sample, state) pairs — with a TRAINED
    model: the oracle (CPU-torch saliency of the same weights, numpy search) against the HIP chain.
    Every differing displacement must be a proven near-tie of the reference objective, and the
    rate is reported (record_property) and bounded."""
    from pcgmix_amd import synthetic
    B, C, T = 256, 4, 5000
    net, cpu = _trained_potes(device, B, T)
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=77)
    x[labels == 1, 2] *= 3.0
    # reference side: same weights on the CPU through torch's own kernels, the oracle's post-processing
    sal_ref = O.saliency_post(O.input_gradient(cpu, x, labels), frames)
    # HIP side
    saliency.set_saliency_model(net)
    try:
        data = torch.from_numpy(x).to(device)
        tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
        sal_gpu = saliency.get_saliency_maps(Args(mode + "durratiomixup"), device, data, tgt, frames)
    finally:
        saliency.set_saliency_model(None)
    eps = float(np.abs(sal_gpu.cpu().numpy() - sal_ref).max())
    assert eps <= 1e-5
    step = 3
    mix = O.mix_indices(mode + "durratiomixup", labels, wav, step)
    lam = np.float32(O.get_lambda(1.0, step))
    lam_np = np.full((1, 1), lam, dtype=np.float32)
    fr, mx = dev_i32(frames, device), dev_i32(mix, device)
    m = 0 if "env" in mode else 1
    disp_gpu = saliency.optimal_displacements(sal_gpu, fr.data_ptr(), mx.data_ptr(), float(lam), m, B, T)
    disp_gpu = disp_gpu.cpu().numpy().astype(np.int64)
    disp_ref = np.stack([O.salopt_displacements(sal_ref[i], sal_ref[mix[i]], frames[i], frames[mix[i]],
                                                lam_np, mode) for i in range(B)])
    # the kernel on the REFERENCE's maps is the reference's search, bit for bit
    on_ref = saliency.optimal_displacements(torch.from_numpy(sal_ref).to(device), fr.data_ptr(),
                                            mx.data_ptr(), float(lam), m, B, T)
    assert np.array_equal(on_ref.cpu().numpy().astype(np.int64), disp_ref)
    flips = np.argwhere(disp_gpu != disp_ref)
    searched = int((np.diff(frames, axis=1) != np.diff(frames[mix], axis=1)).sum())
    for i, k in flips:
        j = mix[i]
        s1 = sal_ref[i][frames[i, k]:frames[i, k + 1]]
        s2 = sal_ref[j][frames[j, k]:frames[j, k + 1]]
        j_ref = float(O.displacement_objective(s1, s2, lam_np, int(disp_ref[i, k]), mode))
        j_gpu = float(O.displacement_objective(s1, s2, lam_np, int(disp_gpu[i, k]), mode))
        bound = 2.0 * (len(s1) + len(s2)) * eps + 1e-5 * max(1.0, abs(j_ref))
        assert -1e-5 * max(1.0, abs(j_ref)) <= j_ref - j_gpu <= bound, (i, k, j_ref, j_gpu, bound)
    rate = len(flips) / max(1, searched)
    record_property("flipped_states", int(len(flips)))
    record_property("searched_states", searched)
    record_property("max_abs_saliency_difference", eps)
    import warnings
    warnings.warn(f"[a7 flip rate] {mode} trained model, (256,4,5000): {len(flips)} of {searched} searched "
                  f"states differ from the CPU reference (all proven near-ties), eps={eps:.2e}")
    assert rate <= 0.01, (len(flips), searched)
