"""Round-3 train-path tests on the GPU: BASELINE config 3 (saliency-guided method + 1D-CNN) as an
eager and as a captured step and against the reference's own recorded trajectory; the reference's
ResNet9 in TRAIN mode (1D and 2D) replayed on the HIP path; BASELINE config 4 as one chain."""
import argparse
import os
import sys
import warnings

import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import augmentations, augmentations2d, frontend, hostprep, models, models2d, saliency, \
    synthetic, train_model as tm
from conftest import GOLDEN
from oracle import pcgmix_oracle as O

if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)
import train_cases as TC  # noqa: E402

pytestmark = pytest.mark.gpu


def _write_base_checkpoint(args):
    """The 'base' run's model.pth where saliency.py:26-51 looks for it, in the reference's
    DataParallel key layout, holding the weights of potes_state_seed1234.npz."""
    import copy
    sd = np.load(os.path.join(GOLDEN, "potes_state_seed1234.npz"))
    base = copy.copy(args)
    base.method = "base"
    exp = saliency.experiment_dir(base)
    os.makedirs(exp, exist_ok=True)
    torch.save({"module." + k: torch.from_numpy(sd[k]) for k in sd.files}, os.path.join(exp, "model.pth"))


def _potes(args, device):
    torch.manual_seed(7)
    net = tm.build_model(args)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net.to(device).train()


def _salopt_run(device, tmp_path, mode, batches, args, ref_sal=None):
    """SALOPT_STEPS steps of config 3's method; per step the loss, the saliency maps the step
    used and the displacements recomputed from them (the step's own live in device scratch).
    ``ref_sal`` (steps, B, T): the captured saliency pass still runs, but its static output is
    overwritten with these maps before search and splice read it (the reference's recorded maps:
    takes the last-bit differences of the HIP saliency kernels out of the comparison)."""
    _write_base_checkpoint(args)
    net = _potes(args, device)
    opt, sched = tm.make_optimizer(args, net)
    labels_all = np.concatenate([b[1].numpy() for b in batches])
    crit = tm.SELCLoss(labels_all, 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    B = batches[0][0].shape[0]
    T = batches[0][0].shape[2]
    if mode == "graph":
        gstep = tm.GraphedTrainStep(args, net, opt, sched, crit, device, B, 4, T)
        step = lambda b: gstep.step(b, 1, sc)                                   # noqa: E731
    else:
        step = lambda b: tm.train_step(args, net, b, device, opt, sched, crit, 1, sc)  # noqa: E731
    losses, lrs, sals, disps, mixes, lams = [], [], [], [], [], []
    if ref_sal is not None:
        g0 = saliency.step_graph(args, batches[0][0].to(device), 2)
        ref_dev = torch.from_numpy(np.ascontiguousarray(ref_sal)).to(device)
        orig_replay = g0.replay

        def replay(data):
            orig_replay(data)
            g0.sal.copy_(ref_dev[sc.count])
            return g0.sal
        g0.replay = replay
    for b in batches:
        s = sc.count
        lrs.append(opt.param_groups[0]["lr"])
        dev_b = (b[0].to(device),) + tuple(b[1:])
        losses.append(float(step(dev_b)))
        g = saliency.step_graph(args, dev_b[0], 2)          # the captured pass the step replayed
        sal = g.sal.clone()
        mix = hostprep.partner_indices(args.method, b[1].numpy(), b[3], s)
        np.random.seed(s)
        lam = float(np.float32(np.random.beta(1.0, 1.0)))
        mx = torch.from_numpy(mix.astype(np.int32)).to(device)
        disp = saliency.optimal_displacements(sal, g.fr.data_ptr(), mx.data_ptr(), lam, 0, B, T)
        sals.append(sal.cpu().numpy())
        disps.append(disp.cpu().numpy().astype(np.int64))
        mixes.append(mix)
        lams.append(lam)
    torch.cuda.synchronize()
    state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    return dict(losses=np.asarray(losses), lrs=np.asarray(lrs), sal=np.stack(sals), disp=np.stack(disps),
                mix=np.stack(mixes), lam=np.asarray(lams), state=state)


def test_salopt_train_step_graph_matches_eager(device, tmp_path):
    """BASELINE config 3 as a training step: ``(saloptenv)durmixmagwarp(0.2,4)`` at (32,4,2500),
    saliency model loaded from a ``model.pth`` (saliency.py:26-51), dropout off.  The captured
    step (seed/boundaries/payload kernel -> captured saliency pass -> search + splice into the
    training graph's static input -> replay) == the eager ``train_step`` over 5 steps."""
    B, T = 32, 2500
    batches = []
    for i in range(5):
        x, frames, labels, wav = synthetic.make_batch(B, 4, T, sample_rate=1000, seed=700 + i)
        batches.append((torch.from_numpy(x), torch.from_numpy(labels), torch.from_numpy(frames), wav,
                        torch.ones(B, dtype=torch.long), torch.arange(B)))
    res = {}
    for mode in ("step", "graph"):
        args = TC.salopt_traj_args(str(tmp_path / mode))
        args.batch_size = B
        res[mode] = _salopt_run(device, tmp_path, mode, batches, args)
    a, b = res["step"], res["graph"]
    assert np.array_equal(a["mix"], b["mix"]) and np.array_equal(a["disp"], b["disp"])
    assert np.abs(a["sal"] - b["sal"]).max() <= 1e-6
    assert np.allclose(a["losses"], b["losses"], rtol=1e-4, atol=1e-5), (a["losses"], b["losses"])
    for k, v in a["state"].items():
        assert np.allclose(v, b["state"][k], rtol=1e-3, atol=1e-4), k
    assert np.all(np.isfinite(a["losses"])) and len(set(np.round(a["losses"], 6))) > 1


@pytest.mark.parametrize("mode", ["step", "graph"])
def test_salopt_train_step_reproduces_reference_trajectory(mode, device, tmp_path, record_property):
    """The reference's own ``train_epoch`` with ``(saloptenv)durmixmagwarp(0.2,4)`` (5 steps of 8
    cycles, recorded by tests/golden/make_golden_train_r3.py) replayed on the HIP path.

    Partner indices, lambda and learning rates are exact.  The saliency maps come from HIP kernels
    and differ from the CPU reference's in the last bits (asserted <= 1e-5); a displacement — the
    arg-max of a float32 objective — may therefore differ from the recorded one ONLY at a
    near-tie, proven on the REFERENCE's maps exactly as tests/test_saliency_gpu.py does.  While
    every displacement equals the recorded one the loss must agree to 1e-4 and, if that holds to
    the end, every parameter to 1e-3; after a proven near-tie the mixed waveforms legitimately
    differ in that state's samples, and the remaining steps are held to 1e-2."""
    g = np.load(os.path.join(GOLDEN, "train_salopt_ref.npz"))
    args = TC.salopt_traj_args(str(tmp_path))
    batches = TC.salopt_traj_batches()
    r = _salopt_run(device, tmp_path, mode, batches, args)
    assert np.allclose(r["lrs"], g["lrs"], rtol=1e-12, atol=1e-15)
    assert np.array_equal(r["mix"], g["mix"])
    assert np.allclose(r["lam"], g["lam"].astype(np.float32), rtol=0, atol=0)
    diverged, notes = False, []
    for s, b in enumerate(batches):
        frames = b[2].numpy()
        eps = float(np.abs(r["sal"][s] - g["sal"][s]).max())
        assert eps <= 1e-5, (s, eps)         # the saliency model is frozen: independent of the run
        lam_np = np.full((1, 1), np.float32(g["lam"][s]), dtype=np.float32)
        for i, k in np.argwhere(r["disp"][s] != g["disp"][s]):
            j = g["mix"][s][i]
            s1 = g["sal"][s][i][frames[i, k]:frames[i, k + 1]]
            s2 = g["sal"][s][j][frames[j, k]:frames[j, k + 1]]
            j_ref = float(O.displacement_objective(s1, s2, lam_np, int(g["disp"][s][i, k]), args.method))
            j_gpu = float(O.displacement_objective(s1, s2, lam_np, int(r["disp"][s][i, k]), args.method))
            bound = 2.0 * (len(s1) + len(s2)) * eps + 1e-5 * max(1.0, abs(j_ref))
            notes.append(f"step {s} sample {i} state {k}: d_gpu={r['disp'][s][i, k]} "
                         f"d_ref={g['disp'][s][i, k]} dJ={j_ref - j_gpu:.3e} bound={bound:.3e}")
            assert -1e-5 * max(1.0, abs(j_ref)) <= j_ref - j_gpu <= bound, notes[-1]
        if np.any(r["disp"][s] != g["disp"][s]):
            diverged = True                  # this step's batch already differs in those samples
        tol = 1e-2 if diverged else 1e-4
        assert abs(r["losses"][s] - g["losses"][s]) <= tol, (s, r["losses"][s], g["losses"][s], notes)
    worst = 0.0
    for k in g.files:
        if k.startswith("final."):
            d = float(np.abs(r["state"][k[len("final."):]] - g[k]).max())
            worst = max(worst, d)
            assert d <= (1e-2 if diverged else 1e-3), (k, d)
    record_property("near_ties", notes)
    if notes:
        warnings.warn("salopt trajectory: proven near-tie displacement(s): " + "; ".join(notes))
    print(f"[salopt traj {mode}] max loss err {np.abs(r['losses'] - g['losses']).max():.2e}, "
          f"max param err {worst:.2e}, near ties {len(notes)}")


@pytest.mark.parametrize("mode", ["step", "graph"])
def test_salopt_train_step_on_reference_saliency_is_strict(mode, device, tmp_path):
    """The same replay with the REFERENCE's recorded saliency maps in place of the GPU's own (the
    captured pass runs, its output is overwritten): now nothing may differ — every displacement
    equals the recorded one, every loss agrees to 1e-4, every parameter to 1e-3 after step 5.
    Together with the test above (own maps: <= 1e-5 from the reference's, differing displacements
    proven near-ties) this pins config 3's train step to the reference's train_epoch."""
    g = np.load(os.path.join(GOLDEN, "train_salopt_ref.npz"))
    args = TC.salopt_traj_args(str(tmp_path))
    r = _salopt_run(device, tmp_path, mode, TC.salopt_traj_batches(), args, ref_sal=g["sal"])
    assert np.array_equal(r["sal"], g["sal"])
    assert np.array_equal(r["mix"], g["mix"]) and np.array_equal(r["disp"], g["disp"])
    assert np.allclose(r["lrs"], g["lrs"], rtol=1e-12, atol=1e-15)
    err = float(np.abs(r["losses"] - g["losses"]).max())
    assert err <= 1e-4, (r["losses"], g["losses"])
    worst = 0.0
    for k in g.files:
        if k.startswith("final."):
            d = float(np.abs(r["state"][k[len("final."):]] - g[k]).max())
            worst = max(worst, d)
            assert d <= 1e-3, (k, d)
    print(f"[salopt traj on reference saliency, {mode}] max loss err {err:.2e}, max param err {worst:.2e}")


def _digest_check(tag, name, got, want, atol, frac_loose=0.0, loose=0.0):
    d = np.abs(TC.tensor_digest(got)[2:] - want[2:])
    n_bad = int((d > atol).sum())
    assert n_bad <= max(1.0 if frac_loose else 0.0, frac_loose * d.size) and (d.max() <= loose if n_bad else True), \
        (tag, name, float(d.max()), n_bad, d.size)
    return float(d.max())


@pytest.mark.parametrize("tag", ["r1d", "r2d"])
def test_resnet9_train_mode_reproduces_reference(tag, device):
    """The reference's ResNet9 in TRAIN mode — ``models.ResNet9(4,2)`` at (8,4,2500) with
    ``durmixmagwarp(0.2,4)``, ``models2d.ResNet9(2)`` at (4,1,128,128) with 2D ``durratiomixup`` —
    3 steps of its own ``train_epoch`` (tests/golden/make_golden_train_r3.py), replayed through
    ``train_step`` on the HIP path: NHWC MIOpen convolutions and the hand-written BatchNorm /
    ReLU / pool kernels (``pcgmix_bnrp.hip``: batch statistics, running-stat updates, backward),
    ClipAdam.  Losses 1e-4 relative, BN buffers 1e-4, parameters on the recorded digest.

    The goldens run at lr_max = 1e-4 (train_cases.RESNET_LR_MAX: at the reference's 0.01 three
    steps of this network amplify last-bit convolution differences to 4e-3 of the loss), so the
    three updates move every weight by sum(lr) = 2.2e-5 in all — the parameter tolerance is set
    against THAT, not against the weights: 98 % of a tensor's sampled elements within 2e-6.
    Adam's first updates are sign-like (m/sqrt(v) = +-1): an element whose gradient is at
    rounding-noise level may legitimately move the other way (2 lr per step), hence the 2 %
    (measured: up to 1 % of a 2D residual block's weights at batch 4) that may differ by up to
    5e-5."""
    g = np.load(os.path.join(GOLDEN, "train_resnet_ref.npz"))
    if tag == "r1d":
        args, batches = TC.resnet1d_args(), TC.resnet1d_batches()
    else:
        args, batches = TC.resnet2d_args(), TC.resnet2d_batches()
    torch.manual_seed(7)
    net = tm.build_model(args)
    for k, v in net.state_dict().items():                   # same seed -> same initial weights
        if f"{tag}_ini.{k}" in g.files:
            assert np.allclose(TC.tensor_digest(v.numpy())[:2], g[f"{tag}_ini.{k}"], rtol=1e-6), k
    net = net.to(device).train()
    opt, sched = tm.make_optimizer(args, net)
    assert isinstance(opt, tm.ClipAdam)
    labels_all = np.concatenate([b[1].numpy() for b in batches])
    crit = tm.SELCLoss(labels_all, 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    losses, lrs = [], []
    for b in batches:
        lrs.append(opt.param_groups[0]["lr"])
        losses.append(float(tm.train_step(args, net, (b[0].to(device),) + tuple(b[1:]), device, opt,
                                          sched, crit, 1, sc)))
    assert np.allclose(lrs, g[f"{tag}_lrs"], rtol=1e-12)
    rel = np.abs(np.asarray(losses) - g[f"{tag}_losses"]) / np.abs(g[f"{tag}_losses"])
    assert rel.max() <= 1e-4, (losses, g[f"{tag}_losses"])
    state = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}
    worst_b = worst_p = 0.0
    for k in g.files:
        if k.startswith(f"{tag}_buf."):
            name = k[len(tag) + 5:]
            want, got = g[k], state[name]
            if "num_batches" in name:
                assert int(got) == int(want), name
                continue
            d = float(np.abs(got - want).max() / max(1.0, float(np.abs(want).max())))
            worst_b = max(worst_b, d)
            assert d <= 1e-4, (name, d)
        elif k.startswith(f"{tag}_par."):
            name = k[len(tag) + 5:]
            if name.endswith(".0.bias"):
                # A convolution bias in front of a BatchNorm has an exactly zero gradient (the
                # normalisation cancels it); this build folds it into the BatchNorm and gives it that
                # zero, the reference's autograd returns rounding noise of the size of
                # the weight-decay term 1e-4 * b, which Adam's normalisation turns into steps of
                # +-lr.  The bias has no effect on the network function; it may differ by the
                # whole distance three steps can cover, 2 * sum(lr) = 4.4e-5.
                _digest_check(tag, name, state[name], g[k], 4.6e-5)
                continue
            worst_p = max(worst_p, _digest_check(tag, name, state[name], g[k], 2e-6, 0.02, 5e-5))
    print(f"[{tag}] loss rel err {rel.max():.2e}, buffers {worst_b:.2e}, params {worst_p:.2e}")


def test_cfg4_chain_logmel_splice_train_step(device):
    """BASELINE config 4 as ONE path under test: waveform (B,1,5000) -> ``frontend.logmel`` ->
    ``augmentations2d.augment`` (inside ``train_step``, dataset 'PhysioNet(spec128)',
    train_model.py:504-505) -> ResNet9-2D forward/backward/ClipAdam on the HIP path, against the
    CPU chain oracle log-mel -> oracle 2D splice -> the same module in float64 with torch's Adam.
    (Log-mel is parity-unpinned — librosa is not importable offline, DESIGN §4 — so the yardstick
    for the front end is the oracle restatement; everything behind it is pinned elsewhere.)"""
    import copy
    B, T = 8, 5000
    x, frames, labels, wav = synthetic.make_batch(B, 1, T, sample_rate=2000, seed=31)
    args = TC.resnet2d_args()
    args.batch_size, args.num_steps = B, 40
    torch.manual_seed(11)
    ref = models2d.ResNet9(2).train()
    net = copy.deepcopy(ref).to(device).train()
    opt, sched = tm.make_optimizer(args, net)
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    wave = torch.from_numpy(x).to(device)
    spec, fspec = frontend.logmel(wave, frames)
    assert spec.shape == (B, 1, 128, 128)
    batch = (spec, torch.from_numpy(labels), torch.from_numpy(fspec), wav, None, torch.arange(B))
    loss = float(tm.train_step(args, net, batch, device, opt, sched, crit, 1, sc))
    # the CPU chain in float64
    spec_o, fspec_o = O.logmel(x[:, 0, :], frames)
    assert np.array_equal(fspec_o, fspec)
    assert np.abs(spec.cpu().numpy()[:, 0] - spec_o).max() <= 1e-4
    y_o = O.augment("durratiomixup", spec_o[:, None].astype(np.float32), labels, fspec_o, wav, 0)["y"]
    ref = ref.double()
    opt_r = torch.optim.Adam(ref.parameters(), lr=args.lr_max, weight_decay=args.weight_decay)
    sched_r = torch.optim.lr_scheduler.OneCycleLR(opt_r, max_lr=args.lr_max, total_steps=args.num_steps)
    out = ref(torch.from_numpy(y_o).double())
    t = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).double()
    loss_r = -(torch.log_softmax(out, 1) * t).sum(1).mean()
    loss_r.backward()
    torch.nn.utils.clip_grad_value_(ref.parameters(), args.grad_clip)
    opt_r.step()
    sched_r.step()
    assert abs(loss - float(loss_r)) <= 1e-4 * max(1.0, abs(float(loss_r))), (loss, float(loss_r))
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        d = (p.detach().cpu().double() - q.detach()).abs()
        # one Adam step at lr0 = 4e-4 moves every element by +-lr0: an element whose gradient is
        # rounding noise may go the other way (2 lr0); allow that for 0.5 % of a tensor
        assert float((d > 1e-4).double().mean()) <= 0.005 and float(d.max()) <= 1e-3, (k, float(d.max()))
    for (k, v), (_, w) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.allclose(v.cpu().double(), w.double(), rtol=1e-4, atol=1e-5), k


def test_graphed_step_refuses_a_stale_autograd_graph(device):
    """An autograd graph of an earlier eager pass that is still referenced keeps AccumulateGrad
    nodes bound to the legacy default stream; backward under capture would pull that stream into
    the capture and hipStreamEndCapture crashes the process.  GraphedTrainStep sees
    the stream mismatch in its warm-up — outside any capture — and raises instead; once the
    stale graph is dropped it builds and steps."""
    B, C, T = 16, 4, 2500
    args = TC.traj_args()
    args.batch_size = B
    net = _potes(args, device)
    opt, sched = tm.make_optimizer(args, net)
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=1000, seed=3)
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1, device=device)
    data = torch.from_numpy(x).to(device)
    stale = net(data, depth=0, pass_part="second").square().sum()       # eager, default stream
    stale.backward()
    opt.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="autograd graph from an earlier"):
        tm.GraphedTrainStep(args, net, opt, sched, crit, device, B, C, T)
    del stale
    g = tm.GraphedTrainStep(args, net, opt, sched, crit, device, B, C, T)
    batch = (data, torch.from_numpy(labels), torch.from_numpy(frames), wav, None, torch.arange(B))
    assert np.isfinite(float(g.step(batch, 1, tm.step_counter_class())))


def test_train_model_driver_on_spectrograms(device, tmp_path):
    """BASELINE config 4 through the run driver: a dataset dictionary of log-mel images (computed
    here by ``frontend.logmel`` from separable synthetic cycles, laid out as the reference's
    spectrogram container: dataloader_physionet2d.py) -> ``train_model(args.dataset =
    'PhysioNet(spec128)')`` -> 2D durratiomixup + ResNet9-2D -> evaluation -> model.pth."""
    rs = np.random.RandomState(0)
    ds = {}
    for split, n_rec in (("train", 16), ("test", 8)):
        d = {"data": [], "label": [], "frames": [], "wav": [], "sig_qual": []}
        for r in range(n_rec):
            wav, label = f"{'abcdef'[r % 6]}{r:04d}", (r // 2) % 2
            fr = synthetic.make_frames(3, 2.0, rs)
            x = rs.standard_normal((3, 5000)).astype(np.float32)
            if label:       # a spectral tilt (the per-cycle dB reference cancels a plain gain)
                x = np.cumsum(x, axis=1).astype(np.float32)
                x -= x.mean(axis=1, keepdims=True)
            for j in range(3):
                x[j, fr[j, 4]:] = 0
            spec, fspec = frontend.logmel(torch.from_numpy(x).to(device), fr)
            for j in range(3):
                d["data"].append(spec[j, 0].cpu().numpy())
                d["label"].append(label); d["frames"].append(fspec[j]); d["wav"].append(wav)
                d["sig_qual"].append(1)
        ds[split] = d
    args = argparse.Namespace(dataset="PhysioNet(spec128)", model="resnet9", method="durratiomixup+0.7",
                              num_epochs=3, batch_size=16, op="adam", use_sched=True, lr_max=0.002,
                              weight_decay=1e-4, grad_clip=0.1, seed=4, seed_data=1100001, n_fraction=1.0,
                              train_balance=True, num_classes=2, sample_rate=2000, num_channels=1,
                              valid=False, depth=0, EXPERIMENTS=str(tmp_path))
    perf = tm.train_model(args, ds, device, log=None)
    assert perf["steps"][-1] == args.num_steps == 3 * (48 // 16)
    assert all(np.isfinite(v) for v in perf["train_loss"])
    assert perf["train_loss"][-1] < perf["train_loss"][0]
    import glob
    assert len(glob.glob(str(tmp_path / "*" / "model.pth"))) == 1


@pytest.mark.parametrize("busy", [False, True])
@pytest.mark.parametrize("method", ["durmixmagwarp(0.2,4)+0.8", "(saloptenv)durmixmagwarp(0.2,4)"])
def test_pipelined_step_is_bit_identical_to_the_sequential_one(method, busy, device, tmp_path):
    """PipelinedTrainStep (augmentation of batch k+1 on a side stream while the captured graph of
    batch k replays; two slots) == GraphedTrainStep called batch after batch: same kernels, same
    values, only the stream of the augmentation launches differs — every loss and every parameter
    bit for bit over 7 steps (dropout ON: the keys advance in the same host order).

    ``busy``: ~30 ms of GPU work is queued on the main stream in front of the first pipelined
    step, so the host runs far ahead of the GPU and the first (inline, main-stream) prepare is
    still pending when the side stream's prepare of batch 1 is issued — both use the shared
    saliency graph and step context, the hand-over event has to order them."""
    B, T = 32, 2500
    batches = []
    for i in range(7):
        x, frames, labels, wav = synthetic.make_batch(B, 4, T, sample_rate=1000, seed=900 + i)
        batches.append((torch.from_numpy(x).to(device), torch.from_numpy(labels), torch.from_numpy(frames),
                        wav, torch.ones(B, dtype=torch.long), torch.arange(B)))
    res = {}
    for mode in ("sequential", "pipelined"):
        args = TC.salopt_traj_args(str(tmp_path / mode))
        args.method, args.batch_size, args.seed_fix = method, B, 4
        _write_base_checkpoint(args)
        torch.manual_seed(7)
        net = tm.build_model(args).to(device).train()
        opt, sched = tm.make_optimizer(args, net)
        crit = tm.SELCLoss(np.zeros(B * 7, int), 2, es=args.num_epochs + 1, device=device)
        sc = tm.step_counter_class()
        torch.cuda.manual_seed(4)
        cls = tm.PipelinedTrainStep if mode == "pipelined" else tm.GraphedTrainStep
        g = cls(args, net, opt, sched, crit, device, B, 4, T)
        losses = []
        if mode == "pipelined":
            if busy:
                w = torch.full((4096, 4096), 1e-3, device=device)     # (no draw from the device RNG:
                for _ in range(60):                                   #  the dropout keys come from it)
                    w = (w @ w).clamp_(-1, 1)
            for b, nxt in tm.PipelinedTrainStep.pairs(batches):
                losses.append(g.step(b, 1, sc, None, next_batch=nxt).clone())
        else:
            for b in batches:
                losses.append(g.step(b, 1, sc).clone())
        torch.cuda.synchronize()
        res[mode] = (torch.stack(losses).cpu(), [p.detach().cpu().clone() for p in net.parameters()])
        assert sc.count == 7
    assert torch.equal(res["sequential"][0], res["pipelined"][0]), (res["sequential"][0], res["pipelined"][0])
    for a, b in zip(res["sequential"][1], res["pipelined"][1]):
        assert torch.equal(a, b)


def test_train_model_driver_salopt_with_its_own_base_checkpoint(device, tmp_path):
    """The reference's workflow for saliency-guided PCGmix end to end: a 'base' run writes
    model.pth (train_model.py:481-482), the '(saloptenv)…' run of the same configuration loads it as
    its frozen saliency model (saliency.py:26-51) — through ``train_model()``, which pipelines the
    saliency-guided augmentation of batch k+1 with the captured graph of batch k.  Pipelined and
    one-slot runs end with bit-identical parameters."""
    from conftest import learnable_dataset
    ds = learnable_dataset(n_rec=24)
    def make(method):
        return argparse.Namespace(dataset="PhysioNet", model="Potes", method=method, num_epochs=2,
                                  batch_size=16, op="adam", use_sched=True, lr_max=0.003,
                                  weight_decay=1e-4, grad_clip=0.1, seed=4, seed_data=1100001,
                                  n_fraction=1.0, train_balance=True, num_classes=2, sample_rate=1000,
                                  num_channels=4, valid=False, depth=0, EXPERIMENTS=str(tmp_path))
    tm.train_model(make("base"), ds, device, log=None)
    finals = []
    for pipeline in (True, False):
        saliency._LOADED.clear()
        perf = tm.train_model(make("(saloptenv)durmixmagwarp(0.2,4)+0.8"), ds, device, log=None,
                              pipeline=pipeline)
        assert perf["steps"][-1] == 2 * (96 // 16) and all(np.isfinite(v) for v in perf["train_loss"])
        finals.append([p.detach().cpu().clone() for p in perf["model"].parameters()])
    for a, b in zip(*finals):
        assert torch.equal(a, b)


def test_train_epoch_pipelines_the_saliency_guided_step(device, tmp_path):
    """``train_epoch`` — the reference's signature — on a GPU with a saliency-guided method: it
    builds the two-slot pipelined captured step by itself (one batch of lookahead over the loader)
    and returns what the eager epoch (``args.hipgraph = False``) returns: mean loss to 1e-4,
    accuracy and learning rates exactly, parameters to 1e-3; a second epoch re-uses the graph."""
    B, T = 32, 2500
    batches = []
    for i in range(6):
        x, frames, labels, wav = synthetic.make_batch(B, 4, T, sample_rate=1000, seed=740 + i)
        batches.append((torch.from_numpy(x), torch.from_numpy(labels), torch.from_numpy(frames), wav,
                        torch.ones(B, dtype=torch.long), torch.arange(B)))
    out = {}
    for mode in ("eager", "captured"):
        args = TC.salopt_traj_args(str(tmp_path / mode))
        args.batch_size, args.num_steps, args.num_epochs = B, 12, 2
        args.hipgraph = mode == "captured"
        _write_base_checkpoint(args)
        saliency._LOADED.clear()
        net = _potes(args, device)
        opt, sched = tm.make_optimizer(args, net)
        crit = tm.SELCLoss(np.concatenate([b[1].numpy() for b in batches]), 2, es=args.num_epochs + 1,
                           device=device)
        sc = tm.step_counter_class()
        res = [tm.train_epoch(args, net, batches, device, opt, sched, crit, e, sc) for e in (1, 2)]
        step = net.__dict__.get("_pcgmix_epoch_step")
        if mode == "captured":
            assert isinstance(step.step, tm.PipelinedTrainStep)
        else:
            assert step is None
        assert sc.count == 12
        out[mode] = (res, [p.detach().cpu().numpy() for p in net.parameters() if p.requires_grad])
    for (la, aa, lra), (lb, ab, lrb) in zip(out["eager"][0], out["captured"][0]):
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(la)) and aa == ab and lra == lrb
    for a, b in zip(out["eager"][1], out["captured"][1]):
        assert np.allclose(a, b, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("method", ["durratiomixup", "durmixmagwarp(0.2,4)+0.8"])
def test_direct_launches_equal_the_graph_replay(method, device, tmp_path):
    """The fused Potes step issued as its five recorded library launches (the default) == the same
    step replayed as a hipGraph (``GraphedTrainStep.use_tape = False``): every loss and every
    parameter bit for bit over 8 steps, dropout on.  Another model keeps the graph."""
    B, T = 32, 2500
    batches = []
    for i in range(8):
        x, frames, labels, wav = synthetic.make_batch(B, 4, T, sample_rate=1000, seed=820 + i)
        batches.append((torch.from_numpy(x).to(device), torch.from_numpy(labels), torch.from_numpy(frames),
                        wav, torch.ones(B, dtype=torch.long), torch.arange(B)))
    res = {}
    old = tm.GraphedTrainStep.use_tape
    try:
        for use_tape in (True, False):
            tm.GraphedTrainStep.use_tape = use_tape
            args = TC.salopt_traj_args(str(tmp_path / str(use_tape)))
            args.method, args.batch_size, args.seed_fix, args.num_steps = method, B, 4, 8
            torch.manual_seed(7)
            net = tm.build_model(args).to(device).train()
            opt, sched = tm.make_optimizer(args, net)
            crit = tm.SELCLoss(np.zeros(B * 8, int), 2, es=args.num_epochs + 1, device=device)
            sc = tm.step_counter_class()
            torch.cuda.manual_seed(4)
            g = tm.GraphedTrainStep(args, net, opt, sched, crit, device, B, 4, T)
            assert (g.tape is not None) == use_tape
            if use_tape:
                assert [t[0] for t in g.tape] == list(tm.GraphedTrainStep._TAPE_LAUNCHES)
            losses = [g.step(b, 1, sc).clone() for b in batches]
            torch.cuda.synchronize()
            res[use_tape] = (torch.stack(losses).cpu(), [p.detach().cpu().clone() for p in net.parameters()])
    finally:
        tm.GraphedTrainStep.use_tape = old
    assert torch.equal(res[True][0], res[False][0]), (res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert torch.equal(a, b)
    # ResNet9: torch/MIOpen kernels inside the capture — the graph is replayed
    args = TC.salopt_traj_args(str(tmp_path / "r9"))
    args.method, args.model, args.batch_size = "durratiomixup", "resnet9", 4
    net = tm.build_model(args).to(device).train()
    opt, sched = tm.make_optimizer(args, net)
    crit = tm.SELCLoss(np.zeros(8, int), 2, es=args.num_epochs + 1, device=device)
    assert tm.GraphedTrainStep(args, net, opt, sched, crit, device, 4, 4, T).tape is None
