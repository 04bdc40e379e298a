"""Replays of the reference-recorded train-step / evaluation goldens (tests/golden/train_ref.npz,
written by tests/golden/make_golden_train.py from the reference's train_model.py) through this
package, on any device.  Used by test_train_cpu.py (host logic) and test_train_gpu.py (HIP path)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)
import train_cases  # noqa: E402

import pcgmix_amd  # noqa: E402,F401
from pcgmix_amd import models, train_model as tm  # noqa: E402


def golden():
    return np.load(os.path.join(GOLDEN, "train_ref.npz"))


def trajectory(device, mode):
    """10 steps of the reference's train_epoch setup (Potes seed 7, dropout 0, Adam + OneCycleLR,
    clip 0.1, durmixmagwarp(0.2,4)).  mode: 'epoch' = this package's train_epoch, 'step' = eager
    train_step calls, 'graph' = GraphedTrainStep.  Returns (losses, lrs, state_dict, extras)."""
    args = train_cases.traj_args()
    torch.manual_seed(7)
    net = tm.build_model(args)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net = net.to(device).train()
    opt, sched = tm.make_optimizer(args, net)
    batches = train_cases.traj_batches()
    if device.type == "cpu":
        # the product has no CPU augmentation path: the CPU replay (host logic of the step: loss,
        # clip, Adam, OneCycleLR, counters) takes the batches pre-augmented by the oracle
        from oracle import pcgmix_oracle as O
        for i, b in enumerate(batches):
            y = O.augment(args.method, b[0].numpy(), b[1].numpy(), b[2].numpy(), b[3], i)["y"]
            batches[i] = (torch.from_numpy(y),) + b[1:]
        args.method = "base"
    labels = np.concatenate([b[1].numpy() for b in batches])
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    extras = {}
    if mode == "epoch_graph":
        # train_epoch as a caller of the reference's function gets it on a GPU: the captured step,
        # built at the first batch.  It returns the epoch mean only: the per-step trajectory is
        # filled with the recorded one shifted by the mean's error (so check_trajectory's loss
        # bound applies to the mean), learning rates and final parameters are compared as always.
        mean_loss, acc, lrs = tm.train_epoch(args, net, batches, device, opt, sched, crit, 1, sc)
        assert isinstance(net.__dict__["_pcgmix_epoch_step"].step, tm.GraphedTrainStep)
        import copy, pickle
        assert copy.deepcopy(net).__dict__["_pcgmix_epoch_step"] is None      # copies drop the graph
        assert pickle.loads(pickle.dumps(net)).__dict__["_pcgmix_epoch_step"] is None
        extras = {"mean_loss": mean_loss, "acc": acc}
        g = golden()
        losses = g["traj_losses"] + (mean_loss - float(g["traj_mean_loss"]))
    elif mode == "epoch":
        args.hipgraph = False                   # the eager epoch (train_step per batch)
        losses = []
        orig = tm.train_step

        def recording_step(*a, **k):            # train_epoch returns the mean only
            v = orig(*a, **k)
            losses.append(v)
            return v
        tm.train_step = recording_step
        try:
            mean_loss, acc, lrs = tm.train_epoch(args, net, batches, device, opt, sched, crit, 1, sc)
        finally:
            tm.train_step = orig
        extras = {"mean_loss": mean_loss, "acc": acc}
        losses = [float(v) for v in losses]
    else:
        step = None
        if mode == "graph":
            g = tm.GraphedTrainStep(args, net, opt, sched, crit, device, train_cases.TRAJ_B, 4, 2500)
            step = lambda b: g.step(b, 1, sc)                                   # noqa: E731
        else:
            step = lambda b: tm.train_step(args, net, b, device, opt, sched, crit, 1, sc)  # noqa: E731
        losses, lrs = [], []
        for b in batches:
            lrs.append(opt.param_groups[0]["lr"])
            losses.append(float(step(b)))       # the captured step returns its static loss tensor
    assert sc.count == train_cases.TRAJ_STEPS
    return np.asarray(losses), np.asarray(lrs), {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}, extras


def check_trajectory(res, loss_tol, param_tol):
    g = golden()
    losses, lrs, state, extras = res
    assert np.allclose(lrs, g["traj_lrs"], rtol=1e-12, atol=1e-15)
    err = np.abs(losses - g["traj_losses"]).max()
    assert err <= loss_tol, f"loss trajectory off by {err}: {losses} vs {g['traj_losses']}"
    worst = 0.0
    for k in g.files:
        if k.startswith("traj_final."):
            d = float(np.abs(state[k[len("traj_final."):]] - g[k]).max())
            worst = max(worst, d)
            assert d <= param_tol, (k, d)
    if extras:
        assert abs(extras["mean_loss"] - float(g["traj_mean_loss"])) <= loss_tol
        assert abs(100.0 * extras["acc"] - float(g["traj_acc"])) <= 1e-9
    return float(err), worst


def eval_model(device):
    g = golden()
    sd = np.load(os.path.join(GOLDEN, "potes_state_seed1234.npz"))
    m = models.CNN_potes_TS(4, 2, "PhysioNet")
    m.load_state_dict({k: torch.from_numpy(sd[k]) for k in sd.files})
    with torch.no_grad():
        m.linear.bias[0] += float(g["eval_bias0_shift"])
    return m.to(device).eval()


def check_evaluation(device):
    g = golden()
    m = eval_model(device)
    loader = train_cases.eval_loader()
    with torch.no_grad():
        lg = torch.cat([m(b[0].to(device)) for b in loader]).cpu().numpy()
    assert np.abs(lg - g["eval_logits"]).max() <= 1e-4
    crit = tm.SELCLoss(np.zeros(48, int), 2, es=99, device=device)
    for tag, method in (("mean", "base"), ("cm", "base(class_majority)")):
        class A:
            num_classes = 2
        A.method = method
        ev = tm.test_data_accuracy(A, m, loader, device, crit)
        assert ev["recordings"] == 12
        for k in ("accuracy", "specificity", "sensitivity", "f1", "precision", "recall"):
            assert abs(ev[k] - float(g[f"eval_{tag}_{k}"])) <= 1e-9, (tag, k, ev[k], float(g[f"eval_{tag}_{k}"]))
        assert abs(ev["loss"] - float(g[f"eval_{tag}_loss"])) <= 1e-5
        if int(g[f"eval_{tag}_rocauc_raises"]):
            assert ev["rocauc"] is None         # the reference raises IndexError there (:667)
        else:
            assert abs(ev["rocauc"] - float(g[f"eval_{tag}_rocauc"])) <= 1e-12
        # the reference's own call (train_model.py:455): seven positional arguments, results pushed
        # into the performance object, nothing returned; the golden holds what the reference's
        # performance.dict held after that call (make_golden_train.py)
        perf = tm.performance_metrics_class()
        ld = train_cases.ListLoader(loader)
        ld.dataset = range(48)
        if int(g[f"eval_{tag}_rocauc_raises"]):
            with pytest.raises(IndexError):
                tm.test_data_accuracy(A, m, ld, device, crit, 1, perf)
        else:
            assert tm.test_data_accuracy(A, m, ld, device, crit, 1, perf) is None
        assert set(perf.dict) == {"steps", "epochs", "times", "train_loss", "train_accuracy",
                                  "test_loss", "test_accuracy", "test_specificity",
                                  "test_sensitivity", "test_precision", "test_recall", "test_f1",
                                  "test_rocauc"}                        # :180-193
        for k in ("accuracy", "loss", "specificity", "sensitivity", "f1", "precision", "recall",
                  "rocauc"):
            got, want = perf.dict["test_" + k], float(g[f"eval_{tag}_{k}"])
            if np.isnan(want):
                assert got == []                # the reference never appended it (:667 raised)
            else:
                assert len(got) == 1 and abs(got[0] - want) <= (1e-5 if k == "loss" else 1e-9), (tag, k)
    assert float(g["eval_mean_accuracy"]) != float(g["eval_cm_accuracy"])      # the rules differ here


def check_celoss(device):
    g = golden()
    for tag in ("hard", "soft"):
        lg = torch.from_numpy(g["ce_logits"]).to(device).requires_grad_()
        loss = tm.CELoss(2)(lg, torch.from_numpy(g[f"ce_{tag}_target"]).to(device))
        loss.backward()
        assert abs(float(loss.detach()) - float(g[f"ce_{tag}_loss"])) <= 1e-6
        assert np.abs(lg.grad.cpu().numpy() - g[f"ce_{tag}_grad"]).max() <= 1e-7
