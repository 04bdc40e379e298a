"""Training-step machinery on CPU: loss, optimiser/clip parity with a hand-rolled step, the
reference's model goldens, evaluation, and the N>1 path over gloo (world_size 2)."""
import argparse
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import pcgmix_amd  # noqa: F401
from pcgmix_amd import models, models2d, synthetic, train_model as tm
from conftest import GOLDEN
from oracle import pcgmix_oracle as O


def make_args(**kw):
    a = argparse.Namespace(dataset="PhysioNet", model="Potes", method="base", num_epochs=2,
                           batch_size=8, op="adam", use_sched=True, lr_max=0.01, weight_decay=1e-4,
                           grad_clip=0.1, seed=4, num_classes=2, num_channels=4, sig_len=2500,
                           depth=0, num_steps=8, sample_rate=1000)
    a.__dict__.update(kw)
    return a


def no_dropout(model):
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return model


def test_models_match_reference_goldens():
    """Same seed -> same initial weights -> same logits as the reference's own modules
    (goldens written by tests/golden/make_golden.py from reference models.py / models2d.py)."""
    g = np.load(os.path.join(GOLDEN, "models_seed7.npz"))
    x = torch.from_numpy(g["x1d"])
    for build, key, inp in (
        (lambda: models.CNN_potes_TS(4, 2, "PhysioNet"), "potes", x),
        (lambda: models.ResNet9(4, 2), "resnet1d", x),
        (lambda: models2d.ResNet9(2), "resnet2d", torch.from_numpy(g["x2d"])),
    ):
        torch.manual_seed(7)
        m = build().eval()
        assert sum(p.numel() for p in m.parameters()) == int(g[key + "_nparams"])
        out = m(inp, depth=0, pass_part="second").detach().numpy()
        assert np.abs(out - g[key + "_logits"]).max() <= 1e-5
    sd = np.load(os.path.join(GOLDEN, "potes_state_seed1234.npz"))
    assert sorted(sd.files) == sorted(models.CNN_potes_TS().state_dict().keys())
    assert models.potes_flat_features(2500) == 9968 and models.potes_flat_features(5000) == 19968
    assert models.resnet9_flat_features(2500) == 39936 and models.resnet9_flat_features(5000) == 79872


def test_celoss_matches_oracle_and_selc_reduces_to_ce():
    rs = np.random.RandomState(0)
    logits = torch.from_numpy(rs.randn(16, 2).astype(np.float32))
    t = torch.from_numpy((np.eye(2)[rs.randint(0, 2, 16)] * 0.7 + 0.15).astype(np.float32))
    assert abs(float(tm.CELoss(2)(logits, t)) - O.ce_soft(logits.numpy(), t.numpy())) < 1e-6
    selc = tm.SELCLoss(np.zeros(16, int), 2, es=5)
    assert float(selc(logits, t, torch.arange(16), 3, "train")) == float(tm.CELoss(2)(logits, t))
    assert float(selc(logits, t, torch.arange(16), 9, "train")) != float(tm.CELoss(2)(logits, t))
    assert tm.selc_turning_point(make_args(method="durratiomixup", num_epochs=50)) == 51


def test_train_step_equals_manual_adam_clip_step():
    """One train_step == forward, soft CE, backward, clip_grad_value_(0.1), Adam(lr from
    OneCycleLR), as train_model.py:537-569 orders them."""
    args = make_args()
    x, frames, labels, wav = synthetic.make_batch(8, 4, 2500, seed=1)
    batch = (torch.from_numpy(x), torch.from_numpy(labels), torch.from_numpy(frames), wav,
             torch.ones(8, dtype=torch.long), torch.arange(8))
    torch.manual_seed(0)
    m1 = no_dropout(tm.build_model(args)).train()
    torch.manual_seed(0)
    m2 = no_dropout(tm.build_model(args)).train()
    opt, sched = tm.make_optimizer(args, m1)
    sc = tm.step_counter_class()
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1)
    loss = tm.train_step(args, m1, batch, torch.device("cpu"), opt, sched, crit, 0, sc)
    assert sc.count == 1 and torch.isfinite(loss)
    # manual
    opt2, sched2 = tm.make_optimizer(args, m2)
    out = m2(batch[0])
    l2 = -(F.log_softmax(out, 1) * F.one_hot(batch[1], 2)).sum(1).mean()
    l2.backward()
    for p in m2.parameters():
        if p.grad is not None:
            p.grad.clamp_(-0.1, 0.1)
    opt2.step()
    assert abs(float(l2) - float(loss)) < 1e-7
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        assert torch.allclose(p1, p2, atol=1e-7), n1
    # the dead branches never move and never receive gradients (reference models.py:444-455)
    assert all(p.grad is None for p in m1.cnn2.parameters())


def test_train_epoch_runs_and_loss_decreases():
    args = make_args(num_epochs=6, num_steps=6 * 4, lr_max=0.003)
    pool = synthetic.make_batch(32, 4, 2500, seed=2)
    # make the task learnable: class 1 is louder in band 0
    pool[0][pool[2] == 1, 0] *= 3.0
    loader = tm.SyntheticCycleLoader(pool, 8)
    torch.manual_seed(0)
    model = tm.build_model(args)
    opt, sched = tm.make_optimizer(args, model)
    crit = tm.SELCLoss(pool[2], 2, es=args.num_epochs + 1)
    sc = tm.step_counter_class()
    losses = [tm.train_epoch(args, model, loader, torch.device("cpu"), opt, sched, crit, e, sc)[0]
              for e in range(6)]
    assert sc.count == 24 and np.isfinite(losses).all() and losses[-1] < losses[0]
    ev = tm.test_data_accuracy(args, model, tm.SyntheticCycleLoader(pool, 8), torch.device("cpu"), crit)
    assert ev["recordings"] == len(set(pool[3])) and 0 <= ev["accuracy"] <= 100
    assert ev["rocauc"] is None or 0.0 <= ev["rocauc"] <= 1.0


def test_evaluation_majority_vote():
    """Per-recording mean of cycle softmaxes, then argmax (train_model.py:620-632)."""
    class Fixed(torch.nn.Module):
        def forward(self, x):
            return x[:, :2, 0]
    logits = torch.tensor([[2.0, 0.0], [0.0, 0.5], [0.0, 0.5],      # rec a: mean prob favours 0
                           [0.0, 3.0], [1.0, 0.0]])                 # rec b: favours 1
    data = logits[:, :, None].repeat(1, 1, 4)
    loader = [(data, torch.tensor([0, 0, 0, 1, 1]), None, ("a", "a", "a", "b", "b"), None, None)]
    ev = tm.test_data_accuracy(make_args(), Fixed(), loader, torch.device("cpu"))
    assert ev["recordings"] == 2 and ev["accuracy"] == 100.0 and ev["rocauc"] == 1.0


def test_shard_batch_drops_remainder():
    b = (torch.arange(10), tuple("abcdefghij"))
    s0, s1 = tm.shard_batch(b, 0, 3), tm.shard_batch(b, 2, 3)
    assert s0[0].tolist() == [0, 1, 2] and s1[0].tolist() == [6, 7, 8] and s1[1] == ("g", "h", "i")


# ------------------------------------------------------------------ N > 1 over gloo
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, out_path, flat=False):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    args = make_args(batch_size=8)
    x, frames, labels, wav = synthetic.make_batch(16, 4, 2500, seed=3)
    torch.manual_seed(0)
    if flat and rank == 1:
        torch.manual_seed(99)            # FlatGradSync must broadcast rank 0's initial weights
    model = no_dropout(tm.build_model(args)).train()
    sync = None
    if flat:
        sync = tm.FlatGradSync(model, torch.device("cpu"))
    else:
        model = tm.wrap_distributed(model, torch.device("cpu"))
    opt, sched = tm.make_optimizer(args, model)
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1)
    sc = tm.step_counter_class()
    full = (torch.from_numpy(x), torch.from_numpy(labels), torch.from_numpy(frames), wav,
            torch.ones(16, dtype=torch.long), torch.arange(16))
    for _ in range(3):
        tm.train_step(args, model, tm.shard_batch(full, rank, world), torch.device("cpu"), opt,
                      sched, crit, 0, sc, sync=sync)
    if rank == 0:
        inner = model.module if hasattr(model, "module") else model
        torch.save({k: v.clone() for k, v in inner.state_dict().items()}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("flat", [False, True], ids=["ddp", "flat_all_reduce"])
def test_ddp_two_ranks_equals_single_process(tmp_path, flat):
    """2 ranks x batch 8 with averaged gradients == 1 process x batch 16 (same seeds, dropout
    off), within fp32 reduction-order noise — the property SURVEY.md §4 asks for in place of the
    reference's untested DataParallel.  Both averaging paths: DDP hooks (eager step) and
    FlatGradSync (the one the hipGraph step uses)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "ddp.pt")
    mp.spawn(_ddp_worker, args=(2, _free_port(), out, flat), nprocs=2, join=True)
    ddp_state = torch.load(out, weights_only=True)
    args = make_args(batch_size=16)
    x, frames, labels, wav = synthetic.make_batch(16, 4, 2500, seed=3)
    torch.manual_seed(0)
    model = no_dropout(tm.build_model(args)).train()
    opt, sched = tm.make_optimizer(args, model)
    crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1)
    sc = tm.step_counter_class()
    full = (torch.from_numpy(x), torch.from_numpy(labels), torch.from_numpy(frames), wav,
            torch.ones(16, dtype=torch.long), torch.arange(16))
    for _ in range(3):
        tm.train_step(args, model, full, torch.device("cpu"), opt, sched, crit, 0, sc)
    for k, v in model.state_dict().items():
        assert torch.allclose(v, ddp_state[k], atol=2e-6), k


def _seed_worker(rank, world, port, out_path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = tm.step_counter_class()
    seen = []
    for _ in range(3):
        a, b = make_args(), make_args()
        b.rank_seed = True
        seen.append((tm.augmentation_counter(a, sc).count, tm.augmentation_counter(b, sc).count))
        sc.add()
    assert tm.augmentation_counter(make_args(), sc) is sc          # default: the counter itself
    torch.save(seen, f"{out_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_rank_seed_flag_two_ranks(tmp_path):
    """Default (the reference's behaviour): every rank hands augment() the same step count.  With
    ``args.rank_seed`` (non-reference) rank r of w gets count*w + r: distinct on
    every rank and never colliding across steps.  One process: the flag changes nothing."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "seeds")
    mp.spawn(_seed_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert [s[0] for s in r0] == [s[0] for s in r1] == [0, 1, 2]
    assert [s[1] for s in r0] == [0, 2, 4] and [s[1] for s in r1] == [1, 3, 5]
    sc = tm.step_counter_class()
    a = make_args()
    a.rank_seed = True
    assert tm.augmentation_counter(a, sc) is sc                    # no process group


def _driver_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from conftest import learnable_dataset
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    ds = learnable_dataset(n_rec=12, T=2500)
    args = argparse.Namespace(dataset="PhysioNet", model="Potes", method="base", num_epochs=2,
                              batch_size=16, op="adam", use_sched=True, lr_max=0.003,
                              weight_decay=1e-4, grad_clip=0.1, seed=4, seed_data=1100001,
                              n_fraction=1.0, train_balance=True, num_classes=2, sample_rate=1000,
                              num_channels=4, valid=False, depth=0, EXPERIMENTS=out_dir)
    perf = tm.train_model(args, ds, torch.device("cpu"), use_graph=False, log=None)
    inner = perf["model"].module if hasattr(perf["model"], "module") else perf["model"]
    torch.save({"steps": perf["steps"], "params": [p.detach().clone() for p in inner.parameters()]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_train_model_driver_two_ranks(tmp_path):
    """The run driver under torch.distributed (gloo, world 2): every rank trains on its shard of
    each batch, gradients are averaged, so the replicas stay identical; rank 0 alone writes
    model.pth; the step count is the single-process one."""
    import glob
    import torch.multiprocessing as mp
    mp.spawn(_driver_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "rank0.pt"), weights_only=True)
    r1 = torch.load(str(tmp_path / "rank1.pt"), weights_only=True)
    assert r0["steps"] == r1["steps"] and r0["steps"][-1] == 2 * (48 // 16)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    assert len(glob.glob(str(tmp_path / "*" / "model.pth"))) == 1


# ---- reference-recorded goldens (tests/golden/train_ref.npz <- the reference's train_model.py) ----
def test_celoss_matches_reference_golden():
    import train_replay
    train_replay.check_celoss(torch.device("cpu"))


def test_train_epoch_reproduces_reference_trajectory_cpu():
    """train_epoch here == the reference's train_epoch (train_model.py:490-589) run on the same
    10 batches: per-step loss, lr, mean loss, accuracy, every parameter after step 10."""
    import train_replay
    res = train_replay.trajectory(torch.device("cpu"), "epoch")
    train_replay.check_trajectory(res, loss_tol=2e-6, param_tol=2e-5)


def test_evaluation_matches_reference_golden_cpu():
    """test_data_accuracy (mean-probability vote and '(class_majority)') == the reference's
    train_model.py:591-670 on a synthetic 12-recording loader."""
    import train_replay
    train_replay.check_evaluation(torch.device("cpu"))
