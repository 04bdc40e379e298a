"""Train step on the GPU: the HIP augmentation inside the loop, host-label fast path, DDP
wrapper degenerate case."""
import argparse
import copy

import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import augmentations, synthetic, train_model as tm
from conftest import Args, StepCounter, learnable_dataset as _learnable_dataset

pytestmark = pytest.mark.gpu


def make_args(**kw):
    a = argparse.Namespace(dataset="PhysioNet", model="Potes", method="durmixmagwarp(0.2,4)",
                           num_epochs=2, batch_size=32, op="adam", use_sched=True, lr_max=0.01,
                           weight_decay=1e-4, grad_clip=0.1, seed=4, num_classes=2, num_channels=4,
                           sig_len=2500, depth=0, num_steps=16, sample_rate=1000)
    a.__dict__.update(kw)
    return a


def test_host_labels_path_is_identical(device):
    x, frames, labels, wav = synthetic.make_batch(32, 4, 2500, seed=4)
    data = torch.from_numpy(x).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
    a = Args("durmixmagwarp(0.2,4)")
    y1, _, m1, _ = augmentations.augment(a, data, tgt, frames, wav, StepCounter(6), None, device, "")
    y2, _, m2, _ = augmentations.augment(a, data, tgt, frames, wav, StepCounter(6), None, device, "",
                                         host_labels=labels)
    assert np.array_equal(m1, m2) and torch.equal(y1, y2)


@pytest.mark.parametrize("model,method,shape", [("Potes", "durmixmagwarp(0.2,4)", (32, 4, 2500)),
                                                ("resnet9", "durratiomixup+0.5", (16, 4, 2500)),
                                                ("Potes", "durratiomixup", (32, 4, 5000))])
def test_train_epoch_on_gpu(model, method, shape, device):
    B, C, T = shape
    args = make_args(model=model, method=method, batch_size=B, sig_len=T, num_steps=8,
                     sample_rate=2000 if T == 5000 else 1000)
    pool = synthetic.make_batch(4 * B, C, T, sample_rate=args.sample_rate, seed=6)
    loader = tm.SyntheticCycleLoader(pool, B)
    torch.manual_seed(0)
    net = tm.build_model(args).to(device)
    net = tm.wrap_distributed(net, device)           # no process group: returned unchanged
    opt, sched = tm.make_optimizer(args, net)
    crit = tm.SELCLoss(pool[2], 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    before = [p.detach().clone() for p in net.parameters() if p.requires_grad]
    for epoch in range(2):
        loss, acc, lrs = tm.train_epoch(args, net, loader, device, opt, sched, crit, epoch, sc)
        assert np.isfinite(loss) and 0.0 <= acc <= 1.0 and len(lrs) == 4
    assert sc.count == 8
    after = [p.detach() for p in net.parameters() if p.requires_grad]
    assert any(not torch.equal(a, b) for a, b in zip(before, after))
    ev = tm.test_data_accuracy(args, net, tm.SyntheticCycleLoader(pool, B), device, crit)
    assert ev["recordings"] == len(set(pool[3]))


def test_graphed_step_matches_eager(device):
    """hipGraph replay of fwd+loss+bwd+clip with eager augment/Adam == the eager train_step
    (dropout off so both see the same network function)."""
    results = []
    B, C, T = 32, 4, 2500
    pool = synthetic.make_batch(B, C, T, seed=9)
    batch = (torch.from_numpy(pool[0]), torch.from_numpy(pool[2]), torch.from_numpy(pool[1]), pool[3],
             torch.ones(B, dtype=torch.long), torch.arange(B))
    for graphed in (False, True):
        args = make_args(method="durmixmagwarp(0.2,4)+0.7", batch_size=B, num_steps=12)
        torch.manual_seed(0)
        net = tm.build_model(args).to(device)
        for m in net.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        net.train()
        opt, sched = tm.make_optimizer(args, net)
        crit = tm.SELCLoss(pool[2], 2, es=args.num_epochs + 1, device=device)
        sc = tm.step_counter_class()
        if graphed:
            g = tm.GraphedTrainStep(args, net, opt, sched, crit, device, B, C, T)
            losses = [float(g.step(batch, 0, sc)) for _ in range(6)]
        else:
            losses = [float(tm.train_step(args, net, batch, device, opt, sched, crit, 0, sc))
                      for _ in range(6)]
        results.append((losses, [p.detach().clone() for p in net.parameters() if p.requires_grad]))
        assert sc.count == 6
    assert np.allclose(results[0][0], results[1][0], rtol=1e-4, atol=1e-5), results
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("use_graph", [True, False])
def test_train_model_driver_learns_and_saves(use_graph, device, tmp_path):
    """End to end: resident loader -> HIP augmentation -> fused Potes stack -> (graphed) step ->
    evaluation -> checkpoint in the reference's key layout."""
    ds = _learnable_dataset()
    args = argparse.Namespace(dataset="PhysioNet", model="Potes", method="durmixmagwarp(0.2,4)+0.6",
                              num_epochs=6, batch_size=32, op="adam", use_sched=True, lr_max=0.003,
                              weight_decay=1e-4, grad_clip=0.1, seed=4, seed_data=1100001,
                              n_fraction=1.0, train_balance=True, num_classes=2, sample_rate=1000,
                              num_channels=4, valid=False, depth=0, EXPERIMENTS=str(tmp_path))
    perf = tm.train_model(args, ds, device, use_graph=use_graph, log=None)
    assert perf["steps"][-1] == args.num_steps == 6 * (192 // 32)
    assert perf["train_loss"][-1] < perf["train_loss"][0]
    assert perf["test_accuracy"][-1] >= 90.0                  # separable by construction
    import glob
    ck = glob.glob(str(tmp_path / "*" / "model.pth"))
    assert len(ck) == 1
    sd = torch.load(ck[0], weights_only=True)
    assert all(k.startswith("module.") for k in sd) and "module.cnn1.0.0.weight" in sd


def test_clip_adam_matches_torch(device):
    """pcgmix_adam_clip_f32 == clip_grad_value_ + torch.optim.Adam over several steps, with a
    OneCycleLR schedule cycling lr and beta1 on both."""
    torch.manual_seed(0)
    shapes = [(20, 19968), (8, 1, 5), (20,), (2, 20)]
    pa = [torch.nn.Parameter(torch.randn(s, device=device)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = tm.ClipAdam(pa, lr=0.01, weight_decay=1e-4, clip_value=0.1)
    ob = torch.optim.Adam(pb, lr=0.01, weight_decay=1e-4)
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=0.01, total_steps=12)
    sb = torch.optim.lr_scheduler.OneCycleLR(ob, max_lr=0.01, total_steps=12)
    for it in range(10):
        for x, y in zip(pa, pb):
            g = torch.randn_like(x) * (0.3 if it % 2 else 0.05)
            x.grad, y.grad = g.clone(), g.clone()
        torch.nn.utils.clip_grad_value_(pb, 0.1)
        oa.step(); ob.step(); sa.step(); sb.step()
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
        assert oa.param_groups[0]["betas"] == ob.param_groups[0]["betas"]
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6), float((x - y).abs().max())
    for x, y in zip(pa, pb):
        assert torch.allclose(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"], rtol=1e-5, atol=1e-9)


def _graph_rank(rank, world, port, out_path, backend):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)                       # rehearsal: both ranks share the one GPU
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    B, C, T = 32, 4, 2500
    pool = synthetic.make_batch(B, C, T, seed=9)
    full = (torch.from_numpy(pool[0]), torch.from_numpy(pool[2]), torch.from_numpy(pool[1]), pool[3],
            torch.ones(B, dtype=torch.long), torch.arange(B))
    args = make_args(method="base", batch_size=B // world, num_steps=12)
    torch.manual_seed(rank)                             # rank 0's weights must win
    net = tm.build_model(args).to(dev)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.train()
    opt, sched = tm.make_optimizer(args, net)
    crit = tm.SELCLoss(pool[2], 2, es=args.num_epochs + 1, device=dev)
    sc = tm.step_counter_class()
    g = tm.GraphedTrainStep(args, net, opt, sched, crit, dev, B // world, C, T,
                            sync=tm.FlatGradSync(net, dev))
    for _ in range(5):
        g.step(tm.shard_batch(full, rank, world), 0, sc)
    torch.cuda.synchronize()
    if rank == 0:
        torch.save([p.detach().cpu() for p in net.parameters() if p.requires_grad], out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,backend", [(2, "gloo"), (1, "nccl")])
def test_graphed_step_with_flat_all_reduce(world, backend, device, tmp_path):
    """The N>1 training step: hipGraph(fwd+bwd+pack) -> one all-reduce -> ClipAdam.  Two ranks
    sharing this box's GPU over gloo (the card allows it; RCCL wants one GPU per rank), and the
    RCCL call itself with a single rank.  Either must equal one process on the whole batch."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "g.pt")
    mp.spawn(_graph_rank, args=(world, port, out, backend), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)
    B, C, T = 32, 4, 2500
    pool = synthetic.make_batch(B, C, T, seed=9)
    full = (torch.from_numpy(pool[0]), torch.from_numpy(pool[2]), torch.from_numpy(pool[1]), pool[3],
            torch.ones(B, dtype=torch.long), torch.arange(B))
    args = make_args(method="base", batch_size=B, num_steps=12)
    torch.manual_seed(0)
    net = tm.build_model(args).to(device)
    for m in net.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net.train()
    opt, sched = tm.make_optimizer(args, net)
    crit = tm.SELCLoss(pool[2], 2, es=args.num_epochs + 1, device=device)
    sc = tm.step_counter_class()
    g = tm.GraphedTrainStep(args, net, opt, sched, crit, device, B, C, T)
    for _ in range(5):
        g.step(full, 0, sc)
    for a, b in zip(got, [p for p in net.parameters() if p.requires_grad]):
        assert torch.allclose(a.to(device), b, rtol=1e-3, atol=2e-5), float((a.to(device) - b).abs().max())


def test_clip_adam_state_dict_round_trip(device):
    """ClipAdam checkpoints interchange with torch.optim.Adam and survive a reload."""
    torch.manual_seed(1)
    pa = [torch.nn.Parameter(torch.randn(5000, device=device)), torch.nn.Parameter(torch.randn(7, device=device))]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa, ob = tm.ClipAdam(pa, lr=0.01, weight_decay=1e-4, clip_value=0.0), torch.optim.Adam(pb, lr=0.01, weight_decay=1e-4)
    for it in range(4):
        if it == 2:
            oa.load_state_dict(copy.deepcopy(ob.state_dict()))        # torch -> ClipAdam
        for x, y in zip(pa, pb):
            g = torch.randn_like(x)
            x.grad, y.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("train", [True, False])
def test_resnet9_channels_last_path_matches_float64(train, device):
    """ResNet9-1D executed as (B,C,1,L) channels_last conv2d/batch_norm/max_pool2d with the same
    parameters, against a float64 CPU run of the Conv1d/BatchNorm1d/MaxPool1d modules: logits,
    every gradient (relative L2), BN buffers.  (The fp32 module path on this stack goes through
    MIOpen Winograd kernels and is itself 5e-2 away from float64 on some gradients, the
    channels_last path 1e-3: profiles/probes/resnet_grad_noise.py — so float64 is the yardstick.)"""
    import copy
    from pcgmix_amd import models
    torch.manual_seed(5)
    ref = models.ResNet9(4, 2).train(train)
    x = torch.randn(8, 4, 2500)
    m = copy.deepcopy(ref).to(device)
    assert m.nhwc
    ref = ref.double()
    xd = x.to(device)
    out = m(xd, depth=0, pass_part="second")
    want = ref(x.double(), depth=0, pass_part="second")
    assert torch.allclose(out.cpu().double(), want, rtol=1e-4, atol=1e-4)
    mid = m(xd, depth=1, pass_part="first")
    assert mid.shape == (8, 128, 1250)
    ref_mid = ref(x.double(), depth=1, pass_part="first")
    assert torch.allclose(mid.cpu().double(), ref_mid, rtol=1e-3, atol=1e-4)
    tail = m(mid, depth=1, pass_part="second")
    ref(ref_mid, depth=1, pass_part="second")       # same sequence of BatchNorm buffer updates
    if not train:                                   # eval: the two-part pass equals the full pass
        assert torch.allclose(tail, out, rtol=1e-4, atol=1e-4)
    else:
        m.zero_grad(set_to_none=True)
        m(xd).square().sum().backward()
        ref(x.double()).square().sum().backward()
        for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
            n = float(q.grad.norm())
            err = float((p.grad.cpu().double() - q.grad).norm())
            assert err <= 1e-2 * n + 1e-4, (k, err, n)
    for (k, v), (_, w) in zip(m.named_buffers(), ref.named_buffers()):
        assert torch.allclose(v.cpu().double(), w.double(), rtol=1e-4, atol=1e-5), k


def test_clip_adam_channels_last_parameters(device):
    """Conv weights kept channels_last (models2d) update exactly like contiguous ones, whether the
    gradient arrives in the parameter's layout or contiguous."""
    torch.manual_seed(2)
    w = torch.randn(16, 8, 3, 3, device=device)
    pa = [torch.nn.Parameter(w.clone().contiguous(memory_format=torch.channels_last))]
    pb = [torch.nn.Parameter(w.clone())]
    oa, ob = tm.ClipAdam(pa, lr=0.01, weight_decay=1e-4, clip_value=0.1), torch.optim.Adam(pb, lr=0.01, weight_decay=1e-4)
    for it in range(4):
        g = torch.randn_like(w) * 0.2
        pa[0].grad = g.clone().contiguous(memory_format=torch.channels_last) if it % 2 else g.clone()
        pb[0].grad = g.clone()
        torch.nn.utils.clip_grad_value_(pb, 0.1)
        oa.step(); ob.step()
    assert pa[0].is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(pa[0], pb[0], rtol=1e-5, atol=1e-6)


def test_resnet9_2d_matches_float64(device):
    """ResNet9-2D on the HIP path (channels_last, conv bias folded into BN) against a float64 CPU
    run of the plain modules: logits, gradients (relative L2), BN buffers; and eval mode."""
    import copy
    from pcgmix_amd import models2d
    torch.manual_seed(3)
    ref = models2d.ResNet9(2).train()
    x = torch.randn(4, 1, 128, 128)
    m = copy.deepcopy(ref).to(device)
    ref = ref.double()
    out = m(x.to(device))
    want = ref(x.double())
    assert torch.allclose(out.cpu().double(), want, rtol=1e-4, atol=1e-4)
    out.square().sum().backward()
    want.square().sum().backward()
    for (k, p), (_, q) in zip(m.named_parameters(), ref.named_parameters()):
        n = float(q.grad.norm())
        assert float((p.grad.cpu().double() - q.grad).norm()) <= 1e-2 * n + 1e-4, k
    for (k, v), (_, w) in zip(m.named_buffers(), ref.named_buffers()):
        assert torch.allclose(v.cpu().double(), w.double(), rtol=1e-4, atol=1e-5), k
    m.eval(); ref.eval()
    assert torch.allclose(m(x.to(device)).cpu().double(), ref(x.double()), rtol=1e-4, atol=1e-4)


def _driver_rank(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = _learnable_dataset(n_rec=24)
    args = argparse.Namespace(dataset="PhysioNet", model="Potes", method="durratiomixup+0.8",
                              num_epochs=6, batch_size=32, op="adam", use_sched=True, lr_max=0.003,
                              weight_decay=1e-4, grad_clip=0.1, seed=4, seed_data=1100001,
                              n_fraction=1.0, train_balance=True, num_classes=2, sample_rate=1000,
                              num_channels=4, valid=False, depth=0, EXPERIMENTS=out_dir)
    perf = tm.train_model(args, ds, dev, use_graph=True, log=None)
    torch.save({"steps": perf["steps"], "loss": perf["train_loss"],
                "params": [p.detach().cpu() for p in perf["model"].parameters()]},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_train_model_driver_two_ranks_graphed(device, tmp_path):
    """train_model() under torch.distributed with the captured step (hipGraph + FlatGradSync):
    two gloo ranks sharing this box's GPU, each on its half of every batch.  Replicas must stay
    bit-identical (same averaged gradients, same optimiser state), the loss must fall."""
    import glob
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_driver_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "rank0.pt"), weights_only=True)
    r1 = torch.load(str(tmp_path / "rank1.pt"), weights_only=True)
    assert r0["steps"] == r1["steps"] and r0["steps"][-1] == 6 * (96 // 32)
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    # (the reference's reseeding gives every step the SAME dropout masks: the first few steps of
    # a run can go either way — profiles/probes/loss_probe_2rank.py; six epochs do not)
    assert r0["loss"][-1] < 0.75 * r0["loss"][0]
    assert len(glob.glob(str(tmp_path / "*" / "model.pth"))) == 1


def _cfg5_rank(rank, world, port, out_dir):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)                       # rehearsal: both ranks share the one GPU
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, C, T = 256, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=40 + rank)
    batch = (torch.from_numpy(x).to(dev), torch.from_numpy(labels), torch.from_numpy(frames), wav,
             torch.ones(B, dtype=torch.long), torch.arange(B))
    args = make_args(model="resnet9", method="durmixmagwarp(0.2,4)", batch_size=B, sig_len=T,
                     num_steps=12, sample_rate=2000)
    out = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(rank)                         # rank 0's initial weights must win
        net = tm.build_model(args).to(dev).train()
        opt, sched = tm.make_optimizer(args, net)
        crit = tm.SELCLoss(labels, 2, es=args.num_epochs + 1, device=dev)
        sc = tm.step_counter_class()
        sync = tm.FlatGradSync(net, dev)
        start = [p.detach().clone() for p in net.parameters()]
        if mode == "eager":
            losses = [float(tm.train_step(args, net, batch, dev, opt, sched, crit, 0, sc, sync=sync))
                      for _ in range(3)]
        else:
            g = tm.GraphedTrainStep(args, net, opt, sched, crit, dev, B, C, T, sync=sync)
            losses = [float(g.step(batch, 0, sc)) for _ in range(3)]
        torch.cuda.synchronize()
        out[mode] = {"losses": losses, "params": [p.detach().cpu() for p in net.parameters()],
                     "moved": any(not torch.equal(a, b) for a, b in zip(start, net.parameters())),
                     "bn_mean": net.conv1[1].running_mean.detach().cpu()}
    torch.save(out, os.path.join(out_dir, f"cfg5_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg5_per_rank_workload_two_ranks(device, tmp_path):
    """BASELINE.json configs[4], per-rank workload at full size: durmixmagwarp(0.2,4) on
    (256,4,5000) + ResNet9-1D, gradients averaged by ONE flat all-reduce per step
    (``FlatGradSync``; replaces nn.DataParallel, reference train_model.py:385).  Two gloo ranks
    share this box's GPU, each with its own batch; eager step and captured step.  Replicas must
    stay bit-identical (same start, same averaged gradients, same optimiser state) while their
    BatchNorm statistics stay per rank, as under DataParallel."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_cfg5_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(str(tmp_path / "cfg5_rank0.pt"), weights_only=True)
    r1 = torch.load(str(tmp_path / "cfg5_rank1.pt"), weights_only=True)
    for mode in ("eager", "graph"):
        a, b = r0[mode], r1[mode]
        assert a["moved"] and b["moved"]
        assert all(np.isfinite(a["losses"])) and all(np.isfinite(b["losses"]))
        for p, q in zip(a["params"], b["params"]):
            assert torch.equal(p, q), mode
        assert not torch.equal(a["bn_mean"], b["bn_mean"])      # per-rank statistics
        assert a["losses"] != b["losses"]                       # different shards
    # the captured step reproduces the eager one (same seeds, no dropout in ResNet9)
    assert np.allclose(r0["eager"]["losses"], r0["graph"]["losses"], rtol=2e-3, atol=1e-4), \
        (r0["eager"]["losses"], r0["graph"]["losses"])


def test_bench_self_launches_two_ranks(device):
    """`python bench.py --gpus 2` from a bare shell (no WORLD_SIZE): bench.py starts the ranks
    itself as child processes before touching the GPU, relays rank 0's line and returns the
    child's exit code.  Rehearsed with gloo: both ranks share this box's one GPU."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PCGMIX_DIST_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20",
                          "--warmup", "3"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                         env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and "error" not in d
    assert abs(d["value"] - 2 * 256 * 20 / (d["ms_per_step"] * 20 / 1e3)) / d["value"] < 1e-6
    assert d["train"]["global_batch"] == 512 and d["train"]["steps_per_s"] > 0
    assert d["train_cfg5"]["global_batch"] == 512 and d["train_cfg5"]["model"] == "resnet9"
    assert np.isfinite(d["train_cfg5"]["loss"])


# ---- pinned to the reference's own train_model.py (tests/golden/train_ref.npz) ------------------
@pytest.mark.parametrize("mode", ["epoch", "epoch_graph", "step", "graph"])
def test_train_step_reproduces_reference_trajectory(mode, device):
    """The HIP path — HIP augmentation, fused Potes stack/head, soft-CE kernels, ClipAdam — run
    for the 10 steps the reference's ``train_epoch`` (train_model.py:490-589) was recorded on:
    per-step loss within 1e-4, learning rates exact, every trained parameter within 1e-3 after
    the 10th step.  'epoch' = train_epoch with ``args.hipgraph = False`` (eager steps),
    'epoch_graph' = train_epoch as it runs by default on a GPU (it builds and replays the captured
    step), 'step' = eager train_step, 'graph' = the captured step (GraphedTrainStep)."""
    import train_replay
    err, worst = train_replay.check_trajectory(train_replay.trajectory(device, mode),
                                               loss_tol=1e-4, param_tol=1e-3)
    print(f"[traj {mode}] max loss err {err:.2e}, max param err {worst:.2e}")


def test_evaluation_matches_reference_golden(device):
    """test_data_accuracy on the device (mean-probability vote and '(class_majority)') == the
    reference's train_model.py:591-670 on the recorded 12-recording loader; logits of the HIP
    forward within 1e-4 of the reference's."""
    import train_replay
    train_replay.check_evaluation(device)


def test_celoss_kernel_matches_reference_golden(device):
    """pcgmix_soft_ce_{fwd,bwd}_f32 == the reference's CELoss (train_model.py:45-54): value and
    gradient, hard and soft targets."""
    import train_replay
    train_replay.check_celoss(device)


def test_models_match_reference_logits_on_hip_path(device):
    """Same seed -> same weights -> the HIP execution path (fused Potes stack + head; ResNet9 as
    channels_last with the HIP BN/ReLU/pool kernels in train mode, MIOpen convolutions) gives the
    logits the reference's own modules gave (tests/golden/models_seed7.npz), within 1e-4."""
    import os
    from conftest import GOLDEN
    from pcgmix_amd import models, models2d
    g = np.load(os.path.join(GOLDEN, "models_seed7.npz"))
    x = torch.from_numpy(g["x1d"]).to(device)
    for build, key, inp in (
        (lambda: models.CNN_potes_TS(4, 2, "PhysioNet"), "potes", x),
        (lambda: models.ResNet9(4, 2), "resnet1d", x),
        (lambda: models2d.ResNet9(2), "resnet2d", torch.from_numpy(g["x2d"]).to(device)),
    ):
        torch.manual_seed(7)
        m = build().to(device).eval()
        if key == "potes":
            assert m._fused_head(inp)                 # the HIP kernels are what runs
        out = m(inp, depth=0, pass_part="second").detach().cpu().numpy()
        err = np.abs(out - g[key + "_logits"]).max()
        assert err <= 1e-4, (key, err)


def test_graphed_step_matches_eager_with_dropout(device):
    """With dropout ON: the captured step reads its masks from a static buffer of random bytes
    refilled by one eager random_() per step, the eager step draws the same number of bytes from
    the same generator — so with the same seed the two runs see the same masks and must agree."""
    B, C, T = 32, 4, 2500
    pool = synthetic.make_batch(B, C, T, seed=9)
    batch = (torch.from_numpy(pool[0]), torch.from_numpy(pool[2]), torch.from_numpy(pool[1]), pool[3],
             torch.ones(B, dtype=torch.long), torch.arange(B))
    results = []
    for graphed in (False, True):
        args = make_args(method="durratiomixup", batch_size=B, num_steps=12)
        torch.manual_seed(0)
        net = tm.build_model(args).to(device).train()
        opt, sched = tm.make_optimizer(args, net)
        crit = tm.SELCLoss(pool[2], 2, es=args.num_epochs + 1, device=device)
        sc = tm.step_counter_class()
        if graphed:
            g = tm.GraphedTrainStep(args, net, opt, sched, crit, device, B, C, T)
        torch.manual_seed(123)                      # the dropout stream of the 5 steps
        losses = [float(g.step(batch, 0, sc)) if graphed else
                  float(tm.train_step(args, net, batch, device, opt, sched, crit, 0, sc)) for _ in range(5)]
        results.append((losses, [p.detach().clone() for p in net.parameters() if p.requires_grad]))
    assert np.allclose(results[0][0], results[1][0], rtol=1e-4, atol=1e-5), results
    assert len(set(results[0][0])) == 5
    for a, b in zip(results[0][1], results[1][1]):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4)


def test_clip_adam_captured_update_matches_eager(device):
    """ClipAdam as a hipGraph node (scalars read from device memory, computed on the host by
    next_hyper before every replay) == the eager ClipAdam.step, bit for bit, under OneCycleLR
    (lr and beta1 change every step); the step count reaches the state dict."""
    torch.manual_seed(0)
    shapes = [(20, 19968), (8, 1, 5), (20,), (2, 20)]
    pa = [torch.nn.Parameter(torch.randn(s, device=device)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = tm.ClipAdam(pa, lr=0.01, weight_decay=1e-4, clip_value=0.1)
    ob = tm.ClipAdam(pb, lr=0.01, weight_decay=1e-4, clip_value=0.1)
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=0.01, total_steps=12)
    sb = torch.optim.lr_scheduler.OneCycleLR(ob, max_lr=0.01, total_steps=12)
    grads = [torch.zeros_like(p) for p in pa]
    for p, g in zip(pa, grads):
        p.grad = g
    hyper = torch.zeros(8, device=device)
    host = np.zeros(8, dtype=np.float32)
    assert oa.can_capture()
    oa.prepare_capture(pa)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        oa.capture_update(hyper)
    for it in range(10):
        for g, y in zip(grads, pb):
            g.copy_(torch.randn_like(g) * (0.3 if it % 2 else 0.05))
            y.grad = g.clone()
        oa.next_hyper(host)
        hyper.copy_(torch.from_numpy(host))
        graph.replay()
        ob.step(); sa.step(); sb.step()
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
        assert torch.equal(oa.state[x]["exp_avg"], ob.state[y]["exp_avg"])
        assert torch.equal(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"])
    sd = oa.state_dict()
    assert all(float(v["step"]) == 10.0 for v in sd["state"].values())
    with pytest.raises(RuntimeError):
        oa.load_state_dict(sd)                               # the graph holds the old moments


def test_clip_adam_interleaved_captured_and_eager_steps(device):
    """train_epoch mixes captured steps with eager ones for batches of another shape: the step
    count (bias corrections) must run through both kinds — captured, captured, EAGER, captured,
    captured, then a re-capture on the same optimiser — exactly as nine eager steps."""
    torch.manual_seed(1)
    shapes = [(20, 1000), (8, 1, 5), (20,)]
    pa = [torch.nn.Parameter(torch.randn(s, device=device)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = tm.ClipAdam(pa, lr=0.01, weight_decay=1e-4, clip_value=0.1)
    ob = tm.ClipAdam(pb, lr=0.01, weight_decay=1e-4, clip_value=0.1)
    grads = [torch.zeros_like(p) for p in pa]
    for p, g in zip(pa, grads):
        p.grad = g
    hyper, host = torch.zeros(8, device=device), np.zeros(8, dtype=np.float32)

    def capture():
        oa.prepare_capture(pa)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            oa.capture_update(hyper)
        return g
    graph = capture()
    for it, kind in enumerate("ccecc" + "R" + "cec"):
        if kind == "R":
            graph = capture()                          # key change: a new capture, same optimiser
            continue
        for g, y in zip(grads, pb):
            g.copy_(torch.randn_like(g) * 0.2)
            y.grad = g.clone()
        if kind == "c":
            oa.next_hyper(host)
            hyper.copy_(torch.from_numpy(host))
            graph.replay()
        else:
            oa.step()
        ob.step()
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)
        assert torch.equal(oa.state[x]["exp_avg_sq"], ob.state[y]["exp_avg_sq"])
    assert all(float(v["step"]) == 8.0 for v in oa.state_dict()["state"].values())
