"""The armed plain step (csrc/pcgmix_kernels.h ArmedArgs; include/pcgmix_hip.h pcgmix_ctx_armed_*):
with the reference's signature — labels on the device (augmentations.py:501) — a splice of up to 256
samples, plain or with magnitude_warp (augmentations.py:674-683), is ONE kernel, launched before its
index block exists.  Everything here compares it
with the two-launch path (host labels: no read-back, the index block in the kernel arguments) and
with the CPU oracle: partners and waveforms bit for bit.  The ways out of a waiting kernel are
tested too: malformed boundaries after the launch, a host that writes its records too late."""
import ctypes

import numpy as np
import pytest
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import _lib, augmentations, augmentations2d, synthetic
from conftest import Args, StepCounter
from oracle import pcgmix_oracle as O

pytestmark = pytest.mark.gpu


def _stats(device):
    out = (ctypes.c_longlong * 3)()
    _lib.check(_lib.load().pcgmix_ctx_armed_stats(augmentations.step_context(device.index), out), "stats")
    return list(out)


def _step(mod, method, data, labels, frames, wav, step, device, host_labels=None):
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), int(labels.max()) + 1).to(device)
    return mod.augment(Args(method), data, tgt, torch.from_numpy(frames), wav, StepCounter(step), None,
                       device, "", host_labels=host_labels)


@pytest.mark.parametrize("B,C,T,rate", [(256, 4, 5000, 2000), (256, 1, 5000, 2000), (32, 4, 2500, 1000),
                                        (7, 3, 1000, 400), (1, 4, 5000, 2000), (255, 2, 2500, 1000)])
def test_armed_step_equals_two_launch_path_and_oracle(B, C, T, rate, device):
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=rate, seed=B + C)
    data = torch.from_numpy(x).to(device)
    before = _stats(device)
    for step in (0, 3, 41):
        y_a, _, mix_a, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, step, device)
        y_h, _, mix_h, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, step, device,
                                 host_labels=labels)
        assert np.array_equal(mix_a, mix_h)
        assert torch.equal(y_a, y_h) and not torch.equal(y_a, data)
        ref = O.augment("durratiomixup", x, labels, frames, wav, step)
        assert np.array_equal(mix_a, ref["mix"]) and np.array_equal(y_a.cpu().numpy(), ref["y"])
    after = _stats(device)
    assert after[0] - before[0] == 3, "the strict-signature step did not take the armed kernel"
    assert after[2] == before[2], "an armed kernel gave up"


@pytest.mark.parametrize("B,C,T,rate,method", [(256, 4, 5000, 2000, "durmixmagwarp(0.2,4)"),
                                               (256, 1, 5000, 2000, "durmixmagwarp(0.2,4)"),
                                               (31, 4, 2500, 1000, "durmixmagwarp(0.1,3)"),
                                               (64, 3, 5000, 2000, "durmixmagwarp(0.3,8)")])
def test_armed_splice_warp_equals_two_launch_path_and_oracle(B, C, T, rate, method, device):
    """PCGmix+ (augmentations.py:674-683, 924-928) through the strict signature: the armed splice + warp
    kernel reads its knots from the pinned slot while it waits.  Bit-equal to the path with host labels
    (label kernel-free, fetch + launch); against the oracle the warp's usual bar (<= 1 ulp, rare)."""
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=rate, seed=B + C)
    data = torch.from_numpy(x).to(device)
    before = _stats(device)
    for step in (1, 8, 23):
        y_a, _, mix_a, _ = _step(augmentations, method, data, labels, frames, wav, step, device)
        y_h, _, mix_h, _ = _step(augmentations, method, data, labels, frames, wav, step, device,
                                 host_labels=labels)
        assert np.array_equal(mix_a, mix_h) and torch.equal(y_a, y_h) and not torch.equal(y_a, data)
        ref = O.augment(method, x, labels, frames, wav, step)
        got = y_a.cpu().numpy()
        assert np.array_equal(mix_a, ref["mix"]) and np.abs(got - ref["y"]).max() <= 1e-4
        ulp = np.abs(got.view(np.int32).astype(np.int64) - ref["y"].view(np.int32).astype(np.int64))
        assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3
    after = _stats(device)
    assert after[0] - before[0] == 3 and after[2] == before[2]


def test_late_records_for_splice_warp(device):
    """The relaunch after a give-up, for the splice + warp step (it goes through the staged path with the
    labels the call already holds)."""
    lib = _lib.load()
    ctx = augmentations.step_context(device.index)
    B, C, T = 80, 4, 5000
    method = "durmixmagwarp(0.2,4)"
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=13)
    data = torch.from_numpy(x).to(device)
    want, _, mix_w, _ = _step(augmentations, method, data, labels, frames, wav, 4, device, host_labels=labels)
    try:
        before = _stats(device)
        _lib.check(lib.pcgmix_ctx_armed_debug(ctx, 200000, 600), "debug")
        y, _, mix, _ = _step(augmentations, method, data, labels, frames, wav, 4, device)
        after = _stats(device)
        assert np.array_equal(mix, mix_w) and torch.equal(y, want)
        assert after[1] - before[1] == 1 and after[2] - before[2] == 1
    finally:
        _lib.check(lib.pcgmix_ctx_armed_debug(ctx, 0, 0), "debug")
    y, _, mix, _ = _step(augmentations, method, data, labels, frames, wav, 4, device)
    assert torch.equal(y, want) and _stats(device)[2] == after[2]


def test_armed_step_on_spectrograms(device):
    """2D plain splice (augmentations2d.py:206-221): (B, 1, F, W) through the same armed kernel."""
    rs = np.random.RandomState(5)
    B, F, W = 64, 128, 128
    x = rs.standard_normal((B, 1, F, W)).astype(np.float32)
    frames = np.concatenate([np.zeros((B, 1), np.int64), np.sort(rs.randint(1, W + 1, (B, 4)), axis=1)], axis=1)
    labels = rs.randint(0, 2, B).astype(np.int64)
    wav = ["a%04d" % i for i in range(B)]
    data = torch.from_numpy(x).to(device)
    before = _stats(device)
    y_a, _, mix_a, _ = _step(augmentations2d, "durratiomixup", data, labels, frames, wav, 9, device)
    y_h, _, mix_h, _ = _step(augmentations2d, "durratiomixup", data, labels, frames, wav, 9, device,
                             host_labels=labels)
    assert np.array_equal(mix_a, mix_h) and torch.equal(y_a, y_h) and not torch.equal(y_a, data)
    assert _stats(device)[0] - before[0] == 1


def test_armed_steps_back_to_back(device):
    """300 consecutive steps with changing labels, boundaries and step numbers, no synchronisation in
    between (the records of step k+1 are written while the kernel of step k may still be running):
    every output equals the two-launch path's."""
    B, C, T = 256, 4, 5000
    batches = [synthetic.make_batch(B, C, T, sample_rate=2000, seed=s) for s in range(4)]
    datas = [torch.from_numpy(b[0]).to(device) for b in batches]
    tgts = [torch.nn.functional.one_hot(torch.from_numpy(b[2]), 2).to(device) for b in batches]
    frs = [torch.from_numpy(b[1]) for b in batches]
    outs = []
    args = Args("durratiomixup")
    for k in range(300):
        i = k % 4
        y, _, mix, _ = augmentations.augment(args, datas[i], tgts[i], frs[i], batches[i][3], StepCounter(k),
                                             None, device, "")
        outs.append((y, mix))
    torch.cuda.synchronize()
    for k in range(0, 300, 7):
        i = k % 4
        y, _, mix, _ = augmentations.augment(args, datas[i], tgts[i], frs[i], batches[i][3], StepCounter(k),
                                             None, device, "", host_labels=batches[i][2])
        assert np.array_equal(outs[k][1], mix) and torch.equal(outs[k][0], y), k


def test_armed_step_behind_queued_work_and_on_a_side_stream(device):
    """The label write is far away when the call starts (40 ms of matrix products queued in front), and
    the next step comes on ANOTHER stream: the host waits politely, the records of the first kernel
    are not overwritten before it has read them."""
    B, C, T = 128, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=2)
    data = torch.from_numpy(x).to(device)
    want, _, mix_w, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 5, device,
                              host_labels=labels)
    a = torch.randn(4096, 4096, device=device)
    torch.cuda.synchronize()
    for _ in range(40):
        a = (a @ a).clamp_(-1, 1)
    y1, _, mix1, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 5, device)
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        y2, _, mix2, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 5, device)
    torch.cuda.synchronize()
    assert np.array_equal(mix1, mix_w) and np.array_equal(mix2, mix_w)
    assert torch.equal(y1, want) and torch.equal(y2, want)


def test_malformed_boundaries_release_the_waiting_kernel(device):
    """The boundaries are checked after the launch (the check runs while the GPU gets to the kernel):
    a refusal must let the waiting blocks go, and the next step must work."""
    B, C, T = 64, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=8)
    data = torch.from_numpy(x).to(device)
    bad = frames.copy()
    bad[3, 4] = T + 1
    with pytest.raises(ValueError):
        _step(augmentations, "durratiomixup", data, labels, bad, wav, 2, device)
    torch.cuda.synchronize()                   # returns: nobody is waiting any more
    y, _, mix, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 2, device)
    ref = O.augment("durratiomixup", x, labels, frames, wav, 2)
    assert np.array_equal(mix, ref["mix"]) and np.array_equal(y.cpu().numpy(), ref["y"])


def test_late_records_are_noticed_and_the_splice_is_launched_again(device):
    """A host that stalls between the launch and the record write (a debugger, a stopped process):
    the waiting blocks give up after their timeout (2 ms here, 1 s by default), the call sees it —
    records later than 0.4 s after the launch are checked after a stream synchronisation — and
    launches the two-launch path's kernel.  Same output; and a stall the kernel survives (default
    timeout) is only checked, not repeated."""
    lib = _lib.load()
    ctx = augmentations.step_context(device.index)
    B, C, T = 96, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=11)
    data = torch.from_numpy(x).to(device)
    ref = O.augment("durratiomixup", x, labels, frames, wav, 6)
    try:
        before = _stats(device)
        _lib.check(lib.pcgmix_ctx_armed_debug(ctx, 200000, 600), "debug")
        y, _, mix, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 6, device)
        mid = _stats(device)
        assert np.array_equal(mix, ref["mix"]) and np.array_equal(y.cpu().numpy(), ref["y"])
        assert mid[1] - before[1] == 1 and mid[2] - before[2] == 1
        _lib.check(lib.pcgmix_ctx_armed_debug(ctx, 0, 600), "debug")
        y, _, mix, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 6, device)
        after = _stats(device)
        assert np.array_equal(y.cpu().numpy(), ref["y"])
        assert after[1] - mid[1] == 1 and after[2] == mid[2]
    finally:
        _lib.check(lib.pcgmix_ctx_armed_debug(ctx, 0, 0), "debug")
    y, _, mix, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 6, device)
    assert np.array_equal(y.cpu().numpy(), ref["y"]) and _stats(device)[2] == after[2]


def test_begin_without_finish_is_released_by_the_next_begin(device):
    """The two-call form (pcgmix_augment_plain_begin / _finish): a begin whose finish never comes — the
    binding raised between the calls — leaves a kernel waiting; the next begin releases it.  Shapes the
    armed kernel does not take are reported (PCGMIX_NOT_ARMED) with nothing enqueued, and a finish
    without a begin is refused."""
    lib = _lib.load()
    ctx = augmentations.step_context(device.index)
    B, C, T = 48, 4, 5000
    x, frames, labels, wav = synthetic.make_batch(B, C, T, sample_rate=2000, seed=21)
    data = torch.from_numpy(x).to(device)
    tgt = torch.nn.functional.one_hot(torch.from_numpy(labels), 2).to(device)
    lost = torch.empty_like(data)
    st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    assert lib.pcgmix_augment_plain_begin(ctx, data.data_ptr(), lost.data_ptr(), tgt.data_ptr(), 2, B, C, T, st) == 0
    y, _, mix, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 12, device)   # begins again
    torch.cuda.synchronize()                   # returns: the abandoned kernel was released
    ref = O.augment("durratiomixup", x, labels, frames, wav, 12)
    assert np.array_equal(mix, ref["mix"]) and np.array_equal(y.cpu().numpy(), ref["y"])
    odd = torch.empty(B, C, T - 2, device=device)            # T % 4 != 0
    assert lib.pcgmix_augment_plain_begin(ctx, odd.data_ptr(), torch.empty_like(odd).data_ptr(), tgt.data_ptr(), 2,
                                          B, C, T - 2, st) == -3
    fr = np.ascontiguousarray(frames)
    mix_out = np.empty(B, dtype=np.int64)
    assert lib.pcgmix_augment_plain_finish(ctx, fr.ctypes.data, 12, ctypes.c_float(0.5), mix_out.ctypes.data) != 0
    y2, _, mix2, _ = _step(augmentations, "durratiomixup", data, labels, frames, wav, 12, device)
    assert np.array_equal(mix2, ref["mix"]) and np.array_equal(y2.cpu().numpy(), ref["y"])
