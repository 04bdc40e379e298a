"""Inputs of the train-step / evaluation goldens (train_ref.npz), shared by the script that
records them from the reference (make_golden_train.py) and the tests that replay them.  Pure
data builders on this project's synthetic generator: nothing here touches the reference."""
import argparse

import numpy as np
import torch

import pcgmix_amd  # noqa: F401
from pcgmix_amd import synthetic

TRAJ_STEPS, TRAJ_B = 10, 16
EVAL_SEED = 86      # 5 of the 12 recordings vote differently under the two rules


def traj_args():
    """The Namespace fields train_epoch reads (train_model.py:490-589)."""
    return argparse.Namespace(
        dataset="PhysioNet", model="Potes", method="durmixmagwarp(0.2,4)", num_epochs=1,
        batch_size=TRAJ_B, op="adam", use_sched=True, lr_max=0.01, weight_decay=1e-4, grad_clip=0.1,
        seed=4, seed_fix=4, num_classes=2, num_channels=4, sample_rate=1000, depth=0, sig_len=2500,
        num_steps=TRAJ_STEPS, latent_space=False, classical_space=False)


def traj_batches():
    out = []
    for i in range(TRAJ_STEPS):
        x, frames, labels, wav = synthetic.make_batch(TRAJ_B, 4, 2500, sample_rate=1000, seed=200 + i)
        out.append((torch.from_numpy(x), torch.from_numpy(labels), torch.from_numpy(frames), wav,
                    torch.ones(TRAJ_B, dtype=torch.long), torch.arange(TRAJ_B) + TRAJ_B * i))
    return out


def eval_pool(seed=EVAL_SEED):
    """48 cycles, 12 recordings of 2..6 cycles, recording label = label of its cycles."""
    x, frames, labels, _ = synthetic.make_batch(48, 4, 2500, sample_rate=1000, seed=seed)
    sizes = [2, 6, 3, 5, 4, 4, 2, 6, 5, 3, 4, 4]
    wav, lab = [], []
    for r, n in enumerate(sizes):
        wav += [f"{'abcdef'[r % 6]}{r:04d}"] * n
        lab += [r % 2] * n
    labels = np.asarray(lab, dtype=np.int64)
    # interleave so that a recording's cycles are spread over both batches
    order = np.random.RandomState(5).permutation(48)
    return x[order], labels[order], [wav[i] for i in order]


def eval_loader():
    x, labels, wav = eval_pool()
    return [(torch.from_numpy(x[i:i + 24]), torch.from_numpy(labels[i:i + 24]), None,
             tuple(wav[i:i + 24]), None, None) for i in (0, 24)]
