import argparse, sys, tempfile
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import pcgmix_amd  # noqa
from pcgmix_amd import train_model as tm
from conftest import learnable_dataset
dev = torch.device("cuda", 0)
for epochs, bs in ((3, 32), (8, 32), (3, 16), (8, 16)):
    for graph in (True, False):
        ds = learnable_dataset(n_rec=24)
        args = argparse.Namespace(dataset="PhysioNet", model="Potes", method="durratiomixup+0.8",
                                  num_epochs=epochs, batch_size=bs, op="adam", use_sched=True, lr_max=0.003,
                                  weight_decay=1e-4, grad_clip=0.1, seed=4, seed_data=1100001,
                                  n_fraction=1.0, train_balance=True, num_classes=2, sample_rate=1000,
                                  num_channels=4, valid=False, depth=0, EXPERIMENTS=tempfile.mkdtemp())
        perf = tm.train_model(args, ds, dev, use_graph=graph, log=None)
        print(epochs, bs, "graph" if graph else "eager", [round(v, 4) for v in perf["train_loss"]])
