"""Loss trajectory of train_model() on two gloo ranks sharing the GPU (captured step +
FlatGradSync), for a few epoch counts: is a rising loss in a 3-epoch run a bug or chance?"""
import argparse, os, socket, sys, tempfile
import torch
import torch.multiprocessing as mp
sys.path.insert(0, "."); sys.path.insert(0, "tests")


def rank_fn(rank, world, port, epochs, graph):
    import torch.distributed as dist
    import pcgmix_amd  # noqa
    from pcgmix_amd import train_model as tm
    from conftest import learnable_dataset
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ds = learnable_dataset(n_rec=24)
    args = argparse.Namespace(dataset="PhysioNet", model="Potes", method="durratiomixup+0.8",
                              num_epochs=epochs, batch_size=32, op="adam", use_sched=True, lr_max=0.003,
                              weight_decay=1e-4, grad_clip=0.1, seed=4, seed_data=1100001,
                              n_fraction=1.0, train_balance=True, num_classes=2, sample_rate=1000,
                              num_channels=4, valid=False, depth=0, EXPERIMENTS=tempfile.mkdtemp())
    perf = tm.train_model(args, ds, dev, use_graph=graph, log=None)
    print(f"epochs {epochs} graph {graph} rank {rank}:", [round(v, 4) for v in perf["train_loss"]], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    for epochs in (3, 8):
        for graph in (True, False):
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            mp.spawn(rank_fn, args=(2, port, epochs, graph), nprocs=2, join=True)
