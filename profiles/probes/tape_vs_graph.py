"""Is the captured Potes train step faster when its kernels are launched directly than when the
hipGraph is replayed?  The library calls made during the capture are recorded (function, arguments)
and re-issued on the current stream; both forms run 2000 times back to back (same static input,
optimiser included), HIP events around the loop."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import pcgmix_amd, bench
from pcgmix_amd import _lib, train_model as tm, synthetic

dev = torch.device("cuda:0")
real = _lib.load()
tape = []


class Recorder:
    def __getattr__(self, name):
        fn = getattr(real, name)

        def call(*a):
            if a and isinstance(a[-1], ctypes.c_void_p):
                tape.append((name, fn, a))
            return fn(*a)
        return call


args = bench.Args("durratiomixup")
args.model, args.dataset, args.batch_size, args.num_channels, args.sig_len = "Potes", "PhysioNet", 256, 4, 5000
args.num_steps, args.num_epochs, args.lr_max, args.weight_decay, args.grad_clip, args.op, args.use_sched = 100000, 1, 0.01, 1e-4, 0.1, "adam", True
args.seed_fix = 4
torch.manual_seed(0)
net = tm.build_model(args).to(dev).train()
opt, sched = tm.make_optimizer(args, net)
crit = tm.SELCLoss(np.zeros(256, int), 2, es=2, device=dev)
orig = _lib.load
_lib.load = lambda: Recorder()
try:
    g = tm.GraphedTrainStep(args, net, opt, sched, crit, dev, 256, 4, 5000)
finally:
    _lib.load = orig
# the capture is the last pass: keep the calls of the last forward/backward/update only
names = [t[0] for t in tape]
last = len(names) - 1 - names[::-1].index("pcgmix_potes_stack_fwd_save_f32")
tape = tape[last:]
print("recorded launches of the captured pass:", [t[0] for t in tape])
x, frames, labels, wav = synthetic.make_batch(256, 4, 5000, sample_rate=2000, seed=1)
g.x.copy_(torch.from_numpy(x).to(dev))


def timeit(f, n=2000):
    for _ in range(50):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def replay_tape():
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for name, fn, a in tape:
        err = fn(*a[:-1], st)
        if err:
            raise RuntimeError(f"{name}: {err}")


t_graph = timeit(g.graph.replay)
t_tape = timeit(replay_tape)
t_graph2 = timeit(g.graph.replay)
print(f"graph replay {t_graph:.1f} us, direct launches {t_tape:.1f} us, graph replay again {t_graph2:.1f} us per step (no augmentation kernel)")
t0 = time.perf_counter()
for _ in range(2000):
    replay_tape()
host = (time.perf_counter() - t0) / 2000 * 1e6
torch.cuda.synchronize()
print(f"host time of the direct launches: {host:.1f} us per step")
