"""Host enqueue time vs wall time of the captured Potes train step (bs 256, durratiomixup):
is the step GPU-bound or host-bound?  Prints the mean us/step of the Python side alone (loop end
before the final synchronize) and of the whole thing."""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402

dev = torch.device("cuda", 0)
method = sys.argv[1] if len(sys.argv) > 1 else "durratiomixup"
print("method", method)
step, info = bench.build_train_step(method, "Potes", 256, 4, 5000, 2000, dev, 5000, 0)
for _ in range(50):
    step()
torch.cuda.synchronize()
for rep in range(3):
    n = 1000
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: host loop {1e6 * (t1 - t0) / n:7.1f} us/step, with drain {1e6 * (t2 - t0) / n:7.1f} us/step "
          f"({n / (t2 - t0):.0f} step/s)")
# the Python side alone: same loop with the GPU work replaced by nothing is not possible, so
# time the pieces that do not touch the GPU
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
