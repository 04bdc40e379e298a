#!/usr/bin/env python3
"""One captured Potes train step out of a rocprofv3 kernel trace of `bench.py --no-cpu --no-extra`
(profiles/run_r2_bench_profile.sh): the launches from one optimiser kernel to the next in the
graphed train leg, with start offsets and durations, written to profiles/r2_train_step_timeline.txt.

    python profiles/train_timeline.py gpurun_out/prof_r2_bench2/b_kernel_trace.csv
"""
import csv
import os
import statistics
import sys

csv.field_size_limit(1 << 30)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam_clip_multi_kernel" in r["Kernel_Name"]]
# graphed steps: consecutive optimiser launches with a potes_bwd_kernel<true> in between and few launches
steps = []
for a, b in zip(adam, adam[1:]):
    seg = rows[a + 1:b + 1]
    if len(seg) <= 14 and any("potes_bwd" in r["Kernel_Name"] for r in seg):
        steps.append((a, b))
walls = [int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"]) for a, b in steps]
# the eager leg of the bench launches the same kernels, host-bound (~2x the wall): keep the steps
# within 1.3x of the fastest, i.e. the replayed ones
keep = [i for i, w in enumerate(walls) if w <= 1.3 * min(walls)]
steps, walls = [steps[i] for i in keep], [walls[i] for i in keep]
med = statistics.median(walls)
a, b = min(((a, b) for (a, b), w in zip(steps, walls)), key=lambda ab: abs(
    int(rows[ab[1]]["Start_Timestamp"]) - int(rows[ab[0]]["Start_Timestamp"]) - med))
t0 = int(rows[a]["End_Timestamp"])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PCGMIX_ROUND", "r3") + "_train_step_timeline.txt")
with open(out, "w") as f:
    f.write(f"# one captured train step (Potes 1D-CNN, bs 256, durratiomixup) under rocprofv3 --kernel-trace: "
            f"{med / 1e3:.1f} us from optimiser launch to optimiser launch (median of {len(steps)} steps)\n")
    f.write("# start_us  duration_us  kernel      (t = 0: end of the previous step's optimiser kernel)\n")
    busy = 0
    for r in rows[a + 1:b + 1]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d / 1e3:8.1f}  {r['Kernel_Name'][:110]}\n")
    f.write(f"# kernel sum {busy / 1e3:.1f} us\n")
print(open(out).read())
