#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of profiles/run_r2_profiles.sh from gpurun_out/ (scratch) into
profiles/ (tracked):

  r2_bench_kernel_stats.csv             rocprofv3's own per-kernel stats of `python bench.py --no-cpu --no-extra`
  r2_bench_pcgmix_kernels_by_grid.csv   our kernels in that run grouped by launch grid
  r2_resnet{1d,2d}_step_kernels.csv     ONE steady-state training step of the ResNet9 legs (the
                                        kernels between two optimiser launches; the run as a whole
                                        is dominated by MIOpen's find-mode benchmarking)

    python profiles/refresh_profiles_r2.py [gpurun_out]
"""
import collections
import csv
import os
import shutil
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
csv.field_size_limit(1 << 30)


def dur(r):
    return int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


# the bench alone is re-profiled more often than the ResNet probes (run_r2_bench_profile.sh writes
# prof_r2_bench2): take the newer of the two traces
bench_dir = max((os.path.join(src, d) for d in ("prof_r2_bench", "prof_r2_bench2")
                 if os.path.exists(os.path.join(src, d, "b_kernel_trace.csv"))), key=os.path.getmtime)
shutil.copy(os.path.join(bench_dir, "b_kernel_stats.csv"), os.path.join(HERE, "r2_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(os.path.join(bench_dir, "b_kernel_trace.csv"))))
groups = collections.defaultdict(list)
for r in rows:
    if "pcgmix::" in r["Kernel_Name"] or "label_argmax" in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0], r["Grid_Size_X"], r["Grid_Size_Y"], r["VGPR_Count"],
               r["SGPR_Count"], r["LDS_Block_Size"])
        groups[key].append(dur(r))
with open(os.path.join(HERE, "r2_bench_pcgmix_kernels_by_grid.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size_X", "Grid_Size_Y", "VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                "count", "mean_ns", "median_ns", "min_ns", "max_ns"])
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        w.writerow(list(key) + [len(v), round(statistics.mean(v), 1), statistics.median(v), min(v), max(v)])

for tag in ("rn1d", "rn2d"):
    path = os.path.join(src, f"prof_r2_{tag}", "r_kernel_trace.csv")
    if not os.path.exists(path):
        continue
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adam_clip_multi_kernel" in r["Kernel_Name"]]
    ends = [i for j, i in enumerate(idx) if j + 1 == len(idx) or idx[j + 1] - i > 5]
    step = rows[ends[-3] + 1:ends[-2] + 1]
    wall = int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
    g = collections.defaultdict(list)
    for r in step:
        g[r["Kernel_Name"].replace("void ", "")[:140]].append(dur(r))
    busy = sum(sum(v) for v in g.values())
    out = os.path.join(HERE, f"r2_resnet{tag[2:]}_step_kernels.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow([f"# one steady-state step: {len(step)} launches, busy {busy / 1e6:.3f} ms, wall {wall / 1e6:.3f} ms"])
        w.writerow(["Kernel_Name", "calls", "total_ms", "percent", "mean_us"])
        for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([k, len(v), round(sum(v) / 1e6, 4), round(100 * sum(v) / busy, 2), round(statistics.mean(v) / 1e3, 1)])
    print(out, f"busy {busy / 1e6:.2f} ms wall {wall / 1e6:.2f} ms")
