#!/usr/bin/env python3
"""Fold the outputs of profiles/run_r4_profiles.sh (gpurun_out/, scratch) into profiles/ (tracked):

  r4_bench_kernel_stats.csv             rocprofv3's per-kernel stats of the driver's command line
  r4_bench_pcgmix_kernels_by_grid.csv   our kernels in that run grouped by launch grid
  r4_train_step_timeline.txt            one captured Potes train step, kernel by kernel
  r4_cfg3_step_timeline.txt             one saliency-guided augment() step, kernel by kernel
  r4_mix_roofline.json                  splice kernels: durations + FETCH_SIZE / WRITE_SIZE passes

    python profiles/refresh_profiles_r4.py [gpurun_out] [git head]
"""
import collections
import csv
import os
import shutil
import statistics
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
head = sys.argv[2] if len(sys.argv) > 2 else "?"
csv.field_size_limit(1 << 30)
env = dict(os.environ, PCGMIX_ROUND="r4")

bench_dir = os.path.join(src, "prof_r4_bench")
shutil.copy(os.path.join(bench_dir, "b_kernel_stats.csv"), os.path.join(HERE, "r4_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(os.path.join(bench_dir, "b_kernel_trace.csv"))))
groups = collections.defaultdict(list)
for r in rows:
    if "pcgmix::" in r["Kernel_Name"] or "label_argmax" in r["Kernel_Name"] or "seed_frames" in r["Kernel_Name"]:
        key = (r["Kernel_Name"].split("(")[0], r["Grid_Size_X"], r["Grid_Size_Y"], r["VGPR_Count"],
               r["SGPR_Count"], r["LDS_Block_Size"])
        groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(HERE, "r4_bench_pcgmix_kernels_by_grid.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size_X", "Grid_Size_Y", "VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                "count", "mean_ns", "median_ns", "min_ns", "max_ns"])
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        w.writerow(list(key) + [len(v), round(statistics.mean(v), 1), statistics.median(v), min(v), max(v)])
subprocess.run([sys.executable, os.path.join(HERE, "train_timeline.py"), os.path.join(bench_dir, "b_kernel_trace.csv")],
               check=True, env=env)
cfg3 = os.path.join(src, "prof_r4_cfg3", "c_kernel_trace.csv")
if os.path.exists(cfg3):
    subprocess.run([sys.executable, os.path.join(HERE, "cfg3_timeline.py"), cfg3], check=True, env=env)
if os.path.isdir(os.path.join(src, "pmc")):
    subprocess.run([sys.executable, os.path.join(HERE, "summarize_mix_pmc.py"), os.path.join(src, "pmc"), head, "4"],
                   check=True)
