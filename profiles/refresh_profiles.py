#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of the last `rocprofv3 --kernel-trace --stats -- python bench.py
--no-cpu` run from gpurun_out/ (scratch) into profiles/ (tracked):

  r1_bench_kernel_stats.csv            rocprofv3's own per-kernel stats
  r1_bench_pcgmix_kernels_by_grid.csv  pcgmix:: kernels grouped by launch grid (one kernel name
                                       serves several workloads in one bench run)
  r1_mix_kernel_summary.json           kernel_trace block of the headline kernel refreshed (the PMC
                                       block is kept: it comes from separate --pmc passes)

    python profiles/refresh_profiles.py gpurun_out/prof_bench
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys
from collections import defaultdict

HERE = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_bench"


def newest(pattern):
    files = glob.glob(os.path.join(src, "**", pattern), recursive=True)
    if not files:
        raise SystemExit(f"no {pattern} under {src}")
    return max(files, key=os.path.getmtime)


shutil.copy(newest("*kernel_stats.csv"), os.path.join(HERE, "r1_bench_kernel_stats.csv"))
rows = list(csv.DictReader(open(newest("*kernel_trace.csv"))))
groups = defaultdict(list)
for r in rows:
    if "pcgmix::" not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["VGPR_Count"], r["Accum_VGPR_Count"],
           r["SGPR_Count"], r["LDS_Block_Size"])
    groups[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(HERE, "r1_bench_pcgmix_kernels_by_grid.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size_X", "Grid_Size_Y", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count",
                "LDS_Block_Size", "count", "mean", "median", "min", "max"])
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        w.writerow(list(key) + [len(v), statistics.mean(v), statistics.median(v), min(v), max(v)])
summ_path = os.path.join(HERE, "r1_mix_kernel_summary.json")
summ = json.load(open(summ_path))
v = [d for k, d in groups.items() if "mix_warp_kernel<4, false, 2>" in k[0] and k[1] == "2560" and k[2] == "256"]
if v:
    v = v[0]
    summ["kernel_trace"] = {"calls": len(v), "mean_ns": statistics.mean(v), "median_ns": statistics.median(v),
                            "min_ns": min(v), "max_ns": max(v)}
    json.dump(summ, open(summ_path, "w"), indent=1)
print("refreshed", os.listdir(HERE))
