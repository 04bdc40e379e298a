#!/usr/bin/env python3
"""One saliency-guided step (cfg3) out of a rocprofv3 kernel trace of profiles/probes/cfg3_probe.py:
the launches from one splice kernel to the next, written to profiles/r2_cfg3_step_timeline.txt.

    python profiles/cfg3_timeline.py gpurun_out/prof_r2_cfg3/c_kernel_trace.csv
"""
import csv
import os
import statistics
import sys

csv.field_size_limit(1 << 30)
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
mix = [i for i, r in enumerate(rows) if "mix_warp_" in r["Kernel_Name"]]
steps = [(a, b) for a, b in zip(mix, mix[1:])
         if any("salopt_disp" in r["Kernel_Name"] for r in rows[a + 1:b + 1])]
walls = [int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"]) for a, b in steps]
keep = [i for i, w in enumerate(walls) if w <= 1.3 * min(walls)]
steps, walls = [steps[i] for i in keep], [walls[i] for i in keep]
med = statistics.median(walls)
a, b = min(zip(steps, walls), key=lambda sw: abs(sw[1] - med))[0]
t0 = int(rows[a]["End_Timestamp"])
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ.get("PCGMIX_ROUND", "r3") + "_cfg3_step_timeline.txt")
with open(out, "w") as f:
    f.write(f"# one saliency-guided step ((saloptenv)durmixmagwarp(0.2,4), Potes saliency model, bs 256) under "
            f"rocprofv3 --kernel-trace: {med / 1e3:.1f} us from splice end to splice end (median of {len(steps)} steps)\n")
    f.write("# start_us  duration_us  kernel      (t = 0: end of the previous step's splice kernel)\n")
    busy = 0
    for r in rows[a + 1:b + 1]:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        f.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {d / 1e3:8.1f}  {r['Kernel_Name'][:110]}\n")
    f.write(f"# kernel sum {busy / 1e3:.1f} us\n")
print(open(out).read())
