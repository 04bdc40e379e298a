#!/usr/bin/env python3
"""Fold the rocprofv3 passes of profiles/run_mix_pmc.sh (gpurun_out/pmc/<mode>_<B>/{trace,fetch,write})
into profiles/r<round>_mix_roofline.json (round = 3rd argument, default 3): per workload the kernel's mean duration (kernel trace), raw
FETCH_SIZE / WRITE_SIZE per launch (KB as rocprofv3 reports them), the gfx950 correction the guide
prescribes (FETCH_SIZE x 2 for wide coalesced reads, WRITE_SIZE as is), the exact bytes of the
batch, and every fraction of the 8 TB/s roofline one can form from them.

    python profiles/summarize_mix_pmc.py [gpurun_out/pmc] [git head] [round]
"""
import csv
import glob
import json
import os
import statistics
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
head = sys.argv[2] if len(sys.argv) > 2 else "?"
ROUND = int(sys.argv[3]) if len(sys.argv) > 3 else 3
OUT = os.path.join(HERE, f"r{ROUND}_mix_roofline.json")
csv.field_size_limit(1 << 30)
PEAK = 8000.0


def rows(path_glob, name_filter="mix_warp_"):
    out = []
    for path in glob.glob(path_glob):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                if name_filter in r["Kernel_Name"]:
                    out.append(r)
    return out


def summ(v):
    return {"launches": len(v), "mean": statistics.mean(v), "min": min(v), "max": max(v)}


# Workloads collected earlier stay in the file (a later run may re-profile a subset only).
prev = {}
try:
    prev = json.load(open(OUT)).get("workloads", {})
except (OSError, ValueError):
    pass
result = {"round": ROUND, "collected_at": head, "peak_GBs": PEAK,
          "how": "profiles/run_mix_pmc.sh: per workload `rocprofv3 --kernel-trace --stats`, "
                 "`rocprofv3 --pmc FETCH_SIZE`, `rocprofv3 --pmc WRITE_SIZE` (separate passes) around "
                 "profiles/mix_pmc_probe.py (the splice kernel alone, 20 or 200 launches back to back)",
          "correction": "MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE reports half the bytes of wide "
                        "coalesced streaming reads -> doubled; WRITE_SIZE is exact for 16-byte stores. "
                        "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KB (1024 B). The 'copy' "
                        "workloads (no blended range: own read + write only, bytes known exactly) are "
                        "the calibration of that correction on this kernel's own access pattern.",
          "workloads": {}}
for info_path in sorted(glob.glob(os.path.join(src, "mixprobe_*.json"))):
    info = json.load(open(info_path))
    tag = f"{info['mode']}_{info['B']}"
    d = os.path.join(src, tag)
    tr = rows(os.path.join(d, "trace", "*kernel_trace.csv"))
    fe = rows(os.path.join(d, "fetch", "*counter_collection.csv"))
    wr = rows(os.path.join(d, "write", "*counter_collection.csv"))
    if not (tr and fe and wr):
        print("incomplete:", tag, len(tr), len(fe), len(wr))
        continue
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
    dur = dur[len(dur) // 10:]                       # drop the first launches (cold caches, clocks)
    fetch = [float(r["Counter_Value"]) for r in fe if r["Counter_Name"] == "FETCH_SIZE"]
    write = [float(r["Counter_Value"]) for r in wr if r["Counter_Name"] == "WRITE_SIZE"]
    ns = statistics.mean(dur)
    fetch_b, write_b = statistics.mean(fetch) * 1024.0, statistics.mean(write) * 1024.0
    hbm = 2.0 * fetch_b + write_b
    B, C, T = info["B"], info["C"], info["T"]
    name = {"splice": "durratiomixup", "warp": "durmixmagwarp(0.2,4)", "copy": "copy (no blended range)",
            "karg": "durratiomixup [kernarg]"}[info["mode"]]
    w = {"kernel": info["kernel"], "grid": [tr[0]["Grid_Size_X"], tr[0]["Grid_Size_Y"]],
         "vgpr": int(tr[0]["VGPR_Count"]), "launch_ns": summ(dur),
         "FETCH_SIZE_KB": summ(fetch), "WRITE_SIZE_KB": summ(write),
         "fetch_bytes_raw": fetch_b, "fetch_bytes_x2": 2.0 * fetch_b, "write_bytes": write_b,
         "hbm_bytes_per_launch": hbm, "exact_bytes": info["exact_bytes"],
         "own_plus_write_bytes": info["own_plus_write_bytes"],
         "contract_12CT_bytes": info["contract_12CT_bytes"],
         "read_bytes_expected": info["exact_bytes"] - 4.0 * B * C * T,
         "fetch_x2_over_expected_reads": 2.0 * fetch_b / (info["exact_bytes"] - 4.0 * B * C * T),
         "write_over_expected": write_b / (4.0 * B * C * T),
         "GBs_on_counter_bytes": hbm / ns, "frac_on_counter_bytes": hbm / ns / PEAK,
         "GBs_on_exact_bytes": info["exact_bytes"] / ns, "frac_on_exact_bytes": info["exact_bytes"] / ns / PEAK,
         "GBs_on_12CT_model": info["contract_12CT_bytes"] / ns,
         "frac_on_12CT_model": info["contract_12CT_bytes"] / ns / PEAK,
         "infinity_cache_resident": 8.0 * B * C * T < 256 * 2**20}
    result["workloads"][f"{name} ({B},{C},{T})"] = w
    print(f"{tag:14s} {ns / 1e3:9.2f} us  fetch x2 {2 * fetch_b / 1e6:9.1f} MB (expected reads "
          f"{w['read_bytes_expected'] / 1e6:9.1f})  write {write_b / 1e6:9.1f} MB  counter-bytes "
          f"{hbm / ns:7.0f} GB/s = {hbm / ns / PEAK:.3f}  exact {w['frac_on_exact_bytes']:.3f}  12CT "
          f"{w['frac_on_12CT_model']:.3f}")
for k, v in prev.items():
    if k not in result["workloads"]:
        v.setdefault("collected_at", "?")
        result["workloads"][k] = v
for k, v in result["workloads"].items():
    v.setdefault("collected_at", head)
json.dump(result, open(OUT, "w"), indent=1)
