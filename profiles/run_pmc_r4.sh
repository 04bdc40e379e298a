#!/bin/bash
# SQ counters (three passes of <= 8) + kernel trace of ONE kernel, round 4.
#   bash profiles/run_pmc_r4.sh <tag> <probe.py> <kernel-name-substring> [ENV=VALUE ...]
# Writes gpurun_out/pmc_r4/<tag>/summary.json (means over the kernel's launches, first one dropped
# from the duration).  Counters in their own passes, --kernel-trace in its own pass (the pool
# refuses --pmc together with trace domains other than the kernel trace).
set -e -o pipefail
TAG="$1"; PROBE="$2"; KERN="$3"; shift 3
for kv in "$@"; do export "$kv"; done
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$REPO"
export TMPDIR=/tmp
OUT="$REPO/gpurun_out/pmc_r4/$TAG"
mkdir -p "$OUT"
rocprofv3 -f csv --kernel-trace -d "$OUT/trace" -o t -- python3 "$PROBE" > "$OUT/trace.log" 2>&1
rocprofv3 -f csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/p1" -o p -- python3 "$PROBE" > "$OUT/p1.log" 2>&1 || echo "pass 1 failed"
rocprofv3 -f csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d "$OUT/p2" -o p -- python3 "$PROBE" > "$OUT/p2.log" 2>&1 || echo "pass 2 failed"
rocprofv3 -f csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_MISC -d "$OUT/p3" -o p -- python3 "$PROBE" > "$OUT/p3.log" 2>&1 || echo "pass 3 failed"
find "$OUT" -name "*.db" -delete 2>/dev/null || true
KERN="$KERN" TAG="$TAG" python3 - <<'PY'
import csv, glob, json, os, statistics
csv.field_size_limit(1 << 30)
tag, kern = os.environ["TAG"], os.environ["KERN"]
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_r4", tag)
d = {}
tr = [r for p in glob.glob(f"{root}/trace/*kernel_trace.csv") for r in csv.DictReader(open(p))
      if kern in r["Kernel_Name"]]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
if tr:
    d["kernel"] = tr[0]["Kernel_Name"].split("(")[0]
    d["launches"] = len(dur)
    d["launch_us"] = statistics.mean(dur[1:] or dur) / 1e3
    d["vgpr"], d["lds_bytes"] = int(tr[0]["VGPR_Count"]), int(tr[0]["LDS_Block_Size"])
    d["grid"] = [tr[0].get("Grid_Size_X"), tr[0].get("Grid_Size_Y"), tr[0].get("Grid_Size_Z")]
for p in glob.glob(f"{root}/p*/*counter_collection.csv"):
    acc = {}
    for r in csv.DictReader(open(p)):
        if kern in r["Kernel_Name"]:
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, vals in acc.items():
        d[k] = statistics.mean(vals)
json.dump(d, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
PY
