#!/bin/bash
# SQ counters (two passes of 8) + kernel trace of the Potes conv-stack forward, for the VALU kernel
# (PCGMIX_POTES_FWD_VALU=1) and the matrix-core one.  bash profiles/run_potes_fwd_pmc_r3.sh
set -e -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd "$REPO"
export TMPDIR=/tmp
for v in valu mfma; do
  OUT="$REPO/gpurun_out/fwd_pmc_r3/$v"
  mkdir -p "$OUT"
  if [ "$v" = valu ]; then export PCGMIX_POTES_FWD_VALU=1; else unset PCGMIX_POTES_FWD_VALU; fi
  rocprofv3 -f csv --kernel-trace -d "$OUT/trace" -o t -- python3 profiles/probes/potes_fwd_pmc.py > "$OUT/trace.log" 2>&1
  rocprofv3 -f csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/p1" -o p -- python3 profiles/probes/potes_fwd_pmc.py > "$OUT/p1.log" 2>&1
  rocprofv3 -f csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d "$OUT/p2" -o p -- python3 profiles/probes/potes_fwd_pmc.py > "$OUT/p2.log" 2>&1 || echo "pass 2 failed"
  rocprofv3 -f csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_MISC -d "$OUT/p3" -o p -- python3 profiles/probes/potes_fwd_pmc.py > "$OUT/p3.log" 2>&1 || echo "pass 3 failed"
  find "$OUT" -name "*.db" -delete 2>/dev/null || true
done
python3 - <<'PY'
import csv, glob, json, os, statistics
csv.field_size_limit(1 << 30)
root = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "fwd_pmc_r3")
out = {}
for v in ("valu", "mfma"):
    d = {}
    tr = [r for p in glob.glob(f"{root}/{v}/trace/*kernel_trace.csv") for r in csv.DictReader(open(p))
          if "potes_fwd" in r["Kernel_Name"]]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
    d["kernel"] = tr[0]["Kernel_Name"].split("(")[0]
    d["launch_us"] = statistics.mean(dur[1:]) / 1e3
    d["vgpr"], d["lds_bytes"] = int(tr[0]["VGPR_Count"]), int(tr[0]["LDS_Block_Size"])
    for p in glob.glob(f"{root}/{v}/p*/*counter_collection.csv"):
        acc = {}
        for r in csv.DictReader(open(p)):
            if "potes_fwd" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, vals in acc.items():
            d[k] = statistics.mean(vals)
    out[v] = d
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
