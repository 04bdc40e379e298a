#!/bin/bash
# SQ counters of potes_bwd_kernel<true> (two passes of <= 8 SQ counters each) + a kernel trace.
set -e -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/bwd_pmc"
mkdir -p "$OUT"
cd "$REPO"
export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace -d "$OUT/trace" -o t -- python3 profiles/probes/potes_bwd_pmc.py > "$OUT/trace.log" 2>&1
rocprofv3 -f csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/p1" -o p -- python3 profiles/probes/potes_bwd_pmc.py > "$OUT/p1.log" 2>&1
rocprofv3 -f csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -d "$OUT/p2" -o p -- python3 profiles/probes/potes_bwd_pmc.py > "$OUT/p2.log" 2>&1 || echo "pass 2 failed"
