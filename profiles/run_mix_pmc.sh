#!/bin/bash
# rocprofv3 passes behind profiles/r2_mix_roofline.json: per workload one kernel-trace pass and
# one pass per counter (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).  Run on the GPU
# box from the repo root:  bash profiles/run_mix_pmc.sh
set -e -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out/pmc"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# PCGMIX_PMC_WORKLOADS="karg 256;splice 256" re-profiles a subset (the summary keeps the rest)
IFS=';' read -r -a WLS <<< "${PCGMIX_PMC_WORKLOADS:-copy 16384;splice 16384;warp 16384;copy 256;splice 256;warp 256;karg 256}"
for wl in "${WLS[@]}"; do
  set -- $wl
  mode=$1; B=$2
  iters=20; [ "$B" = "256" ] && iters=200
  d="$OUT/${mode}_${B}"
  echo "== $mode $B"; mkdir -p "$d"
  rocprofv3 -f csv --kernel-trace --stats -d "$d/trace" -o t -- python3 "$REPO/profiles/mix_pmc_probe.py" $mode $B $iters "$OUT" > "$d.trace.log" 2>&1
  rocprofv3 -f csv --pmc FETCH_SIZE -d "$d/fetch" -o f -- python3 "$REPO/profiles/mix_pmc_probe.py" $mode $B $iters "$OUT" > "$d.fetch.log" 2>&1
  rocprofv3 -f csv --pmc WRITE_SIZE -d "$d/write" -o w -- python3 "$REPO/profiles/mix_pmc_probe.py" $mode $B $iters "$OUT" > "$d.write.log" 2>&1
  ls "$d"/*/ | head -20
done
# keep the merge-back small: traces of every torch kernel are not needed, only the csv files
find "$OUT" -name "*.db" -delete 2>/dev/null || true
du -sh "$OUT"
