#!/bin/bash
# Round-4 profiles (run on the GPU box from the repo root: bash profiles/run_r4_profiles.sh):
#   1. rocprofv3 --kernel-trace --stats of the DRIVER's command line (bench.py --steps 20 --warmup 5)
#   2. the same of one saliency-guided step loop (BASELINE configs[2])
#   3. kernel-trace + FETCH_SIZE + WRITE_SIZE passes (separate, never with a trace domain) of the
#      splice kernels: the new splice+warp kernel at the saturating batch and at bs 256, the plain
#      splice and the kernarg instantiation for reference
# Summaries are folded into profiles/ by profiles/refresh_profiles_r4.py on the build side.
set -e -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --stats -d "$REPO/gpurun_out/prof_r4_bench" -o b -- python3 "$REPO/bench.py" --steps 20 --warmup 5 --no-cpu --no-extra > "$REPO/gpurun_out/prof_r4_bench.json" 2> "$REPO/gpurun_out/prof_r4_bench.err"
echo "bench profiled"
rocprofv3 -f csv --kernel-trace --stats -d "$REPO/gpurun_out/prof_r4_cfg3" -o c -- python3 "$REPO/profiles/probes/cfg3_probe.py" > "$REPO/gpurun_out/prof_r4_cfg3.log" 2>&1
echo "cfg3 profiled"
cd "$REPO"
PCGMIX_PMC_WORKLOADS="${PCGMIX_PMC_WORKLOADS:-warp 16384;splice 16384;warp 256;karg 256}" bash profiles/run_mix_pmc.sh
find "$REPO/gpurun_out/prof_r4_bench" "$REPO/gpurun_out/prof_r4_cfg3" -name "*.db" -delete 2>/dev/null || true
