set -e -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_r2_rn1d" -o r -- python3 "$REPO/profiles/probes/resnet1d_probe.py" > "$OUT/prof_r2_rn1d.log" 2>&1
rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_r2_rn2d" -o r -- python3 "$REPO/profiles/probes/resnet2d_probe.py" > "$OUT/prof_r2_rn2d.log" 2>&1
