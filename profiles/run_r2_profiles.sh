set -e -o pipefail
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_r2_bench" -o b -- python3 "$REPO/bench.py" --no-cpu --no-extra > "$OUT/prof_r2_bench.json" 2> "$OUT/prof_r2_bench.err"
rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_r2_rn1d" -o r -- python3 "$REPO/profiles/probes/resnet1d_probe.py" > "$OUT/prof_r2_rn1d.log" 2>&1
rocprofv3 -f csv --kernel-trace --stats -d "$OUT/prof_r2_rn2d" -o r -- python3 "$REPO/profiles/probes/resnet2d_probe.py" > "$OUT/prof_r2_rn2d.log" 2>&1
cd "$REPO"
timeout -k 10 300 python3 profiles/probes/resnet_loss_check.py 1d 256 > "$OUT/resnet_loss_1d.json" 2> "$OUT/resnet_loss_1d.err" || echo "loss check 1d failed/timeout"
timeout -k 10 300 python3 profiles/probes/resnet_loss_check.py 2d 256 > "$OUT/resnet_loss_2d.json" 2> "$OUT/resnet_loss_2d.err" || echo "loss check 2d failed/timeout"
