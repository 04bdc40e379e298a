set -e
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --stats -d "$REPO/gpurun_out/prof_r2_bench2" -o b -- python3 "$REPO/bench.py" --no-cpu --no-extra > "$REPO/gpurun_out/prof_r2_bench2.json" 2> "$REPO/gpurun_out/prof_r2_bench2.err"
