set -e
REPO="${GRAFT_REPO_ROOT:-$(pwd)}"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -f csv --kernel-trace --stats -d "$REPO/gpurun_out/prof_r2_cfg3" -o c -- python3 "$REPO/profiles/probes/cfg3_probe.py" > "$REPO/gpurun_out/prof_r2_cfg3.log" 2>&1
