#!/bin/bash
# Kernel durations of one view alone (tools/prof_one_view.py) under rocprofv3 --kernel-trace --stats.
# Usage (on the GPU box, from the repository root): tools/trace_one_view.sh OUTDIR [rows]
out=${1:-gpurun_out/trace}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats -d "$root/$out" --output-format csv -- python3 "$root/tools/prof_one_view.py" > "$root/$out.log" 2>&1 || exit 1
cd "$root" && python tools/kstats.py "$out" ${2:-24}
