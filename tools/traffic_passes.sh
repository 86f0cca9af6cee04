#!/bin/bash
# HBM-side traffic of the kernels of one view (tools/prof_one_view.py, CAM=ring0: view 0 of bench.py's minibatch):
# two rocprofv3 --pmc passes, one per run - read requests by size, bytes written.
# Usage (GPU box, repository root): tools/traffic_passes.sh OUTDIR ; then tools/traffic_json.py OUTDIR
set -e
out=${1:-gpurun_out/traffic}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$root/$out"
cd /tmp && export TMPDIR=/tmp CAM=ring0 REPS=2
timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_128B_sum -d "$root/$out/rdreq" --output-format csv -- python3 "$root/tools/prof_one_view.py" > "$root/$out/rdreq.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d "$root/$out/write" --output-format csv -- python3 "$root/tools/prof_one_view.py" > "$root/$out/write.log" 2>&1
