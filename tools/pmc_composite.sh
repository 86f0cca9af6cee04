#!/bin/bash
# SQ counter passes over tools/prof_one_view.py (one view, forward + backward, 3 repetitions); one pass per
# rocprofv3 run.  Usage: tools/pmc_composite.sh OUTDIR   (run on the GPU box from the repository root)
set -e
out=${1:-gpurun_out/pmc}
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $set -d "$root/$out/pass$i" --output-format csv -- python3 "$root/tools/prof_one_view.py" > "$root/$out/pass$i.log" 2>&1
done
