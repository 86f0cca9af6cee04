#!/bin/bash
# One-rank rehearsal of the multi-rank bench path (process group, RCCL collectives, barrier) without a launcher:
# the environment the launcher would set is given by hand.  Usage: tools/bench_dist1.sh [bench.py flags]
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29611} GSR_BENCH_FORCE_DIST=1
export HSA_ENABLE_IPC_MODE_LEGACY=0
exec python3 bench.py --gpus 1 --no-cpu-baseline "$@"
