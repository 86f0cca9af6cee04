set -e
mkdir -p gpurun_out/r3c
timeout -k 10 800 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3c/suite.log 2>&1 || { tail -30 gpurun_out/r3c/suite.log; exit 1; }
tail -2 gpurun_out/r3c/suite.log
