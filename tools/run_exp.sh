set -e
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_renderer.py tests/test_gpu_hint_flags.py tests/test_gpu_batched_front.py -x -q --timeout 120 --timeout-method thread > gpurun_out/r3c/parity.log 2>&1 || { tail -30 gpurun_out/r3c/parity.log; exit 1; }
tail -2 gpurun_out/r3c/parity.log
