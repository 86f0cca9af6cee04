set -e
mkdir -p gpurun_out/r3d
for lean in 0 1 0 1; do
GSR_ONE_VIEW_LEAN=$lean timeout -k 10 100 python bench.py --no-cpu-baseline --views 1 --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_view']
print('lean=$lean', d['value'], d['ms_per_step'], {k: round(v,4) for k,v in s.items()})"
done > gpurun_out/r3d/views1.txt 2>&1
cat gpurun_out/r3d/views1.txt
