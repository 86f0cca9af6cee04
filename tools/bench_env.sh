#!/bin/bash
# Runs bench.py once per environment setting, back to back on one box.
# Usage: tools/bench_env.sh LIBNAME "A=1 B=2" "A=3" ...   (LIBNAME as in bench_variants.sh; "-" for an empty setting)
lib=$1; shift
if [ "$lib" != base ]; then export GSR_LIB_PATH=$PWD/var/libgsr_$lib.so; fi
for setting in "$@"; do
  line=$( ( if [ "$setting" != "-" ]; then export $setting; fi; timeout -k 10 150 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null ) ) || { echo "$setting: bench failed"; exit 1; }
  echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_view']
print('%-34s %7.1f renders/s  %7.3f ms/step  fwd %.3f  bwd %.3f (in step %.3f)  front %.3f' % ('$setting', d['value'], d['ms_per_step'], s.get('composite_fwd',0), s.get('composite_bwd',0), d['roofline']['avg_launch_ms'], s.get('preprocess_fwd',0)+s.get('bin_offsets',0)+s.get('bin_lists',0)+s.get('tile_order',0)))"
done
