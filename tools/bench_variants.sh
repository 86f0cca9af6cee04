#!/bin/bash
# Runs bench.py once per library variant, back to back on one box (run-to-run differences between boxes are
# larger than most kernel changes).  Usage: tools/bench_variants.sh "base fw4 fw5" [extra bench.py flags]
# A variant NAME loads var/libgsr_NAME.so; "base" loads the in-tree library.
vars=$1; shift
for v in $vars; do
  if [ "$v" = base ]; then unset GSR_LIB_PATH; else export GSR_LIB_PATH=$PWD/var/libgsr_$v.so; fi
  line=$(timeout -k 10 150 python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline "$@" 2>/dev/null) || { echo "$v: bench failed"; exit 1; }
  echo "$line" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['stage_ms_per_view']
print('%-8s %7.1f renders/s  %7.3f ms/step  fwd %.3f  bwd %.3f (in step %.3f)  offs %.4f lists %.4f  front %.3f  sums %.4f chain/view %.4f' % ('$v', d['value'], d['ms_per_step'], s.get('composite_fwd',0), s.get('composite_bwd',0), d['roofline']['avg_launch_ms'], s.get('bin_offsets_views',s.get('bin_offsets',0)), s.get('bin_lists',0), s.get('preprocess_fwd_views',s.get('preprocess_fwd',0))+s.get('bin_offsets_views',s.get('bin_offsets',0))+s.get('bin_lists',0)+s.get('tile_order',0), s.get('pair_reduce',0), s.get('preprocess_bwd_views',0)))"
done
