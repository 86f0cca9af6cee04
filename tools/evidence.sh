#!/bin/bash
# Collects the evidence of a tree in one GPU call: the default bench line, the kernel statistics of the same command
# under rocprofv3, the traffic passes, the SQ counter passes and the side configurations' bench lines.
# Usage (GPU box, repository root): tools/evidence.sh TAG      -> files under gpurun_out/ev_TAG/
set -e
tag=${1:-ev}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out/ev_$tag
mkdir -p "$root/$out"
cd "$root"
timeout -k 10 240 python bench.py > $out/bench_line.json 2> $out/bench_line.err
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$root/$out/prof" --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline > "$root/$out/bench_line_profiled.json" 2> "$root/$out/prof.err")
cp $out/prof/*/*kernel_stats.csv $out/bench_kernel_stats.csv
python tools/trace_timeline.py $out/prof > $out/step_timeline.txt
python tools/trace_gaps.py $out/prof 24 >> $out/step_timeline.txt
python tools/trace_idle.py $out/prof 15 >> $out/step_timeline.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace -d "$root/$out/prof1" --output-format csv -- python3 "$root/bench.py" --no-cpu-baseline --views 1 --steps 20 > /dev/null 2> "$root/$out/prof1.err")
python tools/trace_one_view_step.py $out/prof1 8 > $out/one_view_step_timeline.txt
python tools/kstats.py $out/prof 12
bash tools/traffic_passes.sh $out/traffic
python tools/traffic_json.py $out/traffic $out/traffic.json
CAM=ring0 bash tools/pmc_composite.sh $out/pmc
python tools/pmc_summary.py $out/pmc/pass1 composite > $out/pmc_composite_sq.txt
python tools/pmc_summary.py $out/pmc/pass2 composite >> $out/pmc_composite_sq.txt
python tools/pmc_all_kernels.py $out/pmc/pass1 $out/pmc/pass2 > $out/pmc_all_kernels.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --tile 64 > $out/tile64_line.json 2>/dev/null
timeout -k 10 200 python bench.py --no-cpu-baseline --views 1 --steps 20 > $out/views1_line.json 2>/dev/null
timeout -k 10 200 python bench.py --no-cpu-baseline --views 2 --steps 20 > $out/views2_line.json 2>/dev/null
timeout -k 10 200 python bench.py --no-cpu-baseline --views 4 --steps 10 > $out/views4_line.json 2>/dev/null
timeout -k 10 200 python bench.py --no-cpu-baseline --no-overlap > $out/one_stream_line.json 2>/dev/null
timeout -k 10 200 python bench.py --no-cpu-baseline --train-iter > $out/train_iter_line.json 2>/dev/null
timeout -k 10 200 bash tools/bench_dist1.sh > $out/one_rank_rccl_line.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --train-iter --gaussians 3000000 > $out/config4_train_iter_line.json 2>/dev/null
timeout -k 10 300 python bench.py --no-cpu-baseline --forward-only --gaussians 6000000 --width 3840 --height 2160 --views 2 --steps 3 > $out/config5_forward_line.json 2>/dev/null
rm -rf $out/prof $out/prof1 $out/pmc/pass*/ $out/traffic/rdreq/*/*agent_info.csv $out/traffic/write/*/*agent_info.csv
ls $out
