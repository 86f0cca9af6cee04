set -e
mkdir -p gpurun_out/r3b
for rep in 1 2; do
GSR_REC_FLOATS_EXP=12 STEPS=10 bash tools/bench_variants.sh "rec48"
GSR_REC_FLOATS_EXP=16 STEPS=10 bash tools/bench_variants.sh "hsum"
done > gpurun_out/r3b/rec.txt 2>&1
cat gpurun_out/r3b/rec.txt
