mkdir -p gpurun_out/r3p
for pol in 8192 0; do
PCR_STREAM_MIN_BLOCKS=$pol python tools/train_detail.py > gpurun_out/r3p/train_detail_$pol.log 2>&1; echo "detail rc $?"
PCR_STREAM_MIN_BLOCKS=$pol python bench.py --workload pt128_train --no-cpu-baseline 2>/dev/null | head -c 300; echo
done
