mkdir -p gpurun_out/r3q
python -m pytest tests/test_gpu_train_stream.py -x -q -m gpu > gpurun_out/r3q/test.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r3q/test.log
for pol in 0; do
  echo "== policy $pol"
  PCR_STREAM_MIN=$pol python tools/bench_tdense_fwd.py 32 4096 512
  PCR_STREAM_MIN=$pol python tools/bench_tdense_fwd.py 64 3072 512
  PCR_STREAM_MIN=$pol python tools/bench_tdense_fwd.py 128 1536 512
  PCR_STREAM_MIN=$pol python tools/bench_tdense.py 32 4096 512
  PCR_STREAM_MIN=$pol python tools/bench_tdense.py 64 3072 512
done
python bench.py --workload pt128_train --no-cpu-baseline 2>/dev/null | head -c 300; echo
