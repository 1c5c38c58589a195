mkdir -p gpurun_out/r3o
python -m pytest tests/test_gpu_train_stream.py -x -q -m gpu > gpurun_out/r3o/test.log 2>&1; echo "tests rc $?"; tail -15 gpurun_out/r3o/test.log
for pol in 1000000000 0; do
  echo "== policy $pol"
  PCR_STREAM_MIN=$pol python tools/bench_tdense_fwd.py 32 4096 512
  PCR_STREAM_MIN=$pol python tools/bench_tdense_fwd.py 64 3072 512
  PCR_STREAM_MIN=$pol python tools/bench_tdense.py 32 4096 512
  PCR_STREAM_MIN=$pol python tools/bench_tdense.py 64 3072 512
done
python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_variants.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r3o/test2.log 2>&1; echo "tests2 rc $?"; tail -5 gpurun_out/r3o/test2.log
python bench.py --workload pt128_train --no-cpu-baseline > gpurun_out/r3o/pt128_train.json 2> gpurun_out/r3o/pt128_train.err; echo "bench rc $?"
head -c 400 gpurun_out/r3o/pt128_train.json
