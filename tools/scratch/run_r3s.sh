mkdir -p gpurun_out/r3s
python -m pytest tests/test_gpu_train_stream.py -x -q -m gpu > gpurun_out/r3s/test.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/r3s/test.log
PCR_STREAM_MIN=0 python tools/bench_tdense_fwd.py 128 1536 512
PCR_STREAM_MIN=0 python tools/bench_tdense_fwd.py 64 3072 512
python bench.py --workload pt128_train --no-cpu-baseline 2>/dev/null | head -c 300; echo
