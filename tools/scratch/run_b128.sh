mkdir -p gpurun_out/b128
python -m pytest tests/test_gpu_train_stream.py -x -q -m gpu -k "backward" > gpurun_out/b128/test.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/b128/test.log | cut -c1-300
PCR_STREAM_MIN=1000000000 python tools/bench_tdense.py 128 1536 512
PCR_STREAM_MIN=0 python tools/bench_tdense.py 128 1536 512
python tools/train_detail.py 2>/dev/null | grep -E "L=1536" | cut -c1-110
python bench.py --workload pt128_train --no-cpu-baseline 2>/dev/null | head -c 300; echo
