mkdir -p gpurun_out/r3x
python -m pytest tests/test_gpu_train_stream.py tests/test_gpu_train_ops.py -x -q -m gpu > gpurun_out/r3x/test.log 2>&1; echo "tests rc $?"; tail -8 gpurun_out/r3x/test.log | cut -c1-250
python tools/train_detail.py > gpurun_out/r3x/train_detail.log 2>&1; grep -E "L=1536|L=3072|L=4096|sa_pool|sa_l1" gpurun_out/r3x/train_detail.log | cut -c1-110
python bench.py --workload pt128_train --no-cpu-baseline 2>/dev/null | head -c 300; echo
