mkdir -p gpurun_out/r3v
python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_variants.py tests/test_gpu_attn_split.py -x -q -m gpu > gpurun_out/r3v/test.log 2>&1; echo "tests rc $?"; tail -12 gpurun_out/r3v/test.log | cut -c1-300
