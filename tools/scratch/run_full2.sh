mkdir -p gpurun_out/full
python -m pytest tests -q -m gpu > gpurun_out/full/test.log 2>&1; echo "tests rc $?"; tail -6 gpurun_out/full/test.log | cut -c1-250
python tools/train_detail.py > gpurun_out/full/train_detail.log 2>&1; grep -E "sa_l1|sum" gpurun_out/full/train_detail.log | cut -c1-110
python bench.py --workload pt128_train --no-cpu-baseline 2>/dev/null | head -c 300; echo
