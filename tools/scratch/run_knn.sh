#!/bin/bash
# kNN selection rewrite: bit-exact tests, then the pt1024 step with per-kernel detail
cd /root/repo
mkdir -p gpurun_out/knn
timeout 900 python -m pytest tests/test_gpu_point_ops.py tests/test_gpu_dgcnn.py tests/test_gpu_model.py -x -q -m gpu -k "knn or fuzz or dgcnn or golden or model" 2>&1 | tail -5 > gpurun_out/knn/test.log
timeout 600 python bench.py --workload pt1024 --no-also --no-cpu-baseline --detail > gpurun_out/knn/pt1024.log 2>&1
tail -5 gpurun_out/knn/test.log
grep -i "knn" gpurun_out/knn/pt1024.log | head
tail -1 gpurun_out/knn/pt1024.log | cut -c1-300
