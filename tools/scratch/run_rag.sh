#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/rag
timeout 900 python -m pytest tests/test_gpu_ssg.py -x -q -m gpu 2>&1 | tail -3
for k in 1 2; do
timeout 600 python bench.py --workload ssg1024 --no-also --no-cpu-baseline --detail > gpurun_out/rag/ssg1024.log 2>&1
grep -i "sa_rag" gpurun_out/rag/ssg1024.log | head -4
tail -1 gpurun_out/rag/ssg1024.log | cut -c1-160
done
