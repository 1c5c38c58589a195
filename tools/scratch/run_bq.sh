#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/bq
timeout 900 python -m pytest tests/test_gpu_point_ops.py tests/test_gpu_ssg.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/bq/test.log
timeout 600 python bench.py --no-also --no-cpu-baseline --detail > gpurun_out/bq/ssg1024.log 2>&1
tail -3 gpurun_out/bq/test.log
grep -i "ball\|fps" gpurun_out/bq/ssg1024.log | head
tail -1 gpurun_out/bq/ssg1024.log | cut -c1-200
