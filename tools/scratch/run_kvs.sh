#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/kvs
timeout 1200 python -m pytest tests/test_gpu_attn_split.py tests/test_gpu_model.py tests/test_gpu_ssg.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/kvs/test.log
tail -8 gpurun_out/kvs/test.log
for w in pt1024; do
timeout 600 python bench.py --workload $w --no-also --no-cpu-baseline --detail > gpurun_out/kvs/$w.log 2>&1
grep -i "attn_" gpurun_out/kvs/$w.log | head -16
tail -1 gpurun_out/kvs/$w.log | cut -c1-200
done
