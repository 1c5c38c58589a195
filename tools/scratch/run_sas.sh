#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/sas
timeout 1800 python -m pytest tests/test_gpu_ssg.py tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_config_variants.py tests/test_gpu_precision.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/sas/test.log
tail -12 gpurun_out/sas/test.log
for w in ssg1024 pt1024; do
timeout 600 python bench.py --workload $w --no-also --no-cpu-baseline --detail > gpurun_out/sas/$w.log 2>&1
grep -i "sa_fused\|sa_rag" gpurun_out/sas/$w.log | head -8
tail -1 gpurun_out/sas/$w.log | cut -c1-200
done
