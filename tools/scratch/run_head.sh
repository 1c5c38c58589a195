#!/bin/bash
cd /root/repo
timeout 1200 python -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py tests/test_gpu_config_variants.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python bench.py --workload gallery128 --no-also --no-cpu-baseline --detail > gpurun_out/head_gal.log 2>&1
grep -i "pool_head\|attn_" gpurun_out/head_gal.log | head -6
tail -1 gpurun_out/head_gal.log | cut -c1-160
