mkdir -p gpurun_out/r3f
python -m pytest tests/test_gpu_precision.py tests/test_gpu_model.py tests/test_gpu_config_variants.py tests/test_gpu_ssg.py tests/test_gpu_dgcnn.py tests/test_gpu_pointnet.py tests/test_gpu_distributed.py -q -s -m gpu > gpurun_out/r3f/new.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r3f/new.log | cut -c1-400 | tail -12
grep -o '{"d": .*}' gpurun_out/r3f/new.log | head
for wl in ssg1024 pt1024 pt128 gallery128; do
  python bench.py --workload $wl --no-also --no-cpu-baseline --detail > gpurun_out/r3f/$wl.json 2> gpurun_out/r3f/$wl.err
  echo "== $wl $(python -c "import json;d=json.loads(open('gpurun_out/r3f/$wl.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],2), d.get('max_abs_dlogit_vs_f32_path'), d['roofline']['per_kernel_ms'])")"
done
