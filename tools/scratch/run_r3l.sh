mkdir -p gpurun_out/r3l
python -m pytest tests/test_gpu_precision.py tests/test_gpu_ssg.py tests/test_gpu_model.py tests/test_gpu_config_variants.py tests/test_gpu_train_variants.py -q -m gpu > gpurun_out/r3l/new.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r3l/new.log | cut -c1-400 | tail -8
for wl in ssg1024 pt1024; do
  python bench.py --workload $wl --no-also --no-cpu-baseline --detail > gpurun_out/r3l/$wl.json 2> gpurun_out/r3l/$wl.err
  echo "== $wl $(python -c "import json;d=json.loads(open('gpurun_out/r3l/$wl.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],2), d.get('max_abs_dlogit_vs_f32_path'), d['roofline']['per_kernel_ms'])")"
  grep "^sa_tables" gpurun_out/r3l/$wl.err | awk '{print "   ",$1,$2,$3}'
done
