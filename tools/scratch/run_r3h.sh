mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_train_variants.py tests/test_gpu_precision.py -q -s -m gpu > gpurun_out/r3h/new.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r3h/new.log | cut -c1-600 | tail -12
grep -o '{"tag": "pointnet".*}' gpurun_out/r3h/new.log | cut -c1-3000
for wl in ssg1024 pt1024; do
  python bench.py --workload $wl --no-also --no-cpu-baseline --detail > gpurun_out/r3h/$wl.json 2> gpurun_out/r3h/$wl.err
  echo "== $wl $(python -c "import json;d=json.loads(open('gpurun_out/r3h/$wl.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],2), d.get('max_abs_dlogit_vs_f32_path'), d['roofline']['per_kernel_ms'])")"
done
