mkdir -p gpurun_out/r3m
python -m pytest tests/test_gpu_point_ops.py tests/test_gpu_ssg.py tests/test_gpu_precision.py tests/test_gpu_fuzz.py -q -m gpu > gpurun_out/r3m/new.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r3m/new.log | cut -c1-400 | tail -8
python tools/fuzz_gpu.py --seconds 30 > gpurun_out/r3m/fuzz.log 2>&1; tail -3 gpurun_out/r3m/fuzz.log
for wl in ssg1024; do
  python bench.py --workload $wl --no-also --no-cpu-baseline --detail > gpurun_out/r3m/$wl.json 2> gpurun_out/r3m/$wl.err
  echo "== $wl $(python -c "import json;d=json.loads(open('gpurun_out/r3m/$wl.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],2), d.get('max_abs_dlogit_vs_f32_path'), d['roofline']['per_kernel_ms'])")"
  grep "^fps" gpurun_out/r3m/$wl.err | awk '{print "   ",$1,$2,$3}'
done
