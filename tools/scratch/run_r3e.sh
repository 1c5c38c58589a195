mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_model.py tests/test_gpu_precision.py tests/test_gpu_ssg.py -q -s -m gpu > gpurun_out/r3e/new.log 2>&1; echo "tests rc $?"
grep -E "passed|failed|Error" gpurun_out/r3e/new.log | cut -c1-400 | tail -8
grep -o '{"dlogits.*}\|{"vs_ref32": [0-9].*}' gpurun_out/r3e/new.log | cut -c1-300
for wl in ssg1024 pt1024; do
  python bench.py --workload $wl --no-also --no-cpu-baseline --detail > gpurun_out/r3e/$wl.json 2> gpurun_out/r3e/$wl.err
  echo "== $wl $(python -c "import json;d=json.loads(open('gpurun_out/r3e/$wl.json').read().strip().splitlines()[-1]);print(round(d['value']),round(d['ms_per_step'],2))")"
  grep "^sa_\(fused\|ragged\)" gpurun_out/r3e/$wl.err | awk '{print "   ",$1,$2,$3}'
done
